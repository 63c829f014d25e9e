#!/bin/bash
# usage (GPU box, repo root): bash tools/collect_profiles_r2.sh [label]
# rocprofv3 evidence for the bench line: kernel-trace stats of the default bench command and separate PMC passes (FETCH_SIZE,
# WRITE_SIZE, executed FP64 instructions, SQ busy / wait) for the four kernels the line reports.  Writes
# gpurun_out/<label>/summary.json (+ kernel_stats.csv); copy both into profiles/r2/ to have bench.py read them.
label=${1:-prof_r2}
R=$(pwd); O=$R/gpurun_out/$label; mkdir -p $O
export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $BENCH --steps 200 --warmup 20 > $O/stats.log 2>&1)
declare -A CMD
CMD[config3_modeB]="$R/bench.py --no-cpu-baseline --no-north-star --steps 20 --warmup 5"
CMD[config2_modeB]="$R/bench.py --no-cpu-baseline --no-north-star --workload config2 --steps 20 --warmup 5"
CMD[north_star_obstacles]="$R/tools/tools_run_one.py 0 0 0 modeA 20 19 230 229 8"
CMD[north_star_bundle]="$R/tools/tools_run_one.py 0 0 0 modeB 0 19 230 229 8"
GROUPS_PMC=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM")
for sec in config3_modeB config2_modeB north_star_obstacles north_star_bundle; do
  i=0
  for grp in "${GROUPS_PMC[@]}"; do
    (cd /tmp && rocprofv3 --pmc $grp --output-format csv -d $O/pmc_${sec}_$i -- python3 ${CMD[$sec]} > $O/pmc_${sec}_$i.log 2>&1)
    i=$((i+1))
  done
done
python3 - <<PY
import csv, glob, json, collections
O = "$O"
out = {"kernel_stats_command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 200 --warmup 20"}
rows = []
for f in glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    open(O + "/kernel_stats.csv", "w").write(open(f).read())
out["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev") if k in r}
                       for r in rows if "fx_" in r["Name"]]
cmds = {"config3_modeB": "python3 bench.py --no-cpu-baseline --no-north-star --steps 20 --warmup 5",
        "config2_modeB": "python3 bench.py --no-cpu-baseline --no-north-star --workload config2 --steps 20 --warmup 5",
        "north_star_obstacles": "python3 tools/tools_run_one.py 0 0 0 modeA 20 19 230 229 8",
        "north_star_bundle": "python3 tools/tools_run_one.py 0 0 0 modeB 0 19 230 229 8"}
for sec, cmd in cmds.items():
    agg = collections.defaultdict(list)
    kern = None
    for f in glob.glob(O + f"/pmc_{sec}_*/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fx_eval" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                kern = r["Kernel_Name"]
    out[sec] = {"command": "rocprofv3 --pmc <group> -- " + cmd + "  (one pass per counter group)", "kernel": kern,
                "pmc_per_launch_mean": {c: sum(v) / len(v) for c, v in sorted(agg.items())},
                "pmc_launches": {c: len(v) for c, v in sorted(agg.items())}}
json.dump(out, open(O + "/summary.json", "w"), indent=1)
print(json.dumps({k: (v if k == "kernel_stats" else (v.get("kernel"), len(v.get("pmc_per_launch_mean", {}))) if isinstance(v, dict) else v) for k, v in out.items()}, indent=1)[:3000])
PY
