#!/bin/bash
# usage (GPU box, repo root): bash tools/collect_profiles.sh [label]
# rocprofv3 evidence for the bench line: kernel-trace stats of the default bench command and separate PMC passes (no trace
# domain combined with --pmc) for every fx_* kernel of the workloads the line reports.  Writes gpurun_out/<label>/summary.json
# (+ kernel_stats.csv); copy both into profiles/r<round>/ to have bench.py read them.  Per section the summary holds
# "kernels": {<kernel name as rocprofv3 prints it>: {counter: mean per launch}} -- bench.py looks a kernel up by the name
# fx_step_info_ex reports and marks its figures "stale" when the summary holds another specialisation.
label=${1:-prof}
R=$(pwd); O=$R/gpurun_out/$label; mkdir -p $O
export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $BENCH --steps 200 --warmup 20 > $O/stats.log 2>&1)
declare -A CMD
CMD[config3_modeB]="$R/tools/run_bench_workload.py config3 20"
CMD[config2_modeB]="$R/tools/run_bench_workload.py config2 20"
CMD[north_star_obstacles]="$R/tools/run_bench_workload.py north_star_obstacles 8"
CMD[north_star_bundle]="$R/tools/run_bench_workload.py north_star_bundle 8"
CMD[north_star_bundle_obstacles]="$R/tools/run_bench_workload.py north_star_bundle_obstacles 8"
CMD[config1]="$R/bench.py --workload config1 --steps 200 --warmup 20 --no-cpu-baseline"
CMD[config5_modeA]="$R/bench.py --workload config5 --agents-per-gpu 32 --steps 4 --warmup 1 --no-cpu-baseline --preheat 0"
GROUPS_PMC=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM")
for sec in config3_modeB config2_modeB north_star_obstacles north_star_bundle north_star_bundle_obstacles config1 config5_modeA; do
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
cmds = {"config3_modeB": "python3 tools/run_bench_workload.py config3 20", "config2_modeB": "python3 tools/run_bench_workload.py config2 20",
        "north_star_obstacles": "python3 tools/run_bench_workload.py north_star_obstacles 8",
        "north_star_bundle": "python3 tools/run_bench_workload.py north_star_bundle 8",
        "north_star_bundle_obstacles": "python3 tools/run_bench_workload.py north_star_bundle_obstacles 8",
        "config1": "python3 bench.py --workload config1 --steps 200 --warmup 20 --no-cpu-baseline",
        "config5_modeA": "python3 bench.py --workload config5 --agents-per-gpu 32 --steps 4 --warmup 1 --no-cpu-baseline --preheat 0"}
for sec, cmd in cmds.items():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(O + f"/pmc_{sec}_*/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "fx_eval" in k or "fx_obstacle" in k or "fx_select" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[sec] = {"command": "rocprofv3 --pmc <group> -- " + cmd + "  (one pass per counter group)",
                "kernels": {k: {c: sum(v) / len(v) for c, v in sorted(d.items())} for k, d in agg.items()},
                "launches": {k: {c: len(v) for c, v in sorted(d.items())} for k, d in agg.items()}}
out["config5_modeA"]["units_per_launch"] = 32   # agents per launch of that command (bench.py scales the counts to its own)
json.dump(out, open(O + "/summary.json", "w"), indent=1)
print(json.dumps({k: (v if k == "kernel_stats" else list(v.get("kernels", {})) if isinstance(v, dict) else v) for k, v in out.items()}, indent=1)[:4000])
PY
