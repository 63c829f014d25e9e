#!/bin/bash
# usage (GPU box, repo root): bash tools/collect_profiles.sh <label>
# Runs the headline bench (config2 Mode B with cpu_baseline, Mode A, config3), a rocprofv3 kernel-trace of the
# same bench command and separate PMC passes (FETCH_SIZE, WRITE_SIZE, an SQ group), then writes the summaries
# to gpurun_out/<label>/ (copy the ones to be judged into profiles/).
label=${1:-final}
R=$(pwd); O=$R/gpurun_out/$label; mkdir -p $O
export TMPDIR=/tmp
python3 bench.py --steps 300 --warmup 30 > $O/bench_config2_modeB.json 2> $O/bench_config2_modeB.err
python3 bench.py --steps 300 --warmup 30 --select-only --no-cpu-baseline > $O/bench_config2_modeA.json 2>/dev/null
python3 bench.py --steps 200 --warmup 20 --workload config3 > $O/bench_config3.json 2>/dev/null
python3 bench.py --steps 30 --warmup 3 --workload config5 > $O/bench_config5.json 2>/dev/null
python3 bench.py --steps 60 --warmup 6 --workload config4 > $O/bench_config4.json 2>/dev/null
python3 bench.py --steps 300 --warmup 30 --workload config1 > $O/bench_config1.json 2>/dev/null
python3 tools/e2e_step.py > $O/e2e_step.json 2>/dev/null
python3 tools/obst_split.py > $O/obst_split.txt 2>/dev/null
BENCH="python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $BENCH > $O/stats.log 2>&1)
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-24)
  (cd /tmp && rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$n -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/pmc_$n.log 2>&1)
done
python3 - <<PY
import csv, glob, json, collections
O = "$O"
out = {"command": "python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline (PMC passes: --steps 20 --warmup 5)"}
rows = []
for f in glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    open(O + "/kernel_stats.csv", "w").write(open(f).read())
out["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev") if k in r}
                       for r in rows if r["Name"].startswith("fx_") or "fx_" in r["Name"]]
agg = collections.defaultdict(list)
for f in glob.glob(O + "/pmc_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fx_eval" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            out["pmc_kernel"] = r["Kernel_Name"]
out["pmc_per_launch_mean"] = {c: sum(v) / len(v) for c, v in sorted(agg.items())}
out["pmc_launches"] = {c: len(v) for c, v in sorted(agg.items())}
json.dump(out, open(O + "/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
