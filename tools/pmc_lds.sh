#!/bin/bash
# usage (GPU box, repo root): tools/pmc_lds.sh <label> <workload of run_bench_workload.py> [steps]
# LDS / instruction-fetch / scalar-cache counters of one workload, one rocprofv3 pass per group (a group with a counter this
# build of rocprofv3 does not know fails alone) -> gpurun_out/pmclds_<label>.txt (means per launch of the evaluation kernels)
label=$1; wl=${2:-north_star_obstacles}; steps=${3:-6}
R=$(pwd); O=$R/gpurun_out/pmclds_$label; mkdir -p $O; export TMPDIR=/tmp
i=0
for G in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
         "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN" \
         "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM" \
         "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU" \
         "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"; do
  (cd /tmp && rocprofv3 --pmc $G --output-format csv -d $O/p$i -- python3 $R/tools/run_bench_workload.py $wl $steps > $O/p$i.log 2>&1) || echo "group $i failed: $G"
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fx_eval" in k or "fx_obstacle" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$R/gpurun_out/pmclds_$label.txt", "w") as out:
    for k, d in agg.items():
        out.write(k + "\n"); print(k)
        for c in sorted(d):
            line = f"  {c:28s} {sum(d[c]) / len(d[c]) / 1e6:12.3f} M"
            out.write(line + "\n"); print(line)
PY
