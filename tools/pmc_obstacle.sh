#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_obstacle.sh <label> <workload> <stage> <steps per item>
# PMC passes (one per counter group, no trace domain) of tools/run_split.py; prints per-kernel means.
label=$1; shift
R=$(pwd); O=$R/gpurun_out/$label; mkdir -p $O
export TMPDIR=/tmp
GROUPS_PMC=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS_F64 GRBM_GUI_ACTIVE")
i=0
for grp in "${GROUPS_PMC[@]}"; do
  (cd /tmp && rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$i -- python3 $R/tools/run_split.py "$@" > $O/pmc_$i.log 2>&1) || tail -5 $O/pmc_$i.log
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fx_" in k:
            agg[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
