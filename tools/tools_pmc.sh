#!/bin/bash
# usage: tools_pmc.sh <label> <args to tools_run_one.py...>   (GPU box; writes gpurun_out/pmc_<label>.txt)
label=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp; O=$R/gpurun_out/pmc_$label; mkdir -p $O
cd $R
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum" "TA_BUSY_avr TA_DATA_STALL_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCC_EA0_WRREQ_STALL_sum"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  (cd /tmp && rocprofv3 --pmc $grp --output-format csv -d $O/$n -- python3 $R/tools/tools_run_one.py "$@" > $O/$n.log 2>&1)
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$O/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fx_eval" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            k = r["Kernel_Name"]
print("$label", k)
for c, v in sorted(agg.items()):
    print(f"  {c:28s} {sum(v)/len(v):16.1f}")
PY
