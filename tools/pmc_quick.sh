#!/bin/bash
# usage (GPU box, repo root): tools/pmc_quick.sh <label> <workload of run_bench_workload.py> [steps]
# two rocprofv3 PMC passes (instruction mix, cycle counters) of one workload -> gpurun_out/<label>.txt (means per launch)
label=$1; wl=${2:-north_star_obstacles}; steps=${3:-6}
R=$(pwd); O=$R/gpurun_out/pmcq_$label; mkdir -p $O; export TMPDIR=/tmp
G0="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES"
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"
(cd /tmp && rocprofv3 --pmc $G0 --output-format csv -d $O/p0 -- python3 $R/tools/run_bench_workload.py $wl $steps > $O/p0.log 2>&1)
(cd /tmp && rocprofv3 --pmc $G1 --output-format csv -d $O/p1 -- python3 $R/tools/run_bench_workload.py $wl $steps > $O/p1.log 2>&1)
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fx_eval" in k or "fx_obstacle" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$R/gpurun_out/pmcq_$label.txt", "w") as out:
    for k, d in agg.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        f64 = sum(m.get("SQ_INSTS_VALU_%s_F64" % x, 0) for x in ("ADD", "MUL", "FMA", "TRANS"))
        w = max(m.get("SQ_WAVES", 1), 1)
        line = (f"{k}\n  waves {w:.0f}  VALU {m.get('SQ_INSTS_VALU',0)/1e6:.1f}M ({m.get('SQ_INSTS_VALU',0)/w:.0f}/wave)  FP64 {f64/1e6:.1f}M "
                f"({f64/max(m.get('SQ_INSTS_VALU',1),1):.3f})  add/mul/fma/trans {m.get('SQ_INSTS_VALU_ADD_F64',0)/1e6:.1f}/{m.get('SQ_INSTS_VALU_MUL_F64',0)/1e6:.1f}/"
                f"{m.get('SQ_INSTS_VALU_FMA_F64',0)/1e6:.1f}/{m.get('SQ_INSTS_VALU_TRANS_F64',0)/1e6:.2f}  SALU {m.get('SQ_INSTS_SALU',0)/1e6:.1f}M  LDS {m.get('SQ_INSTS_LDS',0)/1e6:.1f}M  SMEM {m.get('SQ_INSTS_SMEM',0)/1e6:.2f}M\n"
                f"  WAVE_CYCLES {m.get('SQ_WAVE_CYCLES',0)/1e6:.1f}M  BUSY {m.get('SQ_BUSY_CYCLES',0)/1e6:.1f}M  ACTIVE_VALU {m.get('SQ_ACTIVE_INST_VALU',0)/1e6:.1f}M  ACTIVE_ANY {m.get('SQ_ACTIVE_INST_ANY',0)/1e6:.1f}M  "
                f"WAIT_INST_ANY {m.get('SQ_WAIT_INST_ANY',0)/1e6:.1f}M  WAIT_ANY {m.get('SQ_WAIT_ANY',0)/1e6:.1f}M  ACTIVE_LDS {m.get('SQ_ACTIVE_INST_LDS',0)/1e6:.1f}M\n")
        out.write(line); print(line)
PY
