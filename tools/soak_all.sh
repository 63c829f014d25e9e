#!/bin/bash
# every soak mode once (GPU box): usage tools/soak_all.sh [first_case] [n_per_mode]
f=${1:-700000}; n=${2:-2000}
run() { echo "== $1"; shift; env "$@" timeout ${FX_SOAK_TIMEOUT:-1500} python3 tools/soak_parity.py $f $n 2>&1 | grep -v amdgpu | tail -4; f=$((f + 10000)); }
run plain A=1
run tuning FX_SOAK_TUNING=1
run crowded FX_SOAK_MANY=1
run crowded+tuning FX_SOAK_MANY=1 FX_SOAK_TUNING=1
run very-crowded FX_SOAK_MANY=2
run costs FX_SOAK_COSTS=1
run costs+tuning FX_SOAK_COSTS=1 FX_SOAK_TUNING=1
run matrix FX_SOAK_MATRIX=1
run matrix+crowded+tuning FX_SOAK_MATRIX=1 FX_SOAK_MANY=1 FX_SOAK_TUNING=1
run topk+tuning FX_SOAK_TOPK=1 FX_SOAK_TUNING=1
run projection+tuning FX_SOAK_PROJ=1 FX_SOAK_TUNING=1
run tail FX_SOAK_TAIL=1
run tail+tuning FX_SOAK_TAIL=1 FX_SOAK_TUNING=1
run tail+crowded+costs FX_SOAK_TAIL=1 FX_SOAK_MANY=1 FX_SOAK_COSTS=1
run tail+matrix+tuning FX_SOAK_TAIL=1 FX_SOAK_MATRIX=1 FX_SOAK_TUNING=1
run step-kernel FX_SOAK_STEPK=1
run step-kernel+crowded FX_SOAK_STEPK=1 FX_SOAK_MANY=1
echo "== batch"; timeout 900 python3 tools/soak_batch.py $f 300 2>&1 | grep -v amdgpu | tail -3
