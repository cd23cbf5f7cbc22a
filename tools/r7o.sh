#!/bin/bash
# round 6: hidden columns per staged weight chunk of the C = 96 MLP kernels (STJ_MLP_HC96) at 131072 rows and in the train step
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--no-cpu-baseline --no-extra-configs --no-kernel-timing"
{
for i in 1 2; do
  for v in base mlp_hc192 mlp_hc48; do
    if [ $v = base ]; then L=""; else L=strajnet_amd/variants/lib_$v.so; fi
    [ -n "$L" ] && [ ! -f "$L" ] && continue
    STJ_LIB_PATH=$L python bench.py --infer $B --steps 60 --warmup 5 2>/dev/null | line "infer $v"
    STJ_LIB_PATH=$L python bench.py --cfg512 $B --steps 40 --warmup 5 2>/dev/null | line "cfg512 $v"
    STJ_LIB_PATH=$L python bench.py $B --steps 200 --warmup 10 2>/dev/null | line "train $v"
  done
done
} 2>&1 | tee gpurun_out/r07_o_mlp_hc96.txt
