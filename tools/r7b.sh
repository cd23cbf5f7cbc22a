#!/bin/bash
# round 6: MLP half at C = 96 with 2 / 4 row fragments per wave at >= 131072 rows (inference B = 32, cfg-512)
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--no-cpu-baseline --no-extra-configs --no-kernel-timing"
{
STJ_LIB_PATH=strajnet_amd/variants/lib_mlp_f4b2.so python -m pytest tests/test_ops_gpu.py -q -x -k "swin_mlp_fused" 2>&1 | tail -2
for i in 1 2 3; do
  for v in base mlp_f2b2 mlp_f4b2; do
    if [ $v = base ]; then L=""; else L=strajnet_amd/variants/lib_$v.so; fi
    STJ_LIB_PATH=$L python bench.py --infer $B --steps 60 --warmup 5 2>/dev/null | line "infer $v"
  done
done
for i in 1 2 3; do
  for v in base mlp_f2b2 mlp_f4b2; do
    if [ $v = base ]; then L=""; else L=strajnet_amd/variants/lib_$v.so; fi
    STJ_LIB_PATH=$L python bench.py --cfg512 $B --steps 40 --warmup 5 2>/dev/null | line "cfg512 $v"
  done
done
} 2>&1 | tee gpurun_out/r07_b_mlp96_rf.txt
