#!/bin/bash
# round 6: swin_attn_fwd<96> compiled for three waves per SIMD (three windows per CU instead of two) -- inference B = 32, cfg-512, train
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
V=strajnet_amd/variants/lib_attn_occ3.so
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--no-cpu-baseline --no-extra-configs --no-kernel-timing"
{
STJ_LIB_PATH=$V python -m pytest tests/test_ops_gpu.py -q -x -k "swin_attn_half" 2>&1 | tail -1
for i in 1 2 3; do
  STJ_LIB_PATH=$V python bench.py --infer $B --steps 60 --warmup 5 2>/dev/null | line "infer occ3"
  python bench.py --infer $B --steps 60 --warmup 5 2>/dev/null | line "infer base"
done
for i in 1 2; do
  STJ_LIB_PATH=$V python bench.py --cfg512 $B --steps 40 --warmup 5 2>/dev/null | line "cfg512 occ3"
  python bench.py --cfg512 $B --steps 40 --warmup 5 2>/dev/null | line "cfg512 base"
  STJ_LIB_PATH=$V python bench.py $B --steps 200 --warmup 10 2>/dev/null | line "train occ3"
  python bench.py $B --steps 200 --warmup 10 2>/dev/null | line "train base"
done
} 2>&1 | tee gpurun_out/r07_r_attn_occ3.txt
