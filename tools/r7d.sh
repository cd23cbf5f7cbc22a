#!/bin/bash
# round 6: swin_attn_bwd reading the bias table from its per-workgroup copy directly (one LDS copy and one barrier per head less)
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
V=strajnet_amd/variants/lib_attnb_tbv.so
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--no-cpu-baseline --no-extra-configs --no-kernel-timing"
{
STJ_LIB_PATH=$V python -m pytest tests/test_ops_gpu.py -q -x -k "swin_attn_half" 2>&1 | tail -2
for i in 1 2 3; do
  STJ_LIB_PATH=$V python bench.py $B --steps 200 --warmup 10 2>/dev/null | line "train variant"
  python bench.py $B --steps 200 --warmup 10 2>/dev/null | line "train base"
done
for i in 1 2; do
  STJ_LIB_PATH=$V python bench.py --cfg512 $B --steps 40 --warmup 5 2>/dev/null | line "cfg512 variant"
  python bench.py --cfg512 $B --steps 40 --warmup 5 2>/dev/null | line "cfg512 base"
done
} 2>&1 | tee gpurun_out/r07_d_attnb_tbv.txt
