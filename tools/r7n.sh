#!/bin/bash
# round 6: heads per pass of the C = 96 attention forward (STJ_ATTN_HG96) at 131072 rows (inference B = 32, cfg-512)
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
V=strajnet_amd/variants/lib_attn_hg3.so
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--no-cpu-baseline --no-extra-configs --no-kernel-timing"
{
for i in 1 2 3; do
  STJ_LIB_PATH=$V python bench.py --infer $B --steps 60 --warmup 5 2>/dev/null | line "infer hg3"
  python bench.py --infer $B --steps 60 --warmup 5 2>/dev/null | line "infer base"
done
for i in 1 2; do
  STJ_LIB_PATH=$V python bench.py --cfg512 $B --steps 40 --warmup 5 2>/dev/null | line "cfg512 hg3"
  python bench.py --cfg512 $B --steps 40 --warmup 5 2>/dev/null | line "cfg512 base"
done
} 2>&1 | tee gpurun_out/r07_n_attn_hg96.txt
