#!/bin/bash
# round 6: C = 96 MLP kernels at 32768 rows with ALL weights resident (one 384-column chunk, 154 KB of LDS) against two 192-column chunks
cd $GRAFT_REPO_ROOT 2>/dev/null || true
V=strajnet_amd/variants/lib_mlp_w384.so
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--no-cpu-baseline --no-extra-configs --no-kernel-timing"
{
STJ_LIB_PATH=$V python -m pytest tests/test_timed_kernels_gpu.py -q -x -k "swin or mlp" 2>&1 | tail -1
for i in 1 2 3 4; do
  STJ_LIB_PATH=$V python bench.py $B --steps 300 --warmup 10 2>/dev/null | line "train hc384"
  python bench.py $B --steps 300 --warmup 10 2>/dev/null | line "train hc192"
done
} 2>&1 | tee gpurun_out/r07_t_mlp_hc384.txt
