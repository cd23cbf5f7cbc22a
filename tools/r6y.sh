#!/bin/bash
# round 6: upconv_fwd_ws (128 -> 96) with a halo row read once for both tap rows it feeds (WS_ROWREUSE): parity, then alternating same-box runs
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
V=strajnet_amd/variants/lib_wsrow.so
{
STJ_LIB_PATH=$V python -m pytest tests/test_ops_gpu.py tests/test_timed_kernels_gpu.py -q -x -k "upconv" 2>&1 | tail -3
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--no-cpu-baseline --no-extra-configs --no-kernel-timing"
for i in 1 2 3; do
  STJ_LIB_PATH=$V python bench.py --infer $B --steps 30 --warmup 5 2>/dev/null | line "infer rowreuse"
  python bench.py --infer $B --steps 30 --warmup 5 2>/dev/null | line "infer base"
done
for i in 1 2 3; do
  STJ_LIB_PATH=$V python bench.py $B --steps 200 --warmup 10 2>/dev/null | line "train rowreuse"
  python bench.py $B --steps 200 --warmup 10 2>/dev/null | line "train base"
done
} 2>&1 | tee gpurun_out/r06_y_ws_rowreuse.txt
