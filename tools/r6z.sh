#!/bin/bash
# round 6: upconv_fwd_ws2 in halo-row-major order with each output row's epilogue issued in front of the next halo row's MFMAs (WS2_ROWMAJOR)
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
V=${1:-strajnet_amd/variants/lib_ws2row.so}
{
STJ_LIB_PATH=$V python -m pytest tests/test_ops_gpu.py tests/test_timed_kernels_gpu.py -q -x -k "upconv or head or outconv" 2>&1 | tail -3
STJ_LIB_PATH=$V python -m pytest tests/test_model_gpu.py -q -x -k "config4 or fp16 or infer" 2>&1 | tail -3
for i in 1 2; do
echo base; python tools/bench_conv.py --layer 3 --only fwd --iters 20 2>&1 | tail -1
echo variant; STJ_LIB_PATH=$V python tools/bench_conv.py --layer 3 --only fwd --iters 20 2>&1 | tail -1
done
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--no-cpu-baseline --no-extra-configs --no-kernel-timing"
for i in 1 2 3; do
  STJ_LIB_PATH=$V python bench.py --infer $B --steps 60 --warmup 5 2>/dev/null | line "infer variant"
  python bench.py --infer $B --steps 60 --warmup 5 2>/dev/null | line "infer base"
done
for i in 1 2 3; do
  STJ_LIB_PATH=$V python bench.py $B --steps 200 --warmup 10 2>/dev/null | line "train variant"
  python bench.py $B --steps 200 --warmup 10 2>/dev/null | line "train base"
done
} 2>&1 | tee gpurun_out/r06_z7_ws2_for_128_step.txt
