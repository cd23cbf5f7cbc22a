#!/bin/bash
# round 6: fused FG-MSA offset head (csrc/fgoff_fused.hip) A/B, alternating same-box runs; then the model-level tests
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--no-cpu-baseline --no-extra-configs --no-kernel-timing"
for i in 1 2 3; do
  python tools/ab_attr.py fused_fgoff=True -- $B --steps 60 --warmup 10 2>/dev/null | line train_fused
  python tools/ab_attr.py fused_fgoff=False -- $B --steps 60 --warmup 10 2>/dev/null | line train_layerwise
done 2>&1 | tee gpurun_out/r06_j_fgoff.txt
for i in 1 2 3; do
  python tools/ab_attr.py fused_fgoff=True -- $B --infer --steps 30 --warmup 5 2>/dev/null | line infer_fused
  python tools/ab_attr.py fused_fgoff=False -- $B --infer --steps 30 --warmup 5 2>/dev/null | line infer_layerwise
done 2>&1 | tee -a gpurun_out/r06_j_fgoff.txt
python -m pytest tests/test_model_gpu.py tests/test_switches_gpu.py tests/test_timed_kernels_gpu.py -q -x -m gpu 2>&1 | tail -5 | tee gpurun_out/r06_j_tests.txt
