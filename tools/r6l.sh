#!/bin/bash
# round 6: decoder skip junction (stj_skip_junction_bwd) -- tests, then A/B on the training bench in alternating same-box runs
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "upconv_add or linear_z" 2>&1 | tail -5 | tee gpurun_out/r06_l_tests.txt
python -m pytest tests/test_timed_kernels_gpu.py tests/test_fgoff_fused_gpu.py -q -x -m gpu 2>&1 | tail -3 | tee -a gpurun_out/r06_l_tests.txt
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--no-cpu-baseline --no-extra-configs --no-kernel-timing"
for i in 1 2 3 4; do
  python tools/ab_attr.py ops.SKIP_JUNCTION=True -- $B --steps 60 --warmup 10 2>/dev/null | line train_junction
  python tools/ab_attr.py ops.SKIP_JUNCTION=False -- $B --steps 60 --warmup 10 2>/dev/null | line train_separate
done 2>&1 | tee gpurun_out/r06_l_junction.txt
