#!/bin/bash
# round 6: the loss's forward sums and d/dlogits as one pass (stj_loss_fwd_bwd): tests, the loss alone, alternating same-box runs of the step
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
{
python -m pytest tests/test_ops_gpu.py -q -x -k "loss" 2>&1 | tail -3
python -m pytest tests/test_model_gpu.py tests/test_lib_and_dp.py -q -x 2>&1 | tail -3
echo "== loss alone"; python tools/bench_loss.py
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--no-cpu-baseline --no-extra-configs --no-kernel-timing"
for i in 1 2 3; do
  python tools/ab_attr.py ops.LOSS_FUSED_BWD=True -- $B --steps 300 --warmup 10 2>/dev/null | line one_pass_fin_side
  python tools/ab_attr.py ops.LOSS_FUSED_BWD=True ops.LOSS_FIN_SIDE=False -- $B --steps 300 --warmup 10 2>/dev/null | line one_pass_fin_main
  python tools/ab_attr.py ops.LOSS_FUSED_BWD=False -- $B --steps 300 --warmup 10 2>/dev/null | line two_passes
done
} 2>&1 | tee gpurun_out/r06_u_loss_one_pass.txt
