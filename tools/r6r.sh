#!/bin/bash
# round 6: stage-0 mid flush of the Dense weight gradients onto the side stream (the pass's last launch halves), alternating same-box runs
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--no-cpu-baseline --no-extra-configs --no-kernel-timing"
for i in 1 2 3 4; do
  python tools/ab_attr.py wg_mid_flush=True -- $B --steps 60 --warmup 10 2>/dev/null | line mid_flush
  python tools/ab_attr.py wg_mid_flush=False -- $B --steps 60 --warmup 10 2>/dev/null | line end_only
done 2>&1 | tee gpurun_out/r06_r_mid_flush.txt
