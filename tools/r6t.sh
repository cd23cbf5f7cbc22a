#!/bin/bash
# round 6: part of the deferred decoder weight gradients (oldest first: the 256 x 256 level's) released at the two-skip level's junction, alternating same-box runs
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--no-cpu-baseline --no-extra-configs --no-kernel-timing"
for i in 1 2 3; do
  for n in 0 1 2 4; do python tools/ab_attr.py ops.UPWG_EARLY=$n -- $B --steps 60 --warmup 10 2>/dev/null | line early_$n; done
done 2>&1 | tee gpurun_out/r06_t_upwg_early.txt
