#!/bin/bash
# round 6: workgroup budget of the two large deferred up-conv weight-gradient launches, again, at the end of the round (the thin chains beside them are shorter now)
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--no-cpu-baseline --no-extra-configs --no-kernel-timing"
for i in 1 2 3; do
  for n in 128 96 160 192 256; do python tools/ab_attr.py ops.UPWG_BUDGET=$n -- $B --steps 200 --warmup 10 2>/dev/null | line budget_$n; done
done 2>&1 | tee gpurun_out/r07_a_upwg_budget.txt
