#!/bin/bash
# round 6: where the decoder's three skip GEMMs are issued on the side stream (behind the encoder / FG-MSA / the cross-attention)
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--no-cpu-baseline --no-extra-configs --no-kernel-timing"
{
for i in 1 2 3; do
  for m in 0 1 2; do python tools/ab_attr.py skips_issue=$m -- --infer $B --steps 60 --warmup 5 2>/dev/null | line "infer skips_issue=$m"; done
done
for i in 1 2; do
  for m in 0 1 2; do python tools/ab_attr.py skips_issue=$m -- $B --steps 200 --warmup 10 2>/dev/null | line "train skips_issue=$m"; done
done
} 2>&1 | tee gpurun_out/r07_k_skips_issue.txt
