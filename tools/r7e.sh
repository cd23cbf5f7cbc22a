#!/bin/bash
# round 6: cfg-512's C = 384 stage (8192 rows, six blocks): fused split kernels vs the layer-by-layer path, per half
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--cfg512 --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 40 --warmup 5"
for i in 1 2; do
  python tools/ab_attr.py -- $B 2>/dev/null | line fused_both
  python tools/ab_attr.py 'fused_attn_dims=(96,192)' -- $B 2>/dev/null | line attn384_layerwise
  python tools/ab_attr.py 'fused_mlp_dims=(96,192)' -- $B 2>/dev/null | line mlp384_layerwise
  python tools/ab_attr.py 'fused_attn_dims=(96,192)' 'fused_mlp_dims=(96,192)' -- $B 2>/dev/null | line both384_layerwise
done 2>&1 | tee gpurun_out/r07_e_cfg512_c384_layerwise.txt
