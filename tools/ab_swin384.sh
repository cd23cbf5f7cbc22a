#!/bin/bash
# same-box A/B: the C = 384 Swin stage fused (split kernels) vs layer by layer
cd $GRAFT_REPO_ROOT 2>/dev/null || true
line() { python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  timeout 120 python tools/ab_attr.py -- --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>/dev/null | line fused384
  timeout 120 python tools/ab_attr.py 'fused_mlp_dims=(96,192)' 'fused_attn_dims=(96,192)' -- --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>/dev/null | line layerwise384
done
