#!/bin/bash
# inference A/B: decoder skip sums in the wide up-convs' epilogue (default under no_grad) vs separate elementwise adds
cd $GRAFT_REPO_ROOT 2>/dev/null || true
run() {
  python -c "
import sys, runpy, torch
from strajnet_amd import ops
if '$1' == 'separate':
    orig = ops.upconv_add
    def ua(x, pw, pb, r1, r2=None, prep=None):
        y = ops.upconv(x, pw, pb, prep=prep)
        y = y + r1.view(y.shape)
        return y if r2 is None else (y, y + r2.view(y.shape))
    ops.upconv_add = ua
sys.argv = ['bench.py', '--infer', '--no-cpu-baseline', '--no-extra-configs', '--no-kernel-timing', '--steps', '30', '--warmup', '5']
runpy.run_path('bench.py', run_name='__main__')
" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"
}
for i in 1 2 3; do run fused; run separate; done
