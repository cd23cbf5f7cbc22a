#!/bin/bash
# A/B on one box: bench step with the fused FG-MSA attention kernel and with the layer-by-layer path, alternating
cd $GRAFT_REPO_ROOT 2>/dev/null || true
run() {
  timeout 120 python -c "
import sys, runpy
from strajnet_amd import ops
if '$1' == 'off': ops.fg_attn_ok = lambda *a: False
sys.argv = ['bench.py', '--no-cpu-baseline', '--no-extra-configs', '--no-kernel-timing']
runpy.run_path('bench.py', run_name='__main__')
" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"
}
for i in 1 2 3; do run on; run off; done
