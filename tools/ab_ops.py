#!/usr/bin/env python
"""Run bench.py with module constants of strajnet_amd.ops overridden -- same-box A/B runs of a tuning constant.
usage: tools/ab_ops.py UPWG_BUDGET=192 -- --steps 40 --warmup 5 --no-cpu-baseline --no-extra-configs --no-kernel-timing"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
cut = args.index('--') if '--' in args else len(args)
from strajnet_amd import ops
for a in args[:cut]:
    n, v = a.split('=', 1)
    assert hasattr(ops, n), n
    setattr(ops, n, eval(v))
sys.argv = [os.path.join(ROOT, 'bench.py')] + args[cut + 1:]
runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
