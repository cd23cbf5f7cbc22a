#!/usr/bin/env python
"""Run bench.py with attributes of the model overridden after construction -- for same-box A/B runs of a switchable path.
usage: tools/ab_attr.py fused_mlp_dims="(96,192)" fused_fgattn=False ops.SKIP_JUNCTION=False -- --no-cpu-baseline --no-extra-configs --no-kernel-timing
(name=value: attribute of the model; ops.NAME=value: module constant of strajnet_amd.ops)"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
cut = args.index('--') if '--' in args else len(args)
sets = dict(a.split('=', 1) for a in args[:cut])
import strajnet_amd
from strajnet_amd import ops
for n in [k for k in sets if k.startswith('ops.')]:
    assert hasattr(ops, n[4:]), n
    setattr(ops, n[4:], eval(sets.pop(n)))
cls = strajnet_amd.STrajNet
init = cls.__init__


def patched(self, *a, **k):
    init(self, *a, **k)
    for n, v in sets.items():
        assert hasattr(self, n), n
        setattr(self, n, eval(v))


cls.__init__ = patched
sys.argv = [os.path.join(ROOT, 'bench.py')] + args[cut + 1:]
runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
