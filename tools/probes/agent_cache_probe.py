#!/usr/bin/env python
"""Upper bound of what hiding the agent branch (trajNet) completely would buy: bench.py with STrajNet._traj_net answering from a cache
(detached outputs of its first call: no agent kernels in the forward OR the backward pass of the captured step).
usage: python tools/probes/agent_cache_probe.py [0|1] -- <bench args>"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
on = sys.argv[1] == '1'
args = sys.argv[sys.argv.index('--') + 1:] if '--' in sys.argv else []
import strajnet_amd
cls = strajnet_amd.STrajNet
orig = cls._traj_net
cache = {}


def patched(self, obs, occ):
    if not on:
        return orig(self, obs, occ)
    k = id(self)
    if k not in cache:
        cache[k] = tuple(t.detach() for t in orig(self, obs, occ))
    return cache[k]


cls._traj_net = patched
sys.argv = [os.path.join(ROOT, 'bench.py')] + args
runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
