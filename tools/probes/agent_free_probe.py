#!/usr/bin/env python
"""Upper bound of what the agent branch still costs the step: bench.py with STrajNet._traj_net returning a cached (detached) encoding after its
first call, so the captured step contains no agent forward / backward at all.   usage: tools/probes/agent_free_probe.py [bench args]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import strajnet_amd
cls = strajnet_amd.STrajNet
orig = cls._traj_net
cache = {}


def cached(self, obs, occ):
    if 'kv' not in cache:
        k, m = orig(self, obs, occ)
        cache['kv'] = (k.detach().clone(), m.clone())
    return cache['kv']


cls._traj_net = cached
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
