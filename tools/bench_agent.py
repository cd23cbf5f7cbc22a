#!/usr/bin/env python
"""The agent branch alone (STrajNet._traj_net forward + backward, B scenes) in a loop, fused kernels or the layer-by-layer chain: run under
rocprofv3 (tools/prof_py.sh 40 tools/bench_agent.py [B] [fused 0|1] [dtype]) for the per-kernel durations.   Prints the eager loop time (host-bound for the chain)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from strajnet_amd import STrajNet, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
fused = (sys.argv[2] if len(sys.argv) > 2 else '1') != '0'
dtype = {'bf16': torch.bfloat16, 'f32': torch.float32, 'f16': torch.float16}[sys.argv[3] if len(sys.argv) > 3 else 'bf16']
m = STrajNet(bench.CFG256, fg_msa=True, fg=True, large_ogm=False, dtype=dtype, device='cuda:0', seed=0)
m.fused_agent = fused
x = bench.synth_batch(B, 1, 'cuda:0')
G = torch.randn(B, 64, 384, device='cuda:0').to(dtype)
m.dropctx.begin()
def step(train=True):
    m.zero_grad(); ops.use_arena(m._arena); m._sync_compute_weights(); m._agent_pack_stale = True
    m._dctx = None
    if train:
        m.dropctx.n, m.dropctx.sites = 0, {}
        m._dctx = m.dropctx
    key, cmi = m._traj_net(x['obs'], x['occ'])
    if train:
        key.backward(G)
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): step()
e1.record(); torch.cuda.synchronize()
print(f'B={B} fused={fused} {dtype}: eager fwd+bwd loop {e0.elapsed_time(e1) / 20 * 1e3:.0f} us per step (incl. zero_grad / cast / pack)')
with torch.no_grad():
    for _ in range(10): step(False)
torch.cuda.synchronize()
