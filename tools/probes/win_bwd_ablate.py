import os, sys, torch
sys.path.insert(0, '.')
from strajnet_amd.ops import _p, _st, call
B, res, heads = 8, 64, 3
C = heads * 32
qkv = torch.randn(B, res * res, 3 * C, device='cuda').bfloat16()
do = torch.randn(B, res * res, C, device='cuda').bfloat16()
tbl = torch.randn(225, heads, device='cuda')
NP = int(os.environ.get('NP', '1'))
dqkv = torch.empty_like(qkv); dt = torch.zeros(NP, 225, heads, device='cuda')
def f(): call('stj_win_attn_bwd', _p(qkv), _p(tbl), _p(do), _p(dqkv), _p(dt), NP, B, res, heads, 4, 1, _st())
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): f()
g.replay(); torch.cuda.synchronize()
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print('nparts', NP, f'{e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch')
