"""Why is swin_mlp_fwd slower inside the model than in the tight micro-benchmark loop?  Times single launches under different cache /
neighbour conditions."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strajnet_amd import ops
from tools.bench_swin import mk

dt = torch.bfloat16
def once(fn, pre=None, n=10):
    ts = []
    for _ in range(n):
        if pre is not None:
            pre()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

big = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')
for B, res, C in ((8, 64, 96), (8, 32, 192)):
    N = res * res
    pg, pb = mk((C,), dt), mk((C,), dt)
    pw1, pb1, pw2, pb2 = mk((C, 4 * C), dt), mk((4 * C,), dt), mk((4 * C, C), dt), mk((C,), dt)
    x = torch.randn(B, N, C, device='cuda').to(dt)
    def f():
        with torch.no_grad():
            ops.swin_mlp(x, pg, pb, pw1, pb1, pw2, pb2, 1e-5, rows_per_sample=N)
    for _ in range(3): f()
    print(C, 'single launch, warm caches      ', once(f))
    print(C, 'single launch after 512MB memset', once(f, lambda: big.zero_()))
    def f5():
        for _ in range(5): f()
    print(C, '5 back-to-back / 5              ', once(f5) / 5)
