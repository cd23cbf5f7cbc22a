import os, sys
sys.path.insert(0, '/root/repo')
import torch
from strajnet_amd import ops
dt = torch.bfloat16
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M, K, N in [(32768, 96, 384), (32768, 384, 96), (8192, 192, 576)]:
    x = torch.randn(M, K, device='cuda').to(dt); dy = torch.randn(M, N, device='cuda').to(dt)
    gw = torch.zeros(K, N, device='cuda'); gb = torch.zeros(N, device='cuda')
    g = timeit(lambda: ops.gemm(x, dy, gw, K, N, M, (0, 0, 1, K), (0, 0, N, 1), (0, 0, N), 1, c_f32=1, accumulate=1, splitk=0, colsum=gb))
    print(f'wgrad M={M} K={K} N={N}: {g:7.1f} us', flush=True)
