import os, sys
sys.path.insert(0, '/root/repo')
import torch
from strajnet_amd import ops
dt = torch.bfloat16
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M, K, N in [(65536, 1536, 192), (262144, 768, 128), (16384, 1536, 192)]:
    x = torch.randn(M, K, device='cuda').to(dt); wT = (torch.randn(N, K, device='cuda') * 0.1).to(dt)
    y = torch.empty(M, N, device='cuda', dtype=dt)
    for cfg in ('0', '1', '2'):
        os.environ['STJ_GEMM_CFG'] = cfg
        f = timeit(lambda: ops.gemm(x, wT, y, M, N, K, (0, 0, K, 1), (0, 0, 1, K), (0, 0, N), 1))
        print(f'M={M} K={K} N={N} NT cfg{cfg}: {f:8.1f} us  {2*M*K*N/f/1e6:7.1f} TF/s', flush=True)
