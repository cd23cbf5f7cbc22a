#!/usr/bin/env python
"""Micro-benchmark of the batched GEMM at the hot-path Dense shapes (fwd / dgrad / wgrad), bf16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from strajnet_amd import ops
SHAPES = [(2048, 384, 384), (2048, 384, 1536), (2048, 1536, 384), (16384, 384, 512), (16384, 512, 384), (512, 384, 384), (8192, 192, 384), (32768, 96, 288), (32768, 96, 96), (32768, 96, 384), (32768, 384, 96), (8192, 192, 576), (8192, 768, 192), (2048, 384, 1152),
          (2048, 1536, 384), (16384, 384, 126), (16384, 128, 512)]
dt = torch.bfloat16
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M, K, N in SHAPES:
    x = torch.randn(M, K, device='cuda').to(dt); w = (torch.randn(K, N, device='cuda') * 0.1).to(dt)
    b = torch.randn(N, device='cuda'); dy = torch.randn(M, N, device='cuda').to(dt)
    y = torch.empty(M, N, device='cuda', dtype=dt); dx = torch.empty_like(x); gw = torch.zeros(K, N, device='cuda'); gb = torch.zeros(N, device='cuda')
    f = timeit(lambda: ops.gemm(x, w, y, M, N, K, (0, 0, K, 1), (0, 0, N, 1), (0, 0, N), 1, bias=b))
    d = timeit(lambda: ops.gemm(dy, w, dx, M, K, N, (0, 0, N, 1), (0, 0, 1, N), (0, 0, K), 1))
    g = timeit(lambda: ops.gemm(x, dy, gw, K, N, M, (0, 0, 1, K), (0, 0, N, 1), (0, 0, N), 1, c_f32=1, accumulate=1, splitk=0, colsum=gb))
    byts = (M * K + M * N) * 2
    print(f'M={M:6d} K={K:5d} N={N:5d}  fwd {f:7.1f} us  dgrad {d:7.1f} us  wgrad {g:7.1f} us   (stream floor {byts/5e6:6.1f} us, {2*M*K*N/1e9:6.1f} GF)')
