#!/usr/bin/env python
"""One split-K weight-gradient GEMM shape in a loop (for rocprofv3 --pmc): dW[96,384] += x[K,96]^T dy[K,384], K = 32768."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from strajnet_amd import ops
dev = torch.device('cuda', 0)
Kf, N, M = int(os.environ.get('KF', 96)), int(os.environ.get('N', 384)), int(os.environ.get('M', 32768))
x = torch.randn(M, Kf, device=dev).bfloat16()
dy = torch.randn(M, N, device=dev).bfloat16()
gw = torch.zeros(Kf, N, device=dev)
def run():
    ops.gemm(x, dy, gw, Kf, N, M, (0, 0, 1, Kf), (0, 0, N, 1), (0, 0, N), 1, c_f32=1, accumulate=1, splitk=0)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print(f'{Kf}x{N}x{M}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us')
