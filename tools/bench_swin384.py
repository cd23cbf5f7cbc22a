"""Micro-benchmark of the C = 384 Swin stage (16 x 16 tokens per scene, B = 8: 2048 rows, 32 windows): fused split kernels vs layer by layer.
Run under rocprofv3 --kernel-trace for the per-kernel durations (warm caches: the weights stay in L2 / MALL between iterations).
    python tools/bench_swin384.py [B] [res]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strajnet_amd import ops                      # noqa: E402
from bench_swin import mk, timeit                 # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
res = int(sys.argv[2]) if len(sys.argv) > 2 else 16
C, dt = 384, torch.bfloat16
N, heads = res * res, C // 32
pg, pb = mk((C,), dt), mk((C,), dt)
pwq, pbq, pt, pwp, pbp = mk((C, 3 * C), dt), mk((3 * C,), dt), mk((225, heads), dt), mk((C, C), dt), mk((C,), dt)
pw1, pb1, pw2, pb2 = mk((C, 4 * C), dt), mk((4 * C,), dt), mk((4 * C, C), dt), mk((C,), dt)
x = torch.randn(B, N, C, device='cuda').to(dt).requires_grad_(True)
g = torch.randn(B, N, C, device='cuda').to(dt)


def fused(xx):
    y = ops.swin_attn_half(xx, pg, pb, pwq, pbq, pt, pwp, pbp, B, res, 4, 1e-5)
    return ops.swin_mlp(y, pg, pb, pw1, pb1, pw2, pb2, 1e-5, rows_per_sample=N)


def layers(xx):
    h, sk = ops.layernorm_skip(xx, pg, pb, 1e-5)
    a = ops.win_attn(ops.linear(h, pwq, pbq), pt, B, res, heads, 4)
    y = ops.linear(a, pwp, pbp, res=sk)
    h, sk = ops.layernorm_skip(y, pg, pb, 1e-5)
    h = ops.gelu(ops.linear(h, pw1, pb1))
    return ops.linear(h, pw2, pb2, res=sk)


def run(fn, grad):
    def f():
        if grad:
            x.grad = None
            fn(x).backward(g)
        else:
            with torch.no_grad():
                fn(x)
    return f


print(f'B={B} {res}x{res} C=384 block: fwd fused {timeit(run(fused, False)):7.1f} us  layers {timeit(run(layers, False)):7.1f} us | '
      f'fwd+bwd fused {timeit(run(fused, True)):7.1f} us  layers {timeit(run(layers, True)):7.1f} us', flush=True)
