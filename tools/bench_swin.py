"""Micro-benchmark: one Swin block (or just its MLP half) at the model's stage shapes, fused kernels vs the layer-by-layer path.
    python tools/bench_swin.py            (on the GPU box)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strajnet_amd import ops                      # noqa: E402
from strajnet_amd.ops import Param                # noqa: E402


def mk(shape, dt, scale=0.1):
    m = (torch.randn(shape, device='cuda') * scale).requires_grad_(True)
    g = torch.zeros_like(m)
    m.grad = g
    return Param('p', shape, m, m.detach().to(dt), g)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def attn_half(dt):
    for B, res, C, shift in ((8, 64, 96, 0), (8, 64, 96, 4), (8, 32, 192, 4), (8, 128, 96, 4)):
        N = res * res
        heads = C // 32
        pg, pb = mk((C,), dt), mk((C,), dt)
        pwq, pbq, pt, pwp, pbp = mk((C, 3 * C), dt), mk((3 * C,), dt), mk((225, heads), dt), mk((C, C), dt), mk((C,), dt)
        x = torch.randn(B, N, C, device='cuda').to(dt).requires_grad_(True)
        g = torch.randn(B, N, C, device='cuda').to(dt)

        def fused(xx):
            return ops.swin_attn_half(xx, pg, pb, pwq, pbq, pt, pwp, pbp, B, res, shift, 1e-5)

        def unfused(xx):
            h, sk = ops.layernorm_skip(xx, pg, pb, 1e-5)
            a = ops.win_attn(ops.linear(h, pwq, pbq), pt, B, res, heads, shift)
            return ops.linear(a, pwp, pbp, res=sk)

        def run(fn, grad):
            def f():
                if grad:
                    x.grad = None
                    fn(x).backward(g)
                else:
                    with torch.no_grad():
                        fn(x)
            return f
        print(f'B={B} {res}x{res} C={C} shift={shift}: attention half  fwd fused {timeit(run(fused, False)):7.1f} us  layers {timeit(run(unfused, False)):7.1f} us | '
              f'fwd+bwd fused {timeit(run(fused, True)):7.1f} us  layers {timeit(run(unfused, True)):7.1f} us', flush=True)


def main():
    dt = torch.bfloat16
    attn_half(dt)
    for B, res, C in ((8, 64, 96), (8, 32, 192), (8, 16, 384), (8, 128, 96), (32, 64, 96)):
        N = res * res
        pg, pb = mk((C,), dt), mk((C,), dt)
        pw1, pb1, pw2, pb2 = mk((C, 4 * C), dt), mk((4 * C,), dt), mk((4 * C, C), dt), mk((C,), dt)
        x = torch.randn(B, N, C, device='cuda').to(dt).requires_grad_(True)
        g = torch.randn(B, N, C, device='cuda').to(dt)

        def fused_fwd():
            with torch.no_grad():
                return ops.swin_mlp(x, pg, pb, pw1, pb1, pw2, pb2, 1e-5, rows_per_sample=N)

        def unfused(xx):
            h, sk = ops.layernorm_skip(xx, pg, pb, 1e-5)
            h = ops.gelu(ops.linear(h, pw1, pb1))
            return ops.linear(h, pw2, pb2, res=sk)

        def unfused_fwd():
            with torch.no_grad():
                return unfused(x)

        def fused_fb():
            x.grad = None
            ops.swin_mlp(x, pg, pb, pw1, pb1, pw2, pb2, 1e-5, rows_per_sample=N).backward(g)

        def unfused_fb():
            x.grad = None
            unfused(x).backward(g)
        print(f'B={B} {res}x{res} C={C} (M={B * N}): MLP half  fwd fused {timeit(fused_fwd):7.1f} us  layers {timeit(unfused_fwd):7.1f} us | '
              f'fwd+bwd fused {timeit(fused_fb):7.1f} us  layers {timeit(unfused_fb):7.1f} us', flush=True)


if __name__ == '__main__':
    main()
