#!/usr/bin/env python
"""Micro-benchmark of the stem: fused PatchEmbed launches (csrc/patch_embed.hip) vs im2col + dense + LayerNorm launches.
usage: python tools/bench_patch.py [--batch 8] [--iters 50] [--dtype bf16]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from strajnet_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--iters', type=int, default=50)
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--size', type=int, default=256)
a = ap.parse_args()
dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[a.dtype]


def mk(shape, scale=0.1):
    m = (torch.randn(shape) * scale).cuda().requires_grad_(True)
    m.grad = torch.zeros_like(m)
    return ops.Param('p', shape, m, m.detach().to(dt), m.grad)


def timed(name, fn, nbytes):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f'{name:46s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e9:6.2f} TB/s', flush=True)


B, H = a.batch, a.size
M = B * (H // 4) ** 2
for nm, Cin, cs, nch in (('ogm', 11, 2, 22), ('map', 3, 1, 3), ('flow', 2, 1, 2)):
    src = torch.randn(B, H, H, nch, device='cuda')
    w, b, g, be, g2, be2 = mk((4, 4, Cin, 96)), mk((96,)), mk((96,)), mk((96,)), mk((96,)), mk((96,))
    add = torch.randn(M, 96, device='cuda').to(dt)
    es = 2 if dt != torch.float32 else 4
    nb = src.numel() * 4 + M * 96 * es
    with torch.no_grad():
        timed(f'{nm}: fused fwd (inference)', lambda: ops.patch_embed(src, w, b, g, be, Cin, cs, nch, dt), nb)
        timed(f'{nm}: fused fwd + add + LN2 (inference)', lambda: ops.patch_embed(src, w, b, g, be, Cin, cs, nch, dt, 1e-5, add, g2, be2), nb + M * 96 * es)
        timed(f'{nm}: im2col + dense + LN', lambda: ops.layernorm(ops.linear(ops.patch_im2col(src, Cin, cs, nch, dt), w, b), g, be, 1e-5), nb)
        timed(f'{nm}: im2col + dense + LN(+add) + LN', lambda: ops.layernorm(ops.layernorm(ops.linear(ops.patch_im2col(src, Cin, cs, nch, dt), w, b), g, be, 1e-5, res=add), g2, be2, 1e-5), nb + M * 96 * es)
    trig = w.master
    timed(f'{nm}: fused fwd (training: + cols, pre, x2)', lambda: ops.patch_embed(src, w, b, g, be, Cin, cs, nch, dt, 1e-5, add, g2, be2), nb + M * (16 * Cin + 3 * 96) * es)
