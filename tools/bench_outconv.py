#!/usr/bin/env python
"""Micro-benchmark of the output-head conv kernels (48->2, x2 heads) at the cfg-256 B=8 shape, t-major frames.
usage: python tools/bench_outconv.py [--iters N]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from strajnet_amd.ops import _p, _st, call

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--compact', type=int, default=0)
a = ap.parse_args()
B, Tn, H, C = 8, 8, 256, 48
F = B * Tn
x = torch.randn(F, H, H, C, device='cuda').bfloat16()
w = torch.randn(3, 3, C, 2, device='cuda') * 0.1
b = torch.randn(2, device='cuda')
if a.compact:
    y = torch.zeros(F, H, H, 2, device='cuda'); ybs, yts, yps = H * H * 2, B * H * H * 2, 2
else:
    y = torch.zeros(B, H, H, 4 * Tn, device='cuda'); ybs, yts, yps = H * H * 4 * Tn, 4, 4 * Tn
dy = torch.randn_like(y)
dx = torch.empty_like(x)
dw, db = torch.zeros_like(w), torch.zeros_like(b)
from strajnet_amd._lib import lib
ws = torch.empty(int(lib().stj_outconv_bwd_workspace_bytes()), dtype=torch.uint8, device='cuda')


def run(name, fn, nbytes):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f'{name:12s} {ms*1e3:8.1f} us   {nbytes/ms/1e9:6.2f} TB/s algorithmic', flush=True)


run('outconv_fwd', lambda: call('stj_outconv_fwd', _p(x), _p(w), _p(b), _p(y), F, H, H, C, Tn, ybs, yts, yps, 1, _st()), x.numel() * 2 + F * H * H * 8)
for e in (0, 1):
    run(f'outconv_bwd e{e}', lambda: call('stj_outconv_bwd', _p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), F, H, H, C, Tn, ybs, yts, yps, e, _p(ws), ws.numel(), 1, _st()),
        x.numel() * 4 + F * H * H * 8)
# both heads in one launch (the training forward's stj_outconv_pair_fwd): two branch tensors in, whole 128-byte output lines out
x2 = torch.randn(F, H, H, C, device='cuda').bfloat16()
w2 = torch.randn(3, 3, C, 2, device='cuda') * 0.1
yp = torch.zeros(B, H, H, 4 * Tn, device='cuda')
run('pair_fwd', lambda: call('stj_outconv_pair_fwd', _p(x), _p(x2), _p(w), _p(w2), _p(b), _p(b), _p(yp), B, Tn, H, H, C, 1, 1, _st()),
    2 * x.numel() * 2 + yp.numel() * 4)
