#!/usr/bin/env python
"""Kernel-level micro-benchmark of the four fused Swin kernels through the C ABI (no autograd, buffers allocated once), at the stage
shapes of cfg-256 B=8 and cfg-512 B=8.   usage: python tools/bench_swin_k.py [--iters N] [--only mlp|attn] [--cfg512]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from strajnet_amd import ops
from strajnet_amd.ops import _p, _st, call

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--only', default='')
ap.add_argument('--cfg512', action='store_true')
a = ap.parse_args()
dt, dc = torch.bfloat16, 1
dev = 'cuda'


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e3


def r(*s, scale=1.0, d=dt):
    return (torch.randn(*s, device=dev) * scale).to(d)


shapes = [(8, 64, 96), (8, 32, 192), (8, 16, 384)] + ([(8, 128, 96), (8, 64, 192), (8, 32, 384)] if a.cfg512 else [])
for B, res, C in shapes:
    N = res * res; M = B * N; H = C // 32
    x, dy = r(B, N, C), r(B, N, C)
    g, b = r(C, d=torch.float32), r(C, d=torch.float32)
    ws0 = ops._swin_ws(x, M, C)
    for ws in ([ws0, None] if (C == 192 and ws0 is not None) else [ws0]):
        tag = 'split ' if (ws is not None and C == 192) else ''
        if a.only in ('', 'mlp'):
            w1, w2 = r(C, 4 * C, scale=0.05), r(4 * C, C, scale=0.05)
            b1, b2 = r(4 * C, d=torch.float32), r(C, d=torch.float32)
            y, dx = torch.empty_like(x), torch.empty_like(x)
            h, dpre = torch.empty(M, 4 * C, device=dev, dtype=dt), torch.empty(M, 4 * C, device=dev, dtype=dt)
            ln, dys = torch.empty_like(x), torch.empty_like(x)
            dg, db = torch.zeros(32 * C, device=dev), torch.zeros(32 * C, device=dev)
            tf = timeit(lambda: call('stj_swin_mlp_fwd', _p(x), _p(g), _p(b), _p(w1), _p(b1), _p(w2), _p(b2), _p(y), M, C, 1e-5, None, 0, 0.0, N, dc, _p(ws), _st()))
            tb = timeit(lambda: call('stj_swin_mlp_bwd', _p(x), _p(dy), _p(g), _p(b), _p(w1), _p(b1), _p(w2), _p(dx), _p(h), _p(dpre), _p(ln), _p(dys),
                                     _p(dg), _p(db), 32, C, M, C, 1e-5, None, 0, 0.0, N, dc, _p(ws), _st()))
            fl = 2.0 * M * C * 4 * C * 2
            print(f'M={M:6d} C={C} {tag}: mlp fwd {tf:7.1f} us ({fl / tf / 1e6:6.1f} TF/s)   bwd {tb:7.1f} us ({2 * fl / tb / 1e6:6.1f} TF/s)', flush=True)
        if a.only in ('', 'attn'):
            wq, wp = r(C, 3 * C, scale=0.05), r(C, C, scale=0.05)
            bq, bp, tbl = r(3 * C, d=torch.float32), r(C, d=torch.float32), r(225, H, d=torch.float32)
            y, dx = torch.empty_like(x), torch.empty_like(x)
            qkv, dqkv = torch.empty(B, N, 3 * C, device=dev, dtype=dt), torch.empty(B, N, 3 * C, device=dev, dtype=dt)
            att, ln, dys = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
            mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
            dtab = torch.zeros(16 * 225 * H, device=dev)
            dg, db = torch.zeros(32 * C, device=dev), torch.zeros(32 * C, device=dev)
            for shift in (0, 4):
                tf = timeit(lambda: call('stj_swin_attn_fwd', _p(x), _p(g), _p(b), _p(wq), _p(bq), _p(tbl), _p(wp), _p(bp), _p(y), _p(qkv), _p(att), _p(ln),
                                         _p(mean), _p(rstd), B, res, C, shift, 1e-5, None, 0, 0.0, dc, _p(ws), _st()))
                ti = timeit(lambda: call('stj_swin_attn_fwd', _p(x), _p(g), _p(b), _p(wq), _p(bq), _p(tbl), _p(wp), _p(bp), _p(y), None, None, None,
                                         None, None, B, res, C, shift, 1e-5, None, 0, 0.0, dc, _p(ws), _st()))
                tb = timeit(lambda: call('stj_swin_attn_bwd', _p(x), _p(dy), _p(qkv), _p(mean), _p(rstd), _p(g), _p(wq), _p(wp), _p(tbl), _p(dx), _p(dqkv),
                                         _p(dys), _p(dtab), 16, _p(dg), _p(db), 32, C, B, res, C, shift, None, 0, 0.0, dc, _p(ws), _st()))
                print(f'M={M:6d} C={C} {tag}shift={shift}: attn fwd(train) {tf:7.1f} us  fwd(infer) {ti:7.1f} us  bwd {tb:7.1f} us', flush=True)
