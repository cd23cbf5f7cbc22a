#!/usr/bin/env python
"""Micro-benchmark of LayerNorm fwd/bwd at the Swin token shapes (B=8 cfg-256).  usage: python tools/bench_ln.py [--iters N]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from strajnet_amd.ops import _p, _st, call

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=20)
a = ap.parse_args()
for rows, C in [(32768, 96), (8192, 192), (2048, 384), (16384, 384), (16384, 128)]:
    x = torch.randn(rows, C, device='cuda').bfloat16()
    dy = torch.randn(rows, C, device='cuda').bfloat16()
    g, b = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    y, dx = torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(rows, device='cuda'), torch.empty(rows, device='cuda')
    NP = int(os.environ.get('NP', '32'))
    dg, db = torch.zeros(NP * C, device='cuda'), torch.zeros(NP * C, device='cuda')
    def run(name, fn, nbytes):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        print(f'{name} [{rows}x{C}] {ms*1e3:8.1f} us  {nbytes/ms/1e9:6.2f} TB/s', flush=True)
    run('ln_fwd', lambda: call('stj_layernorm_fwd', _p(x), _p(g), _p(b), _p(y), _p(mean), _p(rstd), rows, C, 1e-5, 0, 0, 0, 1, 0, 1, _st()), rows * C * 4)
    run('ln_bwd', lambda: call('stj_layernorm_bwd', _p(dy), _p(x), _p(g), _p(mean), _p(rstd), _p(dx), _p(dg), _p(db), rows, C, 0, 0, 0, 1, 0, None, NP, C, 1, _st()), rows * C * 6)
