#!/usr/bin/env python
"""Micro-benchmark of the decoder conv kernels (fwd / dgrad / wgrad) at the cfg-256 B=8 layer shapes.
usage: python tools/bench_conv.py [--iters N] [--only fwd|dgrad|wgrad] [--layer i] [--dtype bf16|f32]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from strajnet_amd import ops
from strajnet_amd.ops import _p, _st, call

LAYERS = [(64, 16, 384, 192), (64, 32, 192, 128), (64, 64, 128, 96), (64, 128, 96, 48)]
ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--only', default='')
ap.add_argument('--layer', type=int, default=-1)
ap.add_argument('--dtype', default='bf16')
a = ap.parse_args()
dtype = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
dt = 1 if a.dtype == 'bf16' else 0
for li, (F, Hi, Cin, Cout) in enumerate(LAYERS):
    if a.layer >= 0 and li != a.layer:
        continue
    x = torch.randn(F, Hi, Hi, Cin, device='cuda').to(dtype)
    w = torch.randn(3, 3, Cin, Cout, device='cuda') * 0.05
    b = torch.randn(Cout, device='cuda') * 0.1
    wf = torch.empty(16, Cout, Cin, device='cuda', dtype=dtype)
    wd = torch.empty(16, Cin, Cout, device='cuda', dtype=dtype)
    call('stj_upconv_prep', _p(w), _p(wf), _p(wd), Cin, Cout, dt, _st())
    y = torch.empty(F, 2 * Hi, 2 * Hi, Cout, device='cuda', dtype=dtype)
    dp = torch.randn(F, 2 * Hi, 2 * Hi, Cout, device='cuda').to(dtype)
    dx = torch.empty_like(x)
    dweff = torch.zeros(16, Cout, Cin, device='cuda')
    NP = int(os.environ.get('NP', '32'))
    dbp = torch.zeros(NP, Cout, device='cuda')
    flops = 2.0 * 9 * Cin * Cout * 4 * Hi * Hi * F
    def run(name, fn):
        if a.only and a.only != name:
            return
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        print(f'{name:6s} [{Hi}x{Hi},{Cin}->{Cout}] {ms*1e3:9.1f} us  algorithmic {flops/ms/1e9:8.1f} TFLOP/s  executed(folded) {flops/2.25/ms/1e9:8.1f} TFLOP/s')
    run('fwd', lambda: call('stj_upconv_fwd', _p(x), _p(wf), _p(b), _p(y), F, Hi, Hi, Cin, Cout, 2, dt, _st()))
    run('dgrad', lambda: call('stj_upconv_dgrad', _p(dp), _p(wd), _p(dx), None, F, Hi, Hi, Cin, Cout, dt, _st()))
    run('dgradE', lambda: call('stj_upconv_dgrad', _p(dp), _p(wd), _p(dx), _p(x), F, Hi, Hi, Cin, Cout, dt, _st()))
    run('wgrad', lambda: call('stj_upconv_wgrad', _p(x), _p(dp), _p(dweff), _p(None if os.environ.get('NODB') else dbp), NP, F, Hi, Hi, Cin, Cout, 256, dt, _st()))
