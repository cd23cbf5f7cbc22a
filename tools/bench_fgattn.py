#!/usr/bin/env python
"""Micro-benchmark of the fused FG-MSA attention kernels (stj_fg_attn_fwd / stj_fg_attn_bwd) against the layer-by-layer path, warm, alone on
the GPU.   usage: tools/bench_fgattn.py [B] [Hh] [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from strajnet_amd import ops
from test_ops_gpu import mk_param, rnd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Hh = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dt = {'bf16': torch.bfloat16, 'f16': torch.float16}[sys.argv[3] if len(sys.argv) > 3 else 'bf16']
G, gc = 8, 48
HW, C = Hh * Hh, G * gc
q, k, v = (rnd((B, HW, C), dt, 10 + i).requires_grad_(True) for i in range(3))
off = rnd((B, G, HW, 2), dt, 2, 2.0).requires_grad_(True)
go = rnd((B, HW, C), dt, 5)
pt = mk_param((2 * Hh - 1, 2 * Hh - 1, G), dt, 0.5, 1)


def timeit(f, n=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, fn in (('fused', lambda: ops.fg_attn(q, k, v, off, pt, Hh, Hh, gc ** -0.5)),
                 ('layerwise', lambda: ops.mha_core(q, k, v, G, gc, gc ** -0.5, fg_off=off, fg=(pt, Hh, Hh)))):
    tf = timeit(fn)
    def both():
        fn().backward(go)
    tfb = timeit(both)
    print(f'B={B} {Hh}x{Hh} {dt} {name}: fwd {tf:.1f} us   fwd+bwd {tfb:.1f} us  (host-launch bound below ~100 us)')
