#!/usr/bin/env python
"""Element-wise passes of the decoder's skip level alone (stj_unary_bwd, stj_elu_res_bwd, stj_skip_junction_bwd on [64,64,64,128] 16-bit tensors,
67 MB each) as hipGraph replays: us per launch and TB/s of the bytes each moves.   python tools/bench_ew.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from strajnet_amd import ops
from strajnet_amd.ops import call, _p, _st
dt = torch.bfloat16
shape = (64, 64, 64, 128)
n = 64 * 64 * 64 * 128
t = [torch.randn(shape, device='cuda').to(dt) for _ in range(8)]
mb = n * 2 / 1e6
cases = {
    'unary_bwd (ELU)      2R 1W': (3, lambda: call('stj_unary_bwd', _p(t[0]), _p(t[1]), _p(t[2]), n, 2, 0.0, 1, _st())),
    'elu_res_bwd 2 grads  3R 2W': (5, lambda: call('stj_elu_res_bwd', _p(t[0]), _p(t[1]), _p(t[2]), None, _p(t[3]), _p(t[4]), n, 1, _st())),
    'skip_junction 2 skip 5R 3W': (8, lambda: call('stj_skip_junction_bwd', _p(t[0]), _p(t[1]), _p(t[2]), _p(t[3]), _p(t[4]), _p(t[5]), _p(t[6]), _p(t[7]), n, 1, _st())),
    'torch add            2R 1W': (3, lambda: torch.add(t[0], t[1], out=t[2])),
}
for name, (passes, fn) in cases.items():
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f'{name}: {us:7.1f} us  {passes * mb / us / 1e6 * 1e6 / 1e6:5.2f} TB/s')
