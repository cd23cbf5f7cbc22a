#!/usr/bin/env python
"""FG-MSA offset head alone: the fused kernels (csrc/fgoff_fused.hip) against the layer-by-layer chain, forward (inference) and forward +
backward, as hipGraph replays of ITER repetitions (device time, no host gaps).   python tools/bench_fgoff.py [B] [H] [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from strajnet_amd import ops
from test_ops_gpu import mk_param, rnd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dt = {'bf16': torch.bfloat16, 'f32': torch.float32, 'f16': torch.float16}[sys.argv[3] if len(sys.argv) > 3 else 'bf16']
G, GC, C, ITER = 8, 48, 384, 20
ps = [mk_param((3, 3, GC, C), dt, 0.06, 1), mk_param((C,), dt, 0.1, 2), mk_param((C,), dt, 0.2, 3), mk_param((C,), dt, 0.1, 4), mk_param((1, 1, GC, 2), dt, 0.12, 5)]
pw, pb, pg, pbe, p1 = ps
pack = ops.fgoff_pack(pw, dt)
q = rnd((B, H, H, C), dt, 7).requires_grad_(True)
go = rnd((B, G, H * H, 2), dt, 9)


def run(fused, train):
    if fused:
        off = ops.fgoff_chain(q, pw, pb, pg, pbe, p1, pack, H / 2.0, 1e-3)
    else:
        off = ops.fg_offset(ops.gelu(ops.layernorm(ops.grouped_conv3(q, pw, pb, G), pg, pbe, 1e-3)), p1, H / 2.0, G)
    if train:
        q.grad = None
        off.backward(go)


for train in (False, True):
    for fused in (True, False):
        def body():
            if train:
                run(fused, True)
            else:
                with torch.no_grad():
                    run(fused, False)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(ITER):
                body()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        print(f'B={B} H={H} {dt} {"fwd+bwd" if train else "fwd    "} {"fused    " if fused else "layerwise"}: {e0.elapsed_time(e1) / (5 * ITER) * 1e3:7.1f} us')
