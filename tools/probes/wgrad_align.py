#!/usr/bin/env python
"""Does the address alignment of the atomically accumulated folded-tap buffer (dweff) change the time of stj_upconv_wgrad?
Round 5, review item 1: the four decoder weight-gradient shapes, alone, with dweff at several offsets inside one large allocation.
usage: python tools/probes/wgrad_align.py [--iters N]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from strajnet_amd.ops import _p, _st, call

LAYERS = [(64, 16, 384, 192), (64, 32, 192, 128), (64, 64, 128, 96), (64, 128, 96, 48)]
ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=10)
a = ap.parse_args()
dtype, dt = torch.bfloat16, 1
big = torch.zeros(16 << 20, dtype=torch.float32, device='cuda')
base = 13277788 + 4000        # roughly where the model's gradient buffer ends
for F, Hi, Cin, Cout in LAYERS:
    x = torch.randn(F, Hi, Hi, Cin, device='cuda').to(dtype)
    dp = torch.randn(F, 2 * Hi, 2 * Hi, Cout, device='cuda').to(dtype)
    dbp = torch.zeros(32, Cout, device='cuda')
    n = 16 * Cout * Cin
    row = []
    for off in (0, 8, 16, 24, 32, 64, 128, 256, 1024):
        o = (base + 1023) // 1024 * 1024 + off
        dweff = big[o:o + n]
        assert dweff.data_ptr() % 32 == 0
        for budget in (256, 128):
            fn = lambda: call('stj_upconv_wgrad', _p(x), _p(dp), _p(dweff), _p(dbp), 32, F, Hi, Hi, Cin, Cout, budget, dt, _st())
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            row.append(f'off {off * 4:5d} B wg {budget}: {e0.elapsed_time(e1) / a.iters * 1e3:7.1f} us')
    print(f'[{Hi}x{Hi},{Cin}->{Cout}]'); print('   ' + '\n   '.join(row))
