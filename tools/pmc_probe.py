#!/usr/bin/env python
"""Workload for the rocprofv3 --pmc passes: one launch of each decoder conv kernel at the largest layer (96->48 @128x128,
F=64, bf16) preceded by a CALIBRATION copy of known size (256 MiB read + 256 MiB written by torch's vectorised copy kernel)
so that FETCH_SIZE / WRITE_SIZE units and the gfx950 half-count can be checked in the same pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from strajnet_amd.ops import _p, _st, call

cal = torch.empty(128 * 1024 * 1024, dtype=torch.bfloat16, device='cuda').normal_()
dst = torch.empty_like(cal)
F, Hi, Cin, Cout = 64, 128, 96, 48
x = torch.randn(F, Hi, Hi, Cin, device='cuda').bfloat16()
w = torch.randn(3, 3, Cin, Cout, device='cuda') * 0.05
b = torch.randn(Cout, device='cuda') * 0.1
wf = torch.empty(16, Cout, Cin, device='cuda', dtype=torch.bfloat16)
wd = torch.empty(16, Cin, Cout, device='cuda', dtype=torch.bfloat16)
call('stj_upconv_prep', _p(w), _p(wf), _p(wd), Cin, Cout, 1, _st())
y = torch.empty(F, 2 * Hi, 2 * Hi, Cout, device='cuda', dtype=torch.bfloat16)
dp = torch.randn(F, 2 * Hi, 2 * Hi, Cout, device='cuda').bfloat16()
dx = torch.empty_like(x)
dweff = torch.zeros(16, Cout, Cin, device='cuda')
db = torch.zeros(Cout, device='cuda')
torch.cuda.synchronize()
for _ in range(2):
    dst.copy_(cal)                                                     # calibration: 268,435,456 B read, same written
    call('stj_upconv_fwd', _p(x), _p(wf), _p(b), _p(y), F, Hi, Hi, Cin, Cout, 2, 1, _st())
    call('stj_upconv_dgrad', _p(dp), _p(wd), _p(dx), _p(x), F, Hi, Hi, Cin, Cout, 1, _st())
    call('stj_upconv_wgrad', _p(x), _p(dp), _p(dweff), _p(db), 1, F, Hi, Hi, Cin, Cout, 256, 1, _st())
torch.cuda.synchronize()
print('algorithmic bytes: fwd', x.numel() * 2 + y.numel() * 2, 'dgrad(+Xelu)', dp.numel() * 2 + 2 * x.numel() * 2, 'wgrad', x.numel() * 2 + dp.numel() * 2)
