#!/usr/bin/env python
"""Where a tile's time goes in upconv_fwd_ws5_kernel: per-phase cycles of compute wave 0 and mover wave 0 (variant library:
tools/build_variant.sh ws5_stamp conv_ws5.hip "-DSTJ_STAMP").  usage: STJ_AB_WS5=1 STJ_LIB_PATH=strajnet_amd/variants/lib_ws5_stamp.so python tools/probes/ws5_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from strajnet_amd.ops import _p, _st, call
L = ctypes.CDLL(os.environ['STJ_LIB_PATH'])
F, Hi, Cin, Cout = 64, 128, 96, 48
dtype = torch.bfloat16
x = torch.randn(F, Hi, Hi, Cin, device='cuda').to(dtype)
w = torch.randn(3, 3, Cin, Cout, device='cuda') * 0.05
b = torch.randn(Cout, device='cuda') * 0.1
wf = torch.empty(16, Cout, Cin, device='cuda', dtype=dtype); wd = torch.empty(16, Cin, Cout, device='cuda', dtype=dtype)
call('stj_upconv_prep', _p(w), _p(wf), _p(wd), Cin, Cout, 1, _st())
y = torch.empty(F, 2 * Hi, 2 * Hi, Cout, device='cuda', dtype=dtype)
for _ in range(3):
    call('stj_upconv_fwd', _p(x), _p(wf), _p(b), _p(y), F, Hi, Hi, Cin, Cout, 2, 1, _st())
torch.cuda.synchronize()
buf = np.zeros(256 * 16, dtype=np.uint64)
assert L.stj_dbg_ws5_ticks(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(256, 16).astype(np.float64) / 64.0          # per tile (64 tiles per workgroup)
m = np.median(t, axis=0)
print('compute wave, ticks per tile (median over workgroups): poll %.0f | MFMAs %.0f | epilogue %.0f | signal %.0f | sum %.0f' % (m[0], m[1], m[4], m[5], m[:6].sum()))
print('mover wave,   ticks per tile: poll %.0f | unit reads + vmcnt wait %.0f | signal %.0f | DMA issue %.0f | ELU + stores %.0f | sum %.0f' % (m[8], m[9], m[10], m[11], m[13], m[8:14].sum()))
