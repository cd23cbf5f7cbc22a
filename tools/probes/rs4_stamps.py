#!/usr/bin/env python
"""per-phase ticks of wave 0 in upconv_wgrad_rs4_kernel (variant: tools/build_variant.sh rs4_stamp conv_wrs4.hip "-DSTJ_STAMP")"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from strajnet_amd.ops import _p, _st, call
L = ctypes.CDLL(os.environ['STJ_LIB_PATH'])
F, Hi, Cin, Cout = 64, 128, 96, 48
x = torch.randn(F, Hi, Hi, Cin, device='cuda').bfloat16(); dp = torch.randn(F, 2 * Hi, 2 * Hi, Cout, device='cuda').bfloat16()
dweff = torch.zeros(16, Cout, Cin, device='cuda'); dbp = torch.zeros(32, Cout, device='cuda')
for _ in range(3):
    call('stj_upconv_wgrad', _p(x), _p(dp), _p(dweff), _p(dbp), 32, F, Hi, Hi, Cin, Cout, 256, 1, _st())
torch.cuda.synchronize()
buf = np.zeros(256 * 8, dtype=np.uint64)
assert L.stj_dbg_rs4_ticks(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(256, 8).astype(np.float64)
q = t[:, 4]
m = np.median(t[:, :4] / q[:, None], axis=0)
print('units per workgroup %.0f; ticks per unit (median): wait + barrier %.0f | DMA issue %.0f | reads + MFMAs %.0f | loop tail %.0f | sum %.0f' % (np.median(q), m[0], m[1], m[2], m[3], m.sum()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    call('stj_upconv_wgrad', _p(x), _p(dp), _p(dweff), _p(dbp), 32, F, Hi, Hi, Cin, Cout, 256, 1, _st())
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
tot = np.median(t[:, :4].sum(1))
print('kernel %.1f us; loop ticks per workgroup %.0f  ->  %.2f ticks per ns (s_memtime is NOT a cycle counter: tools/probes/clock_probe.hip)' % (us, tot, tot / us / 1e3))
