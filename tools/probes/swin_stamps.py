#!/usr/bin/env python
"""Where the time of swin_mlp_fwd goes: cycle stamps of wave 0 of every workgroup (variant library built with -DSTJ_STAMP:
tools/build_variant.sh stamp swin_fused.hip "-DSTJ_STAMP").   usage: STJ_LIB_PATH=strajnet_amd/variants/lib_stamp.so python tools/probes/swin_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from strajnet_amd import ops, _lib
from strajnet_amd.ops import _p, _st, call

L = ctypes.CDLL(os.environ['STJ_LIB_PATH'])
dt, dc, dev = torch.bfloat16, 1, 'cuda'
def r(*s, scale=1.0, d=dt): return (torch.randn(*s, device=dev) * scale).to(d)
for B, res, C in ((8, 64, 96), (8, 32, 192), (8, 16, 384)):
    N = res * res; M = B * N
    x = r(B, N, C); g, b = r(C, d=torch.float32), r(C, d=torch.float32)
    w1, w2 = r(C, 4 * C, scale=0.05), r(4 * C, C, scale=0.05)
    b1, b2 = r(4 * C, d=torch.float32), r(C, d=torch.float32)
    y = torch.empty_like(x)
    ws = ops._swin_ws(x, M, C)
    for it in range(3):
        if it == 2:
            torch.cuda.synchronize(); assert L.stj_dbg_clear() == 0
        call('stj_swin_mlp_fwd', _p(x), _p(g), _p(b), _p(w1), _p(b1), _p(w2), _p(b2), _p(y), M, C, 1e-5, None, 0, 0.0, N, dc, _p(ws), _st())
    torch.cuda.synchronize()
    buf = np.zeros(8 * 2048, dtype=np.uint64)
    assert L.stj_dbg_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    s = buf.reshape(2048, 8).astype(np.int64)
    nb = int((s[:, 1] != 0).sum())
    s = s[:nb]
    t0 = (s[:, 0] - s[:, 0].min()) * 10.0 / 1e3           # memrealtime: 100 MHz -> us
    d = lambda a, b_: (s[:, a] - s[:, b_]) / 1e3
    fin = s[:, 5] != 0
    tb = np.zeros(8 * 2048, dtype=np.uint64); assert L.stj_dbg_ticks(tb.ctypes.data_as(ctypes.c_void_p)) == 0
    tk = tb.reshape(2048, 8).astype(np.int64)[:nb, :5] / 1e3
    print('   chunk loop, kcycles summed over chunks (median over workgroups): loop-top', np.median(tk[:, 0]).round(2), ' wait barrier 1', np.median(tk[:, 1]).round(2), ' commit', np.median(tk[:, 2]).round(2), ' wait barrier 2', np.median(tk[:, 3]).round(2), ' k-steps', np.median(tk[:, 4]).round(2))
    if not fin.any(): continue
    print(f'M={M} C={C}: {nb} workgroups; start spread {t0.max():.2f} us (median {np.median(t0):.2f}); kcycles median: prologue(rows+LN) {np.median(d(2, 1)):.2f}  '
          f'first chunk ready {np.median(d(6, 1)):.2f}  loop end {np.median(d(3, 1)):.2f}  combine {np.median(d(4, 3)[fin]):.2f}  epilogue {np.median(d(5, 4)[fin]):.2f}  '
          f'total {np.median(d(5, 1)[fin]):.2f} (max {d(5, 1)[fin].max():.2f}); finishing workgroups {int(fin.sum())}')
