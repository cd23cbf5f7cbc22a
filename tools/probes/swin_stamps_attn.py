#!/usr/bin/env python
"""Where the time of swin_attn_bwd goes (cycle stamps of wave 0 of every workgroup; variant library built with -DSTJ_STAMP).
usage: STJ_LIB_PATH=strajnet_amd/variants/lib_stamp.so python tools/probes/swin_stamps_attn.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from strajnet_amd import ops
from strajnet_amd.ops import _p, _st, call

L = ctypes.CDLL(os.environ['STJ_LIB_PATH'])
dt, dc, dev = torch.bfloat16, 1, 'cuda'
def r(*s, scale=1.0, d=dt): return (torch.randn(*s, device=dev) * scale).to(d)
for B, res, C in ((8, 64, 96), (8, 32, 192)):
    N = res * res; M = B * N; H = C // 32
    x, dy = r(B, N, C), r(B, N, C)
    g = r(C, d=torch.float32)
    wq, wp = r(C, 3 * C, scale=0.05), r(C, C, scale=0.05)
    tbl = r(225, H, d=torch.float32)
    dx = torch.empty_like(x)
    qkv, dqkv = r(B, N, 3 * C), torch.empty(B, N, 3 * C, device=dev, dtype=dt)
    dys = torch.empty_like(x)
    mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
    dtab = torch.zeros(16 * 225 * H, device=dev)
    dg, db = torch.zeros(32 * C, device=dev), torch.zeros(32 * C, device=dev)
    ws = ops._swin_ws(x, M, C)
    for it in range(3):
        if it == 2:
            torch.cuda.synchronize(); assert L.stj_dbg_clear() == 0
        call('stj_swin_attn_bwd', _p(x), _p(dy), _p(qkv), _p(mean), _p(rstd), _p(g), _p(wq), _p(wp), _p(tbl), _p(dx), _p(dqkv),
             _p(dys), _p(dtab), 16, _p(dg), _p(db), 32, C, B, res, C, 4, None, 0, 0.0, dc, _p(ws), _st())
    torch.cuda.synchronize()
    buf = np.zeros(8 * 2048, dtype=np.uint64); assert L.stj_dbg_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    s = buf.reshape(2048, 8).astype(np.int64)
    nb = int((s[:, 1] != 0).sum()); s = s[:nb]
    tb = np.zeros(8 * 2048, dtype=np.uint64); assert L.stj_dbg_ticks(tb.ctypes.data_as(ctypes.c_void_p)) == 0
    tk = tb.reshape(2048, 8).astype(np.int64)[:nb, :5] / 1e3
    t0 = (s[:, 0] - s[:, 0].min()) * 10.0 / 1e3
    d = lambda a, b_: (s[:, a] - s[:, b_]) / 1e3
    fin = s[:, 5] != 0
    md = np.median
    print(f'M={M} C={C}: {nb} workgroups stamped, start spread {t0.max():.2f} us (median {md(t0):.2f}); kcycles (median): dy rows + proj product {md(d(2, 1)):.2f}; '
          f'head loop: commit+barrier {md(tk[:, 1]):.2f}  attention of the heads {md(tk[:, 2]):.2f}  copy-out + dLN product {md(tk[:, 3]):.2f}  (between passes {md(tk[:, 0]):.2f}); '
          f'loop end at {md(d(3, 1)):.2f}; LN backward + tail {md(d(5, 3)[fin]) if fin.any() else float("nan"):.2f}; total {md(d(5, 1)[fin]) if fin.any() else float("nan"):.2f}')
