#!/usr/bin/env python
"""Micro-benchmark of the fused cross-attention kernels (stj_xattn_fwd / stj_xattn_bwd) alone on the GPU.   usage: tools/bench_xattn.py [B] [dtype]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from strajnet_amd import STrajNet, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dtype = {'bf16': torch.bfloat16, 'f32': torch.float32, 'f16': torch.float16}[sys.argv[2] if len(sys.argv) > 2 else 'bf16']
m = STrajNet(bench.CFG256, fg_msa=True, fg=True, large_ogm=False, dtype=dtype, device='cuda:0', seed=0)
dev = m.device
query = torch.randn(8, B, 256, 384, device=dev).to(dtype)
key = torch.randn(B, 64, 384, device=dev).to(dtype)
tmask = torch.ones(B, 64, device=dev, dtype=torch.int32)
G = torch.randn(8, B, 256, 384, device=dev).to(dtype)
m._sync_compute_weights(); m._pack_xattn()
zs = m._zstride
def proj(x, suffix):
    p0 = m._zp(suffix)
    return ops.linear_heads_in_z(x, p0.master, p0.c, p0.grad, zs, 8, True)
k, v = proj(key, 'mha/key_kernel'), proj(key, 'mha/value_kernel')
ps = m._xattn_params()
m.dropctx.begin()
for training in (False, True):
    drop = None
    if training:
        m.dropctx.n, m.dropctx.sites = 0, {}
        drop = (0.1, m.dropctx.snap, (m.dropctx.site('a', (8, B, 3, 256, 64), .1), m.dropctx.site('b', (8, B * 256, 512), .1), m.dropctx.site('c', (8, B * 256, 384), .1)))
    q = query.clone().requires_grad_(True)
    def fwd():
        return ops._XAttn.apply(q, k.detach(), v.detach(), ps['wq'].master, tmask, m._xattn_pack, ps, zs, drop)
    y = fwd()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for _ in range(3): fwd()
    torch.cuda.synchronize()
    e[0].record()
    for _ in range(20): y = fwd()
    e[1].record()
    torch.cuda.synchronize()
    tf = e[0].elapsed_time(e[1]) / 20 * 1e3
    # backward: time the C-ABI call only (not the weight-gradient GEMMs): wrap ops.call
    times = []
    raw = ops.call
    def timed(name, *a):
        if name == 'stj_xattn_bwd':
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record(); raw(name, *a); a1.record(); times.append((a0, a1))
        else:
            raw(name, *a)
    ops.call = timed
    for _ in range(8):
        y = fwd(); y.backward(G)
    torch.cuda.synchronize()
    ops.call = raw
    tb = sum(a.elapsed_time(b) for a, b in times[2:]) / len(times[2:]) * 1e3
    print(f'B={B} {dtype} training={training}: xattn_fwd {tf:.1f} us (with saves)   xattn_bwd + dkv reduce {tb:.1f} us')
with torch.no_grad():
    for _ in range(3): ops._XAttn.apply(query, k, v, ps['wq'].master, tmask, m._xattn_pack, ps, zs, None)
    torch.cuda.synchronize()
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record()
    for _ in range(20): ops._XAttn.apply(query, k, v, ps['wq'].master, tmask, m._xattn_pack, ps, zs, None)
    a1.record(); torch.cuda.synchronize()
    print(f'B={B} inference forward {a0.elapsed_time(a1) / 20 * 1e3:.1f} us')
