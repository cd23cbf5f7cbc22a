#!/usr/bin/env python
"""The loss alone at the bench shape (B = 8, 256 x 256, the train.py:195-196 flags): stj_loss_fwd + stj_loss_bwd (two passes and the finalize
launch between them) against stj_loss_fwd_bwd (one pass; coefficients from stj_loss_coef), as hipGraph replays.   python tools/bench_loss.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from strajnet_amd import ops
from strajnet_amd.ops import call, _p, _st
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = W = 256
g = torch.Generator(device='cuda').manual_seed(0)
logits = torch.randn(B, H, W, 32, device='cuda', generator=g) * 2
gt_obs = (torch.rand(B, 8, H, W, 1, device='cuda', generator=g) < 0.05).float()
gt_occ = (torch.rand(B, 8, H, W, 1, device='cuda', generator=g) < 0.02).float()
gt_flow = torch.randn(B, 8, H, W, 2, device='cuda', generator=g) * (torch.rand(B, 8, H, W, 1, device='cuda', generator=g) < 0.05).float()
origin = (torch.rand(B, 8, H, W, 1, device='cuda', generator=g) < 0.05).float()
gate = ops.auc_gate(gt_obs, gt_occ, gt_flow, origin)
w = (1000.0, 1000.0, 1000.0, 1.0)
coef = ops.loss_coef(gt_flow, gate, *w, 1)
sums, sums2 = torch.zeros(32 * 40, device='cuda'), torch.zeros(128 * 40, device='cuda')
loss, coef2, up = torch.empty(5, device='cuda'), torch.empty(32, device='cuda'), torch.ones((), device='cuda')
d1, d2 = torch.empty_like(logits), torch.empty_like(logits)
mb = (logits.numel() * 4 * 2 + (gt_obs.numel() * 2 + gt_flow.numel()) * 4 + origin.numel() * 4) / 1e6      # fused: logits in, dlogits out, ground truth once


def two():
    call('stj_loss_fwd', _p(logits), _p(gt_obs), _p(gt_occ), _p(gt_flow), _p(origin), _p(gate), _p(sums), _p(loss), _p(coef2), B, H, W, *w, 1, _st())
    call('stj_loss_bwd', _p(logits), _p(gt_obs), _p(gt_occ), _p(gt_flow), _p(origin), _p(coef2), _p(up), _p(d1), B, H, W, 1 | 8, _st())


def one():
    call('stj_loss_fwd_bwd', _p(logits), _p(gt_obs), _p(gt_occ), _p(gt_flow), _p(origin), _p(gate), _p(coef), _p(sums2), _p(loss), _p(coef2),
         _p(d2), B, H, W, *w, 1, _st())


def prep():
    ops.loss_coef(gt_flow, gate, *w, 1)


two(); one(); torch.cuda.synchronize()          # (compared here: the replays below accumulate into scratch that must be zero on entry)
diff = f'd/dlogits max |one pass - two passes|: {float((d1 - d2).abs().max()):.3e} of {float(d1.abs().max()):.3e}'
for name, fn in (('fwd + finalize + bwd', two), ('fwd_bwd + finalize', one), ('coef (count + coef)', prep)):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f'{name}: {us:7.1f} us' + (f'  ({mb:.0f} MB: {mb / us:.2f} TB/s)' if fn is one else ''))
print(diff)
