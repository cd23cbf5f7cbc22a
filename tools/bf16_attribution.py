#!/usr/bin/env python
"""Where does the bf16 mode's output error (max-abs ~0.1 on logits of scale ~10) come from?
Runs the eval forward of the bench batch in f32 mode (the parity mode) and in bf16 / fp16 mode with the same weights, taps the stage
boundaries (model.taps) and prints, per tap: rms of the f32 tensor, rms and max-abs of the difference, and the difference relative to
the rms.  Then the same with ONE stage at a time fed from the f32 run is not possible without mixed-precision storage; instead the
second table isolates each stage's own contribution by re-running the 16-bit model from the f32 tap cast to 16 bits (decoder only:
it is a pure function of the cross-attention output and the encoder skips)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from strajnet_amd import STrajNet

dev = torch.device('cuda', 0)
x = bench.synth_batch(8, 1234, dev)


def run(dtype):
    m = STrajNet(bench.CFG256, fg_msa=True, fg=True, large_ogm=False, dtype=dtype, device=dev, seed=0)
    m.serial = True
    m.taps = {}
    with torch.no_grad():
        m(x['ogm'], x['map_img'], training=False, obs=x['obs'], occ=x['occ'], mapt=x['mapt'], flow=x['flow'])
    torch.cuda.synchronize()
    return m.taps


ref = run(torch.float32)
for name, dt in (('bf16', torch.bfloat16), ('fp16', torch.float16)):
    t = run(dt)
    print(f'--- {name} vs f32 (same weights, eval forward, B=8 bench batch) ---')
    print(f'{"tap":28s} {"rms(f32)":>10s} {"rms err":>10s} {"max err":>10s} {"rms err / rms":>14s}')
    for k in ref:
        a, b = ref[k].double(), t[k].double()
        d = (a - b)
        r = float(a.pow(2).mean().sqrt())
        print(f'{k:28s} {r:10.4f} {float(d.pow(2).mean().sqrt()):10.5f} {float(d.abs().max()):10.5f} {float(d.pow(2).mean().sqrt()) / (r + 1e-30):14.5f}')
