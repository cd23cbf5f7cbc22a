#!/usr/bin/env python
"""torch (non-stj) device launches of one eager INFERENCE forward (B = 32, fp16), by op and shape, with the Python frames that issued them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from strajnet_amd import STrajNet

dev = torch.device('cuda:0')
model = STrajNet(bench.CFG256, fg_msa=True, fg=True, large_ogm=False, dtype=torch.float16, device=dev, seed=0)
x = bench.synth_batch(32, 1234, dev, 256)


def step():
    with torch.no_grad():
        return model(x['ogm'], x['map_img'], training=False, obs=x['obs'], occ=x['occ'], mapt=None, flow=x['flow'])


for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = {}
for e in prof.events():
    if e.device_type.name != 'CPU' or not e.name.startswith('aten::'):
        continue
    kern = [k for k in e.kernels] if hasattr(e, 'kernels') else []
    if not kern:
        continue
    stack = [s for s in (e.stack or []) if 'strajnet_amd' in s][:3]
    key = (e.name, str(e.input_shapes)[:90], ' <- '.join(s.split('/')[-1][:60] for s in stack))
    rows.setdefault(key, [0, 0.0])
    rows[key][0] += 1
    rows[key][1] += sum(k.duration for k in kern)
tot = 0
for (name, shapes, stack), (n, us) in sorted(rows.items(), key=lambda r: -r[1][0]):
    tot += n
    print(f'{n:3d} x {name:22s} {us:8.1f} us  {shapes}  {stack}')
print('total torch device launches per forward:', tot)
