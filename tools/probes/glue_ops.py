#!/usr/bin/env python
"""Which torch (non-stj) device kernels does one eager train step launch, and on what shapes?  torch.profiler with record_shapes:
every aten op that launched a GPU kernel, grouped by (op, input shapes), forward and backward separately."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from strajnet_amd import STrajNet, OGMFlow_loss, Nadam
from strajnet_amd.loss import OccupancyFlowTaskConfig, get_pred_waypoint_logits, warpped_gt

dev = torch.device('cuda:0')
model = STrajNet(bench.CFG256, fg_msa=True, fg=True, large_ogm=False, dtype=torch.bfloat16, device=dev, seed=0)
loss_fn = OGMFlow_loss(OccupancyFlowTaskConfig(256, 256, 8), ogm_weight=1000.0, occ_weight=1000.0, flow_weight=1.0, replica=1.0,
                       flow_origin_weight=1000.0, no_use_warp=False, use_pred=False, use_focal_loss=False, use_gt=True)
x = bench.synth_batch(8, 1234, dev, 256)
opt = Nadam.for_model(model, lr=1e-4)


def step():
    model.zero_grad()
    out = model(x['ogm'], x['map_img'], training=True, obs=x['obs'], occ=x['occ'], mapt=x['mapt'], flow=x['flow'])
    d = loss_fn(get_pred_waypoint_logits(out), warpped_gt(x['gt_obs'], x['gt_occ'], x['gt_flow'], x['origin_flow']), None)
    d.total.backward()
    opt.step()


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
    # leaf aten ops that own a device kernel
    kern = [k for k in e.kernels] if hasattr(e, 'kernels') else []
    if not kern:
        continue
    stack = [s for s in (e.stack or []) if 'strajnet_amd' in s or 'bench' in s][:2]
    key = (e.name, str(e.input_shapes), ' <- '.join(s.split('/')[-1] for s in stack))
    rows.setdefault(key, [0, 0.0])
    rows[key][0] += 1
    rows[key][1] += sum(k.duration for k in kern)
tot = 0
for (name, shapes, stack), (n, us) in sorted(rows.items(), key=lambda r: -r[1][0]):
    tot += n
    print(f'{n:3d} x {name:28s} {us:8.1f} us  {shapes[:110]}  {stack}')
print('total torch device launches per step:', tot)
