#!/usr/bin/env python
"""Every (C-ABI kernel, shape) of one serial B=8 bf16 train step with its HIP-event time: `python tools/prof_keys.py [substr]`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from strajnet_amd import STrajNet, OGMFlow_loss, OccupancyFlowTaskConfig, get_pred_waypoint_logits, warpped_gt, prof

dev = torch.device('cuda', 0)
model = STrajNet(bench.CFG256, fg_msa=True, fg=True, large_ogm=False, dtype=torch.bfloat16, device=dev, seed=0)
model.serial = True
loss_fn = OGMFlow_loss(OccupancyFlowTaskConfig(256, 256, 8), replica=1.0, use_focal_loss=False, use_gt=True)
x = bench.synth_batch(8, 1234, dev)


def step():
    model.zero_grad()
    out = model(x['ogm'], x['map_img'], training=True, obs=x['obs'], occ=x['occ'], mapt=x['mapt'], flow=x['flow'])
    d = loss_fn(get_pred_waypoint_logits(out), warpped_gt(x['gt_obs'], x['gt_occ'], x['gt_flow'], x['origin_flow']), None)
    d.total.backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
prof.enable()
N = 5
for _ in range(N):
    step()
torch.cuda.synchronize()
keys, fams = prof.summarize(prof.disable(), N)
pat = sys.argv[1] if len(sys.argv) > 1 else ''
tot = 0.0
for k, v in sorted(keys.items(), key=lambda kv: -kv[1]['ms_per_step']):
    if pat in k:
        tot += v['ms_per_step']
        print(f"{v['ms_per_step'] * 1e3:8.1f} us/step  x{v['launches']:4.1f}  avg {v['avg_ms'] * 1e3:7.1f} us  {k}")
print(f'total {tot * 1e3:.0f} us/step')
