#!/usr/bin/env python
"""Which torch (non-HIP-library) ops does one eager train step still launch, and from where?
Runs one B=8 bf16 step under torch.profiler and prints aten ops that own GPU time, grouped by python stack."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from strajnet_amd import STrajNet, OGMFlow_loss, OccupancyFlowTaskConfig, get_pred_waypoint_logits, warpped_gt

dev = torch.device('cuda', 0)
model = STrajNet(bench.CFG256, fg_msa=True, fg=True, large_ogm=False, dtype=torch.bfloat16, device=dev, seed=0)
loss_fn = OGMFlow_loss(OccupancyFlowTaskConfig(256, 256, 8), replica=1.0, use_focal_loss=False, use_gt=True)
x = bench.synth_batch(8, 1234, dev)


def step():
    model.zero_grad()
    out = model(x['ogm'], x['map_img'], training=True, obs=x['obs'], occ=x['occ'], mapt=x['mapt'], flow=x['flow'])
    d = loss_fn(get_pred_waypoint_logits(out), warpped_gt(x['gt_obs'], x['gt_occ'], x['gt_flow'], x['origin_flow']), None)
    (d['observed_xe'] + d['occluded_xe'] + d['flow'] + d['flow_warp_xe']).backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
from collections import defaultdict
agg = defaultdict(lambda: [0.0, 0])
for e in prof.events():
    if 'emcpy' in e.name or 'emset' in e.name:
        pass
    elif not e.name.startswith('aten::') or e.self_device_time_total <= 0:
        continue
    chain, q = [], e.cpu_parent
    while q is not None and len(chain) < 5:
        if not q.name.startswith('aten::'):
            chain.append(q.name.replace('autograd::engine::evaluate_function: ', 'bwd:'))
        else:
            chain.append(q.name)
        q = q.cpu_parent
    q, src = e, ''
    while q is not None and not src:
        for fr in (q.stack or []):
            if 'strajnet_amd/' in fr:
                src = fr.split('strajnet_amd/')[-1].strip()
                break
        q = q.cpu_parent
    key = (e.name, ' <- '.join(chain[:2]) + '  @ ' + src)
    agg[key][0] += e.self_device_time_total
    agg[key][1] += 1
tot = sum(v[0] for v in agg.values())
print(f'aten ops with GPU time: {tot:.0f} us in one step')
for (name, chain), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:80]:
    print(f'{t:8.1f} us  x{n:3d}  {name:16s} {chain}')
