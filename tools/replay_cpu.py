#!/usr/bin/env python
"""Host-side cost of one hipGraph replay of the train step (is the replay enqueue-bound?): wall time of graph.replay() itself,
of replay + synchronize, and of N back-to-back replays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from strajnet_amd import STrajNet, OGMFlow_loss, OccupancyFlowTaskConfig, Nadam
from strajnet_amd.graph import GraphedTrainStep

dev = torch.device('cuda', 0)
model = STrajNet(bench.CFG256, fg_msa=True, fg=True, large_ogm=False, dtype=torch.bfloat16, device=dev, seed=0)
loss_fn = OGMFlow_loss(OccupancyFlowTaskConfig(256, 256, 8), replica=1.0, use_focal_loss=False, use_gt=True)
x = bench.synth_batch(8, 1234, dev)
step = GraphedTrainStep(model, loss_fn, x)
for _ in range(3):
    step()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print('replay() host call: %.2f ms   replay + sync: %.2f ms' % (1e3 * min(a for a, _ in ts), 1e3 * min(b for _, b in ts)))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('20 back-to-back: host %.2f ms/step, total %.2f ms/step' % (1e3 * (t1 - t0) / 20, 1e3 * (t2 - t0) / 20))
