"""Experiment: one B=8 graph replay per step vs two concurrent B=4 replays (two model instances, two streams)."""
import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
import bench
from strajnet_amd import STrajNet, OGMFlow_loss, OccupancyFlowTaskConfig
from strajnet_amd.graph import GraphedTrainStep
dev = torch.device('cuda', 0)
def make(B, seed):
    m = STrajNet(bench.CFG256, fg_msa=True, fg=True, large_ogm=False, dtype=torch.bfloat16, device=dev, seed=0)
    lf = OGMFlow_loss(OccupancyFlowTaskConfig(256, 256, 8), replica=1.0, use_focal_loss=False, use_gt=True)
    return GraphedTrainStep(m, lf, bench.synth_batch(B, seed, dev))
g8 = make(8, 1)
for _ in range(3): g8()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): g8()
torch.cuda.synchronize(); t8 = (time.perf_counter() - t0) / 20
print(f'1 x B=8: {t8*1e3:.3f} ms/step  {8/t8:.1f} scenes/s', flush=True)
ga, gb = make(4, 2), make(4, 3)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def pair():
    with torch.cuda.stream(sa): ga()
    with torch.cuda.stream(sb): gb()
for _ in range(3): pair()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): pair()
torch.cuda.synchronize(); t4 = (time.perf_counter() - t0) / 20
print(f'2 x B=4 concurrent: {t4*1e3:.3f} ms/pair  {8/t4:.1f} scenes/s', flush=True)
g4 = ga
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): g4()
torch.cuda.synchronize(); t1 = (time.perf_counter() - t0) / 20
print(f'1 x B=4: {t1*1e3:.3f} ms/step  {4/t1:.1f} scenes/s', flush=True)
