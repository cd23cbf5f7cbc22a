#!/usr/bin/env python
"""Is the host's packet submission of the replayed step graph on the GPU's critical path?  Replay the captured train step with a spin kernel
(torch.cuda._sleep) captured at its head: while the GPU spins, the host finishes submitting the whole graph, so the step then runs with every
packet already queued.  step_with_sleep - sleep_alone vs the plain step = what submission order / latency costs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from strajnet_amd import STrajNet, OGMFlow_loss, Nadam
from strajnet_amd.loss import OccupancyFlowTaskConfig
from strajnet_amd.graph import GraphedTrainStep

CYC = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000


class Sleepy(GraphedTrainStep):
    spin = 0

    def _eager(self):
        if Sleepy.spin:
            torch.cuda._sleep(Sleepy.spin)
        return super()._eager()


def timeit(f, n=20):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


dev = torch.device('cuda:0')
x = bench.synth_batch(8, 1234, dev, 256)
res = {}
for spin in (0, CYC):
    Sleepy.spin = spin
    model = STrajNet(bench.CFG256, fg_msa=True, fg=True, large_ogm=False, dtype=torch.bfloat16, device=dev, seed=0)
    loss_fn = OGMFlow_loss(OccupancyFlowTaskConfig(256, 256, 8), ogm_weight=1000.0, occ_weight=1000.0, flow_weight=1.0, replica=1.0,
                           flow_origin_weight=1000.0, no_use_warp=False, use_pred=False, use_focal_loss=False, use_gt=True)
    opt = Nadam.for_model(model, lr=1e-4)
    g = Sleepy(model, loss_fn, x)

    def step():
        g()
        opt.step()
    res[spin] = timeit(step)
    del g, model, opt
gs = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    torch.cuda._sleep(CYC)
    torch.cuda.synchronize()
    with torch.cuda.graph(gs):
        torch.cuda._sleep(CYC)
sl = timeit(gs.replay)
print(f'plain step {res[0]:.3f} ms | with a {sl:.3f} ms spin at its head {res[CYC]:.3f} ms -> step with everything queued {res[CYC] - sl:.3f} ms')
