import sys, time, torch
sys.path.insert(0, '.')
import bench
from strajnet_amd import STrajNet, OGMFlow_loss, OccupancyFlowTaskConfig
from strajnet_amd.graph import GraphedTrainStep
dev = torch.device('cuda', 0)
model = STrajNet(bench.CFG256, fg_msa=True, fg=True, large_ogm=False, dtype=torch.bfloat16, device=dev, seed=0)
loss_fn = OGMFlow_loss(OccupancyFlowTaskConfig(256, 256, 8), ogm_weight=1000.0, occ_weight=1000.0, replica=1.0, flow_origin_weight=1000.0, use_focal_loss=False, use_gt=True)
x = bench.synth_batch(8, 1234, dev)
g = GraphedTrainStep(model, loss_fn, x)
for _ in range(5): g()
torch.cuda.synchronize()
hs = []
t0 = time.perf_counter()
for _ in range(20):
    a = time.perf_counter(); g(); hs.append(time.perf_counter() - a)
host_total = time.perf_counter() - t0
torch.cuda.synchronize()
total = time.perf_counter() - t0
print('host time per replay() call: median %.3f ms, first %.3f ms; 20 replays host %.2f ms, until sync %.2f ms (%.3f ms/step)' % (sorted(hs)[10] * 1e3, hs[0] * 1e3, host_total * 1e3, total * 1e3, total / 20 * 1e3))
# single replay latency: sync before and after
ls = []
for _ in range(10):
    torch.cuda.synchronize(); a = time.perf_counter(); g(); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
    ls.append((b - a, c - a))
print('isolated replay: host call %.3f ms, call+sync %.3f ms' % (sorted(l[0] for l in ls)[5] * 1e3, sorted(l[1] for l in ls)[5] * 1e3))
