"""bench.input_feed (data.HostFeed) next to the resident-input step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from strajnet_amd import STrajNet, OGMFlow_loss, OccupancyFlowTaskConfig, Nadam
from strajnet_amd.graph import GraphedTrainStep
dev = torch.device('cuda:0')
model = STrajNet(bench.CFG256, fg_msa=True, fg=True, large_ogm=False, dtype=torch.bfloat16, device=dev, seed=0, dropout_seed=0)
loss_fn = OGMFlow_loss(OccupancyFlowTaskConfig(256, 256, 8), ogm_weight=1000.0, occ_weight=1000.0, flow_weight=1.0, replica=1.0,
                       flow_origin_weight=1000.0, no_use_warp=False, use_pred=False, use_focal_loss=False, use_gt=True)
x = bench.synth_batch(8, 1234, dev, 256)
opt = Nadam.for_model(model, lr=1e-4)
g0 = GraphedTrainStep(model, loss_fn, x)
def t(fn, n=40):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('resident %.3f ms' % t(lambda: (g0(), opt.step())))
r = bench.input_feed(g0, x, 40, opt.step, dev); print({k: v for k, v in r.items() if k != 'note'})
print('resident %.3f ms' % t(lambda: (g0(), opt.step())))
