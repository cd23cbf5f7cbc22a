import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from strajnet_amd import STrajNet, OGMFlow_loss, OccupancyFlowTaskConfig
from strajnet_amd.graph import GraphedTrainStep
dev = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model = STrajNet(bench.CFG256, fg_msa=True, fg=True, large_ogm=False, dtype=torch.bfloat16, device=dev, seed=0)
loss_fn = OGMFlow_loss(OccupancyFlowTaskConfig(256, 256, 8), replica=1.0, use_focal_loss=False, use_gt=True)
x = bench.synth_batch(B, 1234, dev)
g = GraphedTrainStep(model, loss_fn, x)
print('capture losses', g.losses)
e = g._eager(); torch.cuda.synchronize()
print('eager losses', e, 'grad norm', float(model.flat_grads().norm()))
for i in range(3):
    l = g(); torch.cuda.synchronize()
    print('replay', i, l.tolist(), 'grad norm', float(model.flat_grads().norm()), 'nan grads', int(torch.isnan(model.flat_grads()).sum()))
print('---- weight corruption check')
model2 = STrajNet(bench.CFG256, fg_msa=True, fg=True, large_ogm=False, dtype=torch.bfloat16, device=dev, seed=0)
w0 = model2._flat.clone()
diff = (model._flat - w0)
bad = torch.nonzero(diff != 0).flatten()
nanw = torch.isnan(model._flat).sum()
print('changed weights', bad.numel(), 'nan weights', int(nanw))
if bad.numel():
    lo, hi = int(bad.min()), int(bad.max())
    print('range', lo, hi)
    for n, o in model._offs.items():
        k = 1
        for s_ in model.params[n].shape: k *= s_
        if o <= lo < o + k or o <= hi < o + k:
            print('  in', n, o, k)
