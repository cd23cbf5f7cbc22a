import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from strajnet_amd import STrajNet, OGMFlow_loss, OccupancyFlowTaskConfig, get_pred_waypoint_logits, warpped_gt, ops
dev = torch.device('cuda', 0)
B = 2
model = STrajNet(bench.CFG256, fg_msa=True, fg=True, large_ogm=False, dtype=torch.bfloat16, device=dev, seed=0)
x = bench.synth_batch(B, 1234, dev)
def fwd():
    with torch.no_grad():
        out = model(x['ogm'], x['map_img'], training=True, obs=x['obs'], occ=x['occ'], mapt=x['mapt'], flow=x['flow'])
        gate, auc = ops.auc_gate(x['gt_obs'], x['gt_occ'], x['gt_flow'], x['origin_flow'], return_auc=True)
    return out, gate, auc
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): fwd()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out, gate, auc = fwd()
for i in range(3):
    g.replay(); torch.cuda.synchronize()
    o = out.view(-1, 8, 4)
    print(i, 'nan per channel', [int(torch.isnan(o[..., c]).sum()) for c in range(4)], 'gate', gate.tolist(), 'auc', [round(v, 4) for v in auc.tolist()])
