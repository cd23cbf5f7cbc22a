import sys, os, torch
sys.path.insert(0, '.')
import bench
from strajnet_amd import STrajNet, OGMFlow_loss, OccupancyFlowTaskConfig, get_pred_waypoint_logits, warpped_gt, ops, _lib
dev = torch.device('cuda', 0)
model = STrajNet(bench.CFG256, fg_msa=True, fg=True, large_ogm=False, dtype=torch.bfloat16, device=dev, seed=0)
loss_fn = OGMFlow_loss(OccupancyFlowTaskConfig(256, 256, 8), replica=1.0, use_focal_loss=False, use_gt=True)
x = bench.synth_batch(8, 1234, dev)
order = []
raw = ops._raw_call
def traced(name, *a):
    order.append((name, torch.cuda.current_stream().cuda_stream))
    return raw(name, *a)
ops._raw_call = traced
model.zero_grad()
out = model(x['ogm'], x['map_img'], training=True, obs=x['obs'], occ=x['occ'], mapt=x['mapt'], flow=x['flow'])
d = loss_fn(get_pred_waypoint_logits(out), warpped_gt(x['gt_obs'], x['gt_occ'], x['gt_flow'], x['origin_flow']), None)
n_fwd = len(order)
d.total.backward()
torch.cuda.synchronize()
b = order[n_fwd:]
streams = sorted(set(s for _, s in order))
sid = {s: i for i, s in enumerate(streams)}
print('backward launches:', len(b))
for i, (n, s) in enumerate(b):
    if any(k in n for k in ('xattn_bwd', 'fg_offset_bwd', 'agent_sum_bwd', 'agent_mix_bwd', 'maxpool_bwd', 'fg_bias_bwd', 'swin_mlp_bwd', 'upconv_wgrad', 'win_attn_bwd', 'col2im3')):
        print(i, n, 'stream', sid[s])
