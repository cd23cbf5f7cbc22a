import numpy as np, torch, sys
sys.path.insert(0, '.')
from strajnet_amd import OGMFlow_loss, OccupancyFlowTaskConfig, get_pred_waypoint_logits, warpped_gt
from oracle import np_ref, torch_ref
cfg = dict(input_size=(128, 128), window_size=8, embed_dim=96, depths=[2, 2, 2], num_heads=[3, 6, 12])
x = np_ref.make_inputs(cfg, 2)
Hg = x['gt_obs'].shape[2]
logits = np.random.default_rng(0).normal(0, 2, (2, Hg, Hg, 32)).astype(np.float32)
gt = {k: torch.as_tensor(x[k]).cuda() for k in ('gt_obs', 'gt_occ', 'gt_flow', 'origin_flow')}
for flags in [dict(use_focal_loss=False), dict(use_focal_loss=True), dict(use_focal_loss=False, use_pred=True), dict(use_focal_loss=True, use_pred=True)]:
    lt = torch.as_tensor(logits).cuda().requires_grad_(True)
    d = OGMFlow_loss(OccupancyFlowTaskConfig(Hg, Hg, 8), replica=2.0, use_gt=False, **flags)(get_pred_waypoint_logits(lt), warpped_gt(gt['gt_obs'], gt['gt_occ'], gt['gt_flow'], gt['origin_flow']), None)
    lr = torch.as_tensor(logits).double().requires_grad_(True)
    gtr = {k: torch.as_tensor(x[k]).double() for k in gt}
    dr = torch_ref.loss(lr, gtr['gt_obs'], gtr['gt_occ'], gtr['gt_flow'], gtr['origin_flow'], replica=2.0, use_gt=False, **flags)
    print(flags, {k: (float(d[k]), float(dr[k])) for k in dr})
    for name in dr:
        lt.grad = None; lr.grad = None
        d[name].backward(retain_graph=True); dr[name].backward(retain_graph=True)
        a, r = lt.grad.double().cpu().view(-1, 8, 4), lr.grad.view(-1, 8, 4)
        for c, cn in enumerate(('obs', 'occ', 'fx', 'fy')):
            den = float(r[..., c].abs().max())
            if den > 0 or float(a[..., c].abs().max()) > 0:
                e = (a[..., c] - r[..., c]).abs()
                i = int(e.argmax())
                print(f'   {name:13s} d/d{cn}: max|ref| {den:.3e} max err {float(e.max()):.3e} at {i}: got {float(a[..., c].reshape(-1)[i]):.4e} ref {float(r[..., c].reshape(-1)[i]):.4e}')
