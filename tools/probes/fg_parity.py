import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, 'tests')
import torch, numpy as np
from strajnet_amd import ops, modules
import test_model_gpu as T
from strajnet_amd import OGMFlow_loss, OccupancyFlowTaskConfig, get_pred_waypoint_logits, warpped_gt
from oracle import torch_ref

def old_fgmsa(self, x):
    B, Hh, Ww, C = x.shape
    G = 8; gc = C // G; HW = Hh * Ww
    q = self._dense(x, 'fg_msa/proj_q'); k = self._dense(x, 'fg_msa/proj_k'); v = self._dense(x, 'fg_msa/proj_v')
    o = ops.grouped_conv3(q, self._p('fg_msa/conv_offset_0/kernel'), self._p('fg_msa/conv_offset_0/bias'), G)
    o = ops.gelu(self._ln(o, 'fg_msa/conv_norm', 1e-3))
    o = o.view(B, HW, G, gc).permute(0, 2, 1, 3).contiguous()
    off = ops.tanh_scale(self._dense(o, 'fg_msa/conv_offset_proj', bias=False), Hh / 2.0)
    fh = self._dense(off, 'fg_msa/conv_offset_proj2')
    bias = ops.fg_bias(off, self._p('fg_msa/warp_attn_rel_table'), Hh, Ww)
    a = ops.mha_core(q.view(B, HW, C), k.view(B, HW, C), v.view(B, HW, C), G, gc, gc ** -0.5, bias=bias)
    y = self._dense(a.view(B, Hh, Ww, C), 'fg_msa/proj_out')
    xy = x + y
    query = xy.reshape(1, B, HW, C).expand(8, B, HW, C) + fh.reshape(B, 8, HW, C).permute(1, 0, 2, 3)
    return xy, query.contiguous()

new_fgmsa = modules.STrajNet._fgmsa
ref = None
for name, fn in (('old', old_fgmsa), ('new', new_fgmsa)):
    modules.STrajNet._fgmsa = fn
    model, w, x, xt = T._setup(T.CFG128, 2, torch.float32)
    model.zero_grad()
    out = T._fwd(model, xt)
    out.retain_grad()
    loss_fn = OGMFlow_loss(OccupancyFlowTaskConfig(128, 128, 8), replica=1.0, use_focal_loss=False, use_gt=True)
    d = loss_fn(get_pred_waypoint_logits(out), warpped_gt(xt['gt_obs'], xt['gt_occ'], xt['gt_flow'], xt['origin_flow']), None)
    sum(d.values()).backward()
    if ref is None:
        pr = torch_ref.to_torch(w, torch.float64, requires_grad=True)
        xr = torch_ref.to_torch(x, torch.float64)
        yr = torch_ref.forward(pr, T.CFG128, xr['ogm'], xr['map_img'], xr['obs'], xr['occ'], xr['flow'])
        yr.retain_grad()
        dr = torch_ref.loss(yr, xr['gt_obs'], xr['gt_occ'], xr['gt_flow'], xr['origin_flow'], replica=1.0, use_gt=True)
        sum(dr.values()).backward()
        ref = pr
    errs = []
    for n, p in model.params.items():
        g, gr = p.grad.double().cpu(), ref[n].grad
        diff = (g - gr).abs()
        scale = float(gr.abs().max()) + 1e-12
        e = float(diff.max()) / scale
        errs.append((e, n, int((diff > 0.1 * diff.max()).sum()), diff.numel()))
    errs.sort(reverse=True)
    dg = (out.grad.double().cpu() - yr.grad).abs()
    top = torch.topk(dg.flatten(), 6)
    print('  dlogits: max |grad| %.3e; top diffs' % float(yr.grad.abs().max()), [(float(v), tuple(int(i) for i in np.unravel_index(int(ix), dg.shape))) for v, ix in zip(top.values, top.indices)])
    print(name, 'fwd err', float((out.double().cpu() - yr.detach()).abs().max()))
    for e in errs[:6]:
        print('   %.3e %-40s n_big=%d of %d' % e)
