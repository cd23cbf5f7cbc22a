"""Second, independently formulated restatement of the STrajNet hot path in PyTorch-CPU.

TEST INFRASTRUCTURE (see oracle/__init__.py): never imported by strajnet_amd.
PARITY UNPINNED by the reference (no TensorFlow here).

Purpose: (1) cross-check oracle/np_ref.py (float64, agreement ~1e-10) using different
primitives (F.conv2d NCHW, F.layer_norm, F.grid_sample, torch.roll, unfold-free window
indexing, bucketised AUC); (2) gradient oracle via torch autograd (the reference obtains
gradients from tf.GradientTape, train.py:217-223); (3) the reported, non-target CPU
baseline of bench.py ("CPU restatement (PyTorch), not the TensorFlow reference").

Reference lines restated are the same as in np_ref.py and cited there; this file cites
only where its formulation differs.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _geom(cfg, large_ogm):
    H = cfg['input_size'][0]
    P = H // 4
    C = cfg['embed_dim']
    res = [P, P // 2, P // 4]
    crop = 2 if large_ogm else 1
    return dict(H=H, P=P, C=C, res=res, dim=[C, 2 * C, 4 * C], skip=[r // crop for r in res],
                hb=res[2] // crop, ws=cfg['window_size'], heads=cfg['num_heads'], depths=cfg['depths'],
                large=large_ogm, map_size=H // 2 if large_ogm else H)


def _ln(x, p, name, eps):
    return F.layer_norm(x, (x.shape[-1],), p[name + '/gamma'], p[name + '/beta'], eps)


def _lin(x, p, name, bias=True):
    return F.linear(x, p[name + '/kernel'].t(), p[name + '/bias'] if bias else None)


def _conv_nhwc(x, w_hwio, b=None, stride=1, padding=0, groups=1):
    """NHWC in/out through F.conv2d (NCHW, OIHW)."""
    lead = x.shape[:-3]
    xx = x.reshape((-1,) + x.shape[-3:]).permute(0, 3, 1, 2)
    y = F.conv2d(xx, w_hwio.permute(3, 2, 0, 1), b, stride=stride, padding=padding, groups=groups)
    y = y.permute(0, 2, 3, 1)
    return y.reshape(lead + y.shape[1:])


def _win_index(res, ws, shift, device):
    """Token indices of each (shifted) window: idx[w, n] = flat index in the un-rolled map of the
    n-th token of window w.  Equivalent to roll(-shift) + window_partition (modules.py:230-239)."""
    r = torch.arange(res, device=device)
    src = (r + shift) % res                       # rolled coordinate -> source coordinate
    hh = src.view(res // ws, ws)
    idx = hh[:, None, :, None] * res + hh[None, :, None, :]     # [nh, nw, ws, ws]
    return idx.reshape(-1, ws * ws)


def _region_id(res, ws, shift, device):
    """Shift-mask region labels in rolled coordinates (modules.py:192-203)."""
    lab = torch.zeros(res, dtype=torch.long, device=device)
    lab[res - ws:res - shift] = 1
    lab[res - shift:] = 2
    return lab[:, None] * 3 + lab[None, :]


_MASKS, _DPR = None, {}


def _dropout(x, name, rate, index=None):
    """Keras Dropout with a supplied 0/1 keep mask (see np_ref.keras_dropout); identity when no masks are set."""
    if _MASKS is None or name not in _MASKS:
        return x
    m = _MASKS[name] if index is None else _MASKS[name][index]
    return x * (1.0 / (1.0 - rate)) * m.to(x.dtype).reshape(x.shape)


def _drop_path(x, name, prob):
    if _MASKS is None or name not in _MASKS or prob == 0.0:
        return x
    return x / (1.0 - prob) * _MASKS[name].to(x.dtype).view(-1, *([1] * (x.dim() - 1)))


def _swin_block(x, p, pre, res, heads, ws, shift):
    dp = _DPR.get(pre, 0.0)
    if res <= ws:
        shift, ws = 0, res
    B, L, C = x.shape
    hd = C // heads
    N = ws * ws
    h = _ln(x, p, pre + '/norm1', 1e-5)
    idx = _win_index(res, ws, shift, x.device)                  # [nW, N]
    nW = idx.shape[0]
    qkv = _lin(h, p, pre + '/attn/qkv')[:, idx]                 # [B, nW, N, 3C]
    qkv = qkv.view(B, nW, N, 3, heads, hd)
    q, k, v = qkv[..., 0, :, :], qkv[..., 1, :, :], qkv[..., 2, :, :]
    att = torch.einsum('bwnhd,bwmhd->bwhnm', q * hd ** -0.5, k)
    # relative position bias: index = (dy+ws-1)*(2ws-1) + (dx+ws-1)
    c = torch.arange(ws, device=x.device)
    yy, xx = torch.meshgrid(c, c, indexing='ij')
    yy, xx = yy.reshape(-1), xx.reshape(-1)
    ridx = (yy[:, None] - yy[None, :] + ws - 1) * (2 * ws - 1) + (xx[:, None] - xx[None, :] + ws - 1)
    rpb = p[pre + '/attn/relative_position_bias_table'][ridx]   # [N,N,h]
    att = att + rpb.permute(2, 0, 1)
    if shift > 0:
        lab = _region_id(res, ws, shift, x.device)              # labels live in rolled coords
        labw = lab.view(res // ws, ws, res // ws, ws).permute(0, 2, 1, 3).reshape(nW, N)
        m = (labw[:, :, None] != labw[:, None, :]).to(x.dtype) * -100.0
        att = att + m[None, :, None]
    att = att.softmax(-1)
    o = torch.einsum('bwhnm,bwmhd->bwnhd', att, v).reshape(B, nW * N, C)
    o = _lin(o, p, pre + '/attn/proj')
    out = torch.zeros_like(x).index_copy(1, idx.reshape(-1), o)
    x = x + _drop_path(out, pre + '/drop_path_attn', dp)
    h = _ln(x, p, pre + '/norm2', 1e-5)
    h = F.gelu(_lin(h, p, pre + '/mlp/fc1'), approximate='tanh')
    return x + _drop_path(_lin(h, p, pre + '/mlp/fc2'), pre + '/drop_path_mlp', dp)


def _merge(x, p, pre, res):
    B, L, C = x.shape
    x = x.view(B, res // 2, 2, res // 2, 2, C)          # [B, i, di, j, dj, C]
    x = x.permute(0, 1, 3, 4, 2, 5).reshape(B, (res // 2) ** 2, 4 * C)   # concat order (dj,di): x00,x10,x01,x11
    x = _ln(x, p, pre + '/downsample/norm', 1e-5)
    return _lin(x, p, pre + '/downsample/reduction', bias=False)


def _layer(x, p, pre, res, depth, heads, ws, down):
    for i in range(depth):
        x = _swin_block(x, p, f'{pre}/blocks{i}', res, heads, ws, 0 if i % 2 == 0 else ws // 2)
    return (_merge(x, p, pre, res), x) if down else (x, x)


def _patch_embed(x, p, name):
    y = _conv_nhwc(x, p[name + '/proj/kernel'], p[name + '/proj/bias'], stride=4)
    return _ln(y.reshape(y.shape[0], -1, y.shape[-1]), p, name + '/norm', 1e-5)


def _encoder(p, g, ogm, map_img, flow):
    P, C = g['P'], g['C']
    fl = _ln(_patch_embed(flow, p, 'patch_embed_flow'), p, 'flow_norm', 1e-5)
    flow_x, flow_res = _layer(fl, p, 'flow_layers0', P, g['depths'][0], g['heads'][0], g['ws'], True)
    x = _patch_embed(ogm[..., 0], p, 'patch_embed_vecicle')
    maps = _patch_embed(map_img, p, 'patch_embed_map')
    if g['large']:
        Pm = g['map_size'] // 4
        pad = (P - Pm) // 2
        maps = F.pad(maps.view(-1, Pm, Pm, C), (0, 0, pad, pad, pad, pad)).reshape(-1, P * P, C)
    x = _ln(x + maps, p, 'all_patch_norm', 1e-5)
    res_list = []

    def crop(t, r, c):
        q = r // 4
        return t.view(-1, r, r, c)[:, q:q + r // 2, q:q + r // 2].reshape(-1, (r // 2) ** 2, c)
    for i in range(3):
        r, c = g['res'][i], g['dim'][i]
        x, res = _layer(x, p, f'layers{i}', r, g['depths'][i], g['heads'][i], g['ws'], i < 2)
        if i == 0:
            x = x + flow_x
            res_list.append(crop(flow_res, r, c) if g['large'] else flow_res)
        res_list.append(crop(res, r, c) if g['large'] else res)
    return res_list


def _sample(image, warp):
    """occu_metric.sample(pixel_type=0, BILINEAR, ZERO) == grid_sample(bilinear, zeros, align_corners=True):
    both lerp against an all-zero exterior and return 0 beyond one pixel outside."""
    B, H, W, C = image.shape
    gx = warp[..., 0] * (2.0 / (W - 1)) - 1.0
    gy = warp[..., 1] * (2.0 / (H - 1)) - 1.0
    grid = torch.stack((gx, gy), -1).reshape(B, -1, 1, 2)
    out = F.grid_sample(image.permute(0, 3, 1, 2), grid, mode='bilinear', padding_mode='zeros', align_corners=True)
    return out.permute(0, 2, 3, 1).reshape(warp.shape[:-1] + (C,))


def _fgmsa(p, x, fg):
    B, H, W, C = x.shape
    ng = nh = 8
    gc = C // ng

    def c1(t, name, bias=True):
        k = p[f'fg_msa/{name}/kernel']
        return F.linear(t, k.reshape(k.shape[2], k.shape[3]).t(), p[f'fg_msa/{name}/bias'] if bias else None)
    q = c1(x, 'proj_q')
    o = _conv_nhwc(q, p['fg_msa/conv_offset_0/kernel'], p['fg_msa/conv_offset_0/bias'], padding=1, groups=ng)
    o = F.gelu(_ln(o, p, 'fg_msa/conv_norm', 1e-3), approximate='tanh')
    o = o.view(B, H, W, ng, gc)
    off = torch.tanh(torch.einsum('bhwgc,co->bghwo', o, p['fg_msa/conv_offset_proj/kernel'][0, 0])) * (H / 2)
    flow_hidden = c1(off, 'conv_offset_proj2') if fg else None          # [B,g,H,W,C]
    ii, jj = torch.meshgrid(torch.arange(H, dtype=x.dtype), torch.arange(W, dtype=x.dtype), indexing='ij')
    ref = torch.stack((jj, ii), -1)                                     # ref[i,j] = (j,i)  (FG_MSA.py:96-100)
    pos = off + ref
    qh = q.view(B, H * W, nh, gc).permute(0, 2, 1, 3)
    kh = c1(x, 'proj_k').view(B, H * W, nh, gc).permute(0, 2, 1, 3)
    vh = c1(x, 'proj_v').view(B, H * W, nh, gc).permute(0, 2, 1, 3)
    att = (qh @ kh.transpose(-1, -2)) * gc ** -0.5
    # bias[b,g,q,k] = bilinear(table_g)[row = dcol - off0[k], col = drow - off1[k]]   (App. D-4)
    qg = ref.view(1, 1, H * W, 1, 2)
    disp = qg - pos.view(B, ng, 1, H * W, 2)
    warp = torch.stack((disp[..., 1], disp[..., 0]), -1)               # (x, y)
    tab = p['fg_msa/warp_attn_rel_table'].permute(2, 0, 1)[None].expand(B, -1, -1, -1)
    tab = tab.reshape(B * ng, 2 * H - 1, 2 * W - 1, 1)
    bias = _sample(tab, warp.reshape(B * ng, H * W, H * W, 2)).view(B, ng, H * W, H * W)
    att = (att + bias).softmax(-1)
    out = (att @ vh).permute(0, 2, 1, 3).reshape(B, H, W, C)
    return c1(out, 'proj_out'), pos, flow_hidden


def _mha(q_in, k_in, p, name, mask, drop=None):
    Wq, Wk, Wv = p[name + '/query_kernel'], p[name + '/key_kernel'], p[name + '/value_kernel']
    Wo, bo = p[name + '/projection_kernel'], p[name + '/projection_bias']
    hs = Wq.shape[-1]
    q = torch.einsum('...ni,hio->...hno', q_in, Wq) / math.sqrt(hs)
    k = torch.einsum('...mi,hio->...hmo', k_in, Wk)
    v = torch.einsum('...mi,hio->...hmo', k_in, Wv)
    lg = q @ k.transpose(-1, -2)
    if mask is not None:
        # reference: logits += -10e9*(1-mask) in f32, where x + (-1e10) == -1e10 exactly (|x| < 512) but the
        # gradient of the ADD still passes through to x (matters only for fully masked rows).
        lg = torch.where(mask.unsqueeze(-3) != 0, lg, lg + (-10e9 - lg).detach())
    coef = lg.softmax(-1)
    if drop is not None:
        coef = _dropout(coef, drop, 0.1)
    o = coef @ v
    return torch.einsum('...hni,hio->...no', o, Wo) + bo


def _xattn(p, pre, query, key, mask):
    v = _mha(query, key, p, pre + '/mha', mask, pre + '/mha/dropout')
    v = _ln(v, p, pre + '/norm1', 1e-3)
    v = _dropout(F.elu(_lin(v, p, pre + '/FFN1')), pre + '/dropout1', 0.1)
    v = _dropout(_lin(v, p, pre + '/FFN2'), pre + '/dropout2', 0.1)
    return _ln(v, p, pre + '/norm2', 1e-3)


def _traj(p, obs_traj, occ_traj):
    pre = 'traj_net/traj_encoder'
    tr = torch.cat([obs_traj, occ_traj], 1)                  # [B,64,11,8] -- batched instead of a 64-way loop
    n_obs = obs_traj.shape[1]
    m = (tr[..., 0] != 0)
    m2 = (m[..., :, None] & m[..., None, :])
    nodes = F.elu(F.linear(tr[..., :5], p[pre + '/node_feature/kernel'][0].t(), p[pre + '/node_feature/bias']))
    nodes = _mha(nodes, nodes, p, pre + '/node_attention', m2, pre + '/node_attention/dropout').amax(-2)
    vec = F.linear(tr[..., 0, 5:], p[pre + '/vector_feature/kernel'].t())
    enc = F.elu(_lin(torch.cat([nodes, vec], -1), p, pre + '/sublayer'))
    seg = p['traj_net/seg_embed/kernel']
    embed = torch.cat([seg[0:1].expand(n_obs, -1), seg[1:2].expand(tr.shape[1] - n_obs, -1)], 0)[None]
    cm = m.any(-1)
    concat = enc * cm[..., None].to(enc.dtype)
    am = cm[:, :, None] & cm[:, None, :]
    value = _xattn(p, 'traj_net/cross_attention', concat + embed, concat, am)
    out = enc + value + embed
    obs = _ln(out[:, :n_obs], p, 'traj_net/obs_norm', 1e-3)
    occ = _ln(out[:, n_obs:], p, 'traj_net/occ_norm', 1e-3)
    return torch.cat([obs, occ], 1), cm


def _upconv(x, p, name):
    lead = x.shape[:-3]
    xx = x.reshape((-1,) + x.shape[-3:]).permute(0, 3, 1, 2)
    xx = F.interpolate(xx, scale_factor=2, mode='nearest')
    y = F.elu(F.conv2d(xx, p[name + '/kernel'].permute(3, 2, 0, 1), p[name + '/bias'], padding=1))
    y = y.permute(0, 2, 3, 1)
    return y.reshape(lead + y.shape[1:])


def _resconv(skip, p, name, r):
    """Conv3D(8,1,1) SAME on the 8x time-repeated skip, via F.conv3d with explicit (3,4) time padding."""
    B, L, C = skip.shape
    x = skip.view(B, 1, r, r, C).expand(B, 8, r, r, C).permute(0, 4, 1, 2, 3)          # NCDHW
    x = F.pad(x, (0, 0, 0, 0, 3, 4))
    w = p[name + '/kernel'].permute(4, 3, 0, 1, 2)                                      # [O,I,8,1,1]
    return F.elu(F.conv3d(x, w, p[name + '/bias'])).permute(0, 2, 3, 4, 1)


def forward(p, cfg, ogm, map_img, obs, occ, flow, fg_msa=True, fg=True, large_ogm=False, masks=None):
    """STrajNet.call.  p: dict name->torch tensor (Keras layouts).  masks=None: training=False; otherwise the 0/1 keep masks
    of every Dropout / DropPath site (same names and shapes as np_ref.strajnet_forward) -> training=True with those draws."""
    global _MASKS, _DPR
    from . import np_ref
    _MASKS = None if masks is None else {k: torch.as_tensor(v) for k, v in masks.items()}
    _DPR = np_ref.drop_path_rates(cfg['depths']) if masks is not None else {}
    try:
        return _forward(p, cfg, ogm, map_img, obs, occ, flow, fg_msa, fg, large_ogm)
    finally:
        _MASKS, _DPR = None, {}


def _forward(p, cfg, ogm, map_img, obs, occ, flow, fg_msa, fg, large_ogm):
    g = _geom(cfg, large_ogm)
    hb, Cb = g['hb'], g['dim'][2]
    res_list = _encoder(p, g, ogm, map_img, flow)
    q = res_list[-1].reshape(-1, hb, hb, Cb)
    B = q.shape[0]
    fh = None
    if fg_msa:
        y, pos, fh = _fgmsa(p, q, fg)
        q = q + y
    query = q.reshape(B, 1, hb * hb, Cb).expand(B, 8, hb * hb, Cb)
    if fg:
        query = query + fh.reshape(B, 8, hb * hb, Cb)
    key, tmask = _traj(p, obs, occ)
    am = tmask[:, None, :].expand(B, hb * hb, -1)
    outs = [_xattn(p, f'cross_attn_obs{i}', query[:, i], key, am) + query[:, i] for i in range(8)]
    x = torch.stack(outs, 1).view(B, 8, hb, hb, Cb)
    flow_res, r0, r1 = res_list[0], res_list[1], res_list[2]
    x = _upconv(x, p, 'decoder/upconv_3_0') + _resconv(r1, p, 'decoder/resconv_3', g['skip'][1])
    x = _upconv(x, p, 'decoder/upconv_2_0') + _resconv(r0, p, 'decoder/resconv_2', g['skip'][0])
    fx = x + _resconv(flow_res, p, 'decoder/resconv_f', g['skip'][0])
    x = _upconv(_upconv(x, p, 'decoder/upconv_1_0'), p, 'decoder/upconv_0_0')
    fx = _upconv(_upconv(fx, p, 'decoder/upconvf_1_0'), p, 'decoder/upconvf_0_0')
    y = _conv_nhwc(x, p['decoder/outconv/kernel'], p['decoder/outconv/bias'], padding=1)
    fy = _conv_nhwc(fx, p['decoder/outconv_f/kernel'], p['decoder/outconv_f/bias'], padding=1)
    out = torch.cat([y, fy], -1)                                   # [B,8,H,W,4]
    Hg = out.shape[2]
    return out.permute(0, 2, 3, 1, 4).reshape(B, Hg, Hg, 32)


def auc_pr_bucketised(y_true, y_pred, n=100):
    """Keras AUC(PR, interpolation) via a bucket histogram (independent of np_ref's dense compare)."""
    eps = 1e-7
    thr = torch.tensor([0.0 - eps] + [(i + 1) / (n - 1) for i in range(n - 2)] + [1.0 + eps], dtype=torch.float32)
    yt = y_true.reshape(-1) != 0
    yp = y_pred.reshape(-1).to(torch.float32)
    bucket = torch.searchsorted(thr, yp, right=False)          # number of thresholds strictly below yp
    hp = torch.bincount(bucket[yt], minlength=n + 1).double()
    hn = torch.bincount(bucket[~yt], minlength=n + 1).double()
    # positive at threshold i  <=>  bucket > i
    tp = hp.sum() - hp.cumsum(0)[:n]
    fp = hn.sum() - hn.cumsum(0)[:n]
    fn = hp.sum() - tp
    dnn = lambda a, b: torch.where(b != 0, a / torch.where(b != 0, b, torch.ones_like(b)), torch.zeros_like(a))
    dtp = tp[:-1] - tp[1:]
    pp = tp + fp
    dp = pp[:-1] - pp[1:]
    slope = dnn(dtp, dp.clamp(min=0))
    icpt = tp[1:] - slope * pp[1:]
    ratio = torch.where((pp[:-1] > 0) & (pp[1:] > 0), dnn(pp[:-1], pp[1:].clamp(min=0)), torch.ones_like(pp[1:]))
    return float(dnn(slope * (dtp + icpt * torch.log(ratio)), (tp[1:] + fn[1:]).clamp(min=0)).sum())


_EPS = float(np.float32(1e-7))
_HI = float(np.float32(1.0) - np.float32(1e-7))


def _bce_prob(y, q):
    qc = q.clamp(_EPS, _HI)                       # torch.clamp, like tf.clip_by_value, passes the gradient on [min, max]
    return -(y * torch.log(qc + _EPS) + (1 - y) * torch.log(1 - qc + _EPS))


def _focal(y, prob, ce):
    p_t = y * prob + (1 - y) * (1 - prob)
    return (y * 0.25 + (1 - y) * 0.75) * (1 - p_t) ** 2 * ce


def loss(logits, gt_obs, gt_occ, gt_flow, origin_flow, replica=1.0, use_gt=True,
         ogm_weight=1000.0, occ_weight=1000.0, flow_origin_weight=1000.0, use_focal_loss=False, use_pred=False,
         no_use_warp=False):
    """OGMFlow_loss (loss.py:50-170); the defaults are the train.py:195-196 flags.  Differentiable twin of
    np_ref.ogm_flow_loss, independently written."""
    B, H, W, _ = logits.shape
    yy, xx = torch.meshgrid(torch.arange(H, dtype=logits.dtype), torch.arange(W, dtype=logits.dtype), indexing='ij')
    ident = torch.stack((xx, yy), -1)[None]
    tot = dict(observed_xe=0.0, occluded_xe=0.0, flow=0.0, flow_warp_xe=0.0)
    fc = 0.0
    for k in range(8):
        po, pc, pf = logits[..., 4 * k:4 * k + 1], logits[..., 4 * k + 1:4 * k + 2], logits[..., 4 * k + 2:4 * k + 4]
        to, tc, tf_, org = gt_obs[:, k], gt_occ[:, k], gt_flow[:, k], origin_flow[:, k]
        n = po.numel() * replica
        xo = F.binary_cross_entropy_with_logits(po, to, reduction='sum')
        xc = F.binary_cross_entropy_with_logits(pc, tc, reduction='sum')
        if use_focal_loss:
            xo = xo + _focal(to, torch.sigmoid(po), F.binary_cross_entropy_with_logits(po, to, reduction='none')).sum()
            xc = xc + _focal(tc, torch.sigmoid(pc), F.binary_cross_entropy_with_logits(pc, tc, reduction='none')).sum()
        tot['observed_xe'] = tot['observed_xe'] + ogm_weight * xo / n
        tot['occluded_xe'] = tot['occluded_xe'] + occ_weight * xc / n
        ta = (to + tc).clamp(0, 1)
        res = 1.0
        if use_gt:
            with torch.no_grad():
                wp = _sample(org, ident + tf_)
                res = float(auc_pr_bucketised(ta, wp * ta) > 0)
        fc += res
        ex = ((tf_[..., :1] != 0) | (tf_[..., 1:] != 0)).to(logits.dtype)
        den = ex.sum() * replica / 2
        fl = ((tf_ - pf) * ex).abs().sum() / den if float(den) != 0 else 0.0
        tot['flow'] = tot['flow'] + res * fl
        if no_use_warp:
            continue
        wpo = _sample(org, ident + pf)
        a, b = (po, pc) if use_pred else (to, tc)
        joint = (torch.sigmoid(a) + torch.sigmoid(b)).clamp(0, 1) * wpo
        bce_mean = _bce_prob(ta, joint).reshape(B, -1).mean(-1).sum()
        if use_pred:
            xw = bce_mean
        elif use_focal_loss:
            xw = _focal(ta, joint, _bce_prob(ta, joint)).sum() + bce_mean
        else:
            xw = F.binary_cross_entropy_with_logits(joint, ta, reduction='sum')
        tot['flow_warp_xe'] = tot['flow_warp_xe'] + res * flow_origin_weight * xw / (ta.numel() * replica)
    return dict(observed_xe=tot['observed_xe'] / 8, occluded_xe=tot['occluded_xe'] / 8,
                flow=tot['flow'] / fc, flow_warp_xe=(tot['flow_warp_xe'] / fc) if not no_use_warp else 0.0)


def to_torch(d, dtype=torch.float64, requires_grad=False):
    out = {}
    for k, v in d.items():
        t = torch.as_tensor(np.asarray(v)).to(dtype) if np.asarray(v).dtype.kind == 'f' else torch.as_tensor(np.asarray(v))
        out[k] = t.requires_grad_(True) if requires_grad and t.is_floating_point() else t
    return out
