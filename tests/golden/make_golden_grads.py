#!/usr/bin/env python
"""Golden fixture of one train step: losses, per-tensor gradient L2 norms and a few small gradients in full.

PARITY UNPINNED, like make_golden.py: the numbers come from the in-repo oracle (oracle/torch_ref.py, float64 autograd over the
independently written differentiable restatement; forward cross-checked against oracle/np_ref.py).  128x128 geometry, B=2,
eval-mode forward (training=False semantics), train.py:195-196 loss flags (use_gt=True).  Inputs / weights come from seeds.

    python tests/golden/make_golden_grads.py             # ~20 s of CPU  -> strajnet_128_b2_grads.npz
    python tests/golden/make_golden_grads.py --cfg256    # BASELINE config 2's own geometry: cfg-256, B=8 (float64 on the CPU: minutes,
                                                         # ~25 GB) -> strajnet_256_b8_grads.npz (4 losses + 299 gradient L2 norms)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import np_ref as R, torch_ref as T   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CFG = dict(input_size=(128, 128), window_size=8, embed_dim=96, depths=[2, 2, 2], num_heads=[3, 6, 12])
FULL = ('decoder/outconv/bias', 'decoder/outconv_f/bias', 'decoder/upconv_0_0/bias', 'layers0/blocks1/attn/relative_position_bias_table',
        'fg_msa/warp_attn_rel_table', 'traj_net/seg_embed/kernel', 'cross_attn_obs3/mha/projection_bias', 'all_patch_norm/gamma')


def compute(weight_seed=0, input_seed=1234, B=2, cfg=None, full=FULL, dtype=torch.float64):
    CFG = cfg or globals()['CFG']
    w = R.make_weights(CFG, weight_seed)
    x = R.make_inputs(CFG, B, seed=input_seed)
    p = T.to_torch(w, dtype, requires_grad=True)
    xt = T.to_torch(x, dtype)
    y = T.forward(p, CFG, xt['ogm'], xt['map_img'], xt['obs'], xt['occ'], xt['flow'])
    d = T.loss(y, xt['gt_obs'], xt['gt_occ'], xt['gt_flow'], xt['origin_flow'], replica=1.0, use_gt=True)
    sum(d.values()).backward()
    names = sorted(p)
    return dict(names=np.array(names), grad_l2=np.array([float(p[n].grad.norm()) for n in names]),
                loss=np.array([float(d[k].detach()) for k in ("observed_xe", "occluded_xe", "flow", "flow_warp_xe")]),
                **{'full:' + n: p[n].grad.numpy().copy() for n in full})


def main():
    if '--cfg256' in sys.argv:
        cfg = dict(CFG, input_size=(256, 256))
        out = compute(B=8, cfg=cfg, full=FULL[:3])
        path = os.path.join(HERE, 'strajnet_256_b8_grads.npz')
        np.savez_compressed(path, weight_seed=0, input_seed=1234, **out)
        print('wrote', path, len(out['names']), 'tensors')
        return
    out = compute()
    np.savez_compressed(os.path.join(HERE, 'strajnet_128_b2_grads.npz'), weight_seed=0, input_seed=1234, **out)
    print('wrote', os.path.join(HERE, 'strajnet_128_b2_grads.npz'), len(out['names']), 'tensors')


if __name__ == '__main__':
    main()
