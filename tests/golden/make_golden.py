#!/usr/bin/env python
"""Generates the golden fixtures in this directory from the in-repo oracle (oracle/np_ref.py, float64).

PARITY UNPINNED: the reference cannot be executed (no TensorFlow), so these vectors are outputs of the oracle's
literal restatement, cross-checked against the independent torch restatement (tests/test_oracle_cross.py).
Inputs and weights are regenerated from seeds (np_ref.make_inputs / make_weights), only outputs are stored.

    python tests/golden/make_golden.py            # ~1.5 min of CPU
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import np_ref as R   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    cfg = dict(input_size=(128, 128), window_size=8, embed_dim=96, depths=[2, 2, 2], num_heads=[3, 6, 12])
    w = R.make_weights(cfg, 0)
    x = R.make_inputs(cfg, 1, seed=1234)
    taps = {}
    y = R.strajnet_forward(w, cfg, x['ogm'], x['map_img'], x['obs'], x['occ'], x['flow'], taps=taps)
    L = R.ogm_flow_loss(y, x['gt_obs'], x['gt_occ'], x['gt_flow'], x['origin_flow'])
    np.savez_compressed(os.path.join(HERE, 'strajnet_128_b1.npz'), weight_seed=0, input_seed=1234,
                        logits=y.astype(np.float32), query_sub=taps['query'][0, :, ::16, ::48],
                        traj_key_sub=taps['traj_key'][0, ::8, ::48],
                        loss=np.array([L[k] for k in ('observed_xe', 'occluded_xe', 'flow', 'flow_warp_xe')]))
    cfg = dict(cfg, input_size=(256, 256))
    w = R.make_weights(cfg, 0)
    x = R.make_inputs(cfg, 1, seed=1234)
    y = R.strajnet_forward(w, cfg, x['ogm'], x['map_img'], x['obs'], x['occ'], x['flow'])
    L = R.ogm_flow_loss(y, x['gt_obs'], x['gt_occ'], x['gt_flow'], x['origin_flow'])
    sub = y[0, ::4, ::4, :].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, 'strajnet_256_b1.npz'), weight_seed=0, input_seed=1234,
                        logits_sub=sub, logits_sub_sum=float(sub.astype(np.float64).sum()),
                        logits_rowsum=y[0].sum((1, 2)), logits_abs_max=float(np.abs(y).max()),
                        loss=np.array([L[k] for k in ('observed_xe', 'occluded_xe', 'flow', 'flow_warp_xe')]))
    print('wrote fixtures to', HERE)


if __name__ == '__main__':
    main()
