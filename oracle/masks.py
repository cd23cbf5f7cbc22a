"""TEST INFRASTRUCTURE (like everything under oracle/): turns the Dropout / DropPath keep masks that the HIP kernels drew in
a model's last training=True forward (exported through stj_dropout_mask) into the site dictionary the oracle takes
(np_ref.dropout_sites naming).  The HIP path batches the 8 per-waypoint cross-attentions and the 64 actor encoders, so their
masks are split / reshaped here."""


def masks_from_model(model, B):
    out = {}
    for name in model.dropctx.sites:
        m = model.dropctx.mask(name).cpu().numpy()
        if name.startswith('cross_attn_obs/'):
            suffix = name[len('cross_attn_obs/'):]
            for i in range(8):
                mi = m[i]
                out[f'cross_attn_obs{i}/{suffix}'] = mi if suffix == 'mha/dropout' else mi.reshape(B, -1, mi.shape[-1])
        elif name == 'traj_net/traj_encoder/node_attention/dropout':
            out[name] = m.reshape(B, -1, *m.shape[1:])
        else:
            out[name] = m
    return out
