"""ctypes binding of libstrajnet_hip.so (the C-ABI HIP library, see include/strajnet_hip.h).

The product path has NO CPU fallback: if the library is missing this raises, loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('STJ_LIB_PATH') or os.path.join(_HERE, 'libstrajnet_hip.so')     # override: A/B runs of two builds

vp, ci, cl, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float

# name -> argument ctypes (return type is always int except where noted)
SIGNATURES = {
    'stj_abi_version': [],
    'stj_gemm': [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci,
                 cl, cl, cl, cl, cl, cl, cl, cl, cl, cl, cl, cl, cl, cl, cl, cl,
                 ci, cf, ci, ci, ci, ci, ci, cl, cl, vp, vp],
    'stj_gemm_group_workspace_bytes': [],
    'stj_gemm_group_begin': [vp],
    'stj_gemm_group_end': [vp, vp],
    'stj_wgrad_job_supported': [vp, ci],
    'stj_wgrad_group': [vp, ci, ci, ci, vp],
    'stj_colsum': [vp, vp, ci, ci, cl, ci, vp],
    'stj_cast': [vp, ci, vp, ci, cl, vp],
    'stj_crc32c': [vp, cl, vp],
    'stj_time_collapse': [vp, vp, cl, ci, vp],
    'stj_agent_prep': [vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, ci, vp],
    'stj_agent_mix_fwd': [vp, vp, vp, vp, vp, ci, ci, ci, ci, vp],
    'stj_agent_mix_bwd': [vp, vp, vp, vp, vp, ci, ci, ci, ci, vp],
    'stj_agent_sum_fwd': [vp, vp, vp, vp, ci, ci, ci, ci, vp],
    'stj_agent_sum_bwd': [vp, vp, ci, ci, ci, ci, vp],
    'stj_time_fold': [vp, vp, cl, vp],
    'stj_fold_parts': [vp, vp, vp, cl, vp],
    'stj_decode_raw': [vp, ci, vp, cl, ci, ci, ci, ci, ci, ci, ci, cf, vp],
    'stj_metrics': [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp],
    'stj_rng_advance': [vp, vp],
    'stj_rng_advance_snap': [vp, vp, vp],
    'stj_dropout': [vp, vp, vp, cl, cl, cf, vp, ci, ci, vp],
    'stj_dropout_mask': [vp, cl, cf, vp, ci, vp],
    'stj_nadam_step': [vp, vp, vp, vp, cl, cf, cf, cf, cf, cf, cf, cf, cf, vp],
    'stj_unary_fwd': [vp, vp, cl, ci, cf, ci, vp],
    'stj_unary_bwd': [vp, vp, vp, cl, ci, cf, ci, vp],
    'stj_maxpool_fwd': [vp, vp, vp, cl, ci, ci, ci, vp],
    'stj_maxpool_bwd': [vp, vp, vp, vp, cl, ci, ci, ci, vp],
    'stj_patch_embed_supported': [ci, ci, ci],
    'stj_patch_embed_fwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cl, ci, ci, cf, ci, vp],
    'stj_layernorm_fwd': [vp, vp, vp, vp, vp, vp, cl, ci, cf, ci, ci, cl, ci, cl, ci, vp],
    'stj_layernorm_res_fwd': [vp, vp, vp, vp, vp, vp, vp, cl, ci, cf, cl, ci, cl, ci, vp],
    'stj_layernorm_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, cl, ci, ci, ci, cl, ci, cl, vp, ci, cl, ci, vp],
    'stj_layernorm_bwd_chain_supported': [ci, ci],
    'stj_layernorm_bwd_chain': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, cl, ci, ci, cl, ci, cl, ci, vp],
    'stj_win_attn_fwd': [vp, vp, vp, ci, ci, ci, ci, ci, vp],
    'stj_win_attn_bwd': [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp],
    'stj_swin_split_workspace_bytes': [cl, ci],
    'stj_swin_mlp_fwd': [vp, vp, vp, vp, vp, vp, vp, vp, cl, ci, cf, vp, ci, cf, cl, ci, vp, vp],
    'stj_swin_mlp_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, cl, cl, ci, cf, vp, ci, cf, cl, ci, vp, vp],
    'stj_swin_attn_fwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, vp, ci, cf, ci, vp, vp],
    'stj_swin_attn_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, vp, vp, ci, cl, ci, ci, ci, ci, vp, ci, cf, ci, vp, vp],
    'stj_xattn_pack_workspace_bytes': [ci],
    'stj_xattn_pack_tail_workspace_bytes': [ci],
    'stj_xattn_pack': [vp, vp, vp, vp, cl, ci, vp, ci, vp],
    'stj_xattn_fwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, cl, vp, vp, vp, vp, vp, ci, ci, ci, vp, ci, ci, ci, cf, ci, vp],
    'stj_xattn_bwd_workspace_bytes': [ci, ci, ci],
    'stj_xattn_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, cl, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                      ci, ci, ci, vp, ci, ci, ci, cf, ci, vp],
    'stj_softmax_fwd': [vp, vp, vp, vp, vp, cl, ci, ci, ci, ci, vp],
    'stj_softmax_bwd': [vp, vp, vp, cl, ci, ci, vp],
    'stj_fg_bias_fwd': [vp, vp, vp, ci, ci, ci, ci, ci, vp],
    'stj_fg_bias_bwd': [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp],
    'stj_small_attn_supported': [ci, ci, ci, ci],
    'stj_small_attn_fwd': [vp, vp, vp, vp, vp, vp, cl, ci, ci, ci, cf, vp, ci, cf, ci, vp],
    'stj_small_attn_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, cl, ci, ci, ci, cf, vp, ci, cf, ci, vp],
    'stj_agent_pack_workspace_bytes': [ci],
    'stj_agent_pack': [vp, vp, ci, vp],
    'stj_agent_enc_supported': [ci, ci, ci, ci],
    'stj_agent_enc_fwd': [vp, vp],
    'stj_agent_enc_bwd': [vp, vp],
    'stj_agent_int_supported': [ci, ci, ci],
    'stj_agent_int_fwd': [vp, vp],
    'stj_agent_int_bwd': [vp, vp],
    'stj_agent_out_fwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, ci, vp],
    'stj_agent_out_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp],
    'stj_fgoff_supported': [ci, ci, ci, ci, ci],
    'stj_fgoff_pack_workspace_bytes': [ci],
    'stj_fgoff_pack': [vp, vp, ci, vp],
    'stj_fgoff_fwd': [vp, vp],
    'stj_fgoff_bwd': [vp, vp],
    'stj_fg_attn_fwd': [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, ci, vp],
    'stj_fg_attn_bwd_workspace_bytes': [ci, ci, ci, ci],
    'stj_fg_attn_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, ci, vp],
    'stj_fg_offset_fwd': [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, cf, ci, ci, vp],
    'stj_fg_offset_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, cf, ci, ci, vp],
    'stj_upconv_prep': [vp, vp, vp, ci, ci, ci, vp],
    'stj_upconv_fold': [vp, vp, ci, ci, vp],
    'stj_upconv_fwd': [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp],
    'stj_upconv_fwd_res': [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp],
    'stj_elu_res_bwd': [vp, vp, vp, vp, vp, vp, cl, ci, vp],
    'stj_skip_junction_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, cl, ci, vp],
    'stj_upconv_dgrad': [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp],
    'stj_upconv_wgrad': [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp],
    'stj_outconv_pair_fwd': [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp],
    'stj_upconv_fwd_head': [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp],
    'stj_outconv_pair_gather': [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp],
    'stj_outconv_fwd': [vp, vp, vp, vp, ci, ci, ci, ci, ci, cl, cl, cl, ci, vp],
    'stj_outconv_bwd': [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, cl, cl, cl, ci, vp, cl, ci, vp],
    'stj_outconv_bwd_workspace_bytes': [],
    'stj_im2col_patch': [vp, vp, ci, ci, ci, ci, cl, ci, ci, vp],
    'stj_im2col3': [vp, vp, ci, ci, ci, ci, ci, ci, vp],
    'stj_col2im3': [vp, vp, ci, ci, ci, ci, ci, ci, vp],
    'stj_loss_auc_gate': [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp],
    'stj_loss_fwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, cf, cf, cf, cf, ci, vp],
    'stj_loss_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp],
    'stj_loss_coef': [vp, vp, vp, vp, ci, ci, ci, cf, cf, cf, cf, ci, vp],
    'stj_loss_finalize': [vp, vp, vp, vp, ci, ci, ci, cf, cf, cf, cf, ci, vp],
    'stj_loss_gate_coef': [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, cf, cf, cf, cf, ci, vp],
    'stj_loss_fwd_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, cf, cf, cf, cf, ci, vp],
}

class WgradJob(ctypes.Structure):
    """struct stj_wgrad_job (include/strajnet_hip.h)"""
    _fields_ = [('x', vp), ('dy', vp), ('dw', vp), ('db', vp),
                ('rows', ci), ('cin', ci), ('cout', ci), ('nb1', ci), ('nb2', ci),
                ('ldx', cl), ('lddy', cl), ('lddw', cl),
                ('sx1', cl), ('sx2', cl), ('sdy1', cl), ('sdy2', cl), ('sdw1', cl), ('sdw2', cl), ('sdb1', cl), ('sdb2', cl)]


class AgentWeights(ctypes.Structure):
    """struct stj_agent_weights (include/strajnet_hip.h)"""
    _fields_ = [(n, vp) for n in ('e_wq', 'e_wk', 'e_wv', 'e_wo', 'e_ws', 'i_wq', 'i_wk', 'i_wv', 'i_wo', 'i_w1', 'i_w2')]


class AgentEncArgs(ctypes.Structure):
    """struct stj_agent_enc_args (include/strajnet_hip.h)"""
    _fields_ = ([('obs', vp), ('occ', vp), ('n_obs', ci), ('n_occ', ci), ('B', ci), ('dtype', ci), ('pack', vp)] +
                [(n, vp) for n in ('wn', 'bn', 'wv3', 'bo', 'bs', 'enc', 'cmi', 's_nodes', 's_qkv', 's_att', 's_pmask', 's_cat', 'rng_state')] +
                [('site', ci), ('p_drop', cf)] +
                [('d_enc', vp), ('d_enc_f32', ci)] +
                [(n, vp) for n in ('wq', 'wk', 'wv', 'wo', 'ws', 'dpre_s', 'dout', 'dqkv', 'dwn', 'dbn', 'dwv3')])


class AgentIntArgs(ctypes.Structure):
    """struct stj_agent_int_args (include/strajnet_hip.h)"""
    _fields_ = ([('enc', vp), ('cmi', vp), ('n_obs', ci), ('n_occ', ci), ('B', ci), ('dtype', ci)] +
                [(n, vp) for n in ('pack', 'seg', 'bo', 'g1', 'be1', 'b1', 'b2', 'g2', 'be2', 'g_obs', 'b_obs', 'g_occ', 'b_occ', 'key', 'ws_v1', 'ws_u2',
                                   's_concat', 's_qin', 's_q', 's_k', 's_v', 's_att', 's_v1', 's_n1', 's_h', 's_u2', 's_out', 'rng_state')] +
                [('site_a', ci), ('site_1', ci), ('site_2', ci), ('p_drop', cf)] +
                [(n, vp) for n in ('dkey', 'wq', 'wk', 'wv', 'wo', 'w1', 'w2', 'd_enc', 'ws_dn1', 'dq', 'dk', 'dv', 'dv1', 'dpre1', 'dz2',
                                   'dseg', 'dg1', 'dbe1', 'dg2', 'dbe2', 'dg_obs', 'db_obs', 'dg_occ', 'db_occ')])

class FgOffArgs(ctypes.Structure):
    """struct stj_fgoff_args (include/strajnet_hip.h)"""
    _fields_ = ([('B', ci), ('H', ci), ('W', ci), ('dtype', ci), ('scale', cf), ('eps', cf)] +
                [(n, vp) for n in ('q', 'pack', 'bias', 'gamma', 'beta', 'w1', 'off', 'cols', 'c', 'mean', 'rstd', 'doff', 'dc', 'dq',
                                   'd_w1', 'd_gamma', 'd_beta', 'd_bias')])


_lib = None


class StjError(RuntimeError):
    pass


def lib():
    """Load (once) and return the C-ABI library.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise StjError(f'{LIB_PATH} not found: build it with `python -m strajnet_amd.build` '
                           '(hipcc --offload-arch=gfx950).  There is no CPU / eager fallback.')
        L = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here == header / library mismatch
            fn.argtypes = args
            fn.restype = ci
        for name in SIGNATURES:
            if name.endswith('_workspace_bytes'):
                getattr(L, name).restype = ctypes.c_longlong
        L.stj_last_error.argtypes = []
        L.stj_last_error.restype = ctypes.c_char_p
        _lib = L
    return _lib


def call(name, *args):
    L = lib()
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise StjError(f'{name} failed ({rc}): {L.stj_last_error().decode()}')
