"""Per-launch timing of EVERY C-ABI call (bench.py's live roofline measurement).

When enabled, each `ops.call(name, ...)` is bracketed by a HIP event pair on the launch stream and filed under a key made of the
kernel family and its problem shape, together with a cost model of that launch:

    flops_alg   algorithmic FLOPs (2 per MAC of the reference's direct form: SURVEY.md 8d / App. E)
    flops_exec  FLOPs the kernel actually issues (up-convs fold the nearest 2x upsample into 2x2 taps: 9 -> 4 taps = / 2.25)
    bytes_alg   algorithmic HBM bytes: every operand read once, every result written once

Only meaningful with the model in serial mode (one stream: an event pair brackets exactly one kernel).  Under hipGraph replay
individual launches cannot carry events, so bench.py runs this in an eager pass after the timed region.
"""
import torch

ACTIVE = None        # dict key -> list of (e0, e1) while enabled
COST = {}            # key -> (family, flops_alg, flops_exec, bytes_alg)


def enable():
    global ACTIVE
    ACTIVE = {}


def disable():
    global ACTIVE
    p, ACTIVE = ACTIVE, None
    return p


def _es(dt):
    return 4 if dt == 0 else 2


def _gemm(a):
    (M, N, K, nb1, nb2) = a[6:11]
    sAb1, sAb2, sAm, sAk, sBb1, sBb2, sBk, sBn = a[11:19]
    act, alpha, dt, c_f32, acc, splitk, nkb = a[27], a[28], a[29], a[30], a[31], a[32], a[33]
    nb = nb1 * nb2
    es = _es(dt)
    na = (nb2 if sAb2 else 1) * (nb1 if sAb1 else 1)
    nbb = (nb2 if sBb2 else 1) * (nb1 if sBb1 else 1)
    fl = 2.0 * M * N * K * nb * nkb
    by = es * (M * K * na + K * N * nbb) * nkb + (4 if c_f32 else es) * M * N * nb
    if a[4] is not None and getattr(a[4], 'value', None):
        by += es * M * N * nb
    form = ('T' if sAm == 1 and sAk != 1 else 'N') + ('T' if sBn == 1 and sBk != 1 else 'N')
    fam = 'gemm_wgrad' if acc else 'gemm'
    key = f'{fam}[{M}x{N}x{K}{"x%d" % nkb if nkb > 1 else ""} b{nb} {form}{" f32out" if c_f32 else ""}{" act%d" % act if act else ""}]'
    return key, fam, fl, fl, by


def _upconv(kind):
    def f(a):
        if kind == 'fwd':
            F, Hi, Wi, Cin, Cout, dt = a[4], a[5], a[6], a[7], a[8], a[10]
        elif kind == 'dgrad':
            F, Hi, Wi, Cin, Cout, dt = a[4], a[5], a[6], a[7], a[8], a[9]
        else:
            F, Hi, Wi, Cin, Cout, dt = a[5], a[6], a[7], a[8], a[9], a[11]
        es = _es(dt)
        fl = 2.0 * 9 * Cin * Cout * 4 * Hi * Wi * F
        by = es * F * Hi * Wi * (Cin + 4 * Cout) + es * 16 * Cin * Cout
        if kind == 'dgrad' and getattr(a[3], 'value', None):
            by += es * F * Hi * Wi * Cin              # ELU' operand (the layer input)
        if kind == 'wgrad':
            by += 4 * 16 * Cin * Cout
        return f'upconv_{kind}[{Hi}x{Wi},{Cin}->{Cout},F{F}]', 'upconv_' + kind, fl, fl / 2.25, by
    return f


def _upconv_head(a):
    """stj_upconv_fwd_head(x, wf, bias, wh, z, F, Hi, Wi, Cin, Cout, dtype, stream): the 96 -> 48 up-conv whose epilogue projects onto the
    two heads' 9 taps: reads x and the folded weights, writes z [F,2Hi,2Wi,20]; FLOPs = the up-conv + 18 x 48 MACs per output pixel."""
    F, Hi, Wi, Cin, Cout, dt = a[5], a[6], a[7], a[8], a[9], a[10]
    es = _es(dt)
    fl = 2.0 * 9 * Cin * Cout * 4 * Hi * Wi * F + 2.0 * 18 * Cout * 4 * Hi * Wi * F
    by = es * F * Hi * Wi * (Cin + 4 * 20) + es * 16 * Cin * Cout
    return f'upconv_fwd_head[{Hi}x{Wi},{Cin}->{Cout}->18,F{F}]', 'upconv_fwd', fl, 2.0 * 4 * Cin * Cout * 4 * Hi * Wi * F + 2.0 * 32 * 64 * 4 * Hi * Wi * F, by


def _pair_gather(a):
    """stj_outconv_pair_gather(zo, zf, b0, b1, out, B, Tn, H, W, t_major, dtype, stream): 9-neighbour sums of two z tensors -> [B,H,W,4Tn] f32."""
    B, Tn, H, W, dt = a[5], a[6], a[7], a[8], a[10]
    return f'outconv_pair_gather[{H}x{W},F{B * Tn}x2]', 'outconv_fwd', 2.0 * 18 * 2 * B * Tn * H * W, 0.0, 2 * _es(dt) * B * Tn * H * W * 20 + 4 * B * H * W * 4 * Tn


def _upconv_res(a):
    F, Hi, Wi, Cin, Cout, dt = a[7], a[8], a[9], a[10], a[11], a[12]
    es = _es(dt)
    two = getattr(a[5], 'value', None)
    fl = 2.0 * 9 * Cin * Cout * 4 * Hi * Wi * F
    by = es * F * Hi * Wi * (Cin + 4 * Cout * (4 if two else 2)) + es * 16 * Cin * Cout
    return f'upconv_fwd[{Hi}x{Wi},{Cin}->{Cout},F{F}]+skip{2 if two else 1}', 'upconv_fwd', fl, fl / 2.25, by


def _elu_res(a):
    n, dt = a[6], a[7]
    two = getattr(a[1], 'value', None)
    return f'elu_res_bwd[{n}]x{2 if two else 1}', 'unary_bwd', 0.0, 0.0, (6 if two else 4) * _es(dt) * n


def _outconv(kind):
    def f(a):
        if kind == 'fwd':
            F, H, W, C, dt = a[4], a[5], a[6], a[7], a[12]
        else:
            F, H, W, C, dt = a[6], a[7], a[8], a[9], a[17]
        es = _es(dt)
        fl = 2.0 * 9 * C * 2 * H * W * F * (1 if kind == 'fwd' else 2)
        by = es * F * H * W * C * (1 if kind == 'fwd' else 2) + 4 * F * H * W * 2
        return f'outconv_{kind}[{H}x{W},{C}->2,F{F}]', 'outconv_' + kind, fl, fl, by
    return f


def _outconv_pair(a):
    B, Tn, H, W, C, dt = a[7], a[8], a[9], a[10], a[11], a[13]
    F = B * Tn
    fl = 2 * 2.0 * 9 * C * 2 * H * W * F
    by = 2 * _es(dt) * F * H * W * C + 4 * B * H * W * 4 * Tn
    return f'outconv_pair_fwd[{H}x{W},{C}->2,F{F}x2]', 'outconv_fwd', fl, fl, by


def _ln(kind):
    def f(a):
        if kind == 'fwd':
            rows, C, dt = a[6], a[7], a[14]
            by = 2 * _es(dt) * rows * C
        elif kind == 'res_fwd':
            rows, C, dt = a[7], a[8], a[13]
            by = 3 * _es(dt) * rows * C
            return f'layernorm_res_fwd[{rows}x{C}]', 'layernorm_fwd', 9.0 * rows * C, 9.0 * rows * C, by
        else:
            rows, C, dt = a[8], a[9], a[18]
            by = (4 if getattr(a[15], 'value', None) else 3) * _es(dt) * rows * C
        return f'layernorm_{kind}[{rows}x{C}]', 'layernorm_' + kind, 8.0 * rows * C, 8.0 * rows * C, by
    return f


def _win(kind):
    def f(a):
        if kind == 'fwd':
            B, res, heads, dt = a[3], a[4], a[5], a[7]
            nmm, tens = 2, 4
        else:
            B, res, heads, dt = a[6], a[7], a[8], a[10]
            nmm, tens = 5, 8
        items = B * (res // 8) ** 2 * heads
        fl = items * nmm * 2.0 * 64 * 64 * 32
        by = _es(dt) * B * res * res * heads * 32 * tens
        return f'win_attn_{kind}[B{B} {res}x{res} h{heads}]', 'win_attn_' + kind, fl, fl, by
    return f


def _elem(fam, n_idx, dt_idx, tensors, f32=False):
    def f(a):
        n = a[n_idx]
        es = 4 if f32 else _es(a[dt_idx])
        return f'{fam}[{n}]', fam, 0.0, 0.0, tensors * es * n
    return f


def _loss(fam, tens):
    def f(a):
        B, H, W = (a[9], a[10], a[11]) if fam == 'loss_fwd' else ((a[11], a[12], a[13]) if fam == 'loss_fwd_bwd' else (a[8], a[9], a[10]))
        by = 4.0 * B * H * W * (32 * tens + 8 * 5)
        return f'{fam}[B{B} {H}x{W}]', fam, 0.0, 0.0, by
    return f


def _patch_embed(a):
    """stj_patch_embed_fwd(src, w, bias, gamma, beta, add, gamma2, beta2, cols, pre, x2, y, mean, rstd, mean2, rstd2, B, H, W, Cin, pix_stride,
    ch_stride, Cout, eps, dtype, stream): reads the f32 raster lines once, writes the tokens (+ cols / pre / x2 in training, + add read)."""
    B, H, W, Cin, pix, Cout, dt = a[16], a[17], a[18], a[19], a[20], a[22], a[24]
    es = _es(dt)
    M = B * (H // 4) * (W // 4)
    nz = lambda i: bool(getattr(a[i], 'value', None))
    by = 4.0 * B * H * W * pix + es * M * Cout * (1 + nz(5) + nz(9) + nz(10)) + es * M * 16 * Cin * nz(8)
    fl = 2.0 * M * 16 * Cin * Cout
    return f'patch_embed[{H}x{W},{Cin}->{Cout},B{B}]', 'patch_embed_fwd', fl, fl, by


def _ln_chain(a):
    """stj_layernorm_bwd_chain(dy, x2, g2, mean2, rstd2, x1, g1, mean1, rstd1, d2, dx1, ..., rows, C, ..., dtype, stream)"""
    rows, C, dt = a[15], a[16], a[21]
    return f'ln_bwd_chain[{rows}x{C}]', 'layernorm_bwd', 0.0, 0.0, _es(dt) * rows * C * (4 + bool(getattr(a[9], 'value', None)))


MODELS = {
    'stj_gemm': _gemm,
    'stj_patch_embed_fwd': _patch_embed, 'stj_layernorm_bwd_chain': _ln_chain,
    'stj_upconv_fwd': _upconv('fwd'), 'stj_upconv_fwd_head': _upconv_head, 'stj_outconv_pair_gather': _pair_gather, 'stj_upconv_fwd_res': _upconv_res, 'stj_elu_res_bwd': _elu_res, 'stj_upconv_dgrad': _upconv('dgrad'), 'stj_upconv_wgrad': _upconv('wgrad'),
    'stj_outconv_fwd': _outconv('fwd'), 'stj_outconv_pair_fwd': _outconv_pair, 'stj_outconv_bwd': _outconv('bwd'),
    'stj_layernorm_fwd': _ln('fwd'), 'stj_layernorm_bwd': _ln('bwd'), 'stj_layernorm_res_fwd': _ln('res_fwd'),
    'stj_win_attn_fwd': _win('fwd'), 'stj_win_attn_bwd': _win('bwd'),
    'stj_unary_fwd': _elem('unary_fwd', 2, 5, 2), 'stj_unary_bwd': _elem('unary_bwd', 3, 6, 3),
    'stj_dropout': _elem('dropout', 3, 8, 2),
    'stj_loss_fwd': _loss('loss_fwd', 1), 'stj_loss_bwd': _loss('loss_bwd', 2), 'stj_loss_fwd_bwd': _loss('loss_fwd_bwd', 2),
    'stj_nadam_step': _elem('nadam', 4, 0, 7, f32=True),
}
EXTRA_MODELS = {}        # fused kernels register their models here (ops.py)


def record(name, args, launch):
    """Time one C-ABI call.  `launch` performs it."""
    m = MODELS.get(name) or EXTRA_MODELS.get(name)
    if m is not None:
        try:
            key, fam, fa, fe, by = m(args)
        except Exception:
            key, fam, fa, fe, by = name, name.replace('stj_', ''), 0.0, 0.0, 0.0
    else:
        key, fam, fa, fe, by = name, name.replace('stj_', ''), 0.0, 0.0, 0.0
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    launch()
    e1.record(st)
    COST[key] = (fam, fa, fe, by)
    ACTIVE.setdefault(key, []).append((e0, e1))


def summarize(events, steps):
    """-> (per-key table, per-family table); times in ms per step."""
    keys = {}
    for k, evs in events.items():
        ms = [a.elapsed_time(b) for a, b in evs]
        fam, fa, fe, by = COST[k]
        keys[k] = dict(family=fam, launches=len(ms) / steps, avg_ms=sum(ms) / len(ms), ms_per_step=sum(ms) / steps,
                       flops_alg=fa, flops_exec=fe, bytes_alg=by)
    fams = {}
    for k, v in keys.items():
        f = fams.setdefault(v['family'], dict(ms_per_step=0.0, launches=0.0, flop_alg=0.0, flop_exec=0.0, bytes=0.0))
        f['ms_per_step'] += v['ms_per_step']
        f['launches'] += v['launches']
        f['flop_alg'] += v['flops_alg'] * v['launches']
        f['flop_exec'] += v['flops_exec'] * v['launches']
        f['bytes'] += v['bytes_alg'] * v['launches']
    return keys, fams
