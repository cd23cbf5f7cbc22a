"""Fused FG-MSA attention core (csrc/fgattn.hip: stj_fg_attn_fwd / stj_fg_attn_bwd) against a float64 statement of FG_MSA.py:150-176
(bias sampled with the oracle's occu_metric.sample restatement) and against the layer-by-layer HIP path mha_core(fg_off=, fg=):
values, and the gradients of q, k, v, the offsets and the relative-position table."""
import pytest
import torch

from test_ops_gpu import mk_param, rnd, ref_of

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _ref64(q, k, v, off, table, B, G, Hh):
    """float64: per group softmax(48^-1/2 q k^T + sample(table_g)(displacement - offset)) v."""
    from oracle.torch_ref import _sample
    HW, gc = Hh * Hh, q.shape[-1] // G
    ii, jj = torch.meshgrid(torch.arange(Hh, dtype=torch.float64), torch.arange(Hh, dtype=torch.float64), indexing='ij')
    ref = torch.stack((jj, ii), -1).view(1, 1, HW, 2)
    pos = off + ref
    disp = ref.view(1, 1, HW, 1, 2) - pos.view(B, G, 1, HW, 2)
    warp = torch.stack((disp[..., 1], disp[..., 0]), -1)
    tab = table.permute(2, 0, 1)[None].expand(B, -1, -1, -1).reshape(B * G, 2 * Hh - 1, 2 * Hh - 1, 1)
    bias = _sample(tab, warp.reshape(B * G, HW, HW, 2)).view(B, G, HW, HW)
    qh, kh, vh = (t.view(B, HW, G, gc).permute(0, 2, 1, 3) for t in (q, k, v))
    P = torch.softmax(qh @ kh.transpose(-1, -2) * gc ** -0.5 + bias, -1)
    return (P @ vh).permute(0, 2, 1, 3).reshape(B, HW, G * gc)


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize('B,Hh', [(2, 16), (3, 8)])
def test_fg_attn_vs_f64_and_layerwise(dt, B, Hh):
    from strajnet_amd import ops
    G, gc = 8, 48
    HW, C = Hh * Hh, G * gc
    q, k, v = (rnd((B, HW, C), dt, 10 + i).requires_grad_(True) for i in range(3))
    # offsets off the integer lattice (there TF's clip gradient and the float64 reference's one-sided one differ); some far outside the table
    # (exactly representable in bf16: integer part below 16, fraction a multiple of 1/8)
    gen = torch.Generator().manual_seed(2)
    off = (torch.randint(-9, 10, (B, G, HW, 2), generator=gen) + torch.randint(2, 7, (B, G, HW, 2), generator=gen) / 8.0).to(dt).cuda()
    off[0, 0, :4] = 40.5
    off = off.requires_grad_(True)
    go = rnd((B, HW, C), dt, 5)
    res = []
    for fused in (True, False):
        pt = mk_param((2 * Hh - 1, 2 * Hh - 1, G), dt, 0.5, 1)
        for t in (q, k, v, off):
            t.grad = None
        if fused:
            o = ops.fg_attn(q, k, v, off, pt, Hh, Hh, gc ** -0.5)
        else:
            o = ops.mha_core(q, k, v, G, gc, gc ** -0.5, fg_off=off, fg=(pt, Hh, Hh))
        o.backward(go)
        torch.cuda.synchronize()
        res.append([o.detach().clone(), q.grad.clone(), k.grad.clone(), v.grad.clone(), off.grad.clone(), pt.grad.clone()])
    q64, k64, v64, o64, t64 = ref_of(q), ref_of(k), ref_of(v), ref_of(off), ref_of(pt.master)
    y64 = _ref64(q64, k64, v64, o64, t64, B, G, Hh)
    y64.backward(go.double().cpu())
    refs = [y64, q64.grad, k64.grad, v64.grad, o64.grad, t64.grad]
    names = ['a', 'dq', 'dk', 'dv', 'doff', 'dtable']
    tol = {torch.bfloat16: 2e-2, torch.float16: 3e-3, torch.float32: 2e-5}[dt]       # (f32: the kernel of the 1e-3 parity gate; exact-f32 MFMA)
    rep = [f'{n}: fused {_rel(f, r):.2e} layerwise {_rel(u, r):.2e}' for n, f, u, r in zip(names, res[0], res[1], refs)]
    print('\n'.join(rep))
    for n, f, u, r in zip(names, res[0], res[1], refs):
        assert _rel(f, r) <= tol, rep
        assert _rel(f, r) <= 1.5 * _rel(u, r) + 1e-3, rep            # never noticeably further from float64 than the layer-by-layer path


def test_fg_attn_eval_has_no_lse_and_geometry_limits():
    from strajnet_amd import ops
    from strajnet_amd._lib import StjError
    B, Hh, G, gc = 1, 8, 8, 48
    HW, C = Hh * Hh, G * gc
    pt = mk_param((2 * Hh - 1, 2 * Hh - 1, G), torch.bfloat16, 0.5, 1)
    q, k, v = (rnd((B, HW, C), torch.bfloat16, 20 + i) for i in range(3))
    off = rnd((B, G, HW, 2), torch.bfloat16, 2, 2.0)
    a = ops.fg_attn(q, k, v, off, pt, Hh, Hh, gc ** -0.5)             # no grad needed: the log-sum-exp output is skipped
    b = ops.mha_core(q, k, v, G, gc, gc ** -0.5, fg_off=off, fg=(pt, Hh, Hh))
    assert _rel(a, b) < 1e-2
    # f32 runs the same kernel (round 5); maps other than 8 x 8 / 16 x 16 do not
    assert ops.fg_attn_ok(torch.float32, Hh, Hh, gc) and not ops.fg_attn_ok(torch.bfloat16, 32, 32, gc)
    a32 = ops.fg_attn(q.float(), k.float(), v.float(), off.float(), mk_param((2 * Hh - 1, 2 * Hh - 1, G), torch.float32, 0.5, 1), Hh, Hh, gc ** -0.5)
    assert _rel(a32, a) < 1e-2
    with pytest.raises(StjError):
        ops.fg_attn(q[:, :36], k[:, :36], v[:, :36], off[:, :, :36], pt, 6, 6, gc ** -0.5)
