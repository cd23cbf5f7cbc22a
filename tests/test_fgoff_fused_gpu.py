"""Fused FG-MSA offset head (csrc/fgoff_fused.hip: stj_fgoff_pack / stj_fgoff_fwd / stj_fgoff_bwd) against a float64 statement of
FG_MSA.py:84-92,109-123 (grouped 3x3 conv -> LayerNorm(1e-3) -> gelu -> per-group 1x1 conv 48 -> 2 -> tanh * H/2) and against the
layer-by-layer HIP chain it replaces (grouped_conv3 + layernorm + gelu + fg_offset): the offsets and the gradients of q and of the five
parameters."""
import pytest
import torch

from test_ops_gpu import mk_param, rnd, ref_of

pytestmark = pytest.mark.gpu
G, GC, C = 8, 48, 384


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _params(dt):
    ps = [mk_param((3, 3, GC, C), dt, 0.06, 1), mk_param((C,), dt, 0.1, 2), mk_param((C,), dt, 0.2, 3), mk_param((C,), dt, 0.1, 4),
          mk_param((1, 1, GC, 2), dt, 0.12, 5)]
    with torch.no_grad():
        ps[2].master.add_(1.0)          # gamma around 1
    return ps


def _ref64(q, w, b, gam, bet, w1, scale):
    B, H, W, _ = q.shape
    o = torch.nn.functional.conv2d(q.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), b, padding=1, groups=G).permute(0, 2, 3, 1)
    o = torch.nn.functional.layer_norm(o, (C,), gam, bet, 1e-3)
    o = 0.5 * o * (1 + torch.tanh(0.7978845608028654 * (o + 0.044715 * o ** 3)))
    s = torch.einsum('bpgi,ij->bgpj', o.reshape(B, H * W, G, GC), w1.reshape(GC, 2))
    return torch.tanh(s) * scale


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize('B,H,W', [(2, 16, 16), (32, 16, 16), (1, 5, 16), (2, 32, 32), (3, 8, 8)])       # tiles of 1 x 16, 2 x 16 (16-bit), 1 x 16, 1 x 32, 2 x 8 pixels
def test_fgoff_vs_f64_and_layerwise(dt, B, H, W):
    from strajnet_amd import ops
    if not ops.fgoff_ok(dt, H, W, C, G):
        assert dt == torch.float32 and W == 32           # the f32 parity mode keeps the layer-by-layer chain at 32 x 32
        pytest.skip('geometry served by the layer-by-layer chain')
    scale = H / 2.0
    q = rnd((B, H, W, C), dt, 7).requires_grad_(True)
    go = rnd((B, G, H * W, 2), dt, 9)
    res = []
    for fused in (True, False):
        pw, pb, pg, pbe, p1 = ps = _params(dt)
        q.grad = None
        if fused:
            pack = ops.fgoff_pack(pw, dt)
            off = ops.fgoff_chain(q, pw, pb, pg, pbe, p1, pack, scale, 1e-3)
        else:
            o = ops.gelu(ops.layernorm(ops.grouped_conv3(q, pw, pb, G), pg, pbe, 1e-3))
            off = ops.fg_offset(o, p1, scale, G)
        off.backward(go)
        torch.cuda.synchronize()
        res.append([off.detach().clone(), q.grad.clone()] + [p.grad.clone() for p in ps])
    q64 = ref_of(q)
    w64 = [ref_of(ps[0].c), ref_of(ps[1].master), ref_of(ps[2].master), ref_of(ps[3].master), ref_of(ps[4].c)]
    y64 = _ref64(q64, *w64, scale)
    y64.backward(go.double().cpu())
    refs = [y64, q64.grad] + [t.grad for t in w64]
    names = ['off', 'dq', 'dW', 'dbias', 'dgamma', 'dbeta', 'dW1']
    tol = {torch.bfloat16: 3e-2, torch.float16: 4e-3, torch.float32: 3e-5}[dt]
    rep = [f'{n}: fused {_rel(f, r):.2e} layerwise {_rel(u, r):.2e} fused-vs-layerwise {_rel(f, u):.2e}' for n, f, u, r in zip(names, res[0], res[1], refs)]
    print('\n'.join(rep))
    for n, f, u, r in zip(names, res[0], res[1], refs):
        assert _rel(f, r) <= tol, rep
        assert _rel(f, r) <= 1.5 * _rel(u, r) + 1e-3, rep            # never noticeably further from float64 than the layer-by-layer chain


def test_fgoff_inference_writes_nothing_else_and_rejects_bad_geometry():
    from strajnet_amd import ops
    dt = torch.bfloat16
    pw, pb, pg, pbe, p1 = _params(dt)
    pack = ops.fgoff_pack(pw, dt)
    q = rnd((2, 16, 16, C), dt, 3)
    with torch.no_grad():
        a = ops.fgoff_chain(q, pw, pb, pg, pbe, p1, pack, 8.0, 1e-3)
        o = ops.gelu(ops.layernorm(ops.grouped_conv3(q, pw, pb, G), pg, pbe, 1e-3))
        b = ops.fg_offset(o, p1, 8.0, G)
    torch.cuda.synchronize()
    assert a.shape == (2, G, 256, 2) and _rel(a, b) < 1e-2
    assert not ops.fgoff_ok(dt, 16, 24, C, G) and not ops.fgoff_ok(dt, 16, 16, 192, G) and not ops.fgoff_ok(torch.float32, 32, 32, C, G)
    assert ops.fgoff_ok(dt, 8, 8, C, G) and not ops.fgoff_ok(dt, 5, 8, C, G)         # 8-pixel rows go two to a workgroup
