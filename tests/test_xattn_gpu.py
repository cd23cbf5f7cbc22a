"""Fused Cross_AttentionT kernels (csrc/xattn_fused.hip: stj_xattn_fwd / stj_xattn_bwd) against a float64 restatement of
trajNet.py:189-234,305-317 written here with torch einsums (the kernel-drawn Dropout masks are exported and handed to it), and
against the layer-by-layer HIP chain (model.fused_xattn = False).  Forward values, the gradients of query / key and of all 8 x 19
parameter tensors of the block."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(input_size=(256, 256), window_size=8, embed_dim=96, depths=[2, 2, 2], num_heads=[3, 6, 12])


def _ln(x, g, b, eps):
    m = x.mean(-1, keepdim=True)
    v = ((x - m) ** 2).mean(-1, keepdim=True)
    return (x - m) / torch.sqrt(v + eps) * g + b


def ref_f64(query, key, tmask, W, masks):
    """query [8,B,HW,384], key [B,64,384] (f64, requires_grad), W: name -> f64 tensor, masks: None or dict of keep masks."""
    outs = []
    for z in range(8):
        p = f'cross_attn_obs{z}/'
        q = torch.einsum('bni,hio->bnho', query[z], W[p + 'mha/query_kernel']) / math.sqrt(42.0)
        k = torch.einsum('bmi,hio->bmho', key, W[p + 'mha/key_kernel'])
        v = torch.einsum('bmi,hio->bmho', key, W[p + 'mha/value_kernel'])
        logits = torch.einsum('bnho,bmho->bhnm', q, k)
        # tfa: logits += -10e9 * (1 - mask) in f32: the addend absorbs the logit (value -1e10 exactly), the gradient of the ADD stays 1
        # (it matters for a scene whose agents are all masked: uniform softmax, non-zero dS)
        logits = logits + torch.where(tmask[:, None, None, :] != 0, torch.zeros_like(logits), (-10e9 - logits).detach())
        P = torch.softmax(logits, -1)
        if masks is not None:
            P = P * masks['a'][z].double() / 0.9
        o = torch.einsum('bhnm,bmho->bnho', P, v)
        v1 = torch.einsum('bnho,hoc->bnc', o, W[p + 'mha/projection_kernel']) + W[p + 'mha/projection_bias']
        n1 = _ln(v1, W[p + 'norm1/gamma'], W[p + 'norm1/beta'], 1e-3)
        h = torch.nn.functional.elu(n1 @ W[p + 'FFN1/kernel'] + W[p + 'FFN1/bias'])
        if masks is not None:
            h = h * masks['1'][z].view(h.shape).double() / 0.9
        u = h @ W[p + 'FFN2/kernel'] + W[p + 'FFN2/bias']
        if masks is not None:
            u = u * masks['2'][z].view(u.shape).double() / 0.9
        outs.append(_ln(u, W[p + 'norm2/gamma'], W[p + 'norm2/beta'], 1e-3) + query[z])
    return torch.stack(outs)


def _run(model, query, key, tmask, G, training, fused):
    model.fused_xattn = fused
    model.zero_grad()
    model._sync_compute_weights()
    model._xattn_pack_stale = True
    q = query.clone().requires_grad_(True)
    k = key.clone().requires_grad_(True)
    model._dctx = None
    if training:
        model.dropctx.n, model.dropctx.sites = 0, {}
        model._dctx = model.dropctx
    y = model._cross_attention_z(q, k, tmask)
    (y.float() * G).sum().backward()
    torch.cuda.synchronize()
    names = [n for n in model.params if n.startswith('cross_attn_obs')]
    grads = {n: model.params[n].grad.detach().double().cpu().clone() for n in names}
    return y.detach().double().cpu(), q.grad.double().cpu(), k.grad.double().cpu(), grads


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('training', [False, True])
def test_xattn_fused_vs_f64(dtype, training):
    from strajnet_amd import STrajNet
    torch.manual_seed(0)
    B, HW = 2, 256
    model = STrajNet(CFG, fg_msa=True, fg=True, large_ogm=False, dtype=dtype, device='cuda:0', seed=3)
    # give biases / LN affine parameters non-trivial values (the reference initialises them to 0 / 1)
    with torch.no_grad():
        for n, p in model.params.items():
            if n.startswith('cross_attn_obs') and n.rsplit('/', 1)[-1] in ('bias', 'gamma', 'beta', 'projection_bias'):
                p.master.add_(0.1 * torch.randn_like(p.master))
    dev = model.device
    query = torch.randn(8, B, HW, 384, device=dev).to(dtype)
    key = torch.randn(B, 64, 384, device=dev).to(dtype)
    tmask = (torch.rand(B, 64, device=dev) < 0.7).to(torch.int32)
    tmask[:, 0] = 1
    G = torch.randn(8, B, HW, 384, device=dev)
    if training:
        model.dropctx.begin()
    yf, dqf, dkf, gf = _run(model, query, key, tmask, G, training, True)
    masks = None
    if training:
        masks = {'a': model.dropctx.mask('cross_attn_obs/mha/dropout').cpu(), '1': model.dropctx.mask('cross_attn_obs/dropout1').cpu(),
                 '2': model.dropctx.mask('cross_attn_obs/dropout2').cpu()}
    yu, dqu, dku, gu = _run(model, query, key, tmask, G, training, False)
    if training:        # the layer-by-layer chain drew the very same masks (same sites, same Philox stream)
        assert torch.equal(masks['a'], model.dropctx.mask('cross_attn_obs/mha/dropout').cpu())
        assert torch.equal(masks['2'], model.dropctx.mask('cross_attn_obs/dropout2').cpu())
    # float64 reference on the operands the kernels see (16-bit mode: the rounded weights of the compute copy)
    W = {}
    for n, p in model.params.items():
        if n.startswith('cross_attn_obs'):
            leaf = n.rsplit('/', 1)[-1]
            src = p.c if leaf in ('query_kernel', 'key_kernel', 'value_kernel', 'projection_kernel', 'kernel') else p.master
            W[n] = src.detach().double().cpu().requires_grad_(True)
    q64 = query.double().cpu().requires_grad_(True)
    k64 = key.double().cpu().requires_grad_(True)
    y64 = ref_f64(q64, k64, tmask.cpu(), W, masks)
    (y64 * G.double().cpu()).sum().backward()

    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-30))
    tol_y, tol_g = (2e-5, 2e-4) if dtype == torch.float32 else (1.5e-2, 3e-2)
    rep = [f'y fused {rel(yf, y64.detach()):.2e} unfused {rel(yu, y64.detach()):.2e}',
           f'dq fused {rel(dqf, q64.grad):.2e} unfused {rel(dqu, q64.grad):.2e}',
           f'dk fused {rel(dkf, k64.grad):.2e} unfused {rel(dku, k64.grad):.2e}']
    worst = (0.0, None)
    worst_u = (0.0, None)
    for n in gf:
        e = rel(gf[n], W[n].grad)
        if e > worst[0]:
            worst = (e, n)
        eu = rel(gu[n], W[n].grad)
        if eu > worst_u[0]:
            worst_u = (eu, n)
    rep.append(f'worst param grad fused {worst[0]:.2e} ({worst[1]}) unfused {worst_u[0]:.2e} ({worst_u[1]})')
    print('\n'.join(rep))
    assert float((yf - y64.detach()).abs().max()) <= (1e-4 if dtype == torch.float32 else 0.15), rep
    assert rel(yf, y64.detach()) <= tol_y, rep
    assert rel(dqf, q64.grad) <= tol_g, rep
    assert rel(dkf, k64.grad) <= tol_g, rep
    assert worst[0] <= (tol_g if dtype == torch.float32 else 6e-2), rep
    # and never worse than twice the layer-by-layer chain's own distance from float64 (16-bit mode)
    if dtype != torch.float32:
        assert rel(yf, y64.detach()) <= 2.0 * rel(yu, y64.detach()) + 1e-3, rep


def test_xattn_small_geometry_and_all_masked_scene():
    """HW = 64 (the 128 x 128 smoke geometry: one token tile per scene) and a scene whose agents are ALL masked (uniform softmax)."""
    from strajnet_amd import STrajNet
    torch.manual_seed(1)
    B, HW = 3, 64
    model = STrajNet(CFG, fg_msa=True, fg=True, large_ogm=False, dtype=torch.float32, device='cuda:0', seed=5)
    dev = model.device
    query = torch.randn(8, B, HW, 384, device=dev)
    key = torch.randn(B, 64, 384, device=dev)
    tmask = (torch.rand(B, 64, device=dev) < 0.5).to(torch.int32)
    tmask[1] = 0
    G = torch.randn(8, B, HW, 384, device=dev)
    yf, dqf, dkf, gf = _run(model, query, key, tmask, G, False, True)
    W = {n: p.master.detach().double().cpu().requires_grad_(True) for n, p in model.params.items() if n.startswith('cross_attn_obs')}
    q64 = query.double().cpu().requires_grad_(True)
    k64 = key.double().cpu().requires_grad_(True)
    y64 = ref_f64(q64, k64, tmask.cpu(), W, None)
    (y64 * G.double().cpu()).sum().backward()
    assert float((yf - y64.detach()).abs().max()) < 1e-4
    assert float((dqf - q64.grad).norm() / q64.grad.norm()) < 2e-4
    assert float((dkf - k64.grad).norm() / k64.grad.norm()) < 2e-4
    for n in gf:
        assert float((gf[n] - W[n].grad).norm() / (W[n].grad.norm() + 1e-30)) < 3e-4, n
