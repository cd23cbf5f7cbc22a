"""Grouped stream-K weight gradients (csrc/wgrad_sk.hip: stj_wgrad_group) against float64, for every (rows, cin, cout, batch)
the cfg-256 / cfg-512 train steps issue (list: `bench.py --gemm-trace`), in both orientations, alone and as one grouped launch.

The operands are bf16 / fp16 values, so every product is exact in f32 and the kernel differs from float64 by the f32 accumulation only:
the gate is |err| <= 2^-18 * sum_k |x_k dy_k| per element (measured ~3e-8 relative to that sum) -- an order of magnitude below what a
single dropped or doubled 32-row slab, tile column or flush would produce.  Reference: tape.gradient (train.py:223) of the Keras Dense
kernels / biases, modules.py:36-37,76-83,270-272, trajNet.py:71-77,195-211."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def _lib(lib_built):
    assert torch.cuda.is_available()

TOL = 2.0 ** -18

# rows, cin, cout, nb, shared_x
SHAPES = [
    (32768, 96, 128, 8, True), (32768, 96, 288, 1, False), (32768, 96, 384, 1, False), (32768, 384, 96, 1, False), (32768, 96, 96, 1, False),
    (2048, 384, 384, 1, False), (8192, 768, 192, 1, False), (2048, 1536, 384, 1, False), (2048, 384, 1536, 1, False),
    (2048, 384, 1152, 1, False), (8192, 192, 576, 1, False), (8192, 192, 768, 1, False), (2048, 512, 384, 8, False),
    (8192, 384, 192, 1, False), (8192, 192, 192, 1, False), (8192, 192, 192, 8, True), (2048, 128, 512, 8, False), (512, 384, 384, 1, False),
    (5632, 256, 320, 1, False), (32768, 176, 96, 1, False), (512, 1536, 384, 1, False), (2048, 432, 48, 8, False), (2048, 768, 384, 1, False),
    (512, 384, 1536, 1, False), (32768, 32, 96, 1, False), (32768, 48, 96, 1, False), (5632, 64, 64, 4, False), (512, 384, 64, 6, False),
    # odd corners: one slab, one tile column beyond a tile, widths that end inside a 16-column fragment
    (32, 8, 8, 1, False), (64, 104, 392, 1, False), (96, 392, 104, 3, False), (160, 200, 40, 2, True),
]


def _mk(rows, cin, cout, nb, shared_x, dt, seed, bias=True, ldpad=0, prefill=False):
    from strajnet_amd import ops
    g = torch.Generator(device='cuda'); g.manual_seed(seed)
    ldx, lddy = cin + ldpad, cout + ldpad
    x = torch.randn((1 if shared_x else nb, rows, ldx), device='cuda', generator=g).to(dt)
    dy = torch.randn((nb, rows, lddy), device='cuda', generator=g).to(dt)
    dw = torch.randn((nb, cin, cout), device='cuda', generator=g) if prefill else torch.zeros((nb, cin, cout), device='cuda')
    db = (torch.randn((nb, cout), device='cuda', generator=g) if prefill else torch.zeros((nb, cout), device='cuda')) if bias else None
    dw0, db0 = dw.clone(), (db.clone() if bias else None)
    j = ops.WJob(x, dy, dw, db, rows, cin, cout, ldx, lddy, cout, ops.DTYPE_CODE[dt], nb=(1, nb), sx=(0, 0 if shared_x else rows * ldx),
                 sdy=(0, rows * lddy), sdw=(0, cin * cout), sdb=(0, cout))
    return j, dw0, db0


def _check(j, dw0, db0):
    nb, cin, cout = j.nb[1], j.cin, j.cout
    x = j.x.double()[..., :cin].expand(nb, -1, -1)
    dy = j.dy.double()[..., :cout]
    ref = torch.einsum('zri,zro->zio', x, dy)
    mag = torch.einsum('zri,zro->zio', x.abs(), dy.abs())
    err = ((j.dw.double() - dw0.double() - ref).abs() - TOL * mag - 1e-6 * dw0.abs().double()).max().item()
    assert err <= 0, (j.rows, cin, cout, nb, 'dW', err)
    if j.db is not None:
        errb = ((j.db.double() - db0.double() - dy.sum(1)).abs() - TOL * dy.abs().sum(1) - 1e-6 * db0.abs().double()).max().item()
        assert errb <= 0, (j.rows, cin, cout, nb, 'db', errb)


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
def test_every_model_shape_alone(dt):
    from strajnet_amd import ops
    for i, (rows, cin, cout, nb, sh) in enumerate(SHAPES):
        j, dw0, db0 = _mk(rows, cin, cout, nb, sh, dt, i)
        assert j.supported(), (rows, cin, cout)
        ops.wgrad_group([j])
        torch.cuda.synchronize()
        _check(j, dw0, db0)


@pytest.mark.parametrize('budget', [0, 64, 7])
def test_grouped_launch_accumulates_into_prefilled_gradients(budget):
    """All shapes in one call (two launches of <= 28 jobs), gradients pre-filled (the kernel ADDS), padded row strides, some jobs without
    a bias gradient, on a full / partial / odd number of workgroups."""
    from strajnet_amd import ops
    dt = torch.bfloat16
    jobs = [_mk(r, ci, co, nb, sh, dt, 100 + i, bias=(i % 3 != 0), ldpad=8 * (i % 2), prefill=True) for i, (r, ci, co, nb, sh) in enumerate(SHAPES)]
    ops.wgrad_group([j for j, _, _ in jobs], budget=budget)
    torch.cuda.synchronize()
    for j, dw0, db0 in jobs:
        _check(j, dw0, db0)


# problems at least 192 wide on both sides: the launches that take the 192-column / eight-wave tile geometry when they are long enough
# (>= 32 units per workgroup: cfg-512's 8192-row C = 384 stage, its 32768-row C = 192 stage), incl. ragged widths and one-slab problems
WIDE = [(8192, 384, 1152, 1, False), (8192, 384, 384, 1, False), (8192, 384, 1536, 1, False), (8192, 1536, 384, 1, False),
        (32768, 192, 576, 1, False), (32768, 768, 192, 1, False), (2048, 200, 392, 2, False), (4096, 392, 200, 1, True), (32, 192, 192, 1, False),
        (1024, 584, 776, 1, False)]


@pytest.mark.parametrize('dt,budget', [(torch.bfloat16, 0), (torch.bfloat16, 7), (torch.float16, 64)])
def test_wide_launches_take_the_192_column_tiles(dt, budget):
    """One grouped launch of WIDE (pre-filled gradients, padded strides, some jobs without a bias gradient) -- long enough on every budget
    here to select wgrad_sk_kernel<T, 192> -- and each problem of the cfg-512 kind alone on 7 workgroups."""
    from strajnet_amd import ops
    jobs = [_mk(r, ci, co, nb, sh, dt, 300 + i, bias=(i % 3 != 1), ldpad=8 * (i % 2), prefill=True) for i, (r, ci, co, nb, sh) in enumerate(WIDE)]
    ops.wgrad_group([j for j, _, _ in jobs], budget=budget)
    torch.cuda.synchronize()
    for j, dw0, db0 in jobs:
        _check(j, dw0, db0)
    for i, (r, ci, co, nb, sh) in enumerate(WIDE[:6]):
        j, dw0, db0 = _mk(r, ci, co, nb, sh, dt, 400 + i)
        ops.wgrad_group([j], budget=7)
        torch.cuda.synchronize()
        _check(j, dw0, db0)


def test_same_parameter_twice_and_unsupported_shapes_fall_back():
    """Two jobs adding into ONE gradient (a weight used twice), plus shapes the stream-K kernel refuses (width 42, rows % 32 != 0, f32):
    wgrad_group sends those through stj_gemm and the sum is still right."""
    from strajnet_amd import ops
    dt = torch.bfloat16
    a, dw0, db0 = _mk(2048, 96, 384, 1, False, dt, 7)
    b, _, _ = _mk(4096, 96, 384, 1, False, dt, 8)
    b.dw, b.db = a.dw, a.db
    odd, odw0, odb0 = _mk(2048, 384, 42, 3, False, dt, 9)
    odd2, o2dw0, o2db0 = _mk(1000, 64, 64, 1, False, dt, 10)
    assert not odd.supported() and not odd2.supported()
    f32j, fdw0, fdb0 = _mk(512, 96, 96, 1, False, torch.float32, 11)
    assert not f32j.supported()
    ops.wgrad_group([a, b, odd, odd2])
    ops.wgrad_group([f32j])
    torch.cuda.synchronize()
    ref = a.x.double()[0].T @ a.dy.double()[0] + b.x.double()[0].T @ b.dy.double()[0]
    mag = a.x.double()[0].abs().T @ a.dy.double()[0].abs() + b.x.double()[0].abs().T @ b.dy.double()[0].abs()
    assert ((a.dw.double()[0] - ref).abs() - TOL * mag).max().item() <= 0
    refb = a.dy.double()[0].sum(0) + b.dy.double()[0].sum(0)
    assert ((a.db.double()[0] - refb).abs() - TOL * (a.dy.double()[0].abs().sum(0) + b.dy.double()[0].abs().sum(0))).max().item() <= 0
    for j, d0, b0 in ((odd, odw0, odb0), (odd2, o2dw0, o2db0)):
        x = j.x.double().expand(j.nb[1], -1, -1)
        ref = torch.einsum('zri,zro->zio', x, j.dy.double())
        assert (j.dw.double() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


def test_deferred_queue_in_a_backward_pass_matches_immediate_launches():
    """The same Dense layers with the queue on (STJ_WGRAD_SK default: weight gradients leave at the flush) and off: identical gradients
    up to f32 summation order; and outside a model's backward pass nothing is ever deferred."""
    from strajnet_amd import ops
    from test_ops_gpu import mk_param, rnd
    dt = torch.bfloat16
    grads = []
    for defer in (False, True):
        pw, pb = mk_param((96, 288), dt, 0.1, 1), mk_param((288,), dt, 0.1, 2)
        pw2, pb2 = mk_param((288, 96), dt, 0.1, 3), mk_param((96,), dt, 0.1, 4)
        x = rnd((4096, 96), dt, 5).requires_grad_(True)
        y = ops.linear(ops.linear(x, pw, pb, act=ops.ACT_ELU), pw2, pb2)
        if defer:
            y = ops.join_after_backward(y, (), lambda: None)      # the node a model's output passes through: opens / closes the queue
        y.backward(rnd((4096, 96), dt, 6))
        torch.cuda.synchronize()
        assert not ops._wq()['on'] and not ops._wq()['jobs']
        grads.append([t.grad.clone() for t in (pw, pb, pw2, pb2)] + [x.grad.float().clone()])
    for a, b in zip(*grads):
        assert (a - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item()), (a - b).abs().max().item()
