"""Stream-K grouped weight gradients (stj_wgrad_group) against the split-K tile GEMM (stj_gemm), on the weight-gradient problems the
cfg-256 B=8 train step issues (shapes from `bench.py --gemm-trace`).  Prints us per launch, GB/s of algorithmic bytes, max error vs f64."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from strajnet_amd import ops

dt = torch.bfloat16
dev = 'cuda'


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def make(rows, cin, cout, nb=1, shared_x=False, bias=True, seed=0):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    x = torch.randn((1 if shared_x else nb, rows, cin), device=dev, generator=g).to(dt)
    dy = torch.randn((nb, rows, cout), device=dev, generator=g).to(dt)
    dw = torch.zeros((nb, cin, cout), device=dev)
    db = torch.zeros((nb, cout), device=dev) if bias else None
    j = ops.WJob(x, dy, dw, db, rows, cin, cout, cin, cout, cout, 1, nb=(1, nb), sx=(0, 0 if shared_x else rows * cin), sdy=(0, rows * cout),
                 sdw=(0, cin * cout), sdb=(0, cout))
    return j


def check(j):
    x, dy = j.x.double(), j.dy.double()
    ref = torch.einsum('zri,zro->zio', x.expand(dy.shape[0], -1, -1), dy)
    mag = torch.einsum('zri,zro->zio', x.abs().expand(dy.shape[0], -1, -1), dy.abs())
    e = ((j.dw.double() - ref).abs() / (mag + 1e-30)).max().item()
    eb = 0.0
    if j.db is not None:
        eb = ((j.db.double() - dy.sum(1)).abs() / (dy.abs().sum(1) + 1e-30)).max().item()
    return e, eb


SHAPES = [  # rows, cin, cout, nb, shared_x, count per step
    (32768, 96, 128, 8, True, 2), (32768, 96, 288, 1, False, 4), (32768, 96, 384, 1, False, 4), (32768, 384, 96, 1, False, 4),
    (32768, 96, 96, 1, False, 4), (2048, 384, 384, 1, False, 6), (8192, 768, 192, 1, False, 2), (2048, 1536, 384, 1, False, 2),
    (2048, 384, 1536, 1, False, 2), (2048, 384, 1152, 1, False, 2), (8192, 192, 576, 1, False, 2), (8192, 192, 768, 1, False, 2),
    (2048, 512, 384, 8, False, 1), (8192, 384, 192, 1, False, 2), (8192, 192, 192, 1, False, 2), (8192, 192, 192, 8, True, 1),
    (2048, 128, 512, 8, False, 1), (512, 384, 384, 1, False, 2), (5632, 256, 320, 1, False, 1), (32768, 176, 96, 1, False, 1),
    (512, 1536, 384, 1, False, 1), (2048, 432, 48, 8, False, 1), (2048, 768, 384, 1, False, 1), (512, 384, 1536, 1, False, 1),
    (32768, 32, 96, 1, False, 1), (32768, 48, 96, 1, False, 1), (5632, 64, 64, 4, False, 3), (512, 384, 64, 6, False, 3),
]


def main():
    tot_old = tot_new = 0.0
    alljobs = []
    print(f'{"rows":>6} {"cin":>5} {"cout":>5} {"nb":>3} | {"gemm us":>8} {"sk us":>8} {"GB/s":>7} | rel err dW, db')
    for rows, cin, cout, nb, sh, cnt in SHAPES:
        j = make(rows, cin, cout, nb, sh)
        assert j.supported(), (rows, cin, cout)
        ops.wgrad_group([j]); torch.cuda.synchronize()
        e, eb = check(j)
        t_old = timeit(lambda: j.gemm())
        t_new = timeit(lambda: ops.wgrad_group([j]))
        by = 2 * rows * (cin * (1 if sh else nb) + cout * nb) + 4 * cin * cout * nb
        print(f'{rows:6d} {cin:5d} {cout:5d} {nb:3d} | {t_old:8.1f} {t_new:8.1f} {by / t_new / 1e3:7.0f} | {e:.2e} {eb:.2e}', flush=True)
        tot_old += cnt * t_old; tot_new += cnt * t_new
        for c in range(cnt):
            alljobs.append(make(rows, cin, cout, nb, sh, seed=c + 1))
    print(f'sum over the step, one launch per problem: gemm {tot_old:.0f} us, stream-K {tot_new:.0f} us')
    by = sum(2 * j.rows * (j.cin * (j.nb[1] if j.sx[1] else 1) + j.cout * j.nb[1]) + 4 * j.cin * j.cout * j.nb[1] for j in alljobs)
    for budget in (0, 128, 64):
        t = timeit(lambda: ops.wgrad_group(alljobs, budget=budget), iters=10)
        print(f'all {len(alljobs)} problems of the step grouped (launches of 28), budget {budget or 256} workgroups: {t:.0f} us, {by / t / 1e3:.0f} GB/s')
    st0 = [j for j in alljobs if j.rows == 32768]
    t = timeit(lambda: ops.wgrad_group(st0), iters=10)
    by0 = sum(2 * j.rows * (j.cin * (j.nb[1] if j.sx[1] else 1) + j.cout * j.nb[1]) for j in st0)
    print(f'the {len(st0)} problems over 32768 rows in one launch: {t:.0f} us, {by0 / t / 1e3:.0f} GB/s')


if __name__ == '__main__':
    main()
