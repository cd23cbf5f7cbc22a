"""The stage flushes of the encoder backward (one stj_wgrad_group launch per Swin stage: 2 blocks x 4 Dense weight gradients) against the
launch's workgroup budget: with 2048 / 8192 rows a 256-workgroup stream-K launch cuts every 96 x 384 tile into 2-3 visits that each end in
36864 f32 atomics.   usage: python tools/probes/wgrad_sk_budget.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from strajnet_amd import ops
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_wgrad_sk import make, timeit

for name, rows, C, nblk in (('stage 2 (16x16, C=384)', 2048, 384, 2), ('stage 1 (32x32, C=192)', 8192, 192, 2), ('stage 0 (64x64, C=96)', 32768, 96, 4),
                             ('cfg-512 stage 2 (32x32, C=384, 6 blocks)', 8192, 384, 6), ('cfg-512 stage 1 (64x64, C=192)', 32768, 192, 2)):
    jobs = []
    for b in range(nblk):
        for cin, cout in ((C, 3 * C), (C, C), (C, 4 * C), (4 * C, C)):
            jobs.append(make(rows, cin, cout, seed=len(jobs)))
    by = sum(2 * j.rows * (j.cin + j.cout) + 4 * j.cin * j.cout for j in jobs)
    res = []
    ops.wgrad_group(jobs); torch.cuda.synchronize()
    for budget in (256, 192, 128):
        t = timeit(lambda: ops.wgrad_group(jobs, budget=budget), iters=10)
        res.append(f'{budget}: {t:6.1f} us')
    print(f'{name}: {len(jobs)} jobs, {by / 1e6:.0f} MB   ' + '  '.join(res), flush=True)
