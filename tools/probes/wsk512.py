"""wgrad_sk on the cfg-512 stage shapes: grouped (as the flush issues them) and alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench_wgrad_sk import make, timeit
from strajnet_amd import ops
STAGES = {'s0 131072 rows': [(131072, 96, 288), (131072, 96, 96), (131072, 96, 384), (131072, 384, 96)] * 2,
          's1 32768 rows': [(32768, 192, 576), (32768, 192, 192), (32768, 192, 768), (32768, 768, 192)] * 2,
          's2 8192 rows': [(8192, 384, 1152), (8192, 384, 384), (8192, 384, 1536), (8192, 1536, 384)] * 6,
          'cfg256 s0 32768 rows': [(32768, 96, 288), (32768, 96, 96), (32768, 96, 384), (32768, 384, 96)] * 2}
for name, shapes in STAGES.items():
    jobs = [make(r, ci, co, seed=i) for i, (r, ci, co) in enumerate(shapes)]
    by = sum(r * (ci + co) * 2 for r, ci, co in shapes)
    t = timeit(lambda: ops.wgrad_group(jobs), 10)
    print(f'{name}: group of {len(jobs)}: {t:8.1f} us  {by / t / 1e3:7.1f} GB/s')
    for j, (r, ci, co) in zip(jobs[:4], shapes[:4]):
        t1 = timeit(lambda: ops.wgrad_group([j]), 10)
        print(f'    alone [{r} x {ci} -> {co}]: {t1:8.1f} us  {r * (ci + co) * 2 / t1 / 1e3:7.1f} GB/s')
    del jobs
    torch.cuda.empty_cache()
