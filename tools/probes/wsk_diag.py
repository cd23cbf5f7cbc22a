import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from strajnet_amd import ops
from bench_wgrad_sk import make, timeit
def by(jobs): return sum(2 * j.rows * (j.cin * (j.nb[1] if j.sx[1] else 1) + j.cout * j.nb[1]) for j in jobs)
sets = {
 'stage0 x4 (96x288,96x384,384x96,96x96)': [make(32768, ci, co, 1, False, seed=i) for i in range(4) for ci, co in ((96, 288), (96, 384), (384, 96), (96, 96))],
 '96x384 x8': [make(32768, 96, 384, 1, False, seed=i) for i in range(8)],
 '96x96 x16': [make(32768, 96, 96, 1, False, seed=i) for i in range(16)],
 'resconv (96x128 b8 x2, 192x192 b8)': [make(32768, 96, 128, 8, True), make(32768, 96, 128, 8, True, seed=1), make(8192, 192, 192, 8, True)],
 'stage2 (384-wide, 2048 rows)': [make(2048, ci, co, 1, False, seed=i) for i in range(2) for ci, co in ((384, 1152), (384, 384), (384, 1536), (1536, 384))],
}
for name, jobs in sets.items():
    for budget in (256, 128):
        t = timeit(lambda: ops.wgrad_group(jobs, budget=budget), iters=10)
        print(f'DBG={os.environ.get("STJ_WGRAD_SK_DBG","0")} {name:45s} G={budget}: {t:7.1f} us {by(jobs)/t/1e3:6.0f} GB/s', flush=True)
