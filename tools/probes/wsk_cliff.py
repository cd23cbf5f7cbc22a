import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench_wgrad_sk import make, timeit, check
from strajnet_amd import ops
for (r, ci, co) in [(8192, 768, 192), (16384, 768, 192), (32768, 768, 192), (32768, 384, 192), (32768, 768, 96), (32768, 768, 384), (32768, 192, 768), (32768, 1536, 384), (131072, 384, 96), (65536, 768, 192)]:
    j = make(r, ci, co)
    sup = j.supported()
    t1 = timeit(lambda: ops.wgrad_group([j]), 5)
    j.dw.zero_(); j.db.zero_(); ops.wgrad_group([j]); torch.cuda.synchronize()
    e = check(j)
    print(f'[{r} x {ci} -> {co}] supported={sup}: {t1:8.1f} us  {r * (ci + co) * 2 / t1 / 1e3:7.1f} GB/s  err {e[0]:.1e} {e[1]:.1e}')
    del j
