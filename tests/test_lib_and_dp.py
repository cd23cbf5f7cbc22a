"""CPU-side checks: the C-ABI library builds/loads and exports every symbol include/strajnet_hip.h declares (no compute
calls without a GPU); the product refuses CPU tensors; and the data-parallel semantics (replica-scaled loss + SUM
all-reduce of the flat gradient bucket == single-process global-batch gradient) hold with world_size 2 over gloo."""
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol(lib_built):
    import ctypes
    hdr = open(os.path.join(ROOT, 'include', 'strajnet_hip.h')).read()
    names = set(re.findall(r'\b(stj_[a-z0-9_]+)\s*\(', hdr))
    names -= {'stj_status', 'stj_dtype', 'stj_act', 'stj_unary'}
    assert len(names) >= 30
    L = ctypes.CDLL(lib_built)
    for n in sorted(names):
        assert hasattr(L, n), f'{n} declared in include/strajnet_hip.h but not exported'
    from strajnet_amd import _lib
    assert set(_lib.SIGNATURES) | {'stj_last_error'} == names, (set(_lib.SIGNATURES) | {'stj_last_error'}) ^ names
    assert _lib.lib().stj_abi_version() == 1


def test_header_argument_counts_match_binding():
    """Every prototype in the header has as many parameters as the ctypes signature table."""
    from strajnet_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'strajnet_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    for name, args in _lib.SIGNATURES.items():
        m = re.search(r'\b(?:int|long long)\s+' + name + r'\s*\((.*?)\)\s*;', hdr, flags=re.S)
        assert m, name
        params = [p for p in m.group(1).split(',') if p.strip() and p.strip() != 'void']
        assert len(params) == len(args), (name, len(params), len(args))


def test_product_has_no_cpu_fallback():
    from strajnet_amd import ops
    with pytest.raises(RuntimeError):
        ops.gelu(torch.zeros(16))
    import strajnet_amd.modules as m
    src = open(m.__file__).read() + open(ops.__file__).read()
    assert 'oracle' not in src.replace('the oracle', '')       # the product never imports the checker


def test_shard_batch():
    from strajnet_amd import dp
    b = {'x': torch.arange(8).view(8, 1), 'y': torch.arange(16).view(8, 2)}
    s = dp.shard_batch(b, rank_=1, world_=2)
    assert s['x'].flatten().tolist() == [4, 5, 6, 7]
    with pytest.raises(ValueError):
        dp.shard_batch({'x': torch.zeros(3, 1)}, rank_=0, world_=2)


_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["STJ_ROOT"])
import numpy as np, torch, torch.distributed as dist
from oracle import np_ref, torch_ref
from strajnet_amd import dp
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.set_num_threads(2)
cfg = dict(input_size=(128, 128), window_size=8, embed_dim=96, depths=[2, 2, 2], num_heads=[3, 6, 12])
w = np_ref.make_weights(cfg, 0)
x1 = np_ref.make_inputs(cfg, 1)
x = {k: np.concatenate([v, v], 0) for k, v in x1.items()}   # equal shards: the per-replica flow-pixel count (loss.py:292-294) equals the global mean
names = list(w)
def grads(batch, replica):
    p = torch_ref.to_torch(w, torch.float64, requires_grad=True)
    xt = torch_ref.to_torch(batch, torch.float64)
    y = torch_ref.forward(p, cfg, xt["ogm"], xt["map_img"], xt["obs"], xt["occ"], xt["flow"])
    d = torch_ref.loss(y, xt["gt_obs"], xt["gt_occ"], xt["gt_flow"], xt["origin_flow"], replica=replica, use_gt=False)
    sum(d.values()).backward()
    return torch.cat([p[n].grad.reshape(-1) for n in names])
mine = dp.shard_batch({k: torch.as_tensor(v) for k, v in x.items()})
flat = grads({k: v.numpy() for k, v in mine.items()}, float(world))     # loss pre-scaled by 1/replica (loss.py:200)
dp.allreduce_flat_grads(flat)                                           # SUM over replicas (MirroredStrategy)
if rank == 0:
    full = grads(x, 1.0)
    # every term is a mean over the (global) batch => replica-scaled loss + SUM all-reduce == single-process gradient
    err = float((flat - full).abs().max() / full.abs().max())
    print("DP_REL_ERR", err)
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_dp_world2_gloo(tmp_path):
    script = tmp_path / 'dp_worker.py'
    script.write_text(_WORKER)
    env = dict(os.environ, STJ_ROOT=ROOT, OMP_NUM_THREADS='2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    m = re.search(r'DP_REL_ERR ([0-9.e+-]+)', r.stdout)
    assert m, r.stdout[-2000:]
    assert float(m.group(1)) < 1e-9
