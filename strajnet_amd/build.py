"""Builds libstrajnet_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libstrajnet_hip.so')
SOURCES = ['util.hip', 'gemm.hip', 'wgrad_sk.hip', 'norm.hip', 'patch_embed.hip', 'swin_attn.hip', 'swin_fused.hip', 'xattn_fused.hip', 'fgattn.hip', 'agent_fused.hip', 'fgoff_fused.hip', 'attn.hip', 'conv.hip', 'conv_ws.hip', 'conv_ps.hip', 'loss.hip', 'rng.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-Wno-unused-result', '-Wno-unused-value',
         '-ffp-contract=fast']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    objs, jobs = [], []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(HERE, 'build', s.replace('.hip', '.o'))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc] + FLAGS + ['-c', src, '-o', obj])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        return cmd, r
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for cmd, r in ex.map(run, jobs):
            if verbose and (r.stdout.strip() or r.stderr.strip()):
                sys.stderr.write(r.stdout + r.stderr)
            if r.returncode != 0:
                raise RuntimeError('hipcc failed: ' + ' '.join(cmd) + '\n' + r.stderr[-4000:])
    if jobs or force or _stale(OUT, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-s', '-o', OUT] + objs      # -s: no static symbol table (the C ABI is in .dynsym)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed: ' + r.stderr[-4000:])
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
