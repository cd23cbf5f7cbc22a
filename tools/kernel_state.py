#!/usr/bin/env python
"""Current-state table of the kernels one train step runs, generated from a profile set under profiles/ (DESIGN.md section 0 embeds it).

    python tools/kernel_state.py <tag> [--min-us 25]        e.g.  tools/kernel_state.py r06_k  >  profiles/r06_k_kernel_state.md

Inputs (written by tools/final_artifacts.sh / trace.sh / pmc_step.sh):
    profiles/<tag>_pmc_step.txt                        per kernel ALONE (serial mode): launches, us, MB read / written, %HBM, %MFMA, %LDS conflicts, %wait
    profiles/<tag>_kernel_trace_concurrent_b8_bf16.txt the same kernels as scheduled in the replayed step: us per step, launches per step, average us
Columns: kernel (template arguments kept), launches per step, us alone / in the step, us per step in the step, % of 8 TB/s, % MFMA busy,
what the counters say bounds it alone, and the round that last changed the kernel (LAST_TOUCHED below: maintained by hand with the code)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel-name prefix -> (round that last changed it, one-line note)
LAST_TOUCHED = [
    ('agf::agent_', 6, 'new: fused agent branch (csrc/agent_fused.hip)'),
    ('fgo::fgoff_', 6, 'new: fused FG-MSA offset head (csrc/fgoff_fused.hip)'),
    ('skip_junction_bwd_kernel', 6, 'new: the decoder level\'s three ELU\' products in one pass'),
    ('unary_', 6, 'one vector per thread'), ('elu_res_bwd', 6, 'one vector per thread'), ('cast_kernel', 6, 'one vector per thread'),
    ('upconv_fwd_ps_kernel', 6, 'half-chunk weight staging, 16-row tiles'),
    ('wsk::wgrad_sk_kernel', 5, 'second tile geometry (192-column slices)'),
    ('upconv_wgrad_tr4_kernel', 5, 'accumulation-buffer alignment'),
    ('upconv_wgrad_tr_kernel', 4, 'k-permutation / conflict-free strides'),
    ('outconv_bwd_mfma2_kernel', 4, 'partials + reduce'),
    ('outconv_bwd_reduce_kernel', 4, ''),
    ('upconv_dgrad_ws2_kernel', 6, 'halo-row-major fragment order, 3-deep ring'),
    ('upconv_fwd_ws2_kernel', 6, 'halo-row-major fragment order, epilogue under the next row\'s MFMAs, 6-deep ring'),
    ('loss_fwd_bwd_kernel', 6, 'new: forward sums and d/dlogits in one pass'),
    ('loss_flow_count_kernel', 6, ''), ('loss_coef_kernel', 6, ''),
    ('upconv_dgrad_ws_kernel', 4, 'rolling halo, one barrier per pass'),
    ('upconv_fwd_ws_kernel', 5, 'XCD-aware cout groups'),
    ('outconv_pair_fwd_kernel', 2, ''),
    ('upconv_dgrad_pf_kernel', 2, 'pipelined persistent form'),
    ('swin_attn_', 5, 'in-launch slice combine, prologue latency'),
    ('swin_mlp_', 5, 'eight-wave workgroups, in-launch slice combine'),
    ('swin_split_', 5, ''),
    ('xat::xattn_', 3, 'fused Cross_AttentionT x 8'),
    ('fga::fgattn_', 5, 'f32 instantiation'),
    ('gemm_group_kernel', 2, ''), ('gemm_deepk_kernel', 2, ''), ('gemm_kernel', 2, ''), ('linear_rs_kernel', 2, ''),
    ('pe::patch_embed', 4, 'fused stem'),
    ('loss_', 2, ''), ('auc_', 2, ''), ('nadam_kernel', 2, ''),
    ('fg_offset_', 2, ''), ('ln_', 3, ''), ('unary_', 2, ''), ('elu_res_bwd', 2, ''),
    ('upconv_prep_kernel', 2, ''), ('upconv_fold_kernel', 2, ''), ('cast_kernel', 1, ''),
]


def short(name):
    name = re.sub(r'^void ', '', name.strip())
    name = re.sub(r'\(.*$', '', name)
    name = name.replace('at::native::', 'at::').replace('(anonymous namespace)::', '')
    return name[:70]


def touched(name):
    for pre, rnd, note in LAST_TOUCHED:
        if name.startswith(pre):
            return rnd, note
    return None, ''


def read_pmc(path):
    rows = {}
    for ln in open(path):
        if ln.startswith('#') or '% time' in ln or not ln.strip():
            continue
        f = ln.split(None, 10)
        if len(f) < 11:
            continue
        try:
            rows[short(f[10])] = dict(launches=int(f[1]), us=float(f[2]), rd=float(f[3]), wr=float(f[4]), tbs=float(f[5]), hbm=float(f[6]),
                                      mfma=float(f[7]), confl=float(f[8]), wait=float(f[9]))
        except ValueError:
            continue
    return rows


def read_trace(path):
    rows = {}
    for ln in open(path):
        if ln.startswith('#') or 'us/step' in ln or not ln.strip():
            continue
        f = ln.split(None, 6)
        if len(f) < 7:
            continue
        if '__amd_rocclr' in f[6]:        # runtime blit kernels of the bench's input-feed measurement (host -> device pieces), not of the step
            continue
        try:
            rows[short(f[6])] = dict(us_step=float(f[1]), calls=float(f[2]), avg=float(f[3]))
        except ValueError:
            continue
    return rows


def bound(p):
    if p is None:
        return ''
    if p['hbm'] >= 45 and p['hbm'] >= 1.2 * p['mfma']:
        return 'HBM'
    if p['mfma'] >= 35 and p['mfma'] > p['hbm']:
        return 'MFMA'
    if p['hbm'] >= 25 and p['mfma'] >= 20:
        return 'HBM + MFMA (neither saturated)'
    if p['confl'] >= 30:
        return 'LDS (bank conflicts)'
    if p['wait'] >= 50:
        return 'latency (waves waiting)'
    return 'L2 / issue (no counter saturated)'


def main():
    tag = sys.argv[1]
    min_us = float(sys.argv[sys.argv.index('--min-us') + 1]) if '--min-us' in sys.argv else 25.0
    pmc = read_pmc(os.path.join(ROOT, 'profiles', f'{tag}_pmc_step.txt'))
    tr = read_trace(os.path.join(ROOT, 'profiles', f'{tag}_kernel_trace_concurrent_b8_bf16.txt'))
    total = sum(v['us_step'] for v in tr.values())
    print(f'<!-- generated by tools/kernel_state.py {tag}: do not edit by hand -->')
    print(f'Kernels of one B = 8 bf16 train step with at least {min_us:.0f} us per step in the replayed step (profile set `profiles/{tag}_*`; '
          f'{len(tr)} distinct kernels, {sum(v["calls"] for v in tr.values()):.0f} launches, {total / 1e3:.2f} ms of summed in-step kernel time).')
    print()
    print('| kernel | launches / step | us alone | us in step | us / step (in step) | % of 8 TB/s | % MFMA busy | bound (alone) | last touched |')
    print('|---|---|---|---|---|---|---|---|---|')
    for k, v in sorted(tr.items(), key=lambda kv: -kv[1]['us_step']):
        if v['us_step'] < min_us:
            continue
        p = pmc.get(k)
        rnd, note = touched(k)
        print(f"| `{k}` | {v['calls']:.0f} | {p['us']:.0f} | {v['avg']:.0f} | {v['us_step']:.0f} | {p['hbm']:.0f} | {p['mfma']:.0f} | {bound(p)} | "
              f"round {rnd}{': ' + note if note else ''} |" if p else
              f"| `{k}` | {v['calls']:.0f} | - | {v['avg']:.0f} | {v['us_step']:.0f} | - | - | | {'round %d' % rnd if rnd else ''} |")
    rest = sum(v['us_step'] for v in tr.values() if v['us_step'] < min_us)
    print(f'| (everything below {min_us:.0f} us per step) | {sum(v["calls"] for v in tr.values() if v["us_step"] < min_us):.0f} | | | {rest:.0f} | | | | |')


if __name__ == '__main__':
    main()
