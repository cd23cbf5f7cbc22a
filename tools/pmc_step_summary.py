#!/usr/bin/env python
"""Summarise the three rocprofv3 --pmc passes of tools/pmc_step.sh (whole bench step, serial mode), per kernel and per launch.
usage: pmc_step_summary.py <FETCH_SIZE csv> <WRITE_SIZE csv> <SQ csv> <kernel_trace csv of the SQ pass> [json_out]
Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts a 128-byte read
request as 64 bytes -> doubled here; the calibration copy of bench.py (STJ_BENCH_CALIB=1: 268,435,456 B read and written by torch's
copy kernel) is printed first so the correction can be checked in the same pass."""
import collections, csv, json, re, sys

# kernels that serve exactly one (layer, shape) of the bench step -> the key bench.py files that launch under (roofline.traffic)
BENCH_KEY = {'upconv_dgrad_ws2_kernel<true, true>': 'upconv_dgrad[128x128,96->48,F64]',
             'upconv_dgrad_ws2_kernel<true>': 'upconv_dgrad[128x128,96->48,F64]',
             'upconv_fwd_ws2_kernel<bf16, 3, 3, true>': 'upconv_fwd[128x128,96->48,F64]',
             'upconv_fwd_ws2_kernel<bf16, 3, 3>': 'upconv_fwd[128x128,96->48,F64]',
             'upconv_wgrad_tr4_kernel<3, 6, 32, 128>': 'upconv_wgrad[128x128,96->48,F64]',
             'upconv_wgrad_tr_kernel<3, 6, 32>': 'upconv_wgrad[128x128,96->48,F64]',
             'upconv_wgrad_tr4_kernel<4, 4, 32, 128>': 'upconv_wgrad[64x64,128->96,F64]',
             'outconv_bwd_mfma2_kernel<48>': 'outconv_bwd[256x256,48->2,F64]',
             'outconv_bwd_mfma_kernel<48>': 'outconv_bwd[256x256,48->2,F64]',
             'outconv_pair_fwd_kernel<bf16, 48>': 'outconv_pair_fwd[256x256,48->2,F64x2]',
             'outconv_fwd_mfma_kernel<bf16, 48>': 'outconv_fwd[256x256,48->2,F64]',
             'upconv_fwd_ws_kernel<bf16, 4, 2, 2>': 'upconv_fwd[64x64,128->96,F64]',
             'upconv_dgrad_ws_kernel<3, 4, false, 96>': 'upconv_dgrad[64x64,128->96,F64]',
             # dominant kernels of the extra configurations (tools/pmc_step.sh <tag> --infer / --cfg512)
             'upconv_fwd_ws2_kernel<f16, 3, 3, true, true>': 'upconv_fwd_head[128x128,96->48->18,F256]',
             'swin_mlp_bwd_kernel<bf16, 384, 1, 2, 4, 1>': 'swin_mlp_bwd[8192x384]'}


def short(n):
    n = re.sub(r'\(.*', '', n)
    n = n.replace('void ', '').replace('at::native::', 'at::')
    return n[:90]


def load(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        d[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return d


f, w, s = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[4])):
    dur[short(r['Kernel_Name'])].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
mean = lambda v: sum(v) / len(v) if v else 0.0
rows = []
for k in dur:
    rd = mean(f[k].get('FETCH_SIZE', [])) * 2 * 1024
    wr = mean(w[k].get('WRITE_SIZE', [])) * 1024
    us = mean(dur[k])
    sq = {c: mean(v) for c, v in s[k].items()}
    rows.append((sum(dur[k]), k, len(dur[k]), us, rd, wr, sq))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('# rocprofv3 --pmc, whole bench step in serial mode (every kernel alone), mean per launch; MB = 1e6 bytes; HBM peak 8 TB/s; MFMA busy =')
print('# SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs); conflicts = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES')
for k in f:        # the calibration copy is the largest single copy dispatch of the run
    if 'copy' in k.lower() and f[k].get('FETCH_SIZE') and max(f[k]['FETCH_SIZE']) * 2 * 1024 > 2.0e8:
        print(f"# calibration copy (268.4 MB each way): FETCH_SIZE x2 = {max(f[k]['FETCH_SIZE']) * 2 * 1024 / 1e6:.1f} MB, "
              f"WRITE_SIZE = {max(w[k].get('WRITE_SIZE', [0])) * 1024 / 1e6:.1f} MB  [{k[:40]}]")
print(f'{"% time":>7} {"launches":>8} {"us":>8} {"read MB":>9} {"write MB":>9} {"TB/s":>6} {"%HBM":>5} {"%MFMA":>6} {"%confl":>6} {"%wait":>6}  kernel')
out = {}
for t, k, n, us, rd, wr, sq in rows[:60]:
    tb = (rd + wr) / us / 1e6 if us else 0.0
    mf = sq.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (us * 2400 * 1024) * 100 if us else 0.0
    cf = sq.get('SQ_LDS_BANK_CONFLICT', 0.0) / sq['SQ_LDS_IDX_ACTIVE'] * 100 if sq.get('SQ_LDS_IDX_ACTIVE') else 0.0
    wt = sq.get('SQ_WAIT_ANY', 0.0) / sq['SQ_WAVE_CYCLES'] * 100 if sq.get('SQ_WAVE_CYCLES') else 0.0
    print(f'{t / tot * 100:7.2f} {n:8d} {us:8.1f} {rd / 1e6:9.1f} {wr / 1e6:9.1f} {tb:6.2f} {tb / 8 * 100:5.0f} {mf:6.1f} {cf:6.1f} {wt:6.1f}  {k}')
    if k in BENCH_KEY:
        out.setdefault(BENCH_KEY[k], round(rd + wr))
if len(sys.argv) > 5:
    json.dump(out, open(sys.argv[5], 'w'), indent=1)
