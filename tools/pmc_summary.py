#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes of tools/pmc_probe.py.
usage: python tools/pmc_summary.py <fetch_csv> <write_csv> <sq_csv> [json_out] [kernel_trace_csv of the SQ pass]
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-byte read requests as 64 bytes
(MI355X_MICROARCH.md, HBM section) -> doubled here, and the calibration copy in the same pass (268,435,456 B read and
written by hipMemcpy's copyBuffer kernel) is printed so the correction can be checked."""
import collections, csv, json, sys

KEEP = ('upconv_fwd_ws', 'upconv_dgrad_ws', 'upconv_wgrad_tr', 'copyBuffer')
LABEL = {'upconv_fwd_ws': 'upconv_fwd[128x128,96->48]', 'upconv_dgrad_ws': 'upconv_dgrad[128x128,96->48]', 'upconv_wgrad_tr': 'upconv_wgrad[128x128,96->48]'}


def load(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        for k in KEEP:
            if k in r['Kernel_Name']:
                d[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in d.items()}


f, w, s = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
out = {}
print('# per launch, mean of the launches in the pass; MB = 1e6 bytes')
print(f"calibration copy (268.4 MB each way): FETCH_SIZE x2 = {f[('copyBuffer', 'FETCH_SIZE')] * 2 * 1024 / 1e6:.1f} MB, WRITE_SIZE = {w[('copyBuffer', 'WRITE_SIZE')] * 1024 / 1e6:.1f} MB")
for k in KEEP[:3]:
    rd, wr = f[(k, 'FETCH_SIZE')] * 2 * 1024, w[(k, 'WRITE_SIZE')] * 1024
    out[LABEL[k]] = round(rd + wr)
    sq = {c: v for (kk, c), v in s.items() if kk == k}
    print(f"{k}: read {rd / 1e6:.1f} MB + written {wr / 1e6:.1f} MB = {(rd + wr) / 1e6:.1f} MB")
    wc = sq.get('SQ_WAVE_CYCLES', 0.0)
    for c in sorted(sq):
        print(f"    {c:28s} {sq[c]:14.0f}" + (f"  ({sq[c] / wc * 100:5.1f} % of wave cycles)" if wc and c.startswith(('SQ_WAIT', 'SQ_ACTIVE')) else ''))
    if sq.get('SQ_LDS_IDX_ACTIVE'):
        print(f"    LDS bank-conflict share of LDS cycles: {sq['SQ_LDS_BANK_CONFLICT'] / sq['SQ_LDS_IDX_ACTIVE'] * 100:.1f} %")
if len(sys.argv) > 5:          # durations of the same launches -> HBM rate and MFMA utilisation
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[5])):
        for k in KEEP[:3]:
            if k in r['Kernel_Name']:
                dur[k].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
    print('# derived (duration from the kernel trace of the SQ pass; 256 CUs x 4 SIMDs at 2.4 GHz; HBM peak 8 TB/s):')
    for k in KEEP[:3]:
        if not dur[k]:
            continue
        us = sum(dur[k]) / len(dur[k])
        mfma = s.get((k, 'SQ_VALU_MFMA_BUSY_CYCLES'), 0.0)
        print(f"{k}: {us:.1f} us per launch; traffic {out[LABEL[k]] / us / 1e6:.2f} TB/s = {out[LABEL[k]] / us / 1e6 / 8 * 100:.0f} % of HBM peak; "
              f"MFMA busy {mfma / (us * 2400 * 1024) * 100:.1f} % of SIMD cycles")
if len(sys.argv) > 4 and sys.argv[4] != '-':
    json.dump(out, open(sys.argv[4], 'w'), indent=1)
