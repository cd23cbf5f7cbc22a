#!/usr/bin/env python
"""Dump the kernel timeline of ONE replayed step from a rocprofv3 rocpd database (start offset, duration, stream / queue, kernel name):
which kernels overlap, where the GPU idles, what the critical chain is.   usage: tools/timeline.py results.db [step_index_from_end] [name of
the step's LAST kernel: default 'nadam' (train step); 'outconv_pair_gather' for the inference forward]"""
import sqlite3, sys
db = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
last = sys.argv[3] if len(sys.argv) > 3 else 'nadam'
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
sid = 'stream_id' if 'stream_id' in cols else ('queue_id' if 'queue_id' in cols else None)
q = f"select start, end, {sid if sid else 0}, name from kernels order by start"
rows = list(c.execute(q))
# steps are separated by the Nadam kernel
idx = [i for i, r in enumerate(rows) if last in r[3]]
if len(idx) < back + 1:
    print('not enough steps', len(idx)); sys.exit(1)
a, b = idx[-back - 1] + 1, idx[-back] + 1
step = rows[a:b]
t0 = step[0][0]
busy_end = t0
idle = 0.0
print(f'# {len(step)} kernels, wall {(step[-1][1] - t0) / 1e3:.1f} us, kernel time sum {sum(r[1] - r[0] for r in step) / 1e3:.1f} us; columns: start_us dur_us stream gap_before_us(idle GPU) name')
for s, e, st, n in step:
    gap = max(0.0, (s - busy_end) / 1e3)
    idle += gap
    busy_end = max(busy_end, e)
    print(f'{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {st!s:>6} {gap:7.1f}  {n[:110]}')
print(f'# GPU idle (no kernel running) inside the step: {idle:.1f} us')
# ---- "thin" stretches: intervals in which every running kernel is a short one (< THIN us): dependent chains of small launches
# that run alone, i.e. dispatch-latency-bound time.  Listed when longer than 50 us.
THIN = 16.0
ev = []
for s, e, st, n in step:
    ev.append((s, 1, (e - s) / 1e3 < THIN, n))
    ev.append((e, -1, (e - s) / 1e3 < THIN, n))
ev.sort(key=lambda x: (x[0], x[1]))
big = small = 0
cur_start, cur_n, total_thin = None, 0, 0.0
stretches = []
prev_t = ev[0][0]
names = []
for t, d, is_small, n in ev:
    thin_now = big == 0                      # (idle counts as thin: nothing substantial is running)
    if thin_now:
        total_thin += (t - prev_t) / 1e3
        if cur_start is None:
            cur_start, names = prev_t, []
    elif cur_start is not None:
        stretches.append((cur_start, prev_t, names)); cur_start = None
    prev_t = t
    if is_small:
        small += d
        if d > 0 and big == 0:
            names.append(n.split('(')[0][-40:])
    else:
        big += d
        if big > 0 and cur_start is not None:
            stretches.append((cur_start, t, names)); cur_start = None
print(f'# time with no kernel >= {THIN:.0f} us running: {total_thin:.1f} us')
for a_, b_, nm in stretches:
    if (b_ - a_) / 1e3 > 50:
        print(f'#   thin {(a_ - t0) / 1e3:8.1f} .. {(b_ - t0) / 1e3:8.1f}  ({(b_ - a_) / 1e3:6.1f} us, {len(nm)} small kernels)')
# ---- wall-time attribution: every instant of the step is split equally among the kernels running at it; summed per kernel name.
# (A kernel that always overlaps another costs the step half its duration; the table shows where the WALL time of the concurrent step
# goes, next to each kernel's summed duration.)
import re
pts = sorted(set([s for s, e, st, n in step] + [e for s, e, st, n in step]))
idx = {t: i for i, t in enumerate(pts)}
cnt = [0] * len(pts)
for s, e, st, n in step:
    for i in range(idx[s], idx[e]):
        cnt[i] += 1
share, dur = {}, {}
for s, e, st, n in step:
    k = re.sub(r'\(.*', '', n.replace('void ', ''))[:70]
    w = 0.0
    for i in range(idx[s], idx[e]):
        w += (pts[i + 1] - pts[i]) / cnt[i]
    share[k] = share.get(k, 0.0) + w / 1e3
    dur[k] = dur.get(k, 0.0) + (e - s) / 1e3
print(f'# wall-time attribution (us of the {(step[-1][1] - t0) / 1e3:.0f} us step; summed kernel duration in brackets):')
for k, v in sorted(share.items(), key=lambda kv: -kv[1])[:28]:
    print(f'#   {v:8.1f}  [{dur[k]:8.1f}]  {k}')
