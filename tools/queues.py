#!/usr/bin/env python
"""Which hardware queue each kernel of ONE replayed step ran on (rocprofv3 rocpd database): usage tools/queues.py results.db [t0_us t1_us]"""
import sqlite3, sys
db = sys.argv[1]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print('# columns:', cols)
q = [x for x in ('queue_id', 'stream_id') if x in cols]
rows = list(c.execute(f"select start, end, {', '.join(q)}, name from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'nadam' in r[-1]]
a, b = idx[-3] + 1, idx[-2] + 1
step = rows[a:b]
t0 = step[0][0]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
for r in step:
    s = (r[0] - t0) / 1e3
    if lo <= s <= hi:
        print(f'{s:9.1f} {(r[1] - r[0]) / 1e3:8.1f}  q={r[2:-1]}  {r[-1][:90]}')
