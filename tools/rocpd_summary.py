#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / average, like `--stats`.
usage: python tools/rocpd_summary.py results.db [steps_in_trace] > profiles/summary.txt"""
import sqlite3
import sys

db = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = sqlite3.connect(db)
rows = list(c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                      "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f'# rocprofv3 --kernel-trace summary of {db}; {steps:g} steps in trace')
print(f'# total kernel time {tot/1e3:.3f} ms ({tot/steps/1e3:.3f} ms/step), {sum(r[1] for r in rows)} dispatches, {len(rows)} distinct kernels')
print(f'{"%":>6} {"us/step":>10} {"calls/step":>10} {"avg_us":>9} {"min_us":>9} {"max_us":>9}  kernel')
for r in rows:
    print(f'{r[2]/tot*100:6.2f} {r[2]/steps:10.1f} {r[1]/steps:10.1f} {r[3]:9.1f} {r[4]:9.1f} {r[5]:9.1f}  {r[0][:140]}')
