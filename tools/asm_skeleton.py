#!/usr/bin/env python
"""Skeleton of one kernel of a gfx950 .s file: labels, waits, barriers, branches and global memory instructions, with the count of
MFMA / LDS / VALU / SALU instructions between them.  usage: tools/asm_skeleton.py file.s <mangled-name-substring> [first-label]"""
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end', s, re.S | re.M):
    if sys.argv[2] in m.group(1):
        break
else:
    sys.exit('kernel not found')
print('#', m.group(1))
out, cnt = [], {}
def flush():
    global cnt
    if cnt: out.append('      ' + ' '.join(f'{k}={v}' for k, v in cnt.items()))
    cnt = {}
for i, l in enumerate(m.group(2).split('\n')):
    t = l.strip()
    if not t or t.startswith(';'): continue
    if t.startswith('.LBB'): flush(); out.append(t[:70]); continue
    if t.startswith('.'): continue
    op = t.split()[0]
    key = ('mfma' if op.startswith('v_mfma') else 'ds_r' if op.startswith('ds_read') else 'ds_w' if op.startswith('ds_write') else None)
    if key is None and op.startswith(('s_waitcnt', 's_barrier', 'global_', 'buffer_', 's_cbranch', 's_branch')):
        flush(); out.append(f'{i}: {t[:80]}'); continue
    if key is None: key = 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'other'
    cnt[key] = cnt.get(key, 0) + 1
flush()
k0 = 0
if len(sys.argv) > 3:
    k0 = next((k for k, x in enumerate(out) if x.startswith(sys.argv[3])), 0)
print('\n'.join(out[k0:]))
