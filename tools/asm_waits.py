#!/usr/bin/env python
"""List, per kernel in a gfx950 .s file, the s_waitcnt vmcnt(0) that sit directly in front of MFMA work inside a loop -- the mark of a
global prefetch whose latency is exposed (the waitcnt pass could not prove an older load complete).  usage: tools/asm_waits.py file.s ..."""
import re, sys
for path in sys.argv[1:]:
    s = open(path).read()
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end', s, re.S | re.M):
        name, body = m.group(1), m.group(2).split('\n')
        if 'bf16' not in name and 'Li' not in name: continue
        hits = []
        inloop = False
        for i, l in enumerate(body):
            t = l.strip()
            if 'Loop' in l: inloop = True
            if t.startswith('s_waitcnt') and 'vmcnt(0)' in t and inloop:
                nxt = [x.strip().split()[0] for x in body[i + 1:i + 12] if x.strip() and not x.strip().startswith((';', '.'))]
                if any(x.startswith('v_mfma') for x in nxt[:6]):
                    hits.append(i)
        if hits and '3f16' not in name and 'IfL' not in name and 'IfE' not in name:
            print(path.split('/')[-1], name[:90], hits)
