#!/usr/bin/env python
"""Count, per kernel of a gfx950 .s file, load -> store -> load alternations: the mark of an epilogue whose loads the compiler could not
hoist over the stores (possible aliasing), i.e. a chain of dependent memory round trips.  usage: tools/asm_ldst_chains.py file.s ..."""
import re, sys
for path in sys.argv[1:]:
    s = open(path).read()
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end', s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        seq = []
        for l in body.split('\n'):
            t = l.strip()
            if t.startswith(('global_load', 'buffer_load')): seq.append('L')
            elif t.startswith(('global_store', 'buffer_store')): seq.append('S')
        n = len(re.findall(r'LS(?=L)', ''.join(seq)))
        if n >= 3: print(path.split('/')[-1], name[:90], 'load/store alternations:', n)
