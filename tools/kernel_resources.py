import sys,re
cur=None; d={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); d[cur]={}; continue
    m=re.search(r'remark:\s+([\w \[\]/]+): (\S+)',l)
    if m and cur: d[cur][m.group(1).strip()]=m.group(2)
for k,v in d.items():
    print(f"{k[:75]:77s} V{v.get('VGPRs')} A{v.get('AGPRs')} occ{v.get('Occupancy [waves/SIMD]')} spill{v.get('VGPRs Spill')} lds{v.get('LDS Size [bytes/block]')}")
