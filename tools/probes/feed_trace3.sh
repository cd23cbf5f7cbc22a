cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/ft; FEED_ONLY=${1:-threaded} rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ft -- python tools/probes/feed_probe.py > /tmp/ft.log 2>&1
k=$(find /tmp/ft -name '*kernel_trace.csv' | head -1); m=$(find /tmp/ft -name '*memory_copy_trace.csv' | head -1)
python - "$k" "$m" <<'P'
import csv, sys
ks = list(csv.DictReader(open(sys.argv[1])))
ks.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(ks[0]['Start_Timestamp'])
nad = [k for k in ks if 'nadam' in k['Kernel_Name']]
# steps = intervals between consecutive nadam ends
for a, b in list(zip(nad[:-1], nad[1:]))[-4:]:
    s, e = int(a['End_Timestamp']), int(b['End_Timestamp'])
    inside = [(int(k['Start_Timestamp']), int(k['End_Timestamp']), k['Kernel_Name']) for k in ks if s <= int(k['Start_Timestamp']) < e]
    # union of busy time
    busy, cur_s, cur_e = 0, None, None
    gaps = []
    for ks_, ke_, n in inside:
        if cur_e is None: cur_s, cur_e = ks_, ke_
        elif ks_ > cur_e:
            gaps.append((ks_ - cur_e, (cur_e - s) / 1e6)); busy += cur_e - cur_s; cur_s, cur_e = ks_, ke_
        else: cur_e = max(cur_e, ke_)
    busy += cur_e - cur_s
    gaps.sort(reverse=True)
    print('step %.3f ms: kernels %d, GPU busy %.3f ms, idle %.3f ms; largest gaps (us @ ms into step): %s' % ((e - s) / 1e6, len(inside), busy / 1e6, (e - s - busy) / 1e6, [(round(g / 1e3, 1), round(t, 2)) for g, t in gaps[:6]]))
P
