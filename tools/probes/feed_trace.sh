cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/ft; FEED_ONLY=${1:-base} rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ft -- python tools/probes/feed_probe.py > /tmp/ft.log 2>&1
k=$(find /tmp/ft -name '*kernel_trace.csv' | head -1); m=$(find /tmp/ft -name '*memory_copy_trace.csv' | head -1)
python - "$k" "$m" <<'P'
import csv, sys
ks = list(csv.DictReader(open(sys.argv[1]))); ms = list(csv.DictReader(open(sys.argv[2])))
ks.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(ks[0]['Start_Timestamp'])
big = [r for r in ms if int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 150000]
ev = [(int(k['Start_Timestamp']), int(k['End_Timestamp']), 'K q%s %s' % (k.get('Queue_Id', '?'), k['Kernel_Name'][:50])) for k in ks]
ev += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r['Direction'][12:]) for r in ms]
ev.sort()
# window around the last step's first big copy
s0 = int(big[-3]['Start_Timestamp'])
for s, e, n in ev:
    if s0 - 400000 < s < s0 + 2200000:
        if n.startswith('COPY') and e - s < 20000: continue
        print('%10.3f %8.1f us  %s' % ((s - t0) / 1e6, (e - s) / 1e3, n))
P
