#!/bin/bash
# copy an artifact set from gpurun_out/ (tools/final_artifacts.sh <tag>) into profiles/, merge its PMC traffic figures into profiles/roofline_traffic.json
# (what bench.py quotes as roofline.traffic) and regenerate the kernel table.   usage: tools/adopt_artifacts.sh <tag>
tag=$1
cd "$(dirname "$0")/.."
for f in gpurun_out/${tag}_*; do
  case "$f" in *.err) continue;; esac
  cp "$f" profiles/
done
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
out = json.load(open('profiles/roofline_traffic.json'))
for sfx in ('', '_infer', '_cfg512'):
    try:
        d = json.load(open(f'profiles/{tag}_roofline_traffic{sfx}.json'))
    except FileNotFoundError:
        continue
    for k, v in d.items():
        if k.startswith('_'):
            continue
        if sfx == '' or k not in out or sfx == '_infer' and 'head' in k or sfx == '_cfg512' and k.startswith('swin'):
            out[k] = v
out['_snapshot'] = tag
json.dump(out, open('profiles/roofline_traffic.json', 'w'), indent=1)
PY
python tools/kernel_state.py $tag > profiles/${tag}_kernel_state.md
echo adopted $tag
