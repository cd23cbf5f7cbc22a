#!/bin/bash
# round 6: the two output heads' backward on two streams (the second on its branch's stream), alternating same-box runs; then the tests
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--no-cpu-baseline --no-extra-configs --no-kernel-timing"
for i in 1 2 3 4; do
  python tools/ab_attr.py ops.OUTCONV_BWD_TWO_STREAMS=True -- $B --steps 60 --warmup 10 2>/dev/null | line two_streams
  python tools/ab_attr.py ops.OUTCONV_BWD_TWO_STREAMS=False -- $B --steps 60 --warmup 10 2>/dev/null | line one_stream
done 2>&1 | tee gpurun_out/r06_s_outconv_two_streams.txt
