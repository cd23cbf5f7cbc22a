#!/bin/bash
# round 6: timelines of the captured step with the one-pass and the two-pass loss
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
B="--no-cpu-baseline --no-extra-configs --no-kernel-timing --settle 0"
for v in True False; do
  bash tools/trace_cmd.sh gpurun_out/r06_v_trace_loss_fused_$v.txt 12 python tools/ab_attr.py ops.LOSS_FUSED_BWD=$v -- $B --steps 10 --warmup 2
  db=$(find /tmp/prof_cmd -name '*.db' | head -1)
  python tools/timeline.py $db > gpurun_out/r06_v_timeline_loss_fused_$v.txt 2>&1 || true
done
