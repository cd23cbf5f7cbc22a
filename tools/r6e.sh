cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
for m in 2 0 1 3 4 5; do
  python tools/ab_attr.py fused_agent=True agent_issue_mode=$m -- --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | line fused_mode$m
done
python tools/ab_attr.py fused_agent=False -- --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | line layerwise_mode2
done 2>&1 | tee gpurun_out/r06_e_ab_agent_modes.txt
