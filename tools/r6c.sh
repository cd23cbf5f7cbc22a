cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  python tools/ab_attr.py fused_agent=True fused_agent_int=False -- --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | line enc_only
  python tools/ab_attr.py fused_agent=False -- --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | line layerwise
done 2>&1 | tee gpurun_out/r06_c_ab_agent_enc.txt
