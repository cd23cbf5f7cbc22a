cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee gpurun_out/r06_h_gputests.txt
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  python tools/ab_attr.py fused_agent=True -- --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | line fused
  python tools/ab_attr.py fused_agent=False -- --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | line layerwise
done 2>&1 | tee gpurun_out/r06_h_ab_agent.txt
for i in 1 2; do
  python tools/ab_attr.py fused_agent=True -- --infer --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | line infer_fused
  python tools/ab_attr.py fused_agent=False -- --infer --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | line infer_layerwise
done 2>&1 | tee -a gpurun_out/r06_h_ab_agent.txt
