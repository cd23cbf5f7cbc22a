cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_agent_fused_gpu.py -q -m gpu 2>&1 | tail -5
bash tools/prof_py.sh 30 tools/bench_agent.py 8 1 > gpurun_out/r06_d_agent_fused_prof.txt 2>&1; cat gpurun_out/r06_d_agent_fused_prof.txt | head -24
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  python tools/ab_attr.py fused_agent=True -- --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | line fused
  python tools/ab_attr.py fused_agent=False -- --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | line layerwise
done 2>&1 | tee gpurun_out/r06_d_ab_agent.txt
