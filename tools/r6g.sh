cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_model_gpu.py tests/test_xattn_gpu.py tests/test_lib_and_dp.py -q -x -m gpu 2>&1 | tail -4 | tee gpurun_out/r06_g_tests.txt
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  python tools/ab_attr.py kv_in_agent_branch=True -- --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | line kv_side
  python tools/ab_attr.py kv_in_agent_branch=False -- --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | line kv_main
  python tools/ab_attr.py kv_in_agent_branch=False fused_agent=False -- --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | line r5_path
done 2>&1 | tee gpurun_out/r06_g_ab_kv.txt
