#!/bin/bash
# round 6, first GPU session (the two switches were environment variables then; UPWG_ORDER is a module constant of ops.py now): deferred up-conv weight-gradient order / stream priority A/B on the training bench (alternating same-box runs)
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
run() { env $1 python bench.py --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | line "$1"; }
for i in 1 2 3; do
  run "STJ_UPWG_ORDER=bwd"
  run "STJ_UPWG_ORDER=wide"
  run "STJ_UPWG_ORDER=rev"
  run "STJ_WG_PRIO=1"
  run "STJ_WG_PRIO=1 STJ_UPWG_ORDER=wide"
done 2>&1 | tee gpurun_out/r06_ab_order.txt
python -m pytest tests/test_model_gpu.py -q -x -m gpu -k "graph or pipeline or config4" 2>&1 | tail -3 | tee gpurun_out/r06_a_tests.txt
python -m pytest tests/test_lib_and_dp.py -q -x -m gpu 2>&1 | tail -3 | tee -a gpurun_out/r06_a_tests.txt
