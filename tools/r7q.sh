#!/bin/bash
# round 6: MLP half at C = 96 with 192-column weight chunks at 32768 rows: parity + a timed run
cd $GRAFT_REPO_ROOT 2>/dev/null || true
python -m pytest tests/test_ops_gpu.py tests/test_timed_kernels_gpu.py -q -x -k "swin or mlp" 2>&1 | tail -2
python -m pytest tests/test_model_gpu.py -q -x 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 300 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', d['value'], d['ms_per_step'])"
python bench.py --infer --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('infer', d['value'], d['ms_per_step'])"
