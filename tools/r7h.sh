#!/bin/bash
# sanity after the last python-side changes: loss / model / dp tests + a bench line
cd $GRAFT_REPO_ROOT 2>/dev/null || true
python -m pytest tests/test_ops_gpu.py -q -x -k "loss" 2>&1 | tail -2
python -m pytest tests/test_model_gpu.py tests/test_lib_and_dp.py -q -x 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 200 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', d['value'], d['ms_per_step'])"
