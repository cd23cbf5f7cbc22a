#!/bin/bash
# round 6: the inference heads' projected tensor z with 20 instead of 24 channels per pixel (8-byte pieces): parity, then alternating runs against the previous commit (ab_old/)
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--infer --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 5"
{
python -m pytest tests/test_ops_gpu.py -q -x -k "heads or head or gather or outconv" 2>&1 | tail -2
python -m pytest tests/test_model_gpu.py -q -x -k "config4 or fp16 or infer" 2>&1 | tail -2
for i in 1 2 3; do
  python bench.py $B 2>/dev/null | line "infer z20"
  (cd ab_old && python bench.py $B 2>/dev/null) | line "infer z24"
done
} 2>&1 | tee gpurun_out/r07_j_head_cz20.txt
