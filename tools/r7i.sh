#!/bin/bash
# round 6: upconv_fwd_ps with the skip sums in its epilogue (inference): the skip operands of 4 / 8 items in flight together instead of item by item
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--infer --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 5"
{
python -m pytest tests/test_ops_gpu.py -q -x -k "upconv_add or fused_skip" 2>&1 | tail -2
python -m pytest tests/test_model_gpu.py -q -x -k "config4 or fp16 or infer" 2>&1 | tail -2
for i in 1 2 3; do
  python bench.py $B 2>/dev/null | line "infer batch4"
  STJ_LIB_PATH=strajnet_amd/variants/lib_ps_old.so python bench.py $B 2>/dev/null | line "infer item_by_item"
  STJ_LIB_PATH=strajnet_amd/variants/lib_ps_b8.so python bench.py $B 2>/dev/null | line "infer batch8"
done
} 2>&1 | tee gpurun_out/r07_i_ps_res_batch.txt
