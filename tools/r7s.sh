#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || true
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--no-cpu-baseline --no-extra-configs --no-kernel-timing"
for i in 1 2 3 4; do
  STJ_LIB_PATH=strajnet_amd/variants/lib_attn_occ3.so python bench.py $B --steps 300 --warmup 10 2>/dev/null | line "train occ3"
  python bench.py $B --steps 300 --warmup 10 2>/dev/null | line "train base"
  STJ_LIB_PATH=strajnet_amd/variants/lib_attn_occ4.so python bench.py $B --steps 300 --warmup 10 2>/dev/null | line "train occ4"
done 2>&1 | tee -a gpurun_out/r07_r_attn_occ3.txt
