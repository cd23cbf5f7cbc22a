#!/bin/bash
# round 6: 192-column weight chunks of the C = 96 MLP kernels at 32768 rows (the shipped selection) against 96 everywhere (previous commit), alternating
cd $GRAFT_REPO_ROOT 2>/dev/null || true
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--no-cpu-baseline --no-extra-configs --no-kernel-timing"
for i in 1 2 3 4; do
  python bench.py $B --steps 300 --warmup 10 2>/dev/null | line "train hc192_at_32768"
  STJ_LIB_PATH=strajnet_amd/variants/lib_swin_old.so python bench.py $B --steps 300 --warmup 10 2>/dev/null | line "train hc96"
done 2>&1 | tee -a gpurun_out/r07_o_mlp_hc96.txt
