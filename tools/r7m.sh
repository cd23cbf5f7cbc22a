#!/bin/bash
# the inference trace and timeline again (bench.py --no-kernel-timing no longer appends the non-pipelined comparison loop to the traced run)
cd $GRAFT_REPO_ROOT 2>/dev/null || true
bash tools/trace_infer.sh r06_final > /dev/null 2>&1
head -6 gpurun_out/r06_final_kernel_trace_infer_b32_f16.txt | cut -c1-150
head -8 gpurun_out/r06_final_timeline_infer_b32_f16.txt | cut -c1-120
python bench.py --infer --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['agent_pipeline'][-90:])"
