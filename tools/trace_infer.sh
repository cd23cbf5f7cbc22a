#!/bin/bash
# rocprofv3 kernel trace of the inference bench (config 4).   usage: tools/trace_infer.sh <tag>
tag=$1
cd $GRAFT_REPO_ROOT 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/prof_inf
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_inf -- python bench.py --infer --steps 5 --warmup 2 --settle 0 --no-kernel-timing > gpurun_out/trace_${tag}_infer.json 2> gpurun_out/trace_${tag}_infer.err
db=$(find /tmp/prof_inf -name '*.db' | head -1)
python tools/rocpd_summary.py "$db" 9 > gpurun_out/${tag}_kernel_trace_infer_b32_f16.txt
head -32 gpurun_out/${tag}_kernel_trace_infer_b32_f16.txt | cut -c1-160
python tools/timeline.py "$db" 2 outconv_pair_gather > gpurun_out/${tag}_timeline_infer_b32_f16.txt 2>/dev/null
