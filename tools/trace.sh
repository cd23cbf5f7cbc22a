#!/bin/bash
# rocprofv3 kernel trace of the bench (serial + concurrent), summarised into gpurun_out/.   usage: tools/trace.sh <tag> [extra bench args]
tag=$1; shift
cd $GRAFT_REPO_ROOT 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
for mode in serial concurrent; do
  extra=""; [ $mode = serial ] && extra="--serial"
  rm -rf /tmp/prof_$mode
  rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_$mode -- python bench.py --steps 5 --warmup 2 --settle 0 --no-cpu-baseline --no-kernel-timing --no-extra-configs $extra "$@" > gpurun_out/trace_${tag}_$mode.json 2> gpurun_out/trace_${tag}_$mode.err
  db=$(find /tmp/prof_$mode -name '*.db' | head -1)
  python tools/rocpd_summary.py "$db" 9 > gpurun_out/${tag}_kernel_trace_${mode}_b8_bf16.txt 2>> gpurun_out/trace_${tag}_$mode.err
done
head -45 gpurun_out/${tag}_kernel_trace_serial_b8_bf16.txt
db=$(find /tmp/prof_concurrent -name '*.db' | head -1)
python tools/timeline.py "$db" 2 > gpurun_out/${tag}_timeline_concurrent.txt 2>&1
