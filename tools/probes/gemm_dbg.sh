#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for d in 0 1 2; do
  STJ_GEMM_CFG=$d rocprofv3 --kernel-trace -d gpurun_out/gc$d -o r -- python tools/bench_gemm.py > /dev/null 2>&1
done
