#!/bin/bash
# per-kernel durations (rocprofv3) of the fused Swin kernels at every stage shape of cfg-256 B = 8, warm
cd $GRAFT_REPO_ROOT 2>/dev/null || true
timeout 150 bash tools/prof_py.sh 40 tools/bench_swin.py | grep "swin_\|B="
timeout 100 bash tools/prof_py.sh 30 tools/bench_swin384.py | grep "swin_\|B="
