#!/bin/bash
# rocprofv3 kernel-trace summary of one python command (bounded).   usage: tools/prof_py.sh <lines> <script> [args]
n=$1; shift
cd $GRAFT_REPO_ROOT 2>/dev/null || true
export TMPDIR=/tmp
rm -rf /tmp/prof_py
timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_py -- python "$@" > /tmp/prof_py.out 2>&1
grep -v "amdgpu.ids\|^[WE]2026" /tmp/prof_py.out | tail -5
db=$(find /tmp/prof_py -name '*.db' | head -1)
[ -n "$db" ] && timeout 60 python tools/rocpd_summary.py "$db" 1 | head -$n | cut -c1-170
