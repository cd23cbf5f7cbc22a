#!/bin/bash
# rocprofv3 kernel trace of an arbitrary command, summarised.   usage: tools/trace_cmd.sh <out.txt> <divisor> <command...>
out=$1; div=$2; shift 2
cd $GRAFT_REPO_ROOT 2>/dev/null || true
export TMPDIR=/tmp
rm -rf /tmp/prof_cmd
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_cmd -- "$@" > /tmp/prof_cmd.out 2> /tmp/prof_cmd.err
db=$(find /tmp/prof_cmd -name '*.db' | head -1)
python tools/rocpd_summary.py "$db" $div > $out 2>> /tmp/prof_cmd.err
