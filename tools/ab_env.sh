#!/bin/bash
# A/B of environment settings on the headline bench: usage tools/ab_env.sh "VAR=1" "VAR=2 OTHER=3" ...   (first and last run: no setting)
cd $GRAFT_REPO_ROOT 2>/dev/null || true
run() { timeout 90 env $1 python bench.py --steps 40 --warmup 5 --no-extra-configs --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" || echo "$1 FAILED/TIMEOUT"; }
run "X=0"
for s in "$@"; do run "$s"; done
run "X=0"
