cd $GRAFT_REPO_ROOT
line() { python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
for m in 2 0 1; do
  timeout 120 python tools/ab_attr.py agent_issue_mode=$m -- --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>/dev/null | line mode$m
done; done
