cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10"
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
python bench.py $F 2>/dev/null | line base
STJ_AB_XW=2 python bench.py $F 2>/dev/null | line xw2
python tools/ab_attr.py agent_issue_mode=0 -- $F 2>/dev/null | line mode0
python tools/ab_attr.py agent_issue_mode=1 -- $F 2>/dev/null | line mode1
STJ_AB_XW=2 python tools/ab_attr.py agent_issue_mode=0 -- $F 2>/dev/null | line mode0+xw2
done
