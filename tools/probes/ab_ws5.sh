cd $GRAFT_REPO_ROOT
bash tools/ab_env.sh "STJ_AB_WS5=1" "STJ_AB_WS5=0" "STJ_AB_WS5=1" "STJ_AB_WS5=0" "STJ_AB_WS5=1"
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  STJ_AB_WS5=0 python bench.py --infer --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | line "infer ws2"
  STJ_AB_WS5=1 python bench.py --infer --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | line "infer ws5"
done
