line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  python bench.py --infer --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | line "base"
  STJ_LIB_PATH=$1 python bench.py --infer --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | line "variant"
done
