cd $GRAFT_REPO_ROOT
b() { python bench.py --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
for k in 0 96 128 256; do STJ_AB_WQSIDE=$k b "wqside$k"; done
done
python bench.py --infer --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | tail -1 | cut -c1-200
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -5
