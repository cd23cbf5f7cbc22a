run() { python bench.py --no-cpu-baseline --no-kernel-timing $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do echo -n "base "; run; echo -n "fused2 "; STJ_FUSED_SKIP=2 run; done
