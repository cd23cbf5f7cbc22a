run() { python bench.py --no-cpu-baseline --no-kernel-timing $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do echo -n "new "; run; echo -n "old-movers "; STJ_DGRAD_WS2=2 STJ_WS2_GUARDED=1 run; done
echo -n "infer new "; run --infer; echo -n "infer old "; STJ_WS2_GUARDED=1 run --infer
