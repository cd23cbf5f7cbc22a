run() { python bench.py --no-cpu-baseline --no-kernel-timing $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do echo -n "new "; run; echo -n "no-pf "; STJ_NO_DGRAD_PF=1 run; done
echo -n "cfg512 "; run --cfg512
