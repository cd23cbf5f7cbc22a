run() { python bench.py --no-cpu-baseline --no-kernel-timing $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3 4; do echo -n "new "; run; echo -n "base "; STJ_LIB_PATH=$PWD/tools/probes/libs/base.so run; done
