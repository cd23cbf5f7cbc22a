run() { python bench.py --no-cpu-baseline --no-kernel-timing $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "agent" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -3
for i in 1 2 3; do echo -n "new "; run; done
