run() { python bench.py --no-cpu-baseline --no-kernel-timing $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_timed_kernels_gpu.py -x -q -k "upconv" 2>&1 | tail -2
for l in 0 1; do python tools/bench_conv.py --layer $l --only dgrad --iters 20 2>/dev/null | tail -1; done
for i in 1 2 3; do echo -n "new "; run; echo -n "no-pf "; STJ_NO_DGRAD_PF=1 run; done
