run() { python bench.py --no-cpu-baseline --no-kernel-timing --steps 60 --warmup 5 $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_timed_kernels_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -2
for i in 1 2; do echo -n "new "; run; done
echo -n "cfg512 "; run --cfg512; echo -n "infer "; run --infer
