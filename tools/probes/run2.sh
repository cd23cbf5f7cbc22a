run() { python bench.py --no-cpu-baseline --no-kernel-timing --steps 60 --warmup 5 $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -2
echo -n "new default "; run
echo -n "cfg512 "; run --cfg512
echo -n "cfg512 rows=1024 "; STJ_GEMM_GROUP_ROWS=1024 run --cfg512
echo -n "SPLITK_CAP=48 "; STJ_SPLITK_CAP=48 run
echo -n "SPLITK_CAP=192 "; STJ_SPLITK_CAP=192 run
echo -n "AGENT_LATE=1 "; STJ_AGENT_LATE=1 run
echo -n "new default "; run
