run() { python bench.py --no-cpu-baseline --no-kernel-timing --steps 60 --warmup 5 $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_timed_kernels_gpu.py -x -q -k "swin" 2>&1 | tail -2
for i in 1 2 3; do echo -n "base "; run; for v in a1 a2 a3; do [ -f strajnet_amd/variants/lib_$v.so ] || continue; echo -n "$v "; STJ_LIB_PATH=$PWD/strajnet_amd/variants/lib_$v.so run; done; done
