timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_timed_kernels_gpu.py -x -q -k "upconv" 2>&1 | tail -2
for v in 1 2 1 2; do echo -n "STJ_DGRAD_WS2=$v "; STJ_DGRAD_WS2=$v python tools/bench_conv.py --layer 3 --only dgradE --iters 20 2>/dev/null | tail -1; done
