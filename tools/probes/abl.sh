python tools/bench_conv.py --layer 2 --iters 20 2>&1 | tail -4
timeout 300 python -m pytest tests/test_timed_kernels_gpu.py tests/test_ops_gpu.py -q -x -s -k "upconv" 2>&1 | grep "128->96\|passed\|failed\|Error\|assert" | head
