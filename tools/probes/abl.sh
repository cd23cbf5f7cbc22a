python tools/bench_conv.py --layer 3 --only fwd --iters 20 2>&1 | tail -1
STJ_LIB_PATH=strajnet_amd/variants/lib_old.so python tools/bench_conv.py --layer 3 --only fwd --iters 20 2>&1 | tail -1
python tools/bench_conv.py --layer 3 --only fwd --iters 20 2>&1 | tail -1
timeout 300 python -m pytest tests/test_timed_kernels_gpu.py tests/test_ops_gpu.py -q -x -s -k "upconv" 2>&1 | grep "96->48\|passed\|failed\|Error\|assert" | head
