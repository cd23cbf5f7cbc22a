timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_timed_kernels_gpu.py -x -q -k "upconv" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "train or graph or bucket" 2>&1 | tail -2
