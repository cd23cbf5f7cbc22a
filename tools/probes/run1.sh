timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "fused_skip" 2>&1 | tail -2
