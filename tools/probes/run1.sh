timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "agent" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "edge or forward_parity" 2>&1 | tail -2
