cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_wgrad_sk_gpu.py -x -q 2>&1 | tail -15
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "train_step_parity_f32 or golden or bf16_mode_error" 2>&1 | tail -8
for sk in 0 1; do for i in 1 2; do STJ_WGRAD_SK=$sk timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('SK=$sk', d['value'], d['ms_per_step'], d.get('families',{}).get('gemm_wgrad'), d.get('roofline',{}).get('serial_kernel_ms_per_step'))"; done; done
