cd $GRAFT_REPO_ROOT
b() { env "$@" timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('$*', d['value'], d['ms_per_step'])"; }
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --gemm-trace 2>&1 >/dev/null | grep "gemm\|wgrad" | head -40
b STJ_WGRAD_SK=1
b STJ_WGRAD_SK=1 STJ_WGRAD_SK_POINTS=0
b STJ_WGRAD_SK=1 STJ_WGRAD_SK_WGS=128
b STJ_WGRAD_SK=1 STJ_WGRAD_SK_WGS=192
b STJ_WGRAD_SK=1 STJ_WGRAD_SK_POINTS=0 STJ_WGRAD_SK_WGS=128
b STJ_WGRAD_SK=1
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "train_step_parity_f32 or golden" 2>&1 | tail -3
