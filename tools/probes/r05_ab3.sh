cd $GRAFT_REPO_ROOT
python tools/bench_swin_k.py --only mlp 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "swin" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_switches_gpu.py -q -x 2>&1 | tail -2
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', d['value'], d['ms_per_step'])"
done
