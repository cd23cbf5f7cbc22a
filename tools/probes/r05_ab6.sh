cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "swin" 2>&1 | tail -3
python tools/bench_swin_k.py --cfg512 2>&1 | grep -v amdgpu.ids | tail -12
for i in 1 2; do
python bench.py --cfg512 --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg512 new', d['value'], d['ms_per_step'])"
STJ_LIB_PATH=strajnet_amd/variants/lib_swin_r04.so python bench.py --cfg512 --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg512 old', d['value'], d['ms_per_step'])"
done
