cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for th in 8 16; do for l in 0 1; do STJ_PS_TH=$th python tools/bench_conv.py --only fwd --layer $l --iters 20 | sed "s/^/TH=$th /"; done; done 2>&1 | tee gpurun_out/r06_i_ps.txt
STJ_PS_TH=16 python -m pytest tests/test_ops_gpu.py tests/test_timed_kernels_gpu.py -q -m gpu -k "upconv or conv or decoder" 2>&1 | tail -3
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  STJ_PS_TH=16 python bench.py --infer --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | line infer_th16
  STJ_PS_TH=8 python bench.py --infer --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | line infer_th8
done 2>&1 | tee -a gpurun_out/r06_i_ps.txt
for i in 1 2 3; do
  STJ_PS_TH=16 python bench.py --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | line train_th16
  STJ_PS_TH=8 python bench.py --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | line train_th8
done 2>&1 | tee -a gpurun_out/r06_i_ps.txt
