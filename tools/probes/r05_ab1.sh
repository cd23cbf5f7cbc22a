cd $GRAFT_REPO_ROOT
python tools/probes/wgrad_align.py > gpurun_out/r05_wgrad_align.txt 2>&1
b() { python bench.py --no-cpu-baseline --no-extra-configs --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], {k:v for k,v in d.get('kernel_ms',{}).items() if 'wgrad[' in k}, d.get('families',{}).get('upconv_wgrad'))"; }
for i in 1 2; do
for al in 8 64 1024; do STJ_ARENA_ALIGN=$al b "align$al"; done
done
for i in 1 2 3; do
for k in 0 1 2; do STJ_ARENA_ALIGN=1024 STJ_AB_OUTRED=$k python bench.py --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('outred$k', d['value'], d['ms_per_step'])"; done
done
