cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10"
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
for b in 128 96 160 192 256; do STJ_AB_UPWG=$b python bench.py $F 2>/dev/null | line upwg$b; done
done
