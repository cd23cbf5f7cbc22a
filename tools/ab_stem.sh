for i in 1 2 3; do
for v in True False; do
python tools/ab_attr.py fused_stem=$v -- --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train fused_stem=$v', d['value'], d['ms_per_step'])"
done; done
for i in 1 2; do
for v in True False; do
python tools/ab_attr.py fused_stem=$v -- --infer --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('infer fused_stem=$v', d['value'], d['ms_per_step'])"
done; done
