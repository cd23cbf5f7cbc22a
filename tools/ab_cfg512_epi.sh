line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
  python bench.py --cfg512 --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 30 --warmup 8 2>/dev/null | line "base"
  for c in 512 256; do STJ_LIB_PATH=strajnet_amd/variants/lib_epi$c.so python bench.py --cfg512 --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 30 --warmup 8 2>/dev/null | line "cap$c"; done
done
