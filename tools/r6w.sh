#!/bin/bash
# round 6: is the replayed step clock / power limited?  (a) the one-pass loss A/B cold (no settle, 10 steps) and settled (40 + 10, 200 steps);
# (b) rocm-smi power and clocks sampled under a long run
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="--no-cpu-baseline --no-extra-configs --no-kernel-timing"
{
for i in 1 2 3; do
  python tools/ab_attr.py ops.LOSS_FUSED_BWD=True -- $B --settle 0 --steps 10 --warmup 2 2>/dev/null | line cold_one_pass
  python tools/ab_attr.py ops.LOSS_FUSED_BWD=False -- $B --settle 0 --steps 10 --warmup 2 2>/dev/null | line cold_two_passes
done
for i in 1 2 3; do
  python tools/ab_attr.py ops.LOSS_FUSED_BWD=True -- $B --steps 300 --warmup 10 2>/dev/null | line long_one_pass
  python tools/ab_attr.py ops.LOSS_FUSED_BWD=False -- $B --steps 300 --warmup 10 2>/dev/null | line long_two_passes
done
echo "== rocm-smi under a 3000-step run"
python bench.py $B --steps 3000 --warmup 10 > /tmp/long.json 2>/dev/null &
pid=$!
sleep 6
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk\|Temperature (Sensor junction)\|hotspot" | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/  */ /g'; echo
  sleep 1
done
wait $pid
cat /tmp/long.json | line long3000
echo "== idle"; sleep 3
rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk" | tr '\n' ' '; echo
} 2>&1 | tee gpurun_out/r06_w_clock_power.txt
