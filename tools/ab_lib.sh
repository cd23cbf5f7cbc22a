#!/bin/bash
# same-box A/B of two library builds on the training and inference benches.   usage: tools/ab_lib.sh <variant .so> [rounds]
var=$1; rounds=${2:-3}
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in $(seq $rounds); do
  python bench.py --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | line "train new"
  STJ_LIB_PATH=$var python bench.py --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | line "train old"
done
for i in $(seq 2); do
  python bench.py --infer --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | line "infer new"
  STJ_LIB_PATH=$var python bench.py --infer --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | line "infer old"
done
