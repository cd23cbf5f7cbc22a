#!/bin/bash
# same-box A/B of one environment variable on the training bench.  usage: tools/ab_env2.sh VAR "v1 v2" [rounds] [extra bench args]
var=$1; vals=$2; rounds=${3:-3}; shift 3
for i in $(seq $rounds); do for v in $vals; do
env $var=$v python bench.py --no-cpu-baseline --no-extra-configs --no-kernel-timing --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', d['value'], d['ms_per_step'])"
done; done
