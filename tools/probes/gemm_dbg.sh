#!/bin/bash
for cfg in "384 48" "768 96" "1536 128" "3072 256" "768 24"; do
  set -- $cfg
  echo "tgt $1 cap $2: $(STJ_SPLITK_TGT=$1 STJ_SPLITK_CAP=$2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>&1 | grep -o '"ms_per_step": [0-9.]*')"
done
