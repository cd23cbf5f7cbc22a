#!/bin/bash
# PMC passes over the whole bench step (serial mode: every kernel alone), three separate rocprofv3 runs as MI355X_MICROARCH.md
# prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ counters in their own pass), --kernel-trace only.
#   usage: tools/pmc_step.sh <tag> [bench.py flag: --infer | --cfg512]      -> gpurun_out/<tag>_pmc_step[_infer|_cfg512].txt (+ profiles/roofline_traffic.json style json)
tag=$1; extra=$2; sfx=${extra:+_${extra#--}}
cd $GRAFT_REPO_ROOT 2>/dev/null || true
export TMPDIR=/tmp STJ_BENCH_CALIB=1
mkdir -p gpurun_out
cmd="python bench.py $extra --steps 2 --warmup 1 --settle 0 --no-cpu-baseline --no-kernel-timing --no-extra-configs --serial"
rm -rf /tmp/pmc_f /tmp/pmc_w /tmp/pmc_s
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pmc_f -- $cmd > /dev/null 2> gpurun_out/${tag}_pmc${sfx}.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d /tmp/pmc_w -- $cmd > /dev/null 2>> gpurun_out/${tag}_pmc${sfx}.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d /tmp/pmc_s -- $cmd > /dev/null 2>> gpurun_out/${tag}_pmc${sfx}.err
f=$(find /tmp/pmc_f -name '*counter_collection.csv' | head -1); w=$(find /tmp/pmc_w -name '*counter_collection.csv' | head -1)
s=$(find /tmp/pmc_s -name '*counter_collection.csv' | head -1); k=$(find /tmp/pmc_s -name '*kernel_trace.csv' | head -1)
python tools/pmc_step_summary.py "$f" "$w" "$s" "$k" gpurun_out/${tag}_roofline_traffic${sfx}.json > gpurun_out/${tag}_pmc_step${sfx}.txt 2>> gpurun_out/${tag}_pmc${sfx}.err
head -60 gpurun_out/${tag}_pmc_step${sfx}.txt
