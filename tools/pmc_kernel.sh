#!/bin/bash
# Deep PMC look at ONE micro-benchmark command (several rocprofv3 --pmc passes, --kernel-trace only), per kernel means.
# (SQ counters only: a TCP_* set aborted rocprofv3 and hung the box's session for 20 minutes.)
#   usage: tools/pmc_kernel.sh <tag> <kernel name substring> -- <command ...>      -> gpurun_out/<tag>_pmc_kernel.txt
tag=$1; pat=$2; shift 3
cd $GRAFT_REPO_ROOT 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/${tag}_pmc_kernel.txt; : > $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"; do
  i=$((i+1)); rm -rf /tmp/pk_$i
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/pk_$i -- "$@" > /dev/null 2>> gpurun_out/${tag}_pmc_kernel.err
  c=$(find /tmp/pk_$i -name '*counter_collection.csv' | head -1); k=$(find /tmp/pk_$i -name '*kernel_trace.csv' | head -1)
  python - "$c" "$k" "$pat" >> $out <<'P'
import csv, sys, collections
c, k, pat = sys.argv[1:4]
vals = collections.defaultdict(list)
for r in csv.DictReader(open(c)):
    if pat in r['Kernel_Name']:
        vals[r['Counter_Name']].append(float(r['Counter_Value']))
dur = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(k)) if pat in r['Kernel_Name']]
print('# launches %d, mean duration %.1f us' % (len(dur), sum(dur) / max(1, len(dur)) / 1e3))
for n, v in vals.items():
    print('%-40s %16.0f' % (n, sum(v) / len(v)))
P
done
cat $out
