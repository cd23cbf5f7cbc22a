#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/bench_gemm_pmc.py
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > gpurun_out/pmc_avail.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rm -rf /tmp/pg$i
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/pg$i -- python tools/bench_gemm_pmc.py > /dev/null 2> gpurun_out/pmc_gemm_$i.err
  f=$(find /tmp/pg$i -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
if not sys.argv[1]:
    print('no output'); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemm' in r['Kernel_Name']:
        acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print(f'   {c:34s} {sum(v)/len(v):14.0f}  (n={len(v)})')
PY
done
