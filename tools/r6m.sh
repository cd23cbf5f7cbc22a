#!/bin/bash
# round 6: outconv_bwd v3 (wave strips, short-lived workgroups) against v2 and its own variants; then the tests that cover it
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
for v in "" ocb_nw4 ocb_nw16 ocb_v2; do
  L=""; [ -n "$v" ] && L=strajnet_amd/variants/lib_$v.so
  echo "== ${v:-base(nw8)}"; STJ_LIB_PATH=$L bash tools/prof_py.sh 8 tools/bench_outconv.py --iters 10 2>&1 | grep -i "outconv_bwd"
done 2>&1 | tee gpurun_out/r06_m_outconv_bwd.txt
python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "outconv or heads" 2>&1 | tail -3 | tee gpurun_out/r06_m_tests.txt
