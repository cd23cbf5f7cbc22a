#!/bin/bash
# round 6: fused offset head, tile-shape A/B (W = 16: one or two image rows per workgroup) + tests
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
python -m pytest tests/test_fgoff_fused_gpu.py -q -x -m gpu 2>&1 | tail -3 | tee gpurun_out/r06_k_tests.txt
for v in "" strajnet_amd/variants/lib_fgoff_r1.so; do
  echo "lib=$v"
  STJ_LIB_PATH=$v python tools/bench_fgoff.py 8 16 bf16 2>&1 | grep "B=" | grep "fused  "
  STJ_LIB_PATH=$v python tools/bench_fgoff.py 32 16 f16 2>&1 | grep "B=" | grep "fused  "
done 2>&1 | tee gpurun_out/r06_k_fgoff_tiles.txt
python tools/bench_fgoff.py 8 8 f32 2>&1 | grep "B=" | tee -a gpurun_out/r06_k_fgoff_tiles.txt
python tools/bench_fgoff.py 8 16 f32 2>&1 | grep "B=" | tee -a gpurun_out/r06_k_fgoff_tiles.txt
python -m pytest tests/test_model_gpu.py -q -x -m gpu -k "golden or entry_points" 2>&1 | tail -3 | tee -a gpurun_out/r06_k_tests.txt
