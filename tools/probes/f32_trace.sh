cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_timed_kernels_gpu.py -q -x -k "not bench_step" 2>&1 | tail -5
rm -rf /tmp/prof_f32
timeout 900 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_f32 -- python bench.py --steps 3 --warmup 1 --dtype f32 --serial --no-cpu-baseline --no-kernel-timing --no-extra-configs > gpurun_out/trace_f32.json 2> gpurun_out/trace_f32.err
db=$(find /tmp/prof_f32 -name '*.db' | head -1)
python tools/rocpd_summary.py "$db" 6 > gpurun_out/f32_kernel_trace_serial.txt 2>> gpurun_out/trace_f32.err
head -60 gpurun_out/f32_kernel_trace_serial.txt
tail -3 gpurun_out/trace_f32.json
