#!/bin/bash
# gpurun_out/r6/profile (scripts/profile_r6.sh's capture) -> profiles/r06_* (tracked)
cd "$(dirname "$0")/.."
P=gpurun_out/r6/profile
cp $P/bench.json profiles/r06_bench_default.json
grep -v "amdgpu.ids" $P/bench.err > profiles/r06_bench_default_stderr.txt
cp $P/r06_search_kernel_summary.json profiles/r06_search_kernel_summary.json
cp $P/r06_kernel_stats_trace_serial.csv profiles/r06_kernel_stats_one_launch_at_a_time.csv
cp $P/r06_kernel_stats_trace_stream.csv profiles/r06_kernel_stats_batch_stream.csv
[ -f gpurun_out/r6/gpu_tests_b.log ] && cp gpurun_out/r6/gpu_tests_b.log profiles/r06_gpu_tests.txt
ls -la profiles/r06_*
