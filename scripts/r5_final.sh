timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r5_gpu_tests_final.log
tail -3 gpurun_out/r5_gpu_tests_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/profile_r5.sh
