for L in 30 50 100; do
MODE=share L_PQ=$L BS=10000,5000,2500,1250,625 timeout 600 python scripts/stream_probe.py >> gpurun_out/r5_stream_probe_L.txt 2>> gpurun_out/r5_stream_probe.err
done
cat gpurun_out/r5_stream_probe_L.txt; tail -3 gpurun_out/r5_stream_probe.err
