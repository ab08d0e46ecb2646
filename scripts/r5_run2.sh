MODE=quarters FLAGS=0,32 L_PQ=30 BS=10000,2500 timeout 600 python scripts/stream_probe.py >> gpurun_out/r5_stream_probe_L.txt 2>> gpurun_out/r5_stream_probe.err
cat gpurun_out/r5_stream_probe_L.txt; tail -3 gpurun_out/r5_stream_probe.err
