timeout 900 python scripts/longpool_probe.py --graph-cache /tmp/roar_graph.index --L 1000 2000 --tables 1536 2048 4096 8192 16384 > gpurun_out/r5_long_pools_rows.txt 2> gpurun_out/r5_long_pools_rows.err
cat gpurun_out/r5_long_pools_rows.txt | cut -c1-260
