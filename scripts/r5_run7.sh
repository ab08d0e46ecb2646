timeout 900 python scripts/lat_probe.py --graph-cache /tmp/roar_graph.index --L 1000 --nqs 1 10000 --vis 3 > gpurun_out/r5_lat_probe2.txt 2> gpurun_out/r5_lat_probe2.err
cat gpurun_out/r5_lat_probe2.txt | cut -c1-700
