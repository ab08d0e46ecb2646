timeout 900 python scripts/lat_probe.py --graph-cache /tmp/roar_graph.index --L 50 1000 --nqs 1 1250 10000 --vis 3 > gpurun_out/r5_lat_probe.txt 2> gpurun_out/r5_lat_probe.err
cat gpurun_out/r5_lat_probe.txt | cut -c1-700
