#!/bin/bash
# Which in-flight depth / per-launch workgroup cap serves a K-step run best (the pipeline fills and drains inside the timed region)?
# One rank's share of the batch (--nq 1250 = N 8, 2500 = N 4, ...) and the whole batch, K = 20 (the driver's) and 24.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
G=/tmp/r3_bench_graph.index
[ -f $G ] || python3 bench.py --graph-cache $G --steps 4 --warmup 2 --no-cpu-baseline --no-traffic --no-host-path > /dev/null 2> $O/depth_build.err
: > $O/depth_probe.txt
for K in 20 24; do
for cfg in "10000 6 1024" "10000 4 1536" "10000 4 1024" "10000 5 1280" "1250 6 640" "1250 4 960" "1250 4 768" "1250 5 768" "1250 4 640" "2500 6 640" "2500 4 960" "5000 6 640" "5000 4 960"; do
  set -- $cfg
  python3 bench.py --graph-cache $G --L 50 --nq $1 --in-flight $2 --launch-grid $3 --steps $K --warmup 5 --no-cpu-baseline --no-traffic --no-host-path 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=$K nq=$1 in-flight $2 cap $3: ms/step %.4f  QPS %.0f  (one at a time %.4f ms)' % (d['ms_per_step'], d['value'], d['config'].get('ms_per_step_one_launch_at_a_time', 0)))" >> $O/depth_probe.txt
done; done
cat $O/depth_probe.txt
