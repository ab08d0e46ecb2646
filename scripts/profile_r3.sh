#!/bin/bash
# Round-3 measurement recipe (run on the GPU box through gpurun; the graph is deterministic, so every pass below uses the
# same graph -- config.graph_hash says so):
#   1. default bench line (in-run PMC traffic + L2 hit share, cpu_baseline, host-buffer path)
#   2. rocprofv3 --kernel-trace --stats of the same command with one launch at a time (--in-flight 1: the kernel's own
#      duration, which must agree with roofline.kernel_ms of 1.)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3/profile
mkdir -p $O
cd $R
python3 bench.py --graph-cache /tmp/r3_graph2.index > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err; cat $O/bench.json
export TMPDIR=/tmp
cd /tmp
ARGS="--graph-cache /tmp/r3_graph2.index --steps 8 --warmup 1 --in-flight 1 --no-cpu-baseline --no-host-path --no-traffic --L $(python3 -c "import json;print(json.load(open('$O/bench.json'))['config']['L_pq'])")"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py $ARGS > $O/trace.json 2> $O/trace.err
echo "trace rc=$?"; tail -2 $O/trace.err
cd $R
python3 $R/scripts/summarize_profile.py $O r03 --no-write-profiles --traffic-from-bench > $O/summary.json 2> $O/summary.err; tail -5 $O/summary.err
find $O -name "*kernel_trace.csv" -size +4M -delete
du -sh $O
