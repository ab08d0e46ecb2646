#!/bin/bash
# Round-6 measurement recipe (run on the GPU box through gpurun; the graph is deterministic -- config.graph_hash says so):
#   1. default bench line (value through one batch stream, roofline led by the in-run PMC traffic, cpu_baseline, other_workloads)
#   2. rocprofv3 --kernel-trace --stats, ONE launch at a time (per-launch kernel): its average duration must agree with
#      roofline.one_launch_at_a_time.kernel_ms of 1.
#   3. rocprofv3 --kernel-trace --stats of the stream regime: the stream's workgroups are ONE (or two) dispatches of the batch-stream
#      form of the kernel that live for the whole timed loop -- dispatch duration / steps must agree with roofline.kernel_ms of 1.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6/profile
mkdir -p $O
cd $R
ROAR_LOG_CALIBRATION=1 python3 bench.py --graph-cache /tmp/r6_graph.index > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err; cut -c1-400 $O/bench.json
export TMPDIR=/tmp
cd /tmp
L=$(python3 -c "import json;print(json.load(open('$O/bench.json'))['config']['L_pq'])")
ARGS="--graph-cache /tmp/r6_graph.index --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-traffic --no-share-probe --no-other-workloads --L $L"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_serial -- python3 $R/bench.py $ARGS --in-flight 1 --no-stream > $O/trace_serial.json 2> $O/trace_serial.err
echo "serial trace rc=$?"; tail -1 $O/trace_serial.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_stream -- python3 $R/bench.py $ARGS > $O/trace_stream.json 2> $O/trace_stream.err
echo "stream trace rc=$?"; tail -1 $O/trace_stream.err
cd $R
python3 $R/scripts/summarize_profile_r5.py $O r06 > $O/summary.txt 2>&1; tail -40 $O/summary.txt
find $O -name "*kernel_trace.csv" -size +4M -delete
du -sh $O
