#!/bin/bash
# Round-4 measurement recipe (run on the GPU box through gpurun; the graph is deterministic -- config.graph_hash says so):
#   1. default bench line: value with six launches in flight, roofline in that regime + one launch at a time, in-run PMC
#      traffic for both, kernel-trace span of the overlapped loop, cpu_baseline, host-buffer path, calibration log on stderr
#   2. rocprofv3 --kernel-trace --stats of the same command, ONE launch at a time: the kernel's own average duration
#      (must agree with roofline.one_launch_at_a_time.kernel_ms of 1.)
#   3. rocprofv3 --kernel-trace --stats with the timed loop's SIX launches in flight: start / end of every launch
#      (span / launches must agree with roofline.kernel_ms of 1.)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4/profile
mkdir -p $O
cd $R
ROAR_LOG_CALIBRATION=1 python3 bench.py --graph-cache /tmp/r4_graph.index > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err; cat $O/bench.json | cut -c1-600
export TMPDIR=/tmp
cd /tmp
L=$(python3 -c "import json;print(json.load(open('$O/bench.json'))['config']['L_pq'])")
ARGS="--graph-cache /tmp/r4_graph.index --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-traffic --no-share-probe --L $L"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_serial -- python3 $R/bench.py $ARGS --in-flight 1 > $O/trace_serial.json 2> $O/trace_serial.err
echo "serial trace rc=$?"; tail -1 $O/trace_serial.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_flight -- python3 $R/bench.py $ARGS > $O/trace_flight.json 2> $O/trace_flight.err
echo "in-flight trace rc=$?"; tail -1 $O/trace_flight.err
cd $R
python3 $R/scripts/summarize_profile_r4.py $O r04 > $O/summary.txt 2>&1; tail -30 $O/summary.txt
find $O -name "*kernel_trace.csv" -size +4M -delete
du -sh $O
