#!/bin/bash
# Round-2 measurement recipe (run on the GPU box through gpurun; the graph is deterministic, so every pass below builds
# the same graph -- config.graph_hash says so):
#   1. default bench line (with cpu_baseline and the host-buffer path)
#   2. rocprofv3 --kernel-trace --stats of the same command (no cpu baseline / host path, fewer steps)
#   3. two separate --pmc passes (FETCH_SIZE / WRITE_SIZE) as MI355X_MICROARCH.md prescribes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2
mkdir -p $O
cd $R
rm -f /tmp/r2_graph.index
python3 bench.py --graph-cache /tmp/r2_graph.index > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err; cat $O/bench.json
export TMPDIR=/tmp
ARGS="--graph-cache /tmp/r2_graph.index --steps 8 --warmup 1 --no-cpu-baseline --no-host-path --no-traffic --L $(python3 -c "import json;print(json.load(open('$O/bench.json'))['config']['L_pq'])")"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py $ARGS > $O/trace.json 2> $O/trace.err
echo "trace rc=$?"; tail -2 $O/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $ARGS > $O/pmc_fetch.json 2> $O/pmc_fetch.err
echo "pmc fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py $ARGS > $O/pmc_write.json 2> $O/pmc_write.err
echo "pmc write rc=$?"
python3 $R/scripts/summarize_profile.py $O r02 --no-write-profiles > $O/summary.json 2> $O/summary.err; tail -5 $O/summary.err
# keep only the small summaries (gpurun_out is capped at 64 MiB)
find $O -name "*kernel_trace.csv" -size +4M -delete
find $O -name "*counter_collection.csv" -size +4M -delete
du -sh $O
