#!/bin/bash
# Round 6: what would a screening copy of FEWER bits per code let through?  The bound's row side |q| B grows with the quantisation step
# (8 bits: max / 127; 5 bits: max / 15 = 8.5 x; 4 bits: max / 7 = 18 x); variant libraries inflate that term (-DROAR_SCREEN_SLACK_MUL=n,
#   for n in 4 8 18; do scripts/build_variant.sh libroargraph_amd_slack$n.so -DROAR_SCREEN_SLACK_MUL=$n; done)
# while the estimate itself stays the 8-bit one (a coarser estimate scatters inside the wider bound: the slack is what decides).  Per
# library: rows per query that pass the screen on the T2I-10M benchmark launch (the kernel's own counters) and the stream's rate.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6
{
for lib in "" _slack4 _slack8 _slack18; do
  f=$PWD/mysteryann_amd/csrc/libroargraph_amd$lib.so
  [ -f $f ] || continue
  ROAR_AMD_LIB=$f python3 bench.py --graph-cache /tmp/roar_graph.index --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-share-probe --no-other-workloads --no-traffic --L 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; c=d['config']
print('libroargraph_amd$lib.so: rows on the 8-bit copy per query %.0f, rows on f32 per query %.0f (pass rate %.3f), stream %.3f M QPS, one launch %.3f M, implemented bytes per query %.0f KB' % (r['rows_on_8bit_copy_per_query'], r['rows_on_f32_per_query'], c['screen_pass_rate'], d['value']/1e6, c['value_one_launch_at_a_time']/1e6, r['bytes_per_launch']/r['queries_per_launch']/1e3))
"
done
} 2>&1 | tee gpurun_out/r6/coarser_screen_probe.txt
