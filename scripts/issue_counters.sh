#!/bin/bash
# Is the search launch bound by instruction issue?  Counter passes (rocprofv3 --pmc, one small set per pass, kernel trace only) over
# the timed launches of bench.py's profiled child (same data, graph, L_pq, batches as the bench line).
#   bash scripts/issue_counters.sh [L_pq] [extra bench args]      -> gpurun_out/r3/issue_counters.txt
set -u
L=${1:-50}; shift || true
cd /tmp; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
G=/tmp/r3_bench_graph.index
[ -f $G ] || python3 $ROOT/bench.py --graph-cache $G --steps 4 --warmup 2 --no-cpu-baseline --no-traffic "$@" > $OUT/issue_bench.json 2> $OUT/issue_bench.err
: > $OUT/issue_counters.txt
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" \
           "VALUBusy SALUBusy" "MemUnitBusy MemUnitStalled" "LDSBankConflict VALUUtilization"; do
  D=/tmp/pmc_$$_$(echo $set | tr ' ' '_' | cut -c1-40); rm -rf $D
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $D -- python3 $ROOT/bench.py --pmc-child --graph-cache $G --L $L --steps 8 --warmup 1 \
      --learn 0 --ref-learn 1 --ref-learn2 0 "$@" > /dev/null 2> $D.err || { echo "# set [$set] failed: $(tail -2 $D.err | tr '\n' ' ')" >> $OUT/issue_counters.txt; continue; }
  python3 - "$D" >> $OUT/issue_counters.txt <<'PY'
import csv, glob, sys, collections
f = max(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "roar_search_v2_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    v = v[1:] if len(v) > 1 else v  # (the first launch is the warm-up)
    print(f"{k}: mean over {len(v)} launches {sum(v) / len(v):.6g}")
PY
done
cat $OUT/issue_counters.txt
