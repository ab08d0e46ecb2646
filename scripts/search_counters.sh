#!/bin/bash
# PMC counters of the search kernel at a long pool (separate rocprofv3 --pmc passes, --kernel-trace only): T2I-10M, 10 000 queries,
# L_pq = $1 (default 1000), forgetful table of $2 tags (default 2048).  The graph is built by the first pass and cached in /tmp.
set -u
L=${1:-1000}; T=${2:-2048}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${ROUND_TAG:-r4}/search_counters_L$L
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
python3 $R/scripts/longpool_probe.py --L $L --tables $T --no-exact > $O/plain.log 2>&1
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/scripts/longpool_probe.py --L $L --tables $T --no-exact > $O/p$i.log 2>&1
  echo "pass $i ($set): rc=$?"
done
cd $R
grep L_pq $O/plain.log
python3 - <<PY
import csv, glob
for d in sorted(glob.glob("$O/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "roar_search_v2_kernel" in r["Kernel_Name"]]
        names = sorted(set(r["Counter_Name"] for r in rows))
        for nm in names:
            sel = [r for r in rows if r["Counter_Name"] == nm]
            r = sel[-1]  # the last launch of the script (the timed repeat)
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            print(f"{nm:24s} {float(r['Counter_Value']):.4e}   (last search launch, {dur:.1f} ms under the profiler, grid {r.get('Grid_Size', '?')})")
PY
