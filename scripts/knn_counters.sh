#!/bin/bash
# PMC counters of the bf16 pass kernel of the exact kNN generator (separate rocprofv3 --pmc passes, --kernel-trace only):
# one block of 65 536 training queries x 10 M base rows (scripts/knn_profile.py), the LAST pass (all rows) of the second run.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${ROUND_TAG:-r4}/knn_counters
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/scripts/knn_profile.py > $O/p$i.log 2>&1
  echo "pass $i ($set): rc=$?"
done
cd $R
python3 - <<PY
import csv, glob
for d in sorted(glob.glob("$O/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "knn_pass_bf16" in r["Kernel_Name"]]
        names = sorted(set(r["Counter_Name"] for r in rows))
        for nm in names:
            v = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == nm]
            dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if r["Counter_Name"] == nm]
            # the largest launch = the last pass (all rows)
            j = max(range(len(v)), key=lambda i_: dur[i_])
            print(f"{nm:28s} {v[j]:.4e}   (last pass of the block, {dur[j]:.1f} ms under the profiler)")
PY
