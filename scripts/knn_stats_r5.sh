#!/bin/bash
# Round 5: rocprofv3 --kernel-trace --stats of one 65 536-query block of the builder's lists (scripts/knn_profile.py, twice) ->
# gpurun_out/r5/knn_kernel_stats.csv (the kNN kernels' rows) + the pass kernel's launches of the second run, by duration.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5/knnprof
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/scripts/knn_profile.py > $O.log 2>&1
echo "rc=$?"; tail -2 $O.log
cd $R
python3 - <<PY
import csv, glob, os
O = "$O"
st = max(glob.glob(O + "/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(st)))
with open("$R/gpurun_out/r5/knn_kernel_stats.csv", "w") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader()
    for r in rows[:10]:
        r = dict(r); r["Name"] = r["Name"][:150]; w.writerow(r)
tr = max(glob.glob(O + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
k = [r for r in csv.DictReader(open(tr)) if "knn_pass_bf16" in r["Kernel_Name"]]
k.sort(key=lambda r: int(r["Start_Timestamp"]))
half = k[len(k) // 2:]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in half]
print("pass kernel launches of the second run (ms):", " ".join(f"{x:.1f}" for x in d))
print("last pass = the last 8 launches: %.1f ms; all passes: %.1f ms" % (sum(d[-8:]), sum(d)))
PY
