export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/knnprof -- python3 $GRAFT_REPO_ROOT/scripts/knn_profile.py > /tmp/knnprof.log 2>&1
tail -2 /tmp/knnprof.log
python3 - <<'PY'
import csv, glob
f = max(glob.glob('/tmp/knnprof/**/*kernel_trace.csv', recursive=True))
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sel = [r for r in rows if 'knn_' in r['Kernel_Name']]
# the second rep only: the last half
half = sel[len(sel)//2:]
t0 = int(half[0]['Start_Timestamp'])
for r in half:
    nm = r['Kernel_Name'].split('(')[0][-60:]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e6:9.3f} ms  +{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6:8.3f} ms  grid {r.get('Grid_Size','?'):>9}  {nm}")
PY
