R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/calib; mkdir -p $O; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/scripts/calib_fetch.py > $O/out.txt 2> $O/err.txt
cat $O/out.txt; tail -3 $O/err.txt
python3 - <<'PY'
import csv,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",os.getcwd())+"/gpurun_out/calib"
f=glob.glob(O+"/fetch/*/*_counter_collection.csv")[0]
for r in csv.DictReader(open(f)):
    if "roar_distance" in r["Kernel_Name"]:
        dur=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
        print("roar_distance_kernel FETCH_SIZE(KB)=",r["Counter_Value"],"dur_ms=",dur/1e6, "known GB=",20000*256*800/1e9, "ratio counter*1024/known=", float(r["Counter_Value"])*1024/(20000*256*800), "GB/s=",20000*256*800/dur)
PY
