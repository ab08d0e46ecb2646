#!/bin/bash
# Extra PMC passes (own runs, --kernel-trace only): where the search kernel's waves spend their time,
# L2 hit rate, LDS conflicts; MFMA busy share of the exact-kNN kernel.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r1c; mkdir -p $O; cd $R; export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --L 30"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -- python3 $R/bench.py $ARGS > $O/sq.json 2> $O/sq.err; echo "sq rc=$?"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $O/tcc -- python3 $R/bench.py $ARGS > $O/tcc.json 2> $O/tcc.err; echo "tcc rc=$?"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/knn -- python3 $R/scripts/knn_bench.py > $O/knn.txt 2> $O/knn.err; echo "knn rc=$?"; cat $O/knn.txt
python3 - <<'PY'
import csv,glob,os,collections
O=os.environ.get("GRAFT_REPO_ROOT",os.getcwd())+"/gpurun_out/r1c"
def agg(sub,pat,last):
    fs=glob.glob(f"{O}/{sub}/*/*_counter_collection.csv")
    if not fs: print(sub,"no csv"); return
    rows=[r for r in csv.DictReader(open(fs[0])) if pat in r["Kernel_Name"]]
    by=collections.defaultdict(list)
    for r in rows: by[r["Counter_Name"]].append((int(r["Dispatch_Id"]),float(r["Counter_Value"]),int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
    for k,v in by.items():
        v=sorted(v)[-last:]
        print(sub,k,"mean over last",len(v),"launches:",sum(x[1] for x in v)/len(v),"avg dur ms",sum(x[2] for x in v)/len(v)/1e6)
agg("sq","roar_search_kernel<1, true",5); agg("tcc","roar_search_kernel<1, true",5); agg("knn","knn_pass_kernel",3)
PY
find $O -name "*_kernel_trace.csv" -delete
