cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
BS=10000,2500 python3 $R/scripts/stream_probe.py > $R/gpurun_out/r5_pmc_base.txt 2>&1
for bs in 10000 2500; do
  i=0
  for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    BS=$bs rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_${bs}_$i -o x -- python3 $R/scripts/stream_probe.py > $R/gpurun_out/r5_pmc_${bs}_$i.txt 2>&1
    find /tmp/prof_${bs}_$i -name "*.csv" | head -5
    f=$(find /tmp/prof_${bs}_$i -name "*counter_collection.csv" | head -1)
    python3 - "$f" $bs <<'PY'
import csv,sys
f,bs=sys.argv[1],sys.argv[2]
rows=[r for r in csv.DictReader(open(f)) if 'roar_search_v2_kernel' in r['Kernel_Name'] and 'build' not in r['Kernel_Name']]
agg={}
for r in rows:
    k=(int(r['Dispatch_Id']),r['Counter_Name'])
    agg[k]=agg.get(k,0.0)+float(r['Counter_Value'])
last=sorted(agg.items())[-8:]
print('bs='+bs, [(k[0],k[1],'%.4g'%v) for k,v in last])
PY
  done
done > $R/gpurun_out/r5_pmc_summary.txt 2>&1
cat $R/gpurun_out/r5_pmc_summary.txt
