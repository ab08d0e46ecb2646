timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r5_gpu_tests.log
tail -3 gpurun_out/r5_gpu_tests.log
for D in 96 768; do
timeout 600 python bench.py --workload webvid --dim $D --n 1000000 --steps 10 --warmup 2 --batches 2 --no-cpu-baseline --no-host-path --no-share-probe --no-other-workloads > gpurun_out/r5_bench_dim$D.json 2> gpurun_out/r5_bench_dim$D.err
python3 - $D <<'PY'
import json,sys
D=sys.argv[1]
r=json.load(open(f'gpurun_out/r5_bench_dim{D}.json')); c=r['config']; rf=r['roofline']; one=rf['one_launch_at_a_time']
print(D, 'value %.3fM'%(r['value']/1e6), c['value_regime'][:24], 'ms/step %.3f'%r['ms_per_step'], 'L',c['L_pq'],'@',c['recall_at_10'], 'frac', rf['frac'], 'impl frac', rf.get('implemented_bytes_frac'), 'one: kernel_ms', one['kernel_ms'], 'frac', one['frac'], 'impl', one.get('implemented_bytes_frac'), 'traffic/impl', (rf['traffic'] or 0)/rf['bytes_per_launch'])
PY
done | tee gpurun_out/r5_generic_dims.txt
