timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r5_gpu_tests.log
tail -4 gpurun_out/r5_gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 3 --no-other-workloads --no-host-path --cpu-seconds 3 --graph-cache /tmp/roar_bench_graph.index > gpurun_out/r5_bench_b.json 2> gpurun_out/r5_bench_b.err
python3 - <<'PY'
import json
r=json.load(open('gpurun_out/r5_bench_b.json')); c=r['config']; rf=r['roofline']; one=rf['one_launch_at_a_time']
print('value %.3fM'%(r['value']/1e6), c['value_regime'][:30], 'stream %.3fM'%(c['value_batch_stream']/1e6), 'launches %.3fM'%(c['value_launches_in_flight']/1e6), 'serial %.3fM'%(c['value_one_launch_at_a_time']/1e6))
print('kernel one', one['kernel_ms'], 'frac one', one['frac'], 'frac', rf['frac'], 'traffic', rf['traffic'], 'share', c['emulated_share_of_8']['projected_speedup_over_this_run'], c['emulated_share_of_8']['ms_per_step'])
PY
for L in 200 500 1000 2000; do
timeout 600 python bench.py --steps 6 --warmup 2 --batches 2 --L $L --no-other-workloads --no-host-path --no-cpu-baseline --no-traffic --no-share-probe --graph-cache /tmp/roar_bench_graph.index > gpurun_out/r5_bench_L$L.json 2> gpurun_out/r5_bench_L$L.err
python3 - $L <<'PY'
import json,sys
L=sys.argv[1]
r=json.load(open(f'gpurun_out/r5_bench_L{L}.json')); c=r['config']; rf=r['roofline']; one=rf['one_launch_at_a_time']
print('L',L,'value %.3fM'%(r['value']/1e6), 'serial kernel_ms', one['kernel_ms'], 'cmps ratio', rf['kernel_cmps']/rf['avg_cmps'], 'impl frac one', one['frac'], c['visited'][:20], 'tags', rf.get('lds_bytes'))
PY
done
