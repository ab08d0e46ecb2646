timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r5_gpu_tests.log
tail -12 gpurun_out/r5_gpu_tests.log | cut -c1-300
timeout 600 python bench.py --workload webvid --steps 10 --warmup 2 --batches 2 --cpu-seconds 3 --no-host-path --no-share-probe --no-other-workloads > gpurun_out/r5_bench_webvid.json 2> gpurun_out/r5_bench_webvid.err
python3 - <<'PY'
import json
for f in ('webvid',):
    try:
        r=json.load(open(f'gpurun_out/r5_bench_{f}.json')); c=r['config']; rf=r['roofline']; one=rf['one_launch_at_a_time']
        print(f, 'value %.3fM'%(r['value']/1e6), 'serial %.3fM'%(c['value_one_launch_at_a_time']/1e6), 'L',c['L_pq'], 'kernel_ms one', one['kernel_ms'], 'frac one', one['frac'], 'frac', rf['frac'], 'traffic/impl', (rf['traffic'] or 0)/rf['bytes_per_launch'], 'traffic', rf['traffic'], 'cpu ids', r['cpu_baseline']['gpu_vs_cpu_ids_identical_fraction'])
    except Exception as e: print(f, 'ERR', e)
PY
tail -3 gpurun_out/r5_bench_webvid.err | cut -c1-300
