timeout 600 python bench.py --workload sift --steps 10 --warmup 2 --batches 2 --no-cpu-baseline --no-host-path --no-share-probe --no-other-workloads --no-traffic --force-screen > gpurun_out/r5_bench_sift_screen.json 2> gpurun_out/r5_bench_sift_screen.err
timeout 600 python bench.py --workload sift --steps 10 --warmup 2 --batches 2 --no-cpu-baseline --no-host-path --no-share-probe --no-other-workloads --no-traffic > gpurun_out/r5_bench_sift.json 2> gpurun_out/r5_bench_sift.err
python3 - <<'PY'
import json
for f in ('sift_screen','sift'):
    try:
        r=json.load(open(f'gpurun_out/r5_bench_{f}.json')); c=r['config']; rf=r['roofline']; one=rf['one_launch_at_a_time']
        print(f, 'value %.3fM'%(r['value']/1e6), 'serial %.3fM'%(c['value_one_launch_at_a_time']/1e6), 'L',c['L_pq'], 'screen', c['screen'][:12], 'kernel_ms one', one['kernel_ms'], 'impl frac one', one['frac'], 'pass', c.get('screen_pass_rate'), 'cmps ratio', rf['kernel_cmps']/rf['avg_cmps'])
    except Exception as e: print(f, 'ERR', e)
PY
