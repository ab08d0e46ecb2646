timeout 300 python -m pytest tests/test_search_stream.py -x -q 2>&1 | tail -15 > gpurun_out/r5_stream_tests.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-traffic --no-cpu-baseline --no-host-path > gpurun_out/r5_bench_a.json 2> gpurun_out/r5_bench_a.err
tail -3 gpurun_out/r5_stream_tests.log; grep "batch stream\|timeline" gpurun_out/r5_bench_a.err
python3 -c "
import json
r=json.load(open('gpurun_out/r5_bench_a.json')); c=r['config']
print(r['value'], r['ms_per_step'], c['value_regime'])
print('stream', c['value_batch_stream'], 'launches', c['value_launches_in_flight'], 'serial', c['value_one_launch_at_a_time'])
e=c['emulated_share_of_8']; print({k:e[k] for k in ('regime','ms_per_step','projected_speedup_over_this_run','ms_per_step_batch_stream','ms_per_step_launches_in_flight','batch_stream')})
print(r['roofline']['kernel_ms'])
"
