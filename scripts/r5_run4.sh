ROAR_LOG_CALIBRATION=1 timeout 1500 python bench.py > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err
tail -c 2500 gpurun_out/r5_bench_default.err
python3 -c "
import json
r=json.load(open('gpurun_out/r5_bench_default.json')); c=r['config']
print(r['value'], r['ms_per_step'], c['value_regime'])
print('roofline', {k:r['roofline'][k] for k in ('achieved','frac','kernel_ms','traffic','implemented_bytes_frac')})
print('one', {k:r['roofline']['one_launch_at_a_time'].get(k) for k in ('achieved','frac','kernel_ms','traffic','implemented_bytes_frac')})
e=c['emulated_share_of_8']; print({k:e[k] for k in ('regime','ms_per_step','projected_speedup_over_this_run')})
print(json.dumps(r.get('other_workloads'),indent=0)[:3000])
print(r.get('cpu_baseline',{}).get('value'))
"
