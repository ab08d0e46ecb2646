timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r5_gpu_tests.log
tail -3 gpurun_out/r5_gpu_tests.log
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_long_pools.txt
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'scripts'))
import torch
from _graph import t2i_index
idx, base, queries = t2i_index(10_000_000, 10000, "reference", 0, "/tmp/roar_graph.index", say=lambda *a: None)
q = queries[:10000].contiguous()
for L in (50, 100, 200, 500, 1000, 2000):
    best = 1e9
    for _ in range(4):
        o = idx.SearchRoarGraphBatchDevice(q, 10, L); torch.cuda.synchronize()
        best = min(best, idx.info("last_kernel_ns") / 1e6)
    print(f"L_pq={L:5d} {best:8.3f} ms per 10 000 queries  cmps {o['cmps'].double().mean().item():8.0f} hops {o['hops'].double().mean().item():7.1f} tags {idx.info('last_vis_entries')} grid {idx.info('last_grid')} protect {idx.info('last_tag_protect') if False else ''}", flush=True)
q1 = queries[:1].contiguous()
for L in (50, 1000):
    best = 1e9
    for _ in range(6):
        o = idx.SearchRoarGraphBatchDevice(q1, 10, L); torch.cuda.synchronize()
        best = min(best, idx.info("last_kernel_ns") / 1e3)
    print(f"lone query L_pq={L}: {best:.1f} us", flush=True)
PY
for D in 96 768; do
timeout 600 python bench.py --workload webvid --dim $D --n 1000000 --steps 10 --warmup 2 --batches 2 --no-cpu-baseline --no-host-path --no-share-probe --no-other-workloads > gpurun_out/r5_bench_dim$D.json 2> gpurun_out/r5_bench_dim$D.err
python3 - $D <<'PY'
import json,sys
D=sys.argv[1]
r=json.load(open(f'gpurun_out/r5_bench_dim{D}.json')); c=r['config']; rf=r['roofline']; one=rf['one_launch_at_a_time']
print(D, 'value %.3fM'%(r['value']/1e6), c['value_regime'][:24], 'ms/step %.3f'%r['ms_per_step'], 'L',c['L_pq'],'@',c['recall_at_10'], 'frac', rf['frac'], 'impl frac', rf.get('implemented_bytes_frac'), 'one: kernel_ms', one['kernel_ms'], 'frac', one['frac'], 'impl', one.get('implemented_bytes_frac'), 'traffic/impl', (rf['traffic'] or 0)/rf['bytes_per_launch'])
PY
done | tee gpurun_out/r5_generic_dims.txt
