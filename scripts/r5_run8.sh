timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r5_gpu_tests.log
tail -4 gpurun_out/r5_gpu_tests.log
for D in -1 0 4 16 32; do
python3 - $D <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'scripts'))
import torch
from _graph import t2i_index
D = int(sys.argv[1])
idx, base, queries = t2i_index(10_000_000, 10000, "reference", 0, "/tmp/roar_graph.index", say=lambda *a: None)
q = queries[:10000].contiguous()
idx.set_option("merge_defer", D)
row = [f"merge_defer {D:3d}:"]
for L in (100, 200, 500, 1000, 2000):
    best = 1e9
    for _ in range(4):
        o = idx.SearchRoarGraphBatchDevice(q, 10, L); torch.cuda.synchronize()
        best = min(best, idx.info("last_kernel_ns") / 1e6)
    row.append(f"L={L} {best:7.2f} ms cmps {o['cmps'].double().mean().item():8.0f} tags {idx.info('last_vis_entries')}")
print("  ".join(row), flush=True)
PY
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_merge_defer.txt
