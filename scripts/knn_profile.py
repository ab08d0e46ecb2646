"""One chunk of the builder's learn->base lists (65 536 queries x N rows, k = 100) through the bf16-screened generator, for
rocprofv3 --kernel-trace --stats:  rocprofv3 --kernel-trace --stats -d gpurun_out/knnprof -- python3 scripts/knn_profile.py [N [D]]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mysteryann_amd import index as mi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 200
base, q, learn = synth.t2i_torch(n, 16, d, n_learn=65536)
idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None); idx.SetBaseData(base)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ids, _ = idx.ExactKnnDevice(learn, 100, want_dists=False)
    torch.cuda.synchronize(); print(f"n {n} d {d} rep {rep}: {time.perf_counter() - t0:.3f} s", flush=True)
