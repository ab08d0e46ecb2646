"""Kernel time vs batch size for the two kernel forms (T2I-10M shape, default index)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mysteryann_amd import index as mi, synth
n, d, L = 10_000_000, 200, 30  # bench.py's default graph and operating point
base, queries, learn = synth.t2i_torch(n, 10000, d, n_learn=n)
idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None); idx.SetBaseData(base); idx.BuildBootstrapGraph(learn=learn, max_degree=384, row_keep=128); del learn
for nq in (1, 64, 256, 700, 1000, 1250, 1500, 2000, 2500, 4000, 10000):
    row = [f"nq={nq:6d}"]
    for kern in (1, 3, 2):
        idx.set_option("search_kernel", kern)
        q = queries[:nq].contiguous()
        for _ in range(3):
            out = idx.SearchRoarGraphBatchDevice(q, 10, L)
        torch.cuda.synchronize()
        ms = idx.info("last_kernel_ns") / 1e6
        row.append(f"{ {1: '1-wave', 2: '4-wave', 3: '2-wave'}[kern]}: {ms:7.3f} ms ({nq / ms * 1e3:9.0f} qps, grid {idx.info('last_grid')})")
    print("  ".join(row), flush=True)
