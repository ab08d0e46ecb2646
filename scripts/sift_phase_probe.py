"""Phase timers of the instrumented kernel on the SIFT-1M shape (128-d L2; bench.py --workload sift --graph-cache FILE wrote the graph):
shader-clock cycles per hop of the one wave of a query, for a lone query, 1 250 and 10 000 queries, at the L_pq given.
   python scripts/sift_phase_probe.py /tmp/sift.index 170,50"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mysteryann_amd import index as mi, synth

cache = sys.argv[1]
Ls = [int(x) for x in sys.argv[2].split(",")]
n, d = 1_000_000, 128
dev = torch.device("cuda", 0)
base, q, _ = synth.sift_torch(n, 10_000, d, seed=1234, n_learn=0, device=dev, query_seed=1236)
idx = mi.IndexBipartite(d, n, mi.L2, None)
idx.SetBaseData(base)
idx.LoadProjectionGraph(cache)
names = ["select", "adj", "filter", "score", "merge"]
for L in Ls:
    for nq in (1, 1250, 10000):
        qq = q[:nq].contiguous()
        idx.set_option("profile_phases", 0)
        best = 1e9
        for _ in range(3):
            out = idx.SearchRoarGraphBatchDevice(qq, 10, L); torch.cuda.synchronize()
            best = min(best, idx.info("last_kernel_ns") / 1e3)
        idx.set_option("profile_phases", 1)
        for _ in range(2):
            out = idx.SearchRoarGraphBatchDevice(qq, 10, L); torch.cuda.synchronize()
        pr = idx.LastProfile(nq).astype(np.float64)
        H = out["hops"].double().sum().item()
        ph = " ".join(f"{nm} {pr[:, 1 + i].sum() / H:7.0f}" for i, nm in enumerate(names))
        print(f"L={L} nq={nq:5d} {best:8.1f} us grid {idx.info('last_grid')} tags {idx.info('last_vis_entries')} regpool {idx.info('last_register_pool')} | cycles/hop {pr[:, 0].sum() / H:7.0f}: {ph} | "
              f"per pass: issue {pr[:, 8].sum() / max(pr[:, 11].sum(), 1):6.0f} wait+fma {pr[:, 9].sum() / max(pr[:, 11].sum(), 1):6.0f} tail {pr[:, 10].sum() / max(pr[:, 11].sum(), 1):6.0f}, "
              f"passes/hop {pr[:, 11].sum() / H:.2f} rows/hop {out['cmps'].double().sum().item() / H:.1f} survivors/hop {pr[:, 12].sum() / H:.1f} spec hit {pr[:, 6].sum() / H:.2f}", flush=True)
