"""Distribution of per-query work in the bipartite search (T2I-10M, 2 M training queries)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mysteryann_amd import index as mi, synth
n, d, nq, k, nl, M = 10_000_000, 200, 10_000, 10, 2_000_000, 35
base, queries, learn = synth.t2i_torch(n, nq, d, n_learn=nl)
idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None); idx.SetBaseData(base)
knn = torch.empty((nl, M), dtype=torch.int32, device="cuda")
for a in range(0, nl, 1 << 16):
    b = min(a + (1 << 16), nl); knn[a:b] = idx.ExactKnnDevice(learn[a:b], M, want_dists=False)[0]
idx._learn_base_knn = knn.cpu().numpy().view(np.uint32)
par = mi.Parameters(); par.Set("M_pjbp", M); idx.BuildBipartite(nl, None, n, None, par)
init = torch.from_numpy(mi.IndexBipartite.BipartiteEntryIds(2024, nq, 10, n).astype(np.int32)).cuda()
for L in (10, 100):
    for _ in range(2): out = idx.SearchBipartiteGraphBatchDevice(queries, k, L, init)
    torch.cuda.synchronize()
    c = out["cmps"].float().cpu().numpy(); sc = out["scanned"].float().cpu().numpy()
    print(f"L={L} kernel {idx.info('last_kernel_ns') / 1e6:.3f} ms grid {idx.info('last_grid')}  cmps mean {c.mean():.0f} p50 {np.percentile(c, 50):.0f} "
          f"p90 {np.percentile(c, 90):.0f} p99 {np.percentile(c, 99):.0f} max {c.max():.0f}; scanned mean {sc.mean():.0f} max {sc.max():.0f}; "
          f"zero-cmps queries {(c == 0).mean():.3f}", flush=True)
