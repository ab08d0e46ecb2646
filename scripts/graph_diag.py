"""Diagnostics of a built 10 M graph: reachability from the entry point, in-degree of the true
neighbours the search misses."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import scipy.sparse as sp
from scipy.sparse.csgraph import breadth_first_order
from mysteryann_amd import index as mi, synth
n, d, nq, k = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 200, 10_000, 10
base, queries, learn = synth.t2i_torch(n, nq, d, n_learn=n // 4)
idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None); idx.SetBaseData(base)
idx.BuildBootstrapGraph(learn=learn); del learn
gt = idx.ExactKnnDevice(queries, k, want_dists=False)[0].cpu().numpy().view(np.uint32)
ep, off, nbrs = idx.GetProjectionGraph()
off = off.astype(np.int64)
indeg = np.bincount(nbrs, minlength=n)
outdeg = np.diff(off)
print("avg out", outdeg.mean(), "in-degree: zero", (indeg == 0).sum(), "<=2", (indeg <= 2).sum(), "median", np.median(indeg), "max", indeg.max())
g = sp.csr_matrix((np.ones(len(nbrs), np.int8), nbrs, off), shape=(n, n))
order = breadth_first_order(g, ep, directed=True, return_predecessors=False)
print("reachable from ep:", len(order), "of", n)
reach = np.zeros(n, bool); reach[order] = True
for L in (70, 150, 500):
    out = idx.SearchRoarGraphBatchDevice(queries, k, L)
    ids = out["ids"].cpu().numpy().view(np.uint32)
    hit = (gt[:, :, None] == ids[:, None, :]).any(axis=2)
    missed = gt[~hit]
    found = gt[hit]
    print(f"L={L}: recall {hit.mean():.4f}; missed true neighbours: {len(missed)}, unreachable {(~reach[missed]).sum()}, "
          f"in-degree median missed {np.median(indeg[missed])} vs found {np.median(indeg[found])}; "
          f"missed with indeg<=2: {(indeg[missed] <= 2).mean():.3f}, found with indeg<=2: {(indeg[found] <= 2).mean():.3f}")
    # rank position of missed ones in gt
    pos = np.nonzero(~hit)[1]
    print("   missed by gt rank:", np.bincount(pos, minlength=k).tolist())
bn = base.norm(dim=1).cpu().numpy()
print("norm percentiles of all base:", np.percentile(bn, [50, 90, 99]).round(4), "of gt neighbours:", np.percentile(bn[gt.reshape(-1)], [10, 50, 90]).round(4),
      "of missed(L=500):", np.percentile(bn[missed], [10, 50, 90]).round(4))
