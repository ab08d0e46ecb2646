"""Does grouping similar queries raise the L2 hit rate enough to matter?  Sort the query batch by the
node the search will expand second (best of the entry point's neighbours) and time the same kernel."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mysteryann_amd import index as mi, synth
n, d, L = 10_000_000, 200, 60
base, queries, learn = synth.t2i_torch(n, 10000, d, n_learn=n // 2)
idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None); idx.SetBaseData(base); idx.BuildBootstrapGraph(learn=learn); del learn
ep, off, nbrs = idx.GetProjectionGraph()
def run(q, tag):
    for _ in range(3): out = idx.SearchRoarGraphBatchDevice(q, 10, L)
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        out = idx.SearchRoarGraphBatchDevice(q, 10, L); torch.cuda.synchronize(); ts.append(idx.info("last_kernel_ns") / 1e6)
    print(f"{tag}: kernel {np.median(ts):.3f} ms", flush=True); return out
run(queries, "original order")
en = torch.from_numpy(nbrs[int(off[ep]):int(off[ep + 1])].astype(np.int64)).cuda()
s1 = queries @ base[en].T                                   # hop 1 scores
h2 = en[s1.argmax(1)]                                       # node expanded second
key = h2
order = torch.argsort(key, stable=True)
run(queries[order].contiguous(), f"sorted by hop-2 node ({len(torch.unique(key))} groups)")
# two-level key: hop-2 node, then the best neighbour of that node
o2 = []
hn = h2.cpu().numpy(); offn = off
key2 = torch.zeros(len(queries), dtype=torch.int64, device="cuda")
for node in np.unique(hn):
    sel = torch.from_numpy(np.nonzero(hn == node)[0]).cuda()
    nb = torch.from_numpy(nbrs[int(offn[node]):int(offn[node + 1])].astype(np.int64)).cuda()
    key2[sel] = nb[(queries[sel] @ base[nb].T).argmax(1)]
order2 = torch.argsort(h2 * n + key2, stable=True)
run(queries[order2].contiguous(), f"sorted by (hop-2, hop-3) nodes ({len(torch.unique(h2 * n + key2))} groups)")
perm = torch.randperm(len(queries), device="cuda")
run(queries[perm].contiguous(), "random permutation")
