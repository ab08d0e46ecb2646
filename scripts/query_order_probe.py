"""Does the order in which a batch's queries are dequeued matter (similar queries in flight together -> shared rows in
L2 / MALL)?  Same 10 000 queries, same kernel: original order, sorted by k-means cluster of the query vectors, sorted by the
top-1 result id's cluster, and a random permutation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from _graph import t2i_index
idx, base, queries = t2i_index(10_000_000, 10000, "reference", 0, "/tmp/r2_graph.index")
L = 55
def run(q, tag):
    ts = []
    for _ in range(7):
        out = idx.SearchRoarGraphBatchDevice(q, 10, L); torch.cuda.synchronize(); ts.append(idx.info("last_kernel_ns") / 1e6)
    print(f"{tag}: kernel best {min(ts):.3f} median {np.median(ts):.3f} ms", flush=True); return out
out = run(queries, "original order")
def kmeans(x, k, iters=12):
    g = torch.Generator(device=x.device); g.manual_seed(1)
    c = x[torch.randperm(len(x), generator=g, device=x.device)[:k]].clone()
    for _ in range(iters):
        a = (x @ c.T * 2 - (c * c).sum(1)[None]).argmax(1)
        for j in range(k):
            m = a == j
            if m.any(): c[j] = x[m].mean(0)
    return a
for k in (16, 64, 256, 1024):
    a = kmeans(queries, k)
    order = torch.argsort(a, stable=True)
    run(queries[order].contiguous(), f"sorted by k-means cluster of the query (k={k})")
top1 = base[out["ids"][:, 0].long()]
a = kmeans(top1[:, :200].contiguous(), 256)
run(queries[torch.argsort(a, stable=True)].contiguous(), "sorted by k-means cluster of the top-1 result (k=256)")
run(queries[torch.randperm(10000, device="cuda")].contiguous(), "random permutation")
