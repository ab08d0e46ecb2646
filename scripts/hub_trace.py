"""How concentrated are visited-set probes on high in-degree nodes?  Builds the T2I-shape graph, walks a few queries on the
host (plain best-first search, numpy distances) and reports which fraction of the scanned neighbours / of the revisits
fall into the top-H nodes by in-degree.  Decides whether an exact on-chip bitmap for hub nodes would pay."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mysteryann_amd import index as mi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2_000_000)
ap.add_argument("--nq", type=int, default=12)
ap.add_argument("--Ls", type=int, nargs="+", default=[30, 500, 2000])
args = ap.parse_args()
n, d = args.n, 200
base, queries, learn = synth.t2i_torch(n, 1000, d, n_learn=n)
idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None); idx.SetBaseData(base)
idx.BuildBootstrapGraph(learn=learn, max_degree=384, row_keep=128); torch.cuda.synchronize(); del learn
ep, off, nbrs = idx.GetProjectionGraph()
off = off.astype(np.int64)
indeg = np.bincount(nbrs, minlength=n)
order = np.argsort(-indeg, kind="stable")
rank = np.empty(n, np.int64); rank[order] = np.arange(n)
print(f"n={n} edges={len(nbrs)} in-degree: max {indeg.max()} top-1k share {indeg[order[:1000]].sum() / len(nbrs):.3f} "
      f"top-16k {indeg[order[:16384]].sum() / len(nbrs):.3f} top-64k {indeg[order[:65536]].sum() / len(nbrs):.3f} top-256k {indeg[order[:262144]].sum() / len(nbrs):.3f}", flush=True)
hb = base.cpu().numpy(); hq = queries.cpu().numpy()
for L in args.Ls:
    scanned_all, revisit_all, fresh_all = [], [], []
    for qi in range(args.nq):
        q = hq[qi]
        dist = lambda ids: -(hb[ids] @ q)
        pool = [(float(dist(np.array([ep]))[0]), int(ep), False)]
        visited = set()
        while True:
            cur = next((i for i, e in enumerate(pool) if not e[2]), None)
            if cur is None: break
            dv, v, _ = pool[cur]; pool[cur] = (dv, v, True)
            row = nbrs[off[v]:off[v + 1]]
            scanned_all.append(row)
            mask = np.fromiter((int(u) not in visited for u in row), bool, len(row))
            revisit_all.append(row[~mask]); fr = row[mask]; fresh_all.append(fr)
            visited.update(int(u) for u in fr)
            if len(fr):
                dd = dist(fr)
                for u, du in zip(fr, dd):
                    e = (float(du), int(u), False)
                    if len(pool) == L and e[:2] >= pool[-1][:2]: continue
                    if any(p[1] == e[1] for p in pool): continue
                    pool.append(e)
                pool.sort(key=lambda t: (t[0], t[1])); del pool[L:]
    sc = np.concatenate(scanned_all); rv = np.concatenate(revisit_all); fr = np.concatenate(fresh_all)
    line = f"L={L}: scanned/query {len(sc) / args.nq:.0f} fresh {len(fr) / args.nq:.0f} revisits {len(rv) / args.nq:.0f};"
    for H in (4096, 16384, 65536, 262144):
        line += f" top-{H}: scanned {np.mean(rank[sc] < H):.3f} revisits {np.mean(rank[rv] < H):.3f} fresh {np.mean(rank[fr] < H):.3f};"
    print(line, flush=True)
