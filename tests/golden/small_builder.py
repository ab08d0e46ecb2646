"""Small CPU (numpy) RoarGraph-style builder used ONLY to make test fixtures and small test graphs.

It follows the structure of the reference builder (src/index_bipartite.cpp:1043-1277 LinkProjection,
:1612-1694 PruneBiSearchBaseGetBase, :2004-2041 CalculateProjectionep) in broad strokes -- exact
learn->base kNN, pivot = nearest base point of each training query, occlusion pruning, reverse
edges, degree cap, connectivity supplement of up to 2*M extra edges -- but it is an independent,
simplified implementation: graph *quality* is not part of the search-path parity contract, only the
on-disk format and the search results on a given graph are.
"""
import numpy as np


def pair_dist(metric, a, B):
    """Reference-signed distance of row a to rows B: L2 -> squared L2, IP/COSINE -> -dot (distance.h:223)."""
    if metric == 0:
        d = B - a[None, :]
        return np.einsum("ij,ij->i", d, d).astype(np.float32)
    return (-(B @ a)).astype(np.float32)


def exact_knn(Q, X, k, metric, chunk=2048):
    """Brute-force top-k of every row of Q in X, ordered by (distance, id)."""
    ids = np.zeros((Q.shape[0], k), np.uint32)
    ds = np.zeros((Q.shape[0], k), np.float32)
    xn = np.einsum("ij,ij->i", X, X) if metric == 0 else None
    for i in range(0, Q.shape[0], chunk):
        q = Q[i:i + chunk]
        dot = q @ X.T
        if metric == 0:
            dist = np.einsum("ij,ij->i", q, q)[:, None] - 2 * dot + xn[None, :]
        else:
            dist = -dot
        part = np.argpartition(dist, min(k, dist.shape[1] - 1), axis=1)[:, :k]
        pd = np.take_along_axis(dist, part, 1)
        order = np.lexsort((part, pd), axis=1)
        ids[i:i + chunk] = np.take_along_axis(part, order, 1)
        ds[i:i + chunk] = np.take_along_axis(pd, order, 1)
    return ids, ds


def _prune(target, cand_ids, X, M, metric):
    """Occlusion rule: keep p unless some already-kept r has dist(p, r) < dist(p, target); then
    back-fill with the nearest leftovers up to M (cf. PruneBiSearchBaseGetBase)."""
    cand_ids = np.array([c for c in dict.fromkeys(int(c) for c in cand_ids) if c != target], np.int64)
    if cand_ids.size == 0:
        return []
    C = X[cand_ids]
    dt = pair_dist(metric, X[target], C)
    order = np.lexsort((cand_ids, dt))
    cand_ids, C, dt = cand_ids[order], C[order], dt[order]
    if metric == 0:
        G = C @ C.T
        sq = np.diag(G)
        dcc = sq[:, None] - 2 * G + sq[None, :]
    else:
        dcc = -(C @ C.T)
    kept = [0]
    for p in range(1, len(cand_ids)):
        if len(kept) >= M:
            break
        if not np.any(dcc[p, kept] < dt[p]):
            kept.append(p)
    if len(kept) < M:
        ks = set(kept)
        for p in range(1, len(cand_ids)):
            if len(kept) >= M:
                break
            if p not in ks:
                kept.append(p)
                ks.add(p)
    return [int(cand_ids[p]) for p in kept]


def build_roar_like(base, learn, metric, M_sq=32, M=16, seed=0):
    """-> ep, list of u32 arrays (out-neighbour lists; order = pruned list then supply edges)."""
    n = base.shape[0]
    knn, _ = exact_knn(learn, base, min(M_sq, n), metric)
    graph = [[] for _ in range(n)]
    for q in range(learn.shape[0]):  # projection: pivot <- pruned co-neighbours of one training query
        pivot = int(knn[q, 0])
        graph[pivot] = _prune(pivot, knn[q, 1:], base, M, metric)
    for v in range(n):  # reverse edges
        for u in list(graph[v]):
            if v not in graph[u]:
                graph[u].append(v)
    for v in range(n):  # degree cap
        if len(graph[v]) > M:
            graph[v] = _prune(v, graph[v], base, M, metric)
    # connectivity supplement: exact base->base neighbours stand in for the reference's graph search
    bknn, _ = exact_knn(base, base, min(2 * M + 1, n), metric)
    supply = [_prune(v, bknn[v], base, M, metric) for v in range(n)]
    for v in range(n):
        for u in list(supply[v]):
            if v not in supply[u] and len(supply[u]) < 2 * M:
                supply[u].append(v)
    out = []
    for v in range(n):
        have = set(graph[v])
        extra = []
        for u in supply[v]:
            if len(extra) >= 2 * M:
                break
            if u not in have:
                extra.append(u)
                have.add(u)
        out.append(np.array(graph[v] + extra, np.uint32))
    center = base.mean(axis=0)
    ep = int(np.argmin(((base - center[None, :]) ** 2).sum(axis=1)))
    return ep, out


def to_csr(adj):
    offsets = np.zeros(len(adj) + 1, np.uint64)
    offsets[1:] = np.cumsum([len(a) for a in adj])
    nbrs = np.concatenate(adj).astype(np.uint32) if len(adj) else np.zeros(0, np.uint32)
    return offsets, nbrs
