"""Which visited ids does the beam search ask for again?  Replacement policies for the on-chip tag table (sets x 4 ways),
simulated on the exact probe sequence of real searches on the benchmark graph (T2I-10M shape, reference-shaped graph): the
search's trajectory does not depend on the table (a forgotten node is re-scored and its key dropped), so one exact run per
query gives the probe stream, and every policy is replayed over it.  Reported per L_pq and policy: rows scored / rows the
reference scores (1.000 = nothing forgotten).  Host-side study (numpy); the GPU only builds the graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from _graph import t2i_index

idx, base_t, queries = t2i_index(10_000_000, 10_000, "reference", 0, "/tmp/roar_graph.index")
ep, off, nbrs = idx.GetProjectionGraph()
off = off.astype(np.int64)
base = base_t.cpu().numpy()
NQ = int(os.environ.get("NQ", "10"))
Ls = [int(x) for x in os.environ.get("LS", "500,1000,2000").split(",")]
qs = queries[:NQ].cpu().numpy()


def trace(q, L):
    """exact SearchRoarGraph (index_bipartite.cpp:2311-2420); -> list of hops, each (ids probed in row order, fresh mask, survivor mask)"""
    d0 = -float(base[ep] @ q)
    pool_d, pool_i, pool_f = [d0], [ep], [False]
    visited = set()
    hops = []
    cur = 0
    while cur < len(pool_i):
        v = pool_i[cur]
        pool_f[cur] = True
        while cur < len(pool_i) and pool_f[cur]:
            cur += 1
        row = nbrs[off[v]:off[v + 1]]
        fresh = np.array([int(u) not in visited for u in row], bool)
        # (duplicates inside a row were removed at upload; ids of a row are distinct)
        new = row[fresh]
        visited.update(int(u) for u in new)
        surv = np.zeros(len(row), bool)
        if len(new):
            dd = -(base[new] @ q)
            sv = []
            for u, d in zip(new.tolist(), dd.tolist()):
                if len(pool_i) == L and (d, u) >= (pool_d[-1], pool_i[-1]):
                    sv.append(False)
                    continue
                lo = 0
                hi = len(pool_i)
                while lo < hi:
                    mid = (lo + hi) // 2
                    if (pool_d[mid], pool_i[mid]) < (d, u):
                        lo = mid + 1
                    else:
                        hi = mid
                if lo < len(pool_i) and pool_i[lo] == u and pool_d[lo] == d:
                    sv.append(False)
                    continue
                pool_d.insert(lo, d), pool_i.insert(lo, u), pool_f.insert(lo, False)
                if len(pool_i) > L:
                    pool_d.pop(), pool_i.pop(), pool_f.pop()
                if lo < cur:
                    cur = lo
                sv.append(True)
            surv[np.flatnonzero(fresh)] = sv
        hops.append((row.copy(), fresh, surv))
    return hops


def replay(hops, sets, policy, rng):
    """-> rows scored with this table / rows scored exactly"""
    ways = 4
    tab = np.zeros((sets, ways), np.int64) - 1
    prot = np.zeros((sets, ways), bool)
    age = np.zeros((sets, ways), np.int64)
    clock = 0
    scored = exact = 0
    for row, fresh, surv in hops:
        exact += int(fresh.sum())
        for j, u in enumerate(row.tolist()):
            clock += 1
            s = u % sets
            w = np.flatnonzero(tab[s] == u)
            if len(w):
                age[s, w[0]] = clock
                if policy == "protect" and surv[j]:
                    prot[s, w[0]] = True
                continue
            scored += 1  # fresh, or forgotten and scored again
            e = np.flatnonzero(tab[s] < 0)
            if len(e):
                k = e[0]
            elif policy == "random":
                k = (u // sets) & 3
            elif policy == "lru":
                k = int(np.argmin(age[s]))
            elif policy == "protect":
                un = np.flatnonzero(~prot[s])
                k = un[((u // sets) & 3) % len(un)] if len(un) else (u // sets) & 3
            elif policy == "protect_lru":
                un = np.flatnonzero(~prot[s])
                k = un[int(np.argmin(age[s][un]))] if len(un) else int(np.argmin(age[s]))
            tab[s, k] = u
            age[s, k] = clock
            # a re-scored node's key is a duplicate or worse than the pool's last: it is a survivor only when it is new
            prot[s, k] = bool(surv[j] and fresh[j]) if policy.startswith("protect") else False
    return scored / max(exact, 1)


rng = np.random.default_rng(0)
for L in Ls:
    t0 = time.time()
    traces = [trace(qs[i], L) for i in range(NQ)]
    nprobe = np.mean([sum(len(h[0]) for h in t) for t in traces])
    nfresh = np.mean([sum(int(h[1].sum()) for h in t) for t in traces])
    nsurv = np.mean([sum(int(h[2].sum()) for h in t) for t in traces])
    print(f"L={L}: {NQ} queries, probes {nprobe:.0f} fresh {nfresh:.0f} ever in the pool {nsurv:.0f} hops {np.mean([len(t) for t in traces]):.0f} ({time.time() - t0:.0f}s)", flush=True)
    for sets in (1024, 2048, 3072):
        row = [f"  {sets * 4:6d} tags:"]
        for policy in ("random", "lru", "protect", "protect_lru"):
            r = np.mean([replay(t, sets, policy, rng) for t in traces])
            row.append(f"{policy} x{r:.3f}")
        print("  ".join(row), flush=True)
