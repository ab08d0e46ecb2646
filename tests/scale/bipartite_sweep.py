"""The paper's ablation baseline at scale: the bipartite two-hop search (SearchBipartiteGraph,
src/index_bipartite.cpp:282-356) on the T2I shape, graph built by qbaseNNbipartite (:235-280) from exact
learn->base lists made on the device.  One CSV row per L_pq in the column order of
tests/test_search_bipartite.cpp:226-231 (rderr left out: no distances in the device ground truth here)
  L_pq,qps,avg_cmps,mean_latency_ms,recall,avg_hops
for the GPU (kernel time of one launch over all queries), the CPU reference-kind baseline on the first
--cpu-queries queries with the SAME entry points, and an exact comparison of both sides' ids / distances /
cmps / hops.  Test infrastructure (runs the oracle)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
import bench
from mysteryann_amd import index as mi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--learn", type=int, default=2_000_000)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--cpu-queries", type=int, default=1000)
ap.add_argument("--M", type=int, default=35)
ap.add_argument("--Ls", type=int, nargs="+", default=[10, 20, 50, 100, 200, 500, 1000])
ap.add_argument("--visited", default="auto", choices=["auto", "exact", "lds"])
ap.add_argument("--out", default="gpurun_out/bipartite_sweep.csv")
args = ap.parse_args()
n, d, nq, k = args.n, 200, args.nq, 10
base, queries, learn = synth.t2i_torch(n, nq, d, n_learn=args.learn)
idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None); idx.SetBaseData(base)
t0 = time.time()
knn = torch.empty((args.learn, args.M), dtype=torch.int32, device="cuda")
for a in range(0, args.learn, 200_000):
    knn[a:a + 200_000] = idx.ExactKnnDevice(learn[a:a + 200_000], args.M, want_dists=False)[0]
torch.cuda.synchronize(); t_knn = time.time() - t0
idx._learn_base_knn = knn.cpu().numpy().view(np.uint32); del learn, knn
par = mi.Parameters(); par.Set("M_pjbp", args.M)
t0 = time.time(); idx.BuildBipartite(args.learn, None, n, None, par); t_build = time.time() - t0
print(f"# exact learn->base lists {t_knn:.1f}s, qbaseNNbipartite + upload {t_build:.1f}s, "
      f"{idx.info('bipartite_nodes')} nodes {idx.info('bipartite_edges')} edges", flush=True)
idx.set_option("visited_mode", {"exact": 0, "lds": 1, "auto": 2}[args.visited])
gt = idx.ExactKnnDevice(queries, k, want_dists=False)[0].to(torch.int64)
impl = "ref" if oracle.have_ref() else "port"
threads = min(oracle.max_threads(impl), bench.effective_cpus())
hb = base.cpu().numpy(); off, nbrs = idx.GetBipartiteGraph(); hq = queries.cpu().numpy()
init = mi.IndexBipartite.BipartiteEntryIds(2024, nq, 10, n)
init_d = torch.from_numpy(init.astype(np.int32)).cuda()
rows = ["L_pq,qps,avg_cmps,mean_latency_ms,recall,avg_hops,visited_mode,cpu_qps,cpu_threads,gpu_cpu_identical"]
for L in args.Ls:
    for _ in range(2): out = idx.SearchBipartiteGraphBatchDevice(queries, k, L, init_d)
    torch.cuda.synchronize()
    ms = idx.info("last_kernel_ns") / 1e6
    rec = bench.recall_at_k(torch, out["ids"], gt, k)
    mode = idx.info("last_visited_mode")
    m = args.cpu_queries if L <= 200 else max(100, args.cpu_queries // 5)
    t0 = time.perf_counter()
    o = oracle.search_bipartite_batch(hb, off, nbrs, hq[:m], k, L, 1, init[:m], threads=threads, impl=impl)
    dt = time.perf_counter() - t0
    gi = out["ids"][:m].cpu().numpy().view(np.uint32); gd = out["dists"][:m].cpu().numpy()
    gh = out["hops"][:m].cpu().numpy().view(np.uint32); gc = out["cmps"][:m].cpu().numpy().view(np.uint32)
    same = ((gi == o["ids"]).all(axis=1) & (gd == o["dists"]).all(axis=1) & (gh == o["hops"])
            & ((gc == o["cmps"]) if mode == 0 else (gc >= o["cmps"]))).mean()
    rows.append(f"{L},{nq / ms * 1e3:.0f},{float(out['cmps'].float().mean()):.1f},{ms / nq:.6f},{rec:.4f},"
                f"{float(out['hops'].float().mean()):.2f},{'lds' if mode == 1 else 'exact'},{m / dt:.0f},{threads},{same:.4f}")
    print(rows[-1], flush=True)
os.makedirs(os.path.dirname(args.out), exist_ok=True)
open(args.out, "w").write("\n".join(rows) + "\n")
