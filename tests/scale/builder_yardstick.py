"""Yardstick for the reference-shaped device builder at a size where graph quality shows: the CPU
restatement of the reference's BuildRoarGraph (oracle/roar_builder_oracle.c; checked against the real reference builder
by the round-1 judge) on a seeded T2I-shape set, canonical parameters M_sq = 100, M_pjbp = 35, L_pjpq = 500
(run_roargraph_test.sh:9), exact learn->base lists.  Writes tests/golden/builder_yardstick_<n>.json: degree statistics
and recall@10 / distance evaluations at L_pq = 20, 50, 100 (EXACT counters, the CPU search restatement).  Test
infrastructure; runs on the host only (single thread: several minutes at n = 200 000)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
from mysteryann_amd import synth
from tests.golden.small_builder import exact_knn

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=200_000)
ap.add_argument("--nq", type=int, default=1000)
ap.add_argument("--seed", type=int, default=1234)
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--Ls", type=int, nargs="+", default=[20, 50, 100])
ap.add_argument("--name", default="builder_yardstick", help="fixture written: tests/golden/<name>_<n>.json")
ap.add_argument("--survey-probe", action="store_true",
                help="add the figures SURVEY.md 3.2 / 6 / 8(d) report for the COMPILED reference builder on this recipe at "
                     "n = 50 000 (the only numbers the real builder ever produced here) beside the restated builder's")
args = ap.parse_args()
n = args.n
t0 = time.time()
base, q, learn = synth.t2i_numpy(n, args.nq, 200, seed=args.seed, n_learn=n)


def exact_knn_big(Q, X, k, chunk=1024):
    """Inner-product top-k by torch on the host cores (multi-threaded matmul + topk), ordered by (distance, id): the numpy
    argpartition of tests/golden/small_builder.py needs hours at a million points."""
    import torch

    Xt = torch.from_numpy(X)
    ids = np.zeros((Q.shape[0], k), np.uint32)
    for i in range(0, Q.shape[0], chunk):
        d = -(torch.from_numpy(Q[i:i + chunk]) @ Xt.T)
        v, j = torch.topk(d, k, dim=1, largest=False, sorted=True)
        v, j = v.numpy(), j.numpy()
        order = np.lexsort((j, v), axis=1)
        ids[i:i + chunk] = np.take_along_axis(j, order, 1)
    return ids


if n > 300_000:
    knn = exact_knn_big(learn, base, 100)
    gt = exact_knn_big(q, base, 10)
else:
    knn, _ = exact_knn(learn, base, 100, 1)
    gt, _ = exact_knn(q, base, 10, 1)
print(f"data + exact lists in {time.time() - t0:.0f}s", flush=True)
t0 = time.time()
ep, off, nbrs = oracle.build_roargraph(base, knn, 1, 100, 35, 500)
t_build = time.time() - t0
deg = np.diff(off.astype(np.int64))
print(f"restated reference builder: {t_build:.0f}s, ep {ep}, degree avg {deg.mean():.2f} max {deg.max()} min {deg.min()}", flush=True)
out = {"n": n, "nq": args.nq, "seed": args.seed, "n_learn": n, "M_sq": 100, "M_pjbp": 35, "L_pjpq": 500, "metric": "ip",
       "builder_seconds_cpu_1thread": round(t_build, 1), "ep": int(ep), "avg_degree": float(deg.mean()), "max_degree": int(deg.max()),
       "min_degree": int(deg.min()), "curve": {}}
if args.survey_probe:  # SURVEY.md:137-138 (hops, cmps), :195-199 (degree, recall at 10 / 100 / 500), :302 (recall at 20 / 100)
    out["survey_probe_of_the_compiled_reference"] = {
        "n": 50000, "avg_degree": 33.1, "max_degree": 70,
        "recall_at_10": {"10": 0.792, "20": 0.915, "100": 0.9974, "500": 1.0},
        "avg_hops": {"10": 14.5, "20": 23.7, "50": 52.8, "100": 102.5, "200": 202.2, "500": 502.0},
        "avg_cmps": {"10": 675, "20": 1066, "50": 2107, "100": 3519, "200": 5722, "500": 10295}}
for L in args.Ls:
    r = oracle.search_batch(base, off, nbrs, ep, q, 10, L, 1, threads=args.threads)
    out["curve"][str(L)] = {"recall_at_10": float(oracle.recall(r["ids"], gt, 10)), "avg_cmps": float(r["cmps"].mean()),
                            "avg_hops": float(r["hops"].mean())}
    print(L, out["curve"][str(L)], flush=True)
if args.survey_probe:
    # the `std::vector<Neighbor> prune_queue(pruned_list.size())` quirk of :1438 (the work list of the supply pass's reverse
    # prune starts with size() zero entries: id 0, distance 0): the same build without it, to see whether it moves the curve
    t0 = time.time()
    ep2, off2, nbrs2 = oracle.build_roargraph(base, knn, 1, 100, 35, 500, flags=16)
    deg2 = np.diff(off2.astype(np.int64))
    out["without_the_prune_queue_quirk"] = {"ep": int(ep2), "avg_degree": float(deg2.mean()), "max_degree": int(deg2.max()),
                                            "edges_to_node_0": int((nbrs2 == 0).sum()), "curve": {}}
    out["edges_to_node_0"] = int((nbrs == 0).sum())
    for L in args.Ls:
        r = oracle.search_batch(base, off2, nbrs2, ep2, q, 10, L, 1, threads=args.threads)
        out["without_the_prune_queue_quirk"]["curve"][str(L)] = {
            "recall_at_10": float(oracle.recall(r["ids"], gt, 10)), "avg_cmps": float(r["cmps"].mean()), "avg_hops": float(r["hops"].mean())}
    print(f"without the :1438 quirk ({time.time() - t0:.0f}s):", out["without_the_prune_queue_quirk"], flush=True)
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", f"{args.name}_{n}.json")
json.dump(out, open(path, "w"), indent=1)
print("written", path)
