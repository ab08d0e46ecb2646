"""BASELINE configs[4] data path at full size on ONE MI355X: Text-to-Image-100M shape, base + graph
sharded 8-way (12.5 M points per shard, own graph and entry point per shard, local ids), every shard
searched with all 10 000 queries, ids rebased to global, per-shard top-k merged by (distance, id)
with the HIP merge kernel.  100 M x 800 B = 80 GB of vectors + 8 x 8 GB of graphs fit one GPU's
288 GB, so the eight "ranks" run back to back on device 0; on an 8-GPU node each rank holds one
shard and the only change is torch.distributed.all_gather (RCCL over xGMI) in front of the merge
(mysteryann_amd/parallel.py:sharded_search; world-size-2 gloo test in tests/test_parallel_gloo.py).

Checks: recall@10 of the merged result against exact global ground truth (merge of exact per-shard
kNN, MFMA generator), and -- the oracle of SURVEY 8(e) -- the merge of per-shard CPU searches on the
first --check queries must equal the GPU's merged ids and distances exactly.
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--shards", type=int, default=8)
ap.add_argument("--per-shard", type=int, default=12_500_000)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--L", type=int, nargs="+", default=[15, 20, 30, 40, 55], help="L_pq values, all on one build per shard")
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--check", type=int, default=200)
ap.add_argument("--builder", default="reference", choices=["reference", "bootstrap"])
ap.add_argument("--learn", type=int, default=0, help="training queries per shard (0 = per-shard / 5 for the reference builder, / 10 for bootstrap)")
ap.add_argument("--maxdeg", type=int, default=384)
ap.add_argument("--keep", type=int, default=128)
args = ap.parse_args()

import oracle
from mysteryann_amd import index as mi, parallel, synth

d, k, S, m = 200, args.k, args.shards, args.per_shard
_, queries, _ = synth.t2i_torch(1024, args.nq, d, seed=1234)
hq = queries.cpu().numpy()
Ls = args.L
res_i, res_d = {L: [] for L in Ls}, {L: [] for L in Ls}
cpu_i, cpu_d, kern_ms = {L: [] for L in Ls}, {L: [] for L in Ls}, {L: [] for L in Ls}
gt_i, gt_d, build_s = [], [], []
impl = "ref" if oracle.have_ref() else "port"
for s in range(S):
    t0 = time.time()
    # same maps (seed) for every shard, different latent draws per shard: one 100 M-point distribution
    A, B, sc = synth._maps(d, 1234)
    g = torch.Generator(device="cuda"); g.manual_seed(5000 + s)
    base = torch.empty((m, d), device="cuda")
    At = torch.from_numpy(A).cuda()
    for i in range(0, m, 1 << 20):
        mm = min(1 << 20, m - i)
        base[i:i + mm] = torch.randn((mm, synth.LATENT), generator=g, device="cuda") @ At \
            + (0.3 * float(sc)) * torch.randn((mm, d), generator=g, device="cuda")
    n_learn = args.learn or (m // 5 if args.builder == "reference" else m // 10)
    _, _, learn = synth.t2i_torch(1024, 1, d, seed=1234, n_learn=n_learn)
    idx = mi.IndexBipartite(d, m, mi.INNER_PRODUCT, None)
    idx.SetBaseData(base)
    if args.builder == "reference":  # exact lists of the training queries on this shard -> the reference's LinkProjection, canonical parameters
        lists, _ = idx.ExactKnnDevice(learn, 100, want_dists=False)
        prm = mi.Parameters(); prm.Set("M_sq", 100); prm.Set("M_pjbp", 35); prm.Set("L_pjpq", 500)
        idx.BuildRoarGraphDevice(lists, prm)
        del lists
    else:
        idx.BuildBootstrapGraph(learn=learn, max_degree=args.maxdeg, row_keep=args.keep)
    torch.cuda.synchronize()
    build_s.append(time.time() - t0)
    del learn
    gi, gd = idx.ExactKnnDevice(queries, k)
    gt_i.append(parallel.rebase_ids(gi, s * m)); gt_d.append(gd)
    if args.check:
        ep, off, nbrs = idx.GetProjectionGraph()
        hb = base.cpu().numpy()
    for L in Ls:
        for _ in range(2):
            out = idx.SearchRoarGraphBatchDevice(queries, k, L)
        torch.cuda.synchronize()
        kern_ms[L].append(idx.info("last_kernel_ns") / 1e6)
        res_i[L].append(parallel.rebase_ids(out["ids"], s * m)); res_d[L].append(out["dists"])
        if args.check:  # per-shard CPU search of the first queries on this shard's own graph (the oracle of 8(e))
            o = oracle.search_batch(hb, off, nbrs, ep, hq[:args.check], k, L, 1, threads=16, impl=impl)
            cpu_i[L].append(o["ids"].astype(np.int64) + s * m); cpu_d[L].append(o["dists"])
    if args.check:
        del hb, off, nbrs
    print(f"shard {s}: gen+build {build_s[-1]:.1f}s, avg degree {idx.info('nedges') / m:.1f}, search kernel ms "
          + ", ".join(f"L={L}: {kern_ms[L][-1]:.2f}" for L in Ls), flush=True)
    idx.close(); del idx, base

gti, gtd = parallel.merge_topk_device(torch.stack(gt_i), torch.stack(gt_d))
for L in Ls:
    ids_all, d_all = torch.stack(res_i[L]), torch.stack(res_d[L])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mi_, md_ = parallel.merge_topk_device(ids_all, d_all)
    torch.cuda.synchronize(); merge_ms = (time.perf_counter() - t0) * 1e3
    hit = (gti[:, :, None] == mi_[:, None, :]).any(dim=2).sum().item()
    recall = hit / (k * args.nq)
    parity = None
    if args.check:
        ci, cd = np.stack(cpu_i[L]), np.stack(cpu_d[L])  # [S, check, k]
        want_i = np.zeros((args.check, k), np.int64); want_d = np.zeros((args.check, k), np.float32)
        for q in range(args.check):
            a, b = ci[:, q].reshape(-1), cd[:, q].reshape(-1)
            o = np.lexsort((a, b))[:k]
            want_i[q], want_d[q] = a[o], b[o]
        got_i = mi_[:args.check].cpu().numpy().view(np.uint32).astype(np.int64)
        got_d = md_[:args.check].cpu().numpy()
        parity = {"queries_checked": args.check, "ids_identical_fraction": float((got_i == want_i).all(axis=1).mean()),
                  "max_abs_dist_diff": float(np.abs(got_d - want_d).max()), "cpu_impl": impl}
    total_ms = sum(kern_ms[L]) + merge_ms
    print(json.dumps({
        "config": f"t2i-{S * m // 1_000_000}M-ip index-sharded x{S} (one GPU, shards back to back)", "builder": args.builder, "n_base": S * m,
        "per_shard": m, "n_queries": args.nq, "k": k, "L_pq": L, "recall_at_10_global": recall,
        "search_kernel_ms_per_shard": kern_ms[L], "merge_kernel_ms": merge_ms,
        "single_gpu_serial_qps": args.nq / (total_ms * 1e-3),
        "eight_gpu_projection_qps": args.nq / ((max(kern_ms[L]) + merge_ms) * 1e-3),
        "allgather_bytes_per_rank": args.nq * k * 8, "gen_plus_build_seconds_per_shard": build_s,
        "parity_vs_merge_of_cpu_shard_searches": parity}), flush=True)
