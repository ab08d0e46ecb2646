"""BASELINE configs[4] at full size THROUGH THE C ABI (roar_sharded_*, row E2b) on ONE MI355X: Text-to-Image-100M shape, eight
shards of 12.5 M points (own graph and entry point per shard, local ids) all resident on device 0 at once -- 80 GB of vectors +
8 x 6.4 GB of graphs fit one GPU's 288 GB.  One roar_search_batch_sharded call = every shard searched with all 10 000 queries
back to back on the rank's stream, ids rebased, ncclAllGather (one rank here: the collective runs, over a world of one) and the
merge.  On an 8-GPU node the same call with devices 0..7 runs the shards side by side.

Checks: recall@10 of the merged result against exact global ground truth (merge of exact per-shard kNN), and the merge of the
per-shard CPU searches of the first --check queries must equal the returned ids and distances exactly."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--shards", type=int, default=8)
ap.add_argument("--per-shard", type=int, default=12_500_000)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--L", type=int, nargs="+", default=[20, 30, 40])
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--check", type=int, default=100)
args = ap.parse_args()

import oracle
from mysteryann_amd import index as mi, parallel, synth

d, k, S, m = 200, args.k, args.shards, args.per_shard
_, queries, _ = synth.t2i_torch(1024, args.nq, d, seed=1234)
hq = queries.cpu().numpy()
sh = mi.ShardedIndex(d, mi.INNER_PRODUCT, [0] * S)
impl = "ref" if oracle.have_ref() else "port"
gt_i, gt_d, cpu = [], [], {L: [] for L in args.L}
A, B, sc = synth._maps(d, 1234)
At = torch.from_numpy(A).cuda()
for s in range(S):
    t0 = time.time()
    g = torch.Generator(device="cuda"); g.manual_seed(5000 + s)
    base = torch.empty((m, d), device="cuda")
    for i in range(0, m, 1 << 20):
        mm = min(1 << 20, m - i)
        base[i:i + mm] = torch.randn((mm, synth.LATENT), generator=g, device="cuda") @ At \
            + (0.3 * float(sc)) * torch.randn((mm, d), generator=g, device="cuda")
    _, _, learn = synth.t2i_torch(1024, 1, d, seed=1234, n_learn=m // 5)
    idx = sh.shard(s)
    idx.SetBaseData(base)  # adopted in place; the view keeps the tensor alive
    lists, _ = idx.ExactKnnDevice(learn, 100, want_dists=False)
    prm = mi.Parameters(); prm.Set("M_sq", 100); prm.Set("M_pjbp", 35); prm.Set("L_pjpq", 500)
    idx.BuildRoarGraphDevice(lists, prm)
    del lists, learn
    gi, gd = idx.ExactKnnDevice(queries, k)
    gt_i.append(parallel.rebase_ids(gi, s * m)); gt_d.append(gd)
    if args.check:
        ep, off, nbrs = idx.GetProjectionGraph()
        hb = base.cpu().numpy()
        for L in args.L:
            o = oracle.search_batch(hb, off, nbrs, ep, hq[:args.check], k, L, 1, threads=16, impl=impl)
            cpu[L].append((o["ids"].astype(np.int64) + s * m, o["dists"]))
        del hb, off, nbrs
    torch.cuda.synchronize()
    print(f"shard {s}: generated, indexed and checked on the CPU in {time.time() - t0:.1f}s, avg degree {idx.info('nedges') / m:.1f}, "
          f"free HBM {torch.cuda.mem_get_info()[0] / 2**30:.0f} GiB", flush=True)
gti, _ = parallel.merge_topk_device(torch.stack(gt_i), torch.stack(gt_d))
gti = gti.cpu().numpy().view(np.uint32)
for L in args.L:
    sh.SearchRoarGraphBatch(hq, k, L)  # warm (communicator, buffers, launch plans)
    t0 = time.perf_counter()
    r = sh.SearchRoarGraphBatch(hq, k, L)
    wall_ms = (time.perf_counter() - t0) * 1e3
    recall = float((gti[:, :, None] == r["ids"][:, None, :]).any(axis=2).sum()) / (k * args.nq)
    parity = None
    if args.check:
        want_i = np.zeros((args.check, k), np.int64); want_d = np.zeros((args.check, k), np.float32)
        for q in range(args.check):
            a = np.concatenate([c[0][q] for c in cpu[L]]); b = np.concatenate([c[1][q] for c in cpu[L]])
            o = np.lexsort((a, b))[:k]
            want_i[q], want_d[q] = a[o], b[o]
        parity = {"queries_checked": args.check,
                  "ids_identical_fraction": float((r["ids"][:args.check].astype(np.int64) == want_i).all(axis=1).mean()),
                  "max_abs_dist_diff": float(np.abs(r["dists"][:args.check] - want_d).max()), "cpu_impl": impl}
    print(json.dumps({
        "config": f"t2i-{S * m // 1_000_000}M-ip index-sharded x{S} through roar_search_batch_sharded (one GPU = one RCCL rank, shards back to back)",
        "n_base": S * m, "per_shard": m, "n_queries": args.nq, "k": k, "L_pq": L, "recall_at_10_global": recall,
        "ranks": sh.info("ranks"), "search_ms_all_shards": sh.info("last_search_ns") / 1e6, "allgather_plus_merge_ms": sh.info("last_exchange_ns") / 1e6,
        "host_call_wall_ms": wall_ms, "single_gpu_qps": args.nq / (wall_ms * 1e-3),
        "avg_cmps_summed_over_shards": float(r["cmps"].mean()), "avg_hops_summed_over_shards": float(r["hops"].mean()),
        "allgather_bytes_per_shard": args.nq * k * 8, "parity_vs_merge_of_cpu_shard_searches": parity}), flush=True)
sh.close()
