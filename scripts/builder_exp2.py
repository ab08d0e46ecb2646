"""Builder experiments on the T2I-10M shape with EXACT learn->base lists (MFMA generator): the reference-shaped
build (no skeleton) and the bootstrap build fed with exact lists, against the default bootstrap build.
Prints recall / kernel ms / cmps over a short L_pq sweep for each configuration."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mysteryann_amd import index as mi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--lists", type=int, default=2_000_000)
ap.add_argument("--K", type=int, default=63)
ap.add_argument("--configs", nargs="+", default=["ref35", "ref24", "boot"])
ap.add_argument("--Ls", type=int, nargs="+", default=[20, 30, 40, 50, 60, 70, 80, 100, 150])
args = ap.parse_args()
n, d, nq, k = args.n, 200, 10_000, 10
base, queries, learn = synth.t2i_torch(n, nq, d, n_learn=max(args.lists, n // 2))
idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None); idx.SetBaseData(base)
gt = idx.ExactKnnDevice(queries, k, want_dists=False)[0].to(torch.int64)
t0 = time.time()
lists = torch.empty((args.lists, args.K), dtype=torch.int32, device="cuda")
for a in range(0, args.lists, 1 << 16):
    b = min(a + (1 << 16), args.lists)
    lists[a:b] = idx.ExactKnnDevice(learn[a:b], args.K, want_dists=False)[0]
torch.cuda.synchronize()
print(f"exact lists {args.lists} x {args.K}: {time.time() - t0:.1f}s", flush=True)

def sweep(tag):
    print(f"== {tag}: avg degree {idx.info('nedges') / n:.1f} max {idx.info('max_degree')}", flush=True)
    for L in args.Ls:
        for _ in range(2): out = idx.SearchRoarGraphBatchDevice(queries, k, L)
        torch.cuda.synchronize()
        ms = idx.info("last_kernel_ns") / 1e6
        print(f"   L={L:4d} recall {bench.recall_at_k(torch, out['ids'], gt, k):.4f}  {ms:7.3f} ms  {nq / ms * 1e3:9.0f} qps  "
              f"cmps {float(out['cmps'].float().mean()):7.1f} hops {float(out['hops'].float().mean()):6.1f}", flush=True)

for cfg in args.configs:
    t0 = time.time()
    if cfg.startswith("ref"):      # reference shape: projection from the lists + supply pass, no skeleton
        M = int(cfg[3:5]); sl = int(cfg.split(":")[1]) if ":" in cfg else 500
        idx.BuildBootstrapGraph(M=1, base_cap=1, L_build=8, no_skeleton=True, learn=None, learn_knn=lists, M_sq=args.K, M_pj=M,
                                L_learn=args.K, rounds=0, supply_k=M, supply_cand=63, supply_L=sl, max_degree=3 * M, trim_min_proj=0)
    elif cfg == "boot":            # default bootstrap, projection rounds replaced by the exact lists
        idx.BuildBootstrapGraph(learn=None, learn_knn=lists[:, :48].contiguous(), rounds=0)
    elif cfg == "boot+":           # default bootstrap rounds on the approximate lists AND the exact lists
        idx.BuildBootstrapGraph(learn=learn[:n // 2], learn_knn=lists[:, :48].contiguous())
    elif cfg == "default":
        idx.BuildBootstrapGraph(learn=learn[:n // 2])
    torch.cuda.synchronize()
    print(f"build {cfg}: {time.time() - t0:.1f}s", flush=True)
    sweep(cfg)
