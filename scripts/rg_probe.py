"""Reference-shaped device builder (roar_build_roargraph) against the CPU restatement of the reference builder on the same
data and the same exact lists: degree statistics and recall@10 / distance evaluations per L_pq.  Test infrastructure."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from mysteryann_amd import index as mi, synth
from tests.golden.small_builder import exact_knn

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=20000)
ap.add_argument("--nq", type=int, default=1000)
ap.add_argument("--skip-ref", action="store_true")
ap.add_argument("--seed", type=int, default=1234)
args = ap.parse_args()
n = args.n
base, q, learn = synth.t2i_numpy(n, args.nq, 200, seed=args.seed, n_learn=n)
gt, _ = exact_knn(q, base, 10, 1)
idx = mi.IndexBipartite(200, n, mi.INNER_PRODUCT, None); idx.SetBaseData(base)
t0 = time.time()
lists, _ = idx.ExactKnnDevice(torch.from_numpy(learn).cuda(), 100, want_dists=False)
torch.cuda.synchronize(); print(f"exact lists on the device: {time.time() - t0:.1f}s", flush=True)
p = mi.Parameters(); p.Set("M_sq", 100); p.Set("M_pjbp", 35); p.Set("L_pjpq", 500)
t0 = time.time(); idx.BuildRoarGraphDevice(lists, p); torch.cuda.synchronize(); t_gpu = time.time() - t0
ep, off, nbrs = idx.GetProjectionGraph(); deg = np.diff(off.astype(np.int64))
print(f"device builder: {t_gpu:.1f}s ep {ep} degree avg {deg.mean():.2f} max {deg.max()} min {deg.min()} hash {idx.info('graph_hash') & (2**64-1):016x}", flush=True)
def curve(ep, off, nbrs, tag):
    out = []
    for L in (20, 50, 100):
        r = oracle.search_batch(base, off, nbrs, ep, q, 10, L, 1, threads=8)
        out.append(f"L{L}: {oracle.recall(r['ids'], gt, 10):.4f}@{r['cmps'].mean():.0f}")
    print(tag, "  ".join(out), flush=True)
curve(ep, off, nbrs, "device   ")
y = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", f"builder_yardstick_{n}.json")
if os.path.exists(y) and args.seed == 1234:
    Y = json.load(open(y))
    print("yardstick", "  ".join(f"L{L}: {Y['curve'][L]['recall_at_10']:.4f}@{Y['curve'][L]['avg_cmps']:.0f}" for L in ("20", "50", "100")),
          f"| degree avg {Y['avg_degree']:.2f} max {Y['max_degree']} min {Y['min_degree']} ep {Y['ep']}")
elif not args.skip_ref:
    knn = lists.cpu().numpy().view(np.uint32)
    t0 = time.time(); epr, offr, nbr = oracle.build_roargraph(base, knn, 1, 100, 35, 500); dr = np.diff(offr.astype(np.int64))
    print(f"restated reference builder: {time.time() - t0:.0f}s ep {epr} degree avg {dr.mean():.2f} max {dr.max()} min {dr.min()}")
    curve(epr, offr, nbr, "reference")
