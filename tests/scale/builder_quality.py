"""Graph quality of the in-HBM bootstrap builder against the CPU restatement of the reference's
BuildRoarGraph (oracle/roar_builder_oracle.c) ON THE SAME DATA: degree statistics and the
recall-vs-cmps curve (SURVEY 8(f) #1's parity notion).  Test infrastructure (runs the oracle)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from mysteryann_amd import index as mi, synth
from tests.golden.small_builder import exact_knn

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=20000)
ap.add_argument("--nq", type=int, default=1000)
ap.add_argument("--gpu-args", type=str, nargs="*", default=["{}"])
ap.add_argument("--skip-ref", action="store_true")
args = ap.parse_args()
n = args.n
base, q, learn = synth.t2i_numpy(n, args.nq, n_learn=n)
gt, _ = exact_knn(q, base, 10, 1)
out = {"n": n}

def curve(idx):
    idx.set_option("visited_mode", 0)
    rows = []
    for L in (10, 20, 30, 50, 75, 100, 200):
        r = idx.SearchRoarGraphBatch(q, 10, L)
        rows.append((L, round(oracle.recall(r["ids"], gt, 10), 4), int(r["cmps"].mean()), round(float(r["hops"].mean()), 1)))
    return rows

# (a) restated reference builder on exact learn->base lists, canonical parameters (run_roargraph_test.sh:9)
t0 = time.time()
knn, _ = exact_knn(learn, base, 100, 1)
if not args.skip_ref:
    ep, off, nbrs = oracle.build_roargraph(base, knn, 1, 100, 35, 500)
    t_ref = time.time() - t0
    deg = np.diff(off.astype(np.int64))
    idx = mi.IndexBipartite(200, n, mi.INNER_PRODUCT, None); idx.SetBaseData(base); idx.SetProjectionGraph(ep, off, nbrs)
    out["reference_builder_restated"] = {"seconds_cpu_1thread": round(t_ref, 1), "avg_degree": round(float(deg.mean()), 1), "max_degree": int(deg.max()),
                                         "curve_L_recall_cmps_hops": curve(idx)}
    idx.close()
# (b) the in-HBM bootstrap builder (training queries = the same learn set); each --gpu-args entry is one variant
for gi, ga in enumerate(args.gpu_args):
    kw = json.loads(ga)
    exact = kw.pop("exact", 0)
    idx = mi.IndexBipartite(200, n, mi.INNER_PRODUCT, None); idx.SetBaseData(base)
    if exact:
        kw["learn_knn"] = torch.from_numpy(knn[:, :kw.get("M_sq", 48)].astype(np.int32)).contiguous().cuda()
    t0 = time.time()
    idx.BuildBootstrapGraph(learn=torch.from_numpy(learn).cuda(), **kw)
    torch.cuda.synchronize(); t_gpu = time.time() - t0
    out[f"gpu_{gi}"] = {"args": ga, "seconds": round(t_gpu, 2), "avg_degree": round(idx.info("nedges") / n, 1), "max_degree": idx.info("max_degree"),
                        "curve_L_recall_cmps_hops": curve(idx)}
    idx.close()
print(json.dumps(out))
for k in out:
    if k == "n": continue
    print(k, {kk: vv for kk, vv in out[k].items() if kk != "curve_L_recall_cmps_hops"})
    print("   " + "  ".join("L%d:%.4f@%d" % (r[0], r[1], r[2]) for r in out[k]["curve_L_recall_cmps_hops"]))
