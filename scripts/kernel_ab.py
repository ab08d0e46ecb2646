"""A/B of the search kernels on the T2I shape (bench.py's graph): kernel time per (kernel generation, waves per query,
rows per pass, batch size, L_pq, visited mode), ids / distances / hops checked against the first-generation EXACT run,
and the phase timers of the instrumented kernel.  Output: gpurun_out/kernel_ab.txt"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mysteryann_amd import index as mi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--out", default="gpurun_out/kernel_ab.txt")
ap.add_argument("--Ls", type=int, nargs="+", default=[30, 100, 500, 2000])
ap.add_argument("--nqs", type=int, nargs="+", default=[256, 1250, 2500, 10000])
args = ap.parse_args()
n, d, k = args.n, 200, 10
lines = []
def say(s):
    print(s, flush=True); lines.append(s)
base, queries, learn = synth.t2i_torch(n, 10000, d, n_learn=n)
idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None); idx.SetBaseData(base)
t0 = time.time(); idx.BuildBootstrapGraph(learn=learn, max_degree=384, row_keep=128); torch.cuda.synchronize(); del learn
say(f"# n={n} index built in {time.time() - t0:.1f}s avg degree {idx.info('nedges') / n:.1f} max {idx.info('max_degree')}")

def run(nq, L, ver, kern, rows, vis, reps=4):
    idx.set_option("kernel_version", ver); idx.set_option("search_kernel", kern); idx.set_option("score_rows", rows)
    idx.set_option("visited_mode", vis)
    q = queries[:nq].contiguous()
    best = 1e9
    for _ in range(reps):
        out = idx.SearchRoarGraphBatchDevice(q, k, L)
        torch.cuda.synchronize()
        best = min(best, idx.info("last_kernel_ns") / 1e6)
    return out, best

for L in args.Ls:
    for nq in args.nqs:
        ref, t_ref = run(nq, L, 1, 1, 0, 0, reps=2)  # first generation, one wave, EXACT: the reference counters
        bytes_q = (800.0 * ref["cmps"].double().sum() + 4 * ref["scanned"].double().sum() + 8 * ref["hops"].double().sum()).item() + 880.0 * nq
        for vis in ((1, 0) if L < 1000 else (0,)):
            for ver, kern, rows in ((1, 1, 0), (1, 0, 0), (2, 1, 2), (2, 1, 4), (2, 3, 2), (2, 3, 4), (2, 2, 2), (2, 0, 0)):
                if kern in (2, 3) and nq > 2500: continue
                out, ms = run(nq, L, ver, kern, rows, vis)
                ok = bool((out["ids"] == ref["ids"]).all() and (out["dists"] == ref["dists"]).all() and (out["hops"] == ref["hops"]).all())
                if vis == 0: ok = ok and bool((out["cmps"] == ref["cmps"]).all() and (out["scanned"] == ref["scanned"]).all())
                say(f"L={L:5d} nq={nq:6d} vis={'lds' if vis else 'exact'} v{ver} kern={kern} rows={rows} -> waves/q {idx.info('last_waves_per_query')} "
                    f"grid {idx.info('last_grid'):5d} {ms:8.3f} ms {nq / ms * 1e3:10.0f} qps {bytes_q / ms / 1e9:6.2f} TB/s(ref bytes) "
                    f"cmps {float(out['cmps'].float().mean()):.0f} same={ok}")

# phase timers (instrumented v2)
idx.set_option("profile_phases", 1)
for L, vis in ((30, 1), (100, 1), (500, 1), (500, 0), (2000, 0)):
    for nq in (1, 1250, 10000):
        for kern in (1, 3):
            if kern == 3 and nq > 2500: continue
            out, ms = run(nq, L, 2, kern, 2, vis, reps=2)
            pr = idx.LastProfile(nq).astype(np.float64)
            tot = pr[:, 0].sum()
            say(f"PROF L={L} nq={nq} vis={'lds' if vis else 'exact'} kern={kern}: {ms:.3f} ms; cycles/query {pr[:, 0].mean():.0f}; "
                f"select {pr[:, 1].sum() / tot:.3f} adjacency {pr[:, 2].sum() / tot:.3f} filter {pr[:, 3].sum() / tot:.3f} "
                f"score {pr[:, 4].sum() / tot:.3f} merge {pr[:, 5].sum() / tot:.3f}; spec hits {pr[:, 6].sum() / max(out['hops'].sum().item(), 1):.3f} "
                f"adj round trips/hop {pr[:, 7].sum() / max(out['hops'].sum().item(), 1):.3f} hops {out['hops'].float().mean():.1f}")
idx.set_option("profile_phases", 0)
os.makedirs(os.path.dirname(args.out), exist_ok=True)
open(args.out, "w").write("\n".join(lines) + "\n")
