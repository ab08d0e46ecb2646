"""Kernel forms of the SearchRoarGraph kernel on the T2I shape (bench.py's graph): kernel time per (waves per query,
pool form, batch size, L_pq, visited mode), ids / distances / hops checked against the EXACT one-wave run (whose cmps /
scanned are the reference algorithm's counters), TB/s in reference-algorithm bytes, and the phase timers of the
instrumented kernel.  Output: gpurun_out/kernel_ab.txt"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from mysteryann_amd import index as mi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--out", default="gpurun_out/kernel_ab.txt")
ap.add_argument("--Ls", type=int, nargs="+", default=[55, 100, 500, 2000])
ap.add_argument("--nqs", type=int, nargs="+", default=[256, 1250, 2500, 10000])
ap.add_argument("--builder", default="reference", choices=["bootstrap", "reference"])
ap.add_argument("--ref-learn", type=int, default=0)
ap.add_argument("--graph-cache", default="/tmp/r2_graph.index")
ap.add_argument("--prof-Ls", type=int, nargs="*", default=None)
args = ap.parse_args()
n, d, k = args.n, 200, 10
lines = []
def say(s):
    print(s, flush=True); lines.append(s)
from _graph import t2i_index
idx, base, queries = t2i_index(n, 10000, args.builder, args.ref_learn, args.graph_cache, say)

def run(nq, L, kern, vis, reps=3, ve=0, pool=0):
    idx.set_option("search_kernel", kern); idx.set_option("visited_mode", vis); idx.set_option("vis_entries", ve)
    idx.set_option("pool_form", pool)
    q = queries[:nq].contiguous()
    best = 1e9
    for _ in range(reps):
        out = idx.SearchRoarGraphBatchDevice(q, k, L)
        torch.cuda.synchronize()
        best = min(best, idx.info("last_kernel_ns") / 1e6)
    return out, best

for L in args.Ls:
    for nq in args.nqs:
        ref, t_ref = run(nq, L, 1, 0, reps=2)
        bytes_q = (800.0 * ref["cmps"].double().sum() + 4 * ref["scanned"].double().sum() + 8 * ref["hops"].double().sum()).item() + 880.0 * nq
        cfgs = [(1, 0, 0, 0), (0, 0, 0, 0)]                       # EXACT: one wave, auto
        if L < 1000:
            cfgs += [(1, 1, 0, 0), (0, 1, 0, 0)]                  # LDS table: one wave, auto
            if nq <= 2500: cfgs += [(3, 1, 0, 0), (2, 1, 0, 0)]   # two / four waves
            if L <= 64: cfgs += [(1, 1, 0, 1)]                    # LDS pool instead of the register pool

        for kern, vis, ve, pool in cfgs:
            out, ms = run(nq, L, kern, vis, ve=ve, pool=pool)
            ok = bool((out["ids"] == ref["ids"]).all() and (out["dists"] == ref["dists"]).all() and (out["hops"] == ref["hops"]).all())
            if vis == 0: ok = ok and bool((out["cmps"] == ref["cmps"]).all() and (out["scanned"] == ref["scanned"]).all())
            say(f"L={L:5d} nq={nq:6d} vis={'lds' if vis else 'exact'} kern={kern} ve={ve} pool={pool} -> W {idx.info('last_waves_per_query')} "
                f"grid {idx.info('last_grid'):5d} lds {idx.info('last_lds_bytes'):6d} {ms:8.3f} ms {nq / ms * 1e3:10.0f} qps "
                f"{bytes_q / ms / 1e9:6.2f} TB/s(ref bytes) cmps {float(out['cmps'].float().mean()):.0f} same={ok}")

names = ["select", "adj", "filter", "score", "merge"]
idx.set_option("profile_phases", 1)
prof_cfg = ((55, 1), (500, 1), (2000, 0)) if args.prof_Ls is None else tuple((L, 1 if L < 512 else 0) for L in args.prof_Ls)
for L, vis in prof_cfg:
    for nq in (1, 1250, 10000):
        idx.set_option("hub_mode", 1 if L > 64 else 0)  # (the instrumented hub form exists for one wave; keep the plain form here)
        out, ms = run(nq, L, 1, vis, reps=2)
        pr = idx.LastProfile(nq).astype(np.float64)
        H = out["hops"].double().sum().item()
        ph = " ".join(f"{nm} {pr[:, 1 + i].sum() / H:7.0f}" for i, nm in enumerate(names))
        say(f"PROF L={L} nq={nq} vis={'lds' if vis else 'exact'} W=1: {ms:.3f} ms; cycles/hop {pr[:, 0].sum() / H:7.0f}: {ph} | per pass: "
            f"issue {pr[:, 8].sum() / max(pr[:, 11].sum(), 1):6.0f} wait+fma {pr[:, 9].sum() / max(pr[:, 11].sum(), 1):6.0f} "
            f"tail {pr[:, 10].sum() / max(pr[:, 11].sum(), 1):6.0f}, passes/hop {pr[:, 11].sum() / H:.2f} rows/hop "
            f"{out['cmps'].double().sum().item() / H:.1f} survivors/hop {pr[:, 12].sum() / H:.1f} spec hit {pr[:, 6].sum() / H:.2f} "
            f"query cycles mean {pr[:, 0].mean():.0f} max {pr[:, 0].max():.0f}")
idx.set_option("profile_phases", 0)
os.makedirs(os.path.dirname(args.out), exist_ok=True)
open(args.out, "w").write("\n".join(lines) + "\n")
