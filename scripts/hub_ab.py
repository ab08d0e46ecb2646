"""Hub-bitmap form vs the plain forms at long pools (T2I shape, bench.py's bootstrap graph by default): kernel time per
(L_pq, batch, visited mode, table size, hub on/off), results checked against the EXACT plain run."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mysteryann_amd import index as mi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--Ls", type=int, nargs="+", default=[100, 300, 500, 1000, 2000])
ap.add_argument("--nqs", type=int, nargs="+", default=[1250, 10000])
ap.add_argument("--out", default="gpurun_out/hub_ab.txt")
ap.add_argument("--builder", default="reference")
ap.add_argument("--graph-cache", default="/tmp/r2_graph.index")
args = ap.parse_args()
n, d, k = args.n, 200, 10
lines = []
def say(s):
    print(s, flush=True); lines.append(s)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _graph import t2i_index
idx, base, queries = t2i_index(n, 10000, args.builder, 0, args.graph_cache, say)
idx.set_option("search_kernel", 1)
def run(nq, L, vis, ve, hub, reps=3, tp=0):
    idx.set_option("visited_mode", vis); idx.set_option("vis_entries", ve); idx.set_option("hub_mode", 2 if hub else 1)
    q = queries[:nq].contiguous()
    best = 1e9
    for _ in range(reps):
        out = idx.SearchRoarGraphBatchDevice(q, k, L); torch.cuda.synchronize()
        best = min(best, idx.info("last_kernel_ns") / 1e6)
    return out, best
t0 = time.time(); run(64, 100, 1, 0, True, reps=1); say(f"# hub rows built in {time.time() - t0:.1f}s")
for L in args.Ls:
    for nq in args.nqs:
        ref, _ = run(nq, L, 0, 0, False, reps=1)
        bytes_q = (800.0 * ref["cmps"].double().sum() + 4 * ref["scanned"].double().sum() + 8 * ref["hops"].double().sum()).item() + 880.0 * nq
        cfgs = [(0, 0, False, 0), (0, 0, True, 0), (1, 8192, False, 0), (1, 16384, False, 0), (1, 4096, True, 0), (1, 8192, True, 0)]
        for vis, ve, hub, tp in cfgs:
            out, ms = run(nq, L, vis, ve, hub, tp=tp)
            ok = bool((out["ids"] == ref["ids"]).all() and (out["dists"] == ref["dists"]).all() and (out["hops"] == ref["hops"]).all())
            if vis == 0: ok = ok and bool((out["cmps"] == ref["cmps"]).all() and (out["scanned"] == ref["scanned"]).all())
            say(f"L={L:5d} nq={nq:6d} vis={'lds' if vis else 'exact'} ve={ve:5d} hub={int(hub)} tp={tp} (used {idx.info('last_hub_bitmap')}) grid {idx.info('last_grid'):5d} "
                f"lds {idx.info('last_lds_bytes'):6d} {ms:8.3f} ms {bytes_q / ms / 1e9:6.2f} TB/s(ref bytes) frac {bytes_q / ms / 8e9:.3f} "
                f"cmps {float(out['cmps'].float().mean()):.0f} (ref {float(ref['cmps'].float().mean()):.0f}) same={ok}")
os.makedirs(os.path.dirname(args.out), exist_ok=True)
open(args.out, "w").write("\n".join(lines) + "\n")
