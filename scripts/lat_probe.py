"""Latency regime of the search kernel: phase timers of the instrumented v2 kernel (cycles per hop of wave 0) for a lone
query, one query per CU and the strong-scaling batch (1250), per waves-per-query form.  T2I shape, default 2 M points
(the per-hop latencies are those of the 10 M set: both live in HBM far beyond the caches)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from mysteryann_amd import index as mi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--builder", default="reference")
ap.add_argument("--graph-cache", default="/tmp/roar_graph.index")
ap.add_argument("--L", type=int, nargs="+", default=[55])
ap.add_argument("--nqs", type=int, nargs="+", default=[1, 256, 1250, 2500, 10000])
ap.add_argument("--vis", type=int, default=1)
ap.add_argument("--tags", type=int, default=0, help="option vis_entries (0: the plan's own)")
ap.add_argument("--kernels", type=int, nargs="+", default=[1, 3, 2], help="option search_kernel values (1 one wave per query, 3 two, 2 four)")
args = ap.parse_args()
n, d, k = args.n, 200, 10
from _graph import t2i_index
idx, base, queries = t2i_index(n, 10000, args.builder, 0, args.graph_cache)
idx.set_option("visited_mode", args.vis)
if args.tags:
    idx.set_option("calibrate", 0), idx.set_option("screen", 2), idx.set_option("vis_entries", args.tags)
names = ["select", "adj", "filter", "score", "merge"]
for L in args.L:
    for nq in args.nqs:
        q = queries[:nq].contiguous()
        for kern in args.kernels:
            if kern != 1 and nq > 2500: continue
            idx.set_option("search_kernel", kern)
            idx.set_option("profile_phases", 0)
            best = 1e9
            for _ in range(4):
                out = idx.SearchRoarGraphBatchDevice(q, k, L); torch.cuda.synchronize()
                best = min(best, idx.info("last_kernel_ns") / 1e3)
            idx.set_option("profile_phases", 1)
            for _ in range(2):
                out = idx.SearchRoarGraphBatchDevice(q, k, L); torch.cuda.synchronize()
            tp = idx.info("last_kernel_ns") / 1e3
            pr = idx.LastProfile(nq).astype(np.float64)
            hops = out["hops"].double().cpu().numpy()
            H = hops.sum()
            ph = " ".join(f"{nm} {pr[:, 1 + i].sum() / H:7.0f}" for i, nm in enumerate(names))
            print(f"L={L} nq={nq:5d} W={idx.info('last_waves_per_query')} {best:8.1f} us (instrumented {tp:8.1f}) hops mean {hops.mean():.1f} max {hops.max():.0f} | "
                  f"cycles/hop {pr[:, 0].sum() / H:7.0f}: {ph} | per pass: issue {pr[:, 8].sum() / max(pr[:, 11].sum(), 1):6.0f} "
                  f"wait+fma {pr[:, 9].sum() / max(pr[:, 11].sum(), 1):6.0f} tail {pr[:, 10].sum() / max(pr[:, 11].sum(), 1):6.0f}, "
                  f"passes/hop {pr[:, 11].sum() / H:.2f} rows/hop {out['cmps'].double().sum().item() / H:.1f} survivors/hop {pr[:, 12].sum() / H:.1f} "
                  f"spec hit {pr[:, 6].sum() / H:.2f} | query cycles mean {pr[:, 0].mean():.0f} max {pr[:, 0].max():.0f} | "
                  f"inside the merge (LDS pool): lower bounds {pr[:, 13].sum() / H:.0f} ranks {pr[:, 14].sum() / H:.0f} shift {pr[:, 15].sum() / H:.0f}", flush=True)
