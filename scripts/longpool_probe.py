"""Long pools on the benchmark graph: what forgetting costs in ROWS.  Per L_pq: rows scored on the 8-bit copy and on f32 (kernel
counters) and kernel ms, for the forgetful on-chip table at 2 048 / 4 096 / 8 192 tags, the exact on-chip table (spill lists) and
HBM tags.  The difference of the f32 rows to the exact forms is what re-scored revisits cost in f32 reads."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from _graph import t2i_index
ap = argparse.ArgumentParser()
ap.add_argument("--graph-cache", default="/tmp/roar_graph.index")
ap.add_argument("--L", type=int, nargs="+", default=[200, 500, 1000, 2000])
ap.add_argument("--nq", type=int, default=10000)
ap.add_argument("--tables", type=int, nargs="+", default=[2048, 4096, 8192], help="forgetful on-chip table sizes (tags; 3 * 2^k allowed)")
ap.add_argument("--no-exact", action="store_true")
ap.add_argument("--kernel", type=int, default=0, help="option search_kernel (0 auto, 1 one wave per query, 3 two, 2 four)")
ap.add_argument("--protect", type=int, nargs="+", default=[1], help="option tag_protect values to run the forgetful forms with")
args = ap.parse_args()
idx, base, queries = t2i_index(10_000_000, 10000, "reference", 0, args.graph_cache)
q = queries[:args.nq].contiguous()
blk = torch.zeros((args.nq, 4), dtype=torch.int32, device="cuda")
idx.set_option("calibrate", 0); idx.set_option("screen", 2); idx.set_option("search_kernel", args.kernel)
for L in args.L:
    forms = [(f"forgetful {t}" + ("" if pr else " unprotected"), 3, t, pr) for t in args.tables for pr in args.protect] + \
            ([] if args.no_exact else [("exact table 4096", 1, 4096, 1), ("HBM tags", 0, 0, 1)])
    for label, vm, ve, pr in forms:
        idx.set_option("visited_mode", vm); idx.set_option("vis_entries", ve); idx.set_option("tag_protect", pr)
        if args.kernel in (0, 1):  # (the counting instantiation is a one-wave form)
            idx.SetTrafficCounters(blk)
        o = idx.SearchRoarGraphBatchDevice(q, 10, L); torch.cuda.synchronize()
        idx.SetTrafficCounters(None)
        t = blk.to(torch.float64).sum(0) / args.nq
        best = 1e9
        for _ in range(2):
            idx.SearchRoarGraphBatchDevice(q, 10, L); torch.cuda.synchronize()
            best = min(best, idx.info("last_kernel_ns") / 1e6)
        print(f"L_pq {L:5d} {label:28s}: {best:8.2f} ms  cmps {o['cmps'].double().mean().item():9.1f}  rows 8-bit {t[0].item():9.1f}  rows f32 {t[1].item():8.1f}  "
              f"adjacency words {t[2].item():9.0f}  hops {o['hops'].double().mean().item():7.1f}  LDS {idx.info('last_lds_bytes')} B, grid {idx.info('last_grid')} x {idx.info('last_waves_per_query')} waves", flush=True)
