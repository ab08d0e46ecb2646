"""One launch at a time: does the number of workgroups matter beyond "as many as fit"?  10 000 queries over G workgroups are
ceil(10 000 / G) rounds, the last one ragged: kernel ms (best of 4) per option max_grid, on the SIFT-1M shape (graph file written by
bench.py --workload sift --graph-cache FILE) and / or the T2I-10M benchmark graph.
   python scripts/grid_probe.py --sift /tmp/sift.index [--t2i]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from mysteryann_amd import index as mi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--sift", default="")
ap.add_argument("--t2i", action="store_true")
ap.add_argument("--grids", type=int, nargs="+", default=[0, 2000, 2048, 2500, 2560, 2816, 3072])
args = ap.parse_args()


def run(idx, q, L, tag):
    for g in args.grids:
        idx.set_option("max_grid", g)
        best = 1e9
        for _ in range(4):
            idx.SearchRoarGraphBatchDevice(q, 10, L)
            torch.cuda.synchronize()
            best = min(best, idx.info("last_kernel_ns") / 1e6)
        print(f"{tag} L_pq {L} max_grid {g:5d}: grid {idx.info('last_grid'):5d} tags {idx.info('last_vis_entries')} kernel {best:7.3f} ms", flush=True)
    idx.set_option("max_grid", 0)


if args.sift:
    n, d = 1_000_000, 128
    base, q, _ = synth.sift_torch(n, 10_000, d, seed=1234, n_learn=0, device=torch.device("cuda", 0), query_seed=1236)
    idx = mi.IndexBipartite(d, n, mi.L2, None)
    idx.SetBaseData(base)
    idx.LoadProjectionGraph(args.sift)
    run(idx, q.contiguous(), 170, "sift")
    idx.close()
if args.t2i:
    from _graph import t2i_index
    idx, base, queries = t2i_index(10_000_000, 10000, "reference", 0, "/tmp/roar_graph.index")
    run(idx, queries[:10000].contiguous(), 50, "t2i")
