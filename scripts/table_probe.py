"""On-chip tag-table size vs kernel time at medium and long pools (default graph, 10 000 and 1 250 queries)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from _graph import t2i_index
idx, base, queries = t2i_index(10_000_000, 10000, "reference", 0, "/tmp/r2_graph.index")
idx.set_option("visited_mode", 1); idx.set_option("hub_mode", 1)
for L in (150, 200, 300, 400, 500, 700, 1000, 1400):
    for nq in (10000, 1250):
        q = queries[:nq].contiguous()
        ref = None
        for ve in (8192, 12288, 16384, 24576):
            idx.set_option("vis_entries", ve)
            try:
                best = 1e9
                for _ in range(3):
                    out = idx.SearchRoarGraphBatchDevice(q, 10, L); torch.cuda.synchronize()
                    best = min(best, idx.info("last_kernel_ns") / 1e6)
            except Exception as e:
                print(f"L={L} nq={nq} tags {ve}: {e}"); continue
            if ref is None: ref = out
            same = bool((out["ids"] == ref["ids"]).all() and (out["dists"] == ref["dists"]).all() and (out["hops"] == ref["hops"]).all())
            print(f"L={L} nq={nq} tags {ve} (got {idx.info('last_vis_entries')}) grid {idx.info('last_grid')} lds {idx.info('last_lds_bytes')} "
                  f"{best:.3f} ms cmps {float(out['cmps'].float().mean()):.0f} same={same}", flush=True)
