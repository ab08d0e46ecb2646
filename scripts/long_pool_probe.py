import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from _graph import t2i_index
idx, base, queries = t2i_index(10_000_000, 10000, "reference", 0, "/tmp/r2_graph.index")
idx.set_option("hub_mode", 1)
for L in (1500, 1600, 1700, 1800, 2000):
    for nq in (10000, 1250):
        q = queries[:nq].contiguous()
        for vis, ve in ((0, 0), (1, 8192), (1, 4096), (1, 6144)):
            idx.set_option("visited_mode", vis); idx.set_option("vis_entries", ve)
            best = 1e9
            for _ in range(3):
                out = idx.SearchRoarGraphBatchDevice(q, 10, L); torch.cuda.synchronize()
                best = min(best, idx.info("last_kernel_ns") / 1e6)
            print(f"L={L} nq={nq} {'exact' if vis == 0 else 'lds %d' % ve} grid {idx.info('last_grid')} lds {idx.info('last_lds_bytes')} {best:.3f} ms cmps {float(out['cmps'].float().mean()):.0f}", flush=True)
