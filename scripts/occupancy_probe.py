"""Resident one-wave workgroups per CU vs kernel time of the 10 000-query launch (option waves_per_cu), default graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from _graph import t2i_index
idx, base, queries = t2i_index(10_000_000, 10000, "reference", 0, "/tmp/r3_graph2.index")
for L in (50, 100, 200):
    for nq in (10000, 2500):
        q = queries[:nq].contiguous()
        for w in (4, 5, 6, 7, 8):
            idx.set_option("waves_per_cu", w)
            best = 1e9
            for _ in range(5):
                idx.SearchRoarGraphBatchDevice(q, 10, L); torch.cuda.synchronize()
                best = min(best, idx.info("last_kernel_ns") / 1e6)
            print(f"L={L} nq={nq} waves_per_cu={w} grid {idx.info('last_grid')} {best:.3f} ms", flush=True)
