"""Launch shape with the half-precision screen on: resident workgroups per CU x tag-table size, T2I-10M default graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from _graph import t2i_index
idx, base, queries = t2i_index(10_000_000, 10000)
for L in (50, 100, 500, 1000):
    for nq in (10000, 1250):
        q = queries[:nq].contiguous()
        for ve in (8192, 4096, 2048):
            idx.set_option("vis_entries", ve)
            row = [f"L={L} nq={nq} tags={ve}:"]
            for w in (6, 8, 10, 12):
                idx.set_option("waves_per_cu", w)
                best = 1e9
                for _ in range(4):
                    o = idx.SearchRoarGraphBatchDevice(q, 10, L); torch.cuda.synchronize()
                    best = min(best, idx.info("last_kernel_ns") / 1e6)
                row.append(f"wpc {w} (grid {idx.info('last_grid')}) {best:.3f} ms")
            row.append(f"cmps {float(o['cmps'].float().mean()):.0f}")
            print("  ".join(row), flush=True)
idx.set_option("waves_per_cu", 0); idx.set_option("vis_entries", 0)
for s in (0, 1):
    idx.set_option("screen", s)
    for L in (50, 200, 1000, 2000):
        best = 1e9
        for _ in range(3):
            idx.SearchRoarGraphBatchDevice(queries, 10, L); torch.cuda.synchronize()
            best = min(best, idx.info("last_kernel_ns") / 1e6)
        print(f"auto plan, screen {'off' if s else 'on'}: L={L} {best:.3f} ms grid {idx.info('last_grid')} tags {idx.info('last_vis_entries')} mode {idx.info('last_visited_mode')}", flush=True)
