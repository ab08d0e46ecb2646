import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
import torch
from _graph import t2i_index
idx, base, queries = t2i_index(10_000_000, 10000)
for L in [int(x) for x in os.environ.get("LS", "50,100,200,500,1000,1400,2000").split(",")]:
    row=[f"L={L}"]
    for mode,name in ((2,"auto"),(3,"forget"),(0,"exact")):
        idx.set_option("visited_mode", mode)
        for ve in ((0,) if mode!=3 else (8192,4096,2048)):
            idx.set_option("vis_entries", ve)
            best=1e9
            for _ in range(3):
                o=idx.SearchRoarGraphBatchDevice(queries,10,L); torch.cuda.synchronize()
                best=min(best, idx.info("last_kernel_ns")/1e6)
            row.append(f"{name}{'/'+str(ve) if ve else ''}: {best:.3f} ms (grid {idx.info('last_grid')}, tags {idx.info('last_vis_entries')})")
    idx.set_option("vis_entries", 0)
    print("  ".join(row), flush=True)
