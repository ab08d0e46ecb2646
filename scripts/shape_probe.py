"""Kernel time of one 10 000-query launch on the WebVid / SIFT shape (bench.py --workload webvid|sift --graph-cache FILE wrote the
graph): screen on / off x workgroups per CU x tag-table size, ids compared between all forms.
   python scripts/shape_probe.py webvid /tmp/wv.index 95,200        (ROAR_AMD_LIB=... picks another build of the library)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mysteryann_amd import index as mi, synth

wl, cache = sys.argv[1], sys.argv[2]
Ls = [int(x) for x in sys.argv[3].split(",")]
n, d = {"webvid": (2_500_000, 512), "sift": (1_000_000, 128)}[wl]
gen = {"webvid": synth.webvid_torch, "sift": synth.sift_torch}[wl]
dev = torch.device("cuda", 0)
base, q, _ = gen(n, 10_000, d, seed=1234, n_learn=0, device=dev, query_seed=1236)
idx = mi.IndexBipartite(d, n, mi.L2 if wl == "sift" else mi.INNER_PRODUCT, None)
idx.SetBaseData(base)
idx.LoadProjectionGraph(cache)
print(f"# {wl}: n={n} d={d} avg degree {idx.info('nedges') / n:.1f}  lib {os.environ.get('ROAR_AMD_LIB', 'default')}", flush=True)
for L in Ls:
    ref = None
    for screen, hub in [(int(x), int(y)) for x in os.environ.get("SCREEN", "0,1").split(",") for y in os.environ.get("HUB", "0").split(",")]:
        # option values: screen 0 = auto (on where the shape has one), 1 = off; hub_mode 0 = auto, 1 = off, 2 = on
        idx.set_option("hub_mode", hub)
        idx.set_option("visited_mode", int(os.environ.get("VIS", "2")))
        for wpc in [int(x) for x in os.environ.get("WPC", "0,6,8,12").split(",")]:
            for ve in [int(x) for x in os.environ.get("VE", "0,4096").split(",")]:
                idx.set_option("screen", screen)
                idx.set_option("waves_per_cu", wpc)
                idx.set_option("vis_entries", ve)
                best = 1e9
                for _ in range(3):
                    o = idx.SearchRoarGraphBatchDevice(q, 10, L)
                    torch.cuda.synchronize()
                    best = min(best, idx.info("last_kernel_ns") / 1e6)
                if ref is None:
                    ref = o["ids"].clone()
                same = bool((o["ids"] == ref).all())
                print(f"L={L} screen={'off' if screen else 'auto'}({idx.info('last_screen')}) hub={hub}({idx.info('last_hub_bitmap')}) mode {idx.info('last_visited_mode')} wpc={wpc} tags={ve}: {best:.3f} ms  grid {idx.info('last_grid')} "
                      f"tags {idx.info('last_vis_entries')} kernel cmps {float(o['cmps'].float().mean()):.0f} ids_same={same}", flush=True)
