"""Visited-set forms of the search kernel on the benchmark graph (T2I-10M shape, reference-shaped graph), 10 000 queries:
LDS_FORGETFUL (a full set overwrites a victim), LDS (exact: a full set spills to u16 epoch tags in HBM), EXACT (u8 epoch
tags in HBM only), per L_pq and table size: kernel ms, scored rows per query (cmps), roofline fraction with the
reference's byte count (4 D cmps + 4 scanned + 8 hops + 4 D + 8 k, EXACT-mode counters)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from _graph import t2i_index

idx, base, queries = t2i_index(10_000_000, 10_000, "reference", 0, "/tmp/r3_graph2.index")
q = queries.contiguous()
K = 10
Ls = [int(x) for x in os.environ.get("LS", "55,100,200,300,500,700,1000,1400,2000").split(",")]


def run(L, mode, ve=0, nq=10000, reps=3):
    idx.set_option("visited_mode", mode)
    idx.set_option("vis_entries", ve)
    best, out = 1e9, None
    for _ in range(reps):
        out = idx.SearchRoarGraphBatchDevice(q[:nq], K, L)
        torch.cuda.synchronize()
        best = min(best, idx.info("last_kernel_ns") / 1e6)
    return best, out


for L in Ls:
    t_ex, ex = run(L, 0)
    alg = (800 * ex["cmps"].double().sum() + 4 * ex["scanned"].double().sum() + 8 * ex["hops"].double().sum() + (800 + 80) * 10000).item()
    ref_cmps = ex["cmps"].double().mean().item()
    row = [f"L={L:5d} ref cmps {ref_cmps:8.0f} | EXACT {t_ex:8.3f} ms frac {alg / t_ex / 1e6 / 8000:.3f}"]
    for name, mode in (("FORGET", 3), ("SPILL", 1)):
        for ve in (0, 4096, 8192, 12288, 16384):
            if ve and L < 200:
                continue
            t, o = run(L, mode, ve)
            same = bool((o["ids"] == ex["ids"]).all().item())
            row.append(f"{name} tags {idx.info('last_vis_entries'):5d} grid {idx.info('last_grid')}: {t:8.3f} ms frac {alg / t / 1e6 / 8000:.3f} cmps x{o['cmps'].double().mean().item() / ref_cmps:.3f}{'' if same else ' IDS DIFFER'}"
                       f"{' (forgetful!)' if mode == 1 and idx.info('last_visited_forgetful') else ''}")
    print(" | ".join(row), flush=True)
for nq in (1250, 2500):
    for name, mode in (("FORGET", 3), ("SPILL", 1)):
        t, o = run(55, mode, 0, nq, reps=6)
        print(f"L=55 nq={nq} {name}: {t:.4f} ms", flush=True)
