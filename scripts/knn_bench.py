"""Exact-kNN generator timing, both forms (roar_knn.hip: every score by f32 MFMA; roar_knn_bf16.hip: bf16 MFMA screen + exact f32
re-rank): NQ training queries x N base rows x 200-d, k = 100 (the builder's learn->base lists), and the 10 000-query k = 10
ground-truth job of the bench; ids of the two forms compared.
  python scripts/knn_bench.py [N [NQ]]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mysteryann_amd import index as mi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 262_144
d = 200
base, q, learn = synth.t2i_torch(n, 10_000, d, n_learn=nq)
idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None); idx.SetBaseData(base)
res = {}
for name, queries, k in (("ground truth, 10 000 queries, k = 10", q, 10), (f"training lists, {nq} queries, k = 100", learn, 100)):
    for mode, label in ((0, "bf16 screen + f32 re-rank"), (1, "f32 MFMA")):
        if mode == 1 and queries.shape[0] > 131_072:
            queries_m = queries[:131_072]
        else:
            queries_m = queries
        idx.set_option("knn_mode", mode)
        best = 1e9
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ids, dists = idx.ExactKnnDevice(queries_m, k)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        m = queries_m.shape[0]
        extra = f", rows re-ranked per query {idx.info('last_knn_rerank_rows') / m:.0f}" if mode == 0 else ""
        print(f"{name}: {label}: {m} queries in {best * 1e3:.1f} ms -> {2 * m * n * d / best / 1e12:.1f} TFLOP/s of useful (D = 200) work "
              f"(peaks: bf16 2 500, f32 157){extra}; 10 M lists at this rate: {best / m * 1e7:.0f} s", flush=True)
        res[(name, mode)] = (ids[:131_072].clone(), dists[:131_072].clone())
    a, b = res[(name, 0)], res[(name, 1)]
    mm = min(a[0].shape[0], b[0].shape[0])
    same_rows = (a[0][:mm] == b[0][:mm]).all(1).float().mean().item()
    same_sets = (torch.sort(a[0][:mm].long(), 1).values == torch.sort(b[0][:mm].long(), 1).values).all(1).float().mean().item()
    print(f"{name}: lists identical for {same_rows:.5f} of the queries, as sets for {same_sets:.5f} (the f32 form sums in MFMA order, the "
          f"re-rank in the reference's: near-ties may swap); max |distance difference| {float((a[1][:mm] - b[1][:mm]).abs().max()):.3g}", flush=True)
