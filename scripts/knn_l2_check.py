"""The bf16-screened exact kNN generator against the f32 MFMA form on the SIFT-1M shape (128-d, L2), k = 100: the lists must be the same
sets (near-ties may swap inside a list: the forms sum in different orders) and the distances equal to the last bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mysteryann_amd import index as mi, synth
n, d, nq, k = 1_000_000, 128, 20_000, 100
base, q, learn = synth.sift_torch(n, 100, d, seed=1234, n_learn=nq, device=torch.device("cuda", 0), query_seed=1236)
idx = mi.IndexBipartite(d, n, mi.L2, None); idx.SetBaseData(base)
res = {}
for mode in (0, 1):
    idx.set_option("knn_mode", mode)
    ids, dists = idx.ExactKnnDevice(learn, k)
    torch.cuda.synchronize()
    res[mode] = (ids.clone(), dists.clone(), idx.info("last_knn_bf16"))
a, b = res[0], res[1]
assert a[2] == 1 and b[2] == 0, (a[2], b[2])
same_rows = (a[0] == b[0]).all(1).float().mean().item()
same_sets = (torch.sort(a[0].long(), 1).values == torch.sort(b[0].long(), 1).values).all(1).float().mean().item()
print(f"SIFT shape, {nq} queries, k = {k}: lists identical for {same_rows:.5f} of the queries, as sets for {same_sets:.5f}; "
      f"max |distance difference| {float((a[1] - b[1]).abs().max()):.3g} (distances up to {float(a[1].max()):.3g})")
assert same_sets > 0.999
