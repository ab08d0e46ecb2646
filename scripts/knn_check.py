"""ExactKnnDevice at large query counts: time, and agreement of one 200k-query call with 10k-query calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mysteryann_amd import index as mi, synth
n, d = 10_000_000, 200
base, queries, learn = synth.t2i_torch(n, 1000, d, n_learn=200_000)
idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None); idx.SetBaseData(base)
for nq in (10_000, 50_000, 200_000):
    torch.cuda.synchronize(); t0 = time.time()
    ids, _ = idx.ExactKnnDevice(learn[:nq], 35, want_dists=False)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(nq, f"{dt:.3f}s", f"{2.0 * nq * n * d / dt / 1e12:.1f} TFLOP/s", flush=True)
big = ids
small = torch.cat([idx.ExactKnnDevice(learn[a:a + 10_000], 35, want_dists=False)[0] for a in range(0, 200_000, 10_000)])
print("identical rows:", float((big == small).all(dim=1).float().mean()))
ref = torch.topk(learn[:2000, :d] @ base[:, :d].T, 35, dim=1).indices
print("vs torch topk (sets):", float(sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(ref.cpu(), big[:2000].cpu())) / (2000 * 35)))
ref64 = torch.topk(learn[:64, :d].double() @ base[:, :d].double().T, 35, dim=1).indices
print("vs float64 topk (sets):", float(sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(ref64.cpu(), big[:64].cpu())) / (64 * 35)),
      "torch f32 vs float64:", float(sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(ref64.cpu(), ref[:64].cpu())) / (64 * 35)))
print("distinct top-1 targets among 200k:", int(torch.unique(big[:, 0]).numel()))
