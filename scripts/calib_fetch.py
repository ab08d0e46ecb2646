"""FETCH_SIZE calibration for the row-gather access pattern of the search kernel
(MI355X_MICROARCH.md, HBM section: 'calibrate on a known byte count in your own access pattern').
Runs the distance-only kernel (same score_candidates routine: 4 lanes x 16 B = 64-B pieces of
800-B rows) over uniformly random rows of an 8 GB table: known bytes = nq * m * 800."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mysteryann_amd import index as mi
n, d, nq, m = 10_000_000, 200, 20000, 256
base = torch.randn((n, d), device="cuda")
idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None)
idx.SetBaseData(base)
rng = np.random.default_rng(0)
q = rng.standard_normal((nq, d)).astype(np.float32)
for rep in range(3):
    ids = rng.integers(0, n, (nq, m)).astype(np.uint32)
    t0 = time.time(); out = idx.Distances(q, ids); dt = time.time() - t0
    print(f"rep {rep}: rows {nq*m} bytes {nq*m*800/1e9:.3f} GB wall {dt*1e3:.1f} ms (incl. H2D/D2H)", flush=True)
chk = (base[torch.from_numpy(ids[0, :4].astype(np.int64)).cuda()].cpu().numpy() @ q[0]) * -1
print("check", out[0, :4], chk)
