"""Experiment: collect every proposed edge of the hub rows (row capacity 512), then cut rows longer than 200 back to 200
by (a) the neighbours nearest to the hub, (b) the occlusion rule, instead of first-come-first-kept."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mysteryann_amd import index as mi, synth
n, d, nq, k, CAP, KEEP = 10_000_000, 200, 10_000, 10, 512, 184
base, queries, learn = synth.t2i_torch(n, nq, d, n_learn=n)
idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None); idx.SetBaseData(base)
idx.BuildBootstrapGraph(learn=learn, max_degree=CAP); torch.cuda.synchronize(); del learn
gt = idx.ExactKnnDevice(queries, k, want_dists=False)[0].to(torch.int64)
def sweep(tag):
    print(f"== {tag}: avg degree {idx.info('nedges') / n:.1f} max {idx.info('max_degree')}", flush=True)
    for L in (10, 15, 20, 25, 30):
        for _ in range(2): out = idx.SearchRoarGraphBatchDevice(queries, k, L)
        torch.cuda.synchronize()
        print(f"   L={L:3d} recall {bench.recall_at_k(torch, out['ids'], gt, k):.4f} {idx.info('last_kernel_ns') / 1e6:7.3f} ms "
              f"cmps {float(out['cmps'].float().mean()):7.1f}", flush=True)
sweep("capacity 512, first come first kept")
ep, off, nbrs = idx.GetProjectionGraph()
off = off.astype(np.int64); deg = np.diff(off)
big = np.nonzero(deg > KEEP)[0]
print("rows longer than", KEEP, ":", len(big), "of", n, "; edges in them", int(deg[big].sum()), flush=True)
def cut(mode):
    new_rows = {}
    B = 256
    for a in range(0, len(big), B):
        rows = big[a:a + B]
        W = int(deg[rows].max())
        ids = np.zeros((len(rows), W), np.int64); valid = np.zeros((len(rows), W), bool)
        for i, r in enumerate(rows):
            ids[i, :deg[r]] = nbrs[off[r]:off[r + 1]]; valid[i, :deg[r]] = True
        ti = torch.from_numpy(ids).cuda(); tv = torch.from_numpy(valid).cuda()
        C = base[ti.reshape(-1)][:, :d].reshape(len(rows), W, d)
        h = base[torch.from_numpy(rows).cuda()][:, :d]
        dh = torch.einsum("rwd,rd->rw", C, h).masked_fill(~tv, -1e30)  # dot with the hub: larger = nearer
        order = torch.argsort(dh, dim=1, descending=True)
        if mode == "nearest":
            keep = order[:, :KEEP]
        else:  # occlusion: in order of nearness keep c unless a kept r is nearer to c than the hub is
            G = torch.einsum("rwd,rvd->rwv", C, C)
            kept = torch.zeros((len(rows), W), dtype=torch.bool, device="cuda")
            cnt = torch.zeros(len(rows), dtype=torch.int64, device="cuda")
            ar = torch.arange(len(rows), device="cuda")
            for j in range(W):
                c = order[:, j]
                occl = ((G[ar, c] > dh[ar, c][:, None]) & kept).any(dim=1)
                ok = tv[ar, c] & ~occl & (cnt < KEEP)
                kept[ar, c] |= ok; cnt += ok.long()
            # back-fill with the nearest leftovers up to KEEP
            for j in range(W):
                c = order[:, j]
                ok = tv[ar, c] & ~kept[ar, c] & (cnt < KEEP)
                kept[ar, c] |= ok; cnt += ok.long()
            rank = torch.where(kept, torch.arange(W, device="cuda")[None, :].expand(len(rows), W), torch.full_like(order, W + 1))
            # keep original positions' order by nearness
            pos_in_order = torch.empty_like(order); pos_in_order[ar[:, None], order] = torch.arange(W, device="cuda")[None, :].expand(len(rows), W)
            key = torch.where(kept, pos_in_order, torch.full_like(order, 10 * W))
            keep = torch.argsort(key, dim=1)[:, :KEEP]
        sel = torch.gather(ti, 1, keep).cpu().numpy(); selv = torch.gather(tv, 1, keep).cpu().numpy()
        for i, r in enumerate(rows):
            new_rows[int(r)] = sel[i][selv[i]].astype(np.uint32)
    ndeg = deg.copy()
    for r, v in new_rows.items(): ndeg[r] = len(v)
    noff = np.zeros(n + 1, np.uint64); noff[1:] = np.cumsum(ndeg)
    nn = np.empty(int(noff[-1]), np.uint32)
    small = deg <= KEEP
    # copy untouched rows in bulk
    src_idx = np.repeat(small, deg)
    dst_mask = np.repeat(small, ndeg)
    nn[dst_mask] = nbrs[src_idx]
    for r, v in new_rows.items(): nn[int(noff[r]):int(noff[r + 1])] = v
    idx.SetProjectionGraph(ep, noff, nn)
for mode in ("nearest", "occlusion"):
    t0 = time.time(); cut(mode); print(f"cut by {mode}: {time.time() - t0:.1f}s", flush=True)
    sweep(f"rows cut to {KEEP} by {mode}")
