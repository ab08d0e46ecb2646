"""Where does the 8-bit screen pay?  Pass rate (rows it lets through / rows it scores, from the kernel's own counters) and kernel
time with the screen off / on / AUTO (calibrated) on five distributions of the T2I shape (200-d, inner product) and the WebVid shape:
  t2i            the benchmark's set (32-d latent + noise): the friendly case
  t2i-hard       isotropic base (latent = D): scores tightly bunched
  narrow cone    isotropic + a common component 30 x its norm (centred codes, round 4)
  tight clusters 64 centres, rows within 0.2 % of their centre: a hop's neighbours all score alike
  webvid         2.5 M x 512-d unit-norm rows
n points each (default 2 M), graph by BuildRoarGraph on exact lists of n / 5 training queries, 10 000 queries, L_pq = 100."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mysteryann_amd import index as mi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
L, nq, k = 100, 10_000, 10


def clustered(n, nq, n_learn, d=200, seed=13):
    g = torch.Generator(device="cuda").manual_seed(seed)
    centres = torch.randn((64, d), generator=g, device="cuda") * 0.08

    def draw(m, spread):
        c = centres[torch.randint(0, 64, (m,), generator=g, device="cuda")]
        return (c + spread * 0.08 * torch.randn((m, d), generator=g, device="cuda")).contiguous()
    return draw(n, 0.002), draw(nq, 0.02), draw(n_learn, 0.02)


sets = [("t2i", lambda: synth.t2i_torch(n, nq, 200, n_learn=n // 5), 200),
        ("t2i-hard (isotropic)", lambda: synth.t2i_hard_torch(n, nq, 200, n_learn=n // 5), 200),
        ("narrow cone (offset 30)", lambda: synth.t2i_hard_torch(n, nq, 200, n_learn=n // 5, offset=30.0), 200),
        ("tight clusters", lambda: clustered(n, nq, n // 5), 200),
        ("webvid", lambda: synth.webvid_torch(n, nq, 512, n_learn=n // 5), 512)]
for name, gen, d in sets:
    base, q, learn = gen()
    idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None)
    idx.SetBaseData(base)
    lists, _ = idx.ExactKnnDevice(learn, 100, want_dists=False)
    p = mi.Parameters(); p.Set("M_sq", 100), p.Set("M_pjbp", 35), p.Set("L_pjpq", 500)
    idx.BuildRoarGraphDevice(lists, p)
    del lists, learn

    def best(reps=4):
        b = 1e9
        for _ in range(reps):
            idx.SearchRoarGraphBatchDevice(q, k, L); torch.cuda.synchronize()
            b = min(b, idx.info("last_kernel_ns") / 1e6)
        return b
    blk = torch.zeros((nq, 4), dtype=torch.int32, device="cuda")
    idx.set_option("screen", 2); idx.SetTrafficCounters(blk)
    o = idx.SearchRoarGraphBatchDevice(q, k, L); torch.cuda.synchronize(); idx.SetTrafficCounters(None)
    t = blk.to(torch.float64).sum(0); cm = o["cmps"].double().sum().item()
    rate = (t[1].item() - nq - (cm - t[0].item())) / max(t[0].item(), 1.0)
    res = {}
    for label, scr in (("off", 1), ("on", 2), ("auto", 0)):
        idx.set_option("screen", scr); idx.set_option("calibrate", 2)
        res[label] = best()
        if label == "auto":
            picked = "on" if idx.info("last_screen") else "off"
    print(f"{name:26s} avg degree {idx.info('nedges') / n:5.1f}  cmps/query {cm / nq:7.0f}  screen pass rate {rate:.3f}  kernel ms: screen off {res['off']:7.3f} | on {res['on']:7.3f} | "
          f"auto {res['auto']:7.3f} (took: {picked}; {100 * (res['auto'] / min(res['off'], res['on']) - 1):+.1f} % against the better forced setting)", flush=True)
    idx.close(); del base, q
