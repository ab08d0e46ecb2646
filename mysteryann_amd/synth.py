"""Synthetic data of the shapes BASELINE.json names (no dataset can be downloaded).

Recipes follow SURVEY.md section 8(d); they only fix shapes, scales and the base/query
distribution shift (out-of-distribution queries, as in cross-modal data):

  T2I shape   base  x = z A + 0.3 eps,   z ~ N(0, I_32), A in R^{32 x D} fixed, D = 200, IP,
              query q = (1.5 z' + 0.5) B + 0.1 eps,  B = A + 0.5 Delta, scaled so |ip| <~ 1
  SIFT shape  |N(0,1)| * 40 rounded and clipped to 0..255, D = 128, L2 (prepare_data.sh ranges)
  WebVid shape unit-norm rows (prepare_for_clip_webvid.py:92-98), D = 512, IP / cosine
  T2I-hard    the T2I shape WITHOUT the low-rank structure the 8-bit screen profits from: base x = mu + sigma eps with
              eps ~ N(0, I_D) (latent = D: isotropic, query-base scores tightly bunched), the same row norm as the T2I set
              on average; `offset` > 0 adds a common component mu of |mu| = offset x the isotropic part's norm (embedding
              sets sit in a narrow cone: most of a score is then the same for every row, and a bound in units of |q||x|
              says little); queries q = 0.5 u + 1.5 sigma_q eps' around another direction u (out of distribution)

`numpy` versions make the small fixtures; `torch` versions generate the full-size sets directly
in HBM on the GPU box (torch is plumbing here: RNG + device memory).
"""
import numpy as np

LATENT = 32


def _maps(d, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((LATENT, d)).astype(np.float32)
    Delta = rng.standard_normal((LATENT, d)).astype(np.float32)
    B = (A + 0.5 * Delta).astype(np.float32)
    # scale: E|x|^2 ~ LATENT*d*sA^2 ; choose so typical |<x,q>| is O(1)
    s = np.float32(0.3 / np.sqrt(LATENT * np.sqrt(d)))
    return A * s, B * s, s


def t2i_numpy(n, nq, d=200, seed=1234, n_learn=0):
    A, B, s = _maps(d, seed)
    rng = np.random.default_rng(seed + 1)
    base = (rng.standard_normal((n, LATENT)).astype(np.float32) @ A
            + np.float32(0.3) * s * rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)

    def queries(m, r):
        z = np.float32(1.5) * r.standard_normal((m, LATENT)).astype(np.float32) + np.float32(0.5)
        return (z @ B + np.float32(0.1) * s * r.standard_normal((m, d)).astype(np.float32)).astype(np.float32)

    q = queries(nq, np.random.default_rng(seed + 2))
    learn = queries(n_learn, np.random.default_rng(seed + 3)) if n_learn else None
    return base, q, learn


def sift_numpy(n, nq, d=128, seed=1234, n_learn=0):
    rng = np.random.default_rng(seed)
    centers = np.abs(rng.standard_normal((64, d))).astype(np.float32) * 40

    def draw(m, r):
        c = centers[r.integers(0, 64, m)]
        x = c + np.abs(r.standard_normal((m, d))).astype(np.float32) * 20
        return np.clip(np.rint(x), 0, 255).astype(np.float32)

    base = draw(n, np.random.default_rng(seed + 1))
    q = draw(nq, np.random.default_rng(seed + 2))
    learn = draw(n_learn, np.random.default_rng(seed + 3)) if n_learn else None
    return base, q, learn


def webvid_numpy(n, nq, d=512, seed=1234, n_learn=0):
    base, q, learn = t2i_numpy(n, nq, d, seed, n_learn)

    def unit(x):
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)

    return unit(base), unit(q), (unit(learn) if learn is not None else None)


def t2i_torch(n, nq, d=200, seed=1234, n_learn=0, device="cuda", chunk=1 << 20, row_stride=None, query_seed=None):
    """Full-size T2I-shape set generated in HBM.  Returns (base [n, row_stride], queries [nq, d], learn|None)."""
    import torch

    A, B, s = _maps(d, seed)
    A = torch.from_numpy(A).to(device)
    B = torch.from_numpy(B).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(seed + 1)
    ld = row_stride or d
    base = torch.zeros((n, ld), dtype=torch.float32, device=device)
    for i in range(0, n, chunk):
        m = min(chunk, n - i)
        z = torch.randn((m, LATENT), generator=g, device=device)
        e = torch.randn((m, d), generator=g, device=device)
        base[i:i + m, :d] = z @ A + (0.3 * float(s)) * e

    def queries(m, sd):
        g.manual_seed(sd)
        out = torch.empty((m, d), dtype=torch.float32, device=device)
        for i in range(0, m, chunk):
            mm = min(chunk, m - i)
            z = 1.5 * torch.randn((mm, LATENT), generator=g, device=device) + 0.5
            e = torch.randn((mm, d), generator=g, device=device)
            out[i:i + mm] = z @ B + (0.1 * float(s)) * e
        return out

    q = queries(nq, seed + 2 if query_seed is None else query_seed)
    learn = queries(n_learn, seed + 3) if n_learn else None
    return base, q, learn


def _hard_maps(d, seed, offset):
    rng = np.random.default_rng(seed + 77)
    _, _, s = _maps(d, seed)
    sigma = np.float32(np.sqrt(LATENT) * s)  # the T2I set's per-component scale: |x|^2 ~ LATENT d s^2
    u = rng.standard_normal(d).astype(np.float32)
    u *= np.float32(np.sqrt(d)) * sigma / np.linalg.norm(u)  # |u| = the isotropic part's expected norm
    v = rng.standard_normal(d).astype(np.float32)
    v *= np.float32(np.sqrt(d)) * sigma / np.linalg.norm(v)
    return sigma, (np.float32(offset) * u).astype(np.float32), v


def t2i_hard_numpy(n, nq, d=200, seed=1234, n_learn=0, offset=0.0):
    sigma, mu, v = _hard_maps(d, seed, offset)
    rng = np.random.default_rng(seed + 1)
    base = (mu + sigma * rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)

    def queries(m, r):
        return (np.float32(0.5) * v + np.float32(1.5) * sigma * r.standard_normal((m, d)).astype(np.float32)).astype(np.float32)

    q = queries(nq, np.random.default_rng(seed + 2))
    learn = queries(n_learn, np.random.default_rng(seed + 3)) if n_learn else None
    return base, q, learn


def t2i_hard_torch(n, nq, d=200, seed=1234, n_learn=0, device="cuda", chunk=1 << 20, query_seed=None, offset=0.0):
    """Full-size T2I-hard set generated in HBM (see the module docstring).  -> (base [n, d], queries [nq, d], learn|None)."""
    import torch

    sigma, mu, v = _hard_maps(d, seed, offset)
    mu = torch.from_numpy(mu).to(device)
    v = torch.from_numpy(v).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(seed + 1)
    base = torch.empty((n, d), dtype=torch.float32, device=device)
    for i in range(0, n, chunk):
        m = min(chunk, n - i)
        base[i:i + m] = mu + float(sigma) * torch.randn((m, d), generator=g, device=device)

    def queries(m, sd):
        g.manual_seed(sd)
        out = torch.empty((m, d), dtype=torch.float32, device=device)
        for i in range(0, m, chunk):
            mm = min(chunk, m - i)
            out[i:i + mm] = 0.5 * v + (1.5 * float(sigma)) * torch.randn((mm, d), generator=g, device=device)
        return out

    q = queries(nq, seed + 2 if query_seed is None else query_seed)
    learn = queries(n_learn, seed + 3) if n_learn else None
    return base, q, learn


def webvid_torch(n, nq, d=512, seed=1234, n_learn=0, device="cuda", query_seed=None):
    """WebVid shape: unit-norm rows (prepare_for_clip_webvid.py:92-98), unit-norm shifted queries."""
    import torch

    base, q, learn = t2i_torch(n, nq, d, seed, n_learn, device, query_seed=query_seed)
    base /= base.norm(dim=1, keepdim=True)
    q /= q.norm(dim=1, keepdim=True)
    if learn is not None:
        learn /= learn.norm(dim=1, keepdim=True)
    return base, q, learn


def sift_torch(n, nq, d=128, seed=1234, n_learn=0, device="cuda", query_seed=None):
    """SIFT shape: non-negative integer-valued floats 0..255 around 64 cluster centres, L2."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    centers = torch.randn((64, d), generator=g, device=device).abs() * 40

    def draw(m, sd):
        g.manual_seed(sd)
        c = centers[torch.randint(0, 64, (m,), generator=g, device=device)]
        return torch.clamp(torch.round(c + torch.randn((m, d), generator=g, device=device).abs() * 20), 0, 255)

    base = draw(n, seed + 1)
    q = draw(nq, seed + 2 if query_seed is None else query_seed)
    learn = draw(n_learn, seed + 3) if n_learn else None
    return base, q, learn
