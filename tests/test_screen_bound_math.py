"""The 8-bit screen's bound, restated in numpy f32 and checked on the CPU against the reference's own distance code (the oracle's
`dist_many` = distance.h, summation order and all): for every (row, query) pair, the lower bound the kernels compute
(csrc/roar_search_v2.h screen8i_pass, csrc/roar_kernels.hip roar_make_screen8_kernel) must not exceed the distance the reference
computes.  This pins the derivation where no GPU is present; the kernels themselves are tested pair by pair on the
GPU (tests/test_gpu_parity.py::test_screen_never_drops_a_row_at_or_under_the_threshold)."""
import numpy as np
import pytest

f32 = np.float32


def bf16_up(v):
    b = np.asarray(v, f32).view(np.uint32)
    hi = (b >> 16) + ((b & 0xFFFF) != 0)
    return (hi.astype(np.uint32) << 16).view(f32)


def code_rows(x, mu=None):
    """roar_make_screen8_kernel: c, s, B, N per row (f32 arithmetic as the kernel's); mu: the centre of the codes (they code x - mu)."""
    x = np.asarray(x, f32)
    old = np.seterr(all="ignore")
    if mu is not None:
        x = (x - np.asarray(mu, f32)[None, :]).astype(f32)
    mx = np.abs(x).max(axis=1)
    bad = ~np.isfinite(x).all(axis=1) | (mx > f32(1e30)) | ((mx > 0) & (mx < f32(1e-30)))
    s = (mx / f32(127)).astype(f32)
    inv = np.where(mx > 0, f32(127) / np.where(mx > 0, mx, f32(1)), f32(0)).astype(f32)
    c = np.clip(np.rint(x * inv[:, None]), -127, 127).astype(f32)
    d = ((x - s[:, None] * c) * inv[:, None]).astype(f32)
    e2 = (d * d).sum(axis=1, dtype=f32)
    c2 = (c * c).sum(axis=1, dtype=f32)
    nrm = (s * np.sqrt(c2) * f32(1.0001)).astype(f32)
    bnd = (f32(1.001) * s * np.sqrt(e2) * f32(1.0001) + f32(2.4e-7) * nrm).astype(f32)
    B, N = bf16_up(bnd), bf16_up(nrm)
    B[bad], N[bad], s[bad], c[bad] = np.inf, 0, 0, 0
    np.seterr(**old)
    return c.astype(np.int32), s, B, N


def code_query(q, D, mu=None):
    """screen8_query: k, |q| rounded up, t, E, and qd = the centre's part of the bound (0 without a centre)."""
    q = np.asarray(q, f32)
    qd = f32(0)
    if mu is not None:
        mu = np.asarray(mu, f32)
        mq = (mu * q).sum(dtype=f32)
        ma = (np.abs(mu) * np.abs(q)).sum(dtype=f32)
        m2 = (mu * mu).sum(dtype=f32)
    mx, s2 = np.abs(q).max(), (q * q).sum(dtype=f32)
    t = f32(mx / f32(127))
    inv = f32(127) / mx if mx > 0 else f32(0)
    k = np.clip(np.rint(q * inv), -127, 127).astype(f32)
    d = ((q - t * k) * inv).astype(f32)
    e2 = (d * d).sum(dtype=f32)
    qa = f32(np.sqrt(s2) * f32(1.0001))
    qc = f32(f32(1.001) * t * np.sqrt(e2) + (f32(4.0e-7) + f32(1.25) * f32(D // 16 + 6) * f32(5.9604645e-8)) * qa)
    if mu is not None:
        with np.errstate(over="ignore", invalid="ignore"):
            qd = f32(mq + f32(1.25) * f32(D // 64 + 8) * f32(5.9604645e-8) * f32(1.001) * ma
                     + f32(1.25) * f32(D // 16 + 6) * f32(5.9604645e-8) * f32(1.001) * qa * f32(np.sqrt(m2) * f32(1.0001)))
        if not np.isfinite(qd):
            qa = f32(np.inf)
    if not (mx <= 1e18) or not (s2 <= 3e38) or (0 < mx < 1e-30):
        qa = f32(np.inf)
    return k.astype(np.int32), qa, t, qc, qd


@pytest.mark.parametrize("centre", [None, "mean", "far"])
@pytest.mark.parametrize("d", [128, 200, 512])
def test_lower_bound_never_exceeds_the_reference_distance(d, centre):
    """centre: None = codes of x itself (rounds 1-3); "mean" = codes centred on the mean of the rows, as the library makes them;
    "far" = rows that share a large common component (a narrow cone, |mu| = 40 x the spread) with their mean as the centre -- where
    the uncentred bound was worthless."""
    import oracle

    rng = np.random.default_rng(5)
    n, nq = 3000, 48
    x = rng.standard_normal((n, d)).astype(f32)
    x[100:200, 3] *= 1e4
    x[200:220] = 0
    x[300:400] *= f32(1e-20)
    x[400:500] *= f32(1e15)
    x[500:600] = np.minimum(np.abs(np.rint(x[500:600] * 40)), 255)
    x[600:620] = f32(3.25)
    x[620:700] += f32(100)
    x[700:800] = f32(1e-42)
    q = rng.standard_normal((nq, d)).astype(f32)
    q[0] = 0
    q[1] = 0
    q[1, 3] = 3
    q[2] *= f32(1e-10)
    q[3] *= f32(1e10)
    q[4:8] = x[1000:1004]
    q[8:12] = -x[1000:1004]
    mu = None
    if centre == "far":
        x[1100:] += (rng.standard_normal(d) * 40).astype(f32)[None, :]
    if centre is not None:
        with np.errstate(over="ignore", invalid="ignore"):
            mu = (x[1100:].astype(np.float64).sum(axis=0) / (n - 1100)).astype(f32)  # (any finite vector will do as a centre)
    c, s, B, N = code_rows(x, mu)
    worst = -np.inf
    dropped_at_5pc = []
    for i in range(nq):
        k, qa, t, qc, qd = code_query(q[i], d, mu)
        I = (c.astype(np.int64) @ k.astype(np.int64)).astype(f32)  # exact: |I| < 2^24
        with np.errstate(invalid="ignore", over="ignore"):
            lower = (-((s * t) * I) - (qc * N + qa * B) - qd - f32(1e-30)).astype(f32)
        exact = oracle.dist_many(1, x, np.repeat(q[i][None], n, axis=0))  # -(x . q) as distance.h computes it
        viol = lower > exact  # (NaN compares false: such a row is kept)
        assert not viol.any(), (d, i, np.argwhere(viol)[:5].ravel(), lower[viol][:5], exact[viol][:5])
        if i >= 12:
            ordinary = np.arange(1100, n)
            spread = x[ordinary].astype(np.float64) - (0 if mu is None else mu.astype(np.float64)[None, :])
            margin = 0.05 * np.linalg.norm(q[i].astype(np.float64)) * np.linalg.norm(spread, axis=1)
            dropped_at_5pc.append(np.mean(lower[ordinary] > exact[ordinary] - margin))
    # and it is tight: 5 % of |q||x - mu| under the exact distance, ordinary rows are dropped -- also in the narrow cone, where that is
    # 0.1 % of |q||x|
    assert np.mean(dropped_at_5pc) > 0.98


def code_rows_l2(x):
    """roar_make_screen8_l2_kernel: u (0..255), s, lo, |x~|^2, B per row."""
    x = np.asarray(x, f32)
    old = np.seterr(all="ignore")
    n, d = x.shape
    mx, mn = x.max(axis=1), x.min(axis=1)
    bad = ~np.isfinite(x).all(axis=1) | (np.abs(x).max(axis=1) > f32(1e17))
    rng_ = (mx - mn).astype(f32)
    bad |= (rng_ > 0) & (rng_ < f32(1e-30))
    rng_[bad] = 0
    lo = np.where(bad, f32(0), mn).astype(f32)
    tight = (rng_ / f32(255)).astype(f32)
    mant, ex = np.frexp(tight)
    p2 = np.where(tight > 0, np.ldexp(f32(1), np.where(mant == 0.5, ex - 1, ex)), 0).astype(f32)
    out = []
    for sc in (p2, tight):
        inv = np.where(sc > 0, f32(1) / np.where(sc > 0, sc, f32(1)), f32(0)).astype(f32)
        u = np.clip(np.rint((x - lo[:, None]) * inv[:, None]), 0, 255).astype(f32)
        xh = np.float32(np.float64(sc[:, None]) * np.float64(u) + np.float64(lo[:, None]))  # (the kernel's fma: one rounding)
        dd = ((x - xh) * inv[:, None]).astype(f32)
        e2 = (dd * dd).sum(axis=1, dtype=f32)
        n2 = (xh * xh).sum(axis=1, dtype=f32)
        out.append((u, sc, e2, n2))
    exact = out[0][2] == 0
    u = np.where(exact[:, None], out[0][0], out[1][0])
    sc = np.where(exact, out[0][1], out[1][1]).astype(f32)
    e2 = np.where(exact, out[0][2], out[1][2]).astype(f32)
    n2 = np.where(exact, out[0][3], out[1][3]).astype(f32)
    B = (f32(1.001) * sc * np.sqrt(e2) * f32(1.0001) + f32(2.5e-7) * np.sqrt(n2) * f32(1.0001)).astype(f32)
    B[bad], n2[bad], sc[bad], u[bad] = np.inf, 0, 0, 0
    np.seterr(**old)
    return u.astype(f32), sc, lo, n2, B


@pytest.mark.parametrize("d", [128, 512])
def test_l2_lower_bound_never_exceeds_the_reference_distance(d):
    import oracle

    rng = np.random.default_rng(6)
    n, nq = 2500, 40
    x = rng.standard_normal((n, d)).astype(f32)
    x[100:200, 3] *= 1e4
    x[200:220] = 0
    x[300:400] *= f32(1e-20)
    x[400:500] *= f32(1e15)
    x[500:900] = np.minimum(np.abs(np.rint(x[500:900] * 40)), 255)  # SIFT-like: coded without error
    x[900:920] = f32(3.25)
    x[920:1000] += f32(100)
    q = rng.standard_normal((nq, d)).astype(f32)
    q[0] = 0
    q[2] *= f32(1e-10)
    q[3] *= f32(1e10)
    q[4:8] = x[500:504] + rng.standard_normal((4, d)).astype(f32) * f32(0.5)
    q[8:12] = x[920:924] + rng.standard_normal((4, d)).astype(f32) * f32(1e-3)  # the cancelling case
    q[12:16] = x[1500:1504]
    u, sc, lo, n2, B = code_rows_l2(x)
    assert (B[500:900] < 1e-3).all()  # integer rows: the power-of-two scale codes them exactly
    y = f32(d) ** f32(0.5)
    kappa = f32(1.25 * (2.0 * d * (1.0 + float(y)) + d + 8.0) / 16777216.0 * 1.0001)
    tight = []
    for i in range(nq):
        qa, qb = (q[i] * q[i]).sum(dtype=f32), q[i].sum(dtype=f32)
        acc = (u @ q[i]).astype(f32)  # (any summation order: the allowance kappa is for the worst one)
        with np.errstate(all="ignore"):
            xq = (lo * qb + sc * acc).astype(f32)
            dlo = ((n2 - f32(2) * xq + qa) - kappa * (n2 + qa) - f32(1e-30)).astype(f32)
            tt = (np.sqrt(np.maximum(dlo, 0)) * f32(0.999999) - B).astype(f32)
            lower = np.where(tt > 0, tt * tt * f32(0.99999), 0).astype(f32)
        exact = oracle.dist_many(0, x, np.repeat(q[i][None], n, axis=0))  # sum (x - q)^2 as distance.h computes it
        viol = lower > exact
        assert not viol.any(), (d, i, np.argwhere(viol)[:5].ravel(), lower[viol][:5], exact[viol][:5])
        if i >= 16:
            o = np.arange(1100, n)
            tight.append(np.mean(lower[o] > 0.9 * exact[o]))
    assert np.mean(tight) > 0.98  # tight: within 10 % of the exact distance for ordinary rows
