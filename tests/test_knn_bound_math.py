"""The bound behind the bf16-screened exact kNN generator (csrc/roar_knn_bf16.hip), restated in numpy and checked against the
reference's own distance code (oracle dist = the restatement pinned to include/efanna2e/distance.h): for every (query, row)

    | d^ - d | <= eps(q, r) = C |q| |x_r|                       inner product   (d^ = -(q^ . x^), both sides rounded to bf16)
    | d^ + |q|^2 - d | <= 2 C |q| |x_r| + kappa (|x_r|^2 + |q|^2)     L2        (d^ = fl(|x|^2) - 2 q^ . x^)

with the constants the kernels use, on ordinary rows and on rows chosen to strain it (one huge component, tiny rows, rows that are
multiples of the query, cancelling sums).  The GPU tests check the generator's RESULTS against float64 brute force; this pins the
derivation where there is no GPU."""
import numpy as np
import pytest

import oracle


def bf16_rne(x):
    """round-to-nearest-even to bfloat16, returned as float32 (the integer form of v_cvt_pk_bf16_f32 for finite values)"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def constants(dim):
    ksteps = (dim + 15) // 16
    K = 16.0 * ksteps
    C = np.float32(2.02 / 256.0 + (K + dim / 16.0 + 8.0) * 1.25 / 16777216.0)
    kappa = np.float32((dim + 2.0 * (dim / 16.0 + 8.0)) * 1.5 / 16777216.0)
    return C, kappa


def rows_and_queries(rng, d):
    n = 400
    x = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((40, d)).astype(np.float32)
    x[1] *= 1e-3
    x[2] *= 1e4
    x[3, 0] = 5e3                       # one component dwarfs the others
    x[4] = 0.0
    x[5] = q[0] * np.float32(3.0)       # a multiple of a query
    x[6] = q[1] + rng.standard_normal(d).astype(np.float32) * np.float32(1e-3)  # the cancelling case for L2
    x[7] = np.float32(1.0) + rng.standard_normal(d).astype(np.float32) * np.float32(1e-2)  # large common offset
    x[8] = np.float32(0.00390625) * (1 + 2.0 ** -9)  # every component a rounding tie
    q[2] = 0.0
    q[3] *= 1e3
    q[4, :] = np.float32(1.0 + 2.0 ** -9)
    return x, q


@pytest.mark.parametrize("d", [8, 72, 128, 200, 512])
def test_bf16_interval_contains_the_reference_inner_product(d):
    rng = np.random.default_rng(d)
    x, q = rows_and_queries(rng, d)
    C, _ = constants(d)
    xh, qh = bf16_rne(x), bf16_rne(q)
    assert np.all(np.abs(xh - x) <= np.abs(x) * 2.0 ** -8)  # the rounding the derivation assumes
    approx = -(qh.astype(np.float32) @ xh.astype(np.float32).T)  # (f32 products of bf16 values are exact; f32 sums)
    bn_all = (x.astype(np.float32) ** 2).sum(1, dtype=np.float32)
    xn = np.sqrt(bn_all) * np.float32(1.0002)
    qn = np.sqrt((q.astype(np.float32) ** 2).sum(1, dtype=np.float32)) * np.float32(1.0002)
    worst = 0.0
    for i in range(q.shape[0]):
        exact = np.array([oracle.dist(1, x[j], q[i]) for j in range(x.shape[0])], np.float32)
        eps = C * qn[i] * xn
        err = np.abs(approx[i].astype(np.float64) - exact.astype(np.float64))
        assert np.all(err <= eps.astype(np.float64) + 1e-30), (d, i, float((err - eps).max()))
        # the pass kernel's admission test takes the hardware's square root (v_sqrt_f32: 1 ulp) of max(|x|^2, 1e-30): still a bound
        xn_hw = np.sqrt(np.maximum(bn_all, np.float32(1e-30))) * np.float32(1.0 - 2.0 ** -22) * np.float32(1.0002)
        assert np.all(err <= (C * qn[i] * xn_hw).astype(np.float64) + 1e-30), (d, i)
        with np.errstate(invalid="ignore", divide="ignore"):
            worst = max(worst, float(np.nanmax(np.where(eps > 0, err / eps, 0))))
    assert 0.0 < worst <= 1.0  # the bound holds, and the data comes within sight of it (not vacuous by orders of magnitude)


@pytest.mark.parametrize("d", [8, 128, 200, 512])
def test_bf16_interval_contains_the_reference_l2_distance(d):
    rng = np.random.default_rng(100 + d)
    x, q = rows_and_queries(rng, d)
    C, kappa = constants(d)
    xh, qh = bf16_rne(x), bf16_rne(q)
    bn = (x.astype(np.float32) ** 2).sum(1, dtype=np.float32)  # the kernel's f32 row norms (any f32 summation order: kappa pays for it)
    dot = qh.astype(np.float32) @ xh.astype(np.float32).T
    xn = np.sqrt(bn) * np.float32(1.0002)
    for i in range(q.shape[0]):
        qq64 = float((q[i].astype(np.float64) ** 2).sum())
        qq = np.float32(qq64)
        approx = (bn - np.float32(2.0) * dot[i]).astype(np.float64) + qq64
        exact = np.array([oracle.dist(0, x[j], q[i]) for j in range(x.shape[0])], np.float64)
        eps = (np.float32(2.0) * C * np.sqrt(qq) * np.float32(1.0002) * xn + kappa * bn + kappa * qq * np.float32(1.0002)).astype(np.float64)
        err = np.abs(approx - exact)
        assert np.all(err <= eps + 1e-30), (d, i, float((err - eps).max()))


def test_interval_selection_keeps_the_exact_top_k():
    """Step 3 of the derivation on numbers: with intervals [lo, hi] around the exact distances, the rows with lo <= U (U = the k-th
    smallest hi) contain the exact top k -- for any errors inside the intervals, however adversarial."""
    rng = np.random.default_rng(5)
    for trial in range(200):
        n, k = 300, int(rng.integers(1, 40))
        exact = rng.standard_normal(n) * (10.0 ** rng.integers(-3, 3))
        eps = np.abs(rng.standard_normal(n)) * 0.05 * np.abs(exact).mean()
        approx = exact + eps * rng.uniform(-1, 1, n) * (1.0 if trial % 2 else np.sign(rng.standard_normal(n)))  # anywhere inside, or at the edges
        lo, hi = approx - eps, approx + eps
        U = np.sort(hi)[k - 1]
        kept = set(np.nonzero(lo <= U)[0].tolist())
        top = set(np.argsort(exact, kind="stable")[:k].tolist())
        assert top <= kept
