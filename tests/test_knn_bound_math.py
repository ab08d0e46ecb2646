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


# ---- round 5: the admission test as one more MFMA (knn_pass_bf16_kernel, "The verdict as one more MFMA") ---------------------------------
def bf16_up(v):
    """the bf16 value that is not below v (csrc/roar_knn_bf16.hip bf16_up), as float32"""
    v = np.ascontiguousarray(v, np.float32)
    u = v.view(np.uint32)
    t = u & np.uint32(0xFFFF0000)
    special = ((u & np.uint32(0x7F800000)) == np.uint32(0x7F800000)) | ((u & np.uint32(0xFFFF)) == 0)
    up = np.where((u >> 31) != 0, t, t + np.uint32(0x10000)).astype(np.uint32)
    return np.where(special, t, up).astype(np.uint32).view(np.float32)


def split3_up(v):
    """v as t1 + t2 + t3 >= v, three bf16 values (bf16_split3_up)"""
    v = np.ascontiguousarray(v, np.float32)
    t1 = bf16_rne(v)
    fin = np.isfinite(v)
    with np.errstate(invalid="ignore"):
        r1 = np.where(fin, v - t1, np.float32(0)).astype(np.float32)
        t2 = bf16_rne(r1)
        t3 = bf16_up((r1 - t2).astype(np.float32))
    return t1, np.where(fin, t2, np.float32(0)).astype(np.float32), np.where(fin, t3, np.float32(0)).astype(np.float32)


def f32_sum_worst(terms, sign):
    """sum of float32 terms in f32, every partial sum pushed one ulp the way `sign` says: whatever order and rounding the matrix core
    uses for its f32 accumulation, the result is not further from the exact sum than this"""
    acc = np.float32(0)
    for t in terms:
        acc = np.float32(acc + np.float32(t))
        if np.isfinite(acc):
            acc = np.nextafter(acc, np.float32(sign * np.inf), dtype=np.float32)
    return acc


MARGIN = np.float32(4.0e-5)  # KNN_VERDICT_MARGIN


def test_bf16_helpers_round_the_way_the_verdict_needs():
    rng = np.random.default_rng(11)
    v = np.concatenate([rng.standard_normal(2000) * 10.0 ** rng.integers(-20, 20, 2000), [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf,
                        3.0e38, -3.0e38, 1e-38, -1e-38]]).astype(np.float32)
    up = bf16_up(v)
    assert np.all(up >= v) and np.all((up.view(np.uint32) & 0xFFFF) == 0)
    fin = np.isfinite(up) & (np.abs(v) >= 2.0 ** -100)  # (not the subnormals: they have fewer bits to cut)
    assert np.all(np.abs(up[fin] - v[fin]) <= np.abs(v[fin]) * 2.0 ** -7)
    t1, t2, t3 = split3_up(v)
    for t in (t1, t2, t3):
        assert np.all((t.view(np.uint32) & 0xFFFF) == 0)
    s = t1.astype(np.float64) + t2.astype(np.float64) + t3.astype(np.float64)
    ok = np.isfinite(v)
    assert np.all(s[ok] >= v[ok].astype(np.float64))
    big = ok & (np.abs(v) >= 2.0 ** -100)
    assert np.all(s[big] - v[big] <= np.abs(v[big]).astype(np.float64) * 2.0 ** -22)
    assert np.all(t1[~ok] == v[~ok]) and np.all(t2[~ok] == 0) and np.all(t3[~ok] == 0)


@pytest.mark.parametrize("ip", [1, 0])
@pytest.mark.parametrize("d", [72, 128, 200])
def test_verdict_mfma_admits_everything_the_interval_test_admits(d, ip):
    """S = acc + e^ xn^ + U' (+ the row's -bl'/2 for L2), as the extra MFMA forms it from bf16 columns, is >= 0 wherever the f32 test
    lo = d^ - eps <= U holds -- with the f32 sum of the MFMA rounded against us at every step."""
    rng = np.random.default_rng(1000 * ip + d)
    x, q = rows_and_queries(rng, d)
    C, kappa = constants(d)
    xh, qh = bf16_rne(x), bf16_rne(q)
    bn = (x.astype(np.float32) ** 2).sum(1, dtype=np.float32)
    acc = (qh.astype(np.float32) @ xh.astype(np.float32).T).astype(np.float32)  # the clean accumulator (q^ . x^)
    xn_k = np.sqrt(np.maximum(bn, np.float32(1e-30))) * np.float32(1.0002)      # the kernel's own xn (eps of the lists)
    xn_col = bf16_up(np.sqrt(np.maximum(bn, np.float32(1e-30))) * np.float32(1.0003))  # the row's verdict column
    assert np.all(xn_col >= xn_k)
    qq = (q.astype(np.float32) ** 2).sum(1, dtype=np.float32)
    ea = (np.float32(1.0 if ip else 2.0) * C * np.sqrt(qq) * np.float32(1.0002)).astype(np.float32)  # knn_slack_kernel
    kq = (np.float32(0) if ip else kappa * qq * np.float32(1.0002)).astype(np.float32) * np.ones_like(qq)
    bl = (bn - kappa * bn).astype(np.float32)
    checked = admitted_f32 = 0
    for i in range(q.shape[0]):
        # a threshold in the middle of this query's lo values, and the extremes
        lo_ip = -(acc[i] + ea[i] * xn_k)
        lo_l2 = bl - np.float32(2) * acc[i] - ea[i] * xn_k
        lo = (lo_ip if ip else lo_l2).astype(np.float32)
        for U in [np.float32(np.median(lo)), np.float32(lo.min()), np.float32(np.percentile(lo, 90)), np.float32(np.inf)]:
            thr = np.float32(U + kq[i])
            e_col = bf16_up(np.float32(ea[i] * np.float32(1.001 if ip else 0.5005)))
            thr_m = np.float32(thr + abs(thr) * MARGIN + np.float32(1e-30))
            t = split3_up(np.array([thr_m if ip else np.float32(0.5) * thr_m], np.float32))
            for j in range(x.shape[0]):
                terms = [acc[i, j], np.float32(e_col) * xn_col[j], t[0][0], t[1][0], t[2][0]]
                if not ip:
                    b = split3_up(np.array([np.float32(-0.5) * (bl[j] - MARGIN * bl[j])], np.float32))
                    terms += [b[0][0], b[1][0], b[2][0]]
                S_worst = f32_sum_worst(terms, -1)
                checked += 1
                if lo[j] <= thr:
                    admitted_f32 += 1
                    assert S_worst >= 0 or np.isnan(S_worst) and not np.isfinite(thr), (d, ip, i, j, float(U), float(S_worst))
    assert admitted_f32 > checked // 10
