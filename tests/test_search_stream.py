"""GPU parity of the batch-stream form (roar_search_stream_{open,submit,wait,close}, csrc/roar_stream.cpp; kernel: SRC == 2 in
csrc/roar_search_v2.h): a stream of ragged batches == the same batches as separate launches == the CPU oracle, query by query.

The loop being split is the reference harness's `omp parallel for` over the queries of one batch
(tests/test_search_roargraph.cpp:202-212); a stream serves many such batches with workgroups that cross batch boundaries, so the
tests push on exactly that: batches far smaller and far larger than the number of workgroups, a handful of workgroups serving
dozens of batches, slot reuse (ring), counter-cell reuse (epoch reset), waits in any order, buffers rewritten between batches,
several submitting threads.

Bar: ids / distances / hops bit-exact vs the oracle; cmps >= the oracle's (the stream runs the forgetful on-chip table) and equal to
the per-launch form's with the same table.
"""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K = 10
CASES = {"l2_128": 0, "ip_200": 1}


def _random_case(n, d, nq, seed, maxdeg=40):
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    deg = rng.integers(1, maxdeg, n)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(deg)
    nbrs = rng.integers(0, n, int(off[-1])).astype(np.uint32)
    return base, q, off, nbrs


def _golden_index(case, golden_dir):
    from mysteryann_amd import fileio
    from mysteryann_amd import index as mi

    base = fileio.read_fbin(os.path.join(golden_dir, case + ".base.fbin"))
    idx = mi.IndexBipartite(base.shape[1], base.shape[0], CASES[case], None)
    idx.LoadSearchNeededData(os.path.join(golden_dir, case + ".base.fbin"))
    idx.LoadProjectionGraph(os.path.join(golden_dir, case + ".roar.index"))
    q = fileio.read_fbin(os.path.join(golden_dir, case + ".query.fbin"))
    return idx, q


def _cuts(total, sizes):
    """consecutive (lo, hi) row ranges of the given sizes, wrapping around `total` rows"""
    out, lo = [], 0
    for s in sizes:
        if lo + s > total:
            lo = 0
        out.append((lo, lo + s))
        lo += s
    return out


def _check(r, o, lo, hi, what, cmps_of=None):
    for key in ("ids", "dists", "hops"):
        assert np.array_equal(r[key].cpu().numpy().view(o[key].dtype), o[key][lo:hi]), (what, key, lo, hi)
    c = r["cmps"].cpu().numpy().astype(np.int64)
    assert (c >= o["cmps"][lo:hi].astype(np.int64)).all(), (what, "cmps under the reference's")
    if cmps_of is not None:
        assert np.array_equal(c, cmps_of[lo:hi].astype(np.int64)), (what, "cmps differ from the per-launch form's")


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("L", [10, 50, 100])
def test_stream_of_ragged_batches_equals_separate_launches_and_golden(case, L, golden_dir, built_lib):
    """The reference's goldens, cut into ragged batches (1 ... all queries, repeated), through one stream: every batch equals the
    golden rows (ids, distances, hops exact) and the per-launch form with the same forgetful table (cmps too)."""
    import torch

    from mysteryann_amd import fileio

    exp = np.load(os.path.join(golden_dir, case + ".expected.npz"))
    idx, q = _golden_index(case, golden_dir)
    idx.set_option("calibrate", 0)
    idx.set_option("visited_mode", 3)
    qd = torch.from_numpy(fileio.pad_rows(q)).cuda()
    nq = q.shape[0]
    whole = idx.SearchRoarGraphBatchDevice(qd, K, L)
    torch.cuda.synchronize()
    o = {k_: exp[f"{k_}_{L}"] for k_ in ("ids", "dists", "hops", "cmps")}
    per_launch_cmps = whole["cmps"].cpu().numpy()
    sizes = [1, 7, nq, 3, 30, 1, 1, nq - 1, 2, 17, 5, nq, 9]
    with idx.OpenSearchStream(K, L) as st:
        pend = []
        for lo, hi in _cuts(nq, sizes):
            t, out = st.submit(qd[lo:hi])
            pend.append((t, out, lo, hi))
        for t, out, lo, hi in pend:
            assert st.wait(t) == 0
            _check(out, o, lo, hi, (case, L), per_launch_cmps)
        assert st.info("batches_completed") == len(sizes)
        assert st.info("launches") >= 1 and st.info("resident_capacity") >= 1
    idx.close()


@pytest.mark.parametrize("grid", [3, 64, 0], ids=["3-workgroups", "64-workgroups", "whole-chip"])
@pytest.mark.parametrize("d,metric,L", [(200, 1, 40), (128, 0, 100), (96, 1, 64), (512, 1, 65)])
def test_few_workgroups_cross_many_batch_boundaries(grid, d, metric, L, built_lib):
    """Random graph; 60 batches of 1 ... 300 queries against a stream whose workgroups are capped (option max_grid) at 3 / 64 / the
    chip: with 3 workgroups each one crosses dozens of batch boundaries and batches finish far apart; every query equals the oracle."""
    import oracle
    import torch

    from mysteryann_amd import index as mi

    n, nq = 5000, 1200
    base, q, off, nbrs = _random_case(n, d, nq, seed=d + 3 * metric)
    o = oracle.search_batch(base, off, nbrs, 11, q, K, L, metric, threads=8)
    assert not o["short"]
    idx = mi.IndexBipartite(d, n, metric, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(11, off, nbrs)
    idx.set_option("calibrate", 0)
    if grid:
        idx.set_option("max_grid", grid)
    qd = torch.from_numpy(q).cuda()
    rng = np.random.default_rng(5)
    sizes = [int(x) for x in rng.integers(1, 300, 60)]
    sizes[7] = 1
    sizes[8] = 1
    with idx.OpenSearchStream(K, L) as st:
        pend = []
        for lo, hi in _cuts(nq, sizes):
            t, out = st.submit(qd[lo:hi])
            pend.append((t, out, lo, hi))
        for t, out, lo, hi in reversed(pend):  # (waits in the opposite order)
            assert st.wait(t) == 0
            _check(out, o, lo, hi, (grid, d, metric, L))
        if grid:
            assert st.info("resident_capacity") <= grid
        lo_t = [st.times(t) for t, *_ in pend]
        assert all(b_ >= a_ > 0 for a_, b_ in lo_t)
    idx.close()


def test_slots_and_counter_cells_are_reused(built_lib, monkeypatch):
    """ROAR_STREAM_CAP = 8: a ring of 4 slots and an epoch of 8 batch numbers -- 50 batches reuse every slot a dozen times and reset
    the counter cells six times; submit() blocks on the oldest batch when the ring is full (nothing is waited for until the end)."""
    import oracle
    import torch

    from mysteryann_amd import index as mi

    monkeypatch.setenv("ROAR_STREAM_CAP", "8")
    n, d, nq, L = 4000, 200, 600, 30
    base, q, off, nbrs = _random_case(n, d, nq, seed=21)
    o = oracle.search_batch(base, off, nbrs, 3, q, K, L, 1, threads=8)
    idx = mi.IndexBipartite(d, n, 1, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(3, off, nbrs)
    idx.set_option("calibrate", 0)
    qd = torch.from_numpy(q).cuda()
    rng = np.random.default_rng(9)
    sizes = [int(x) for x in rng.integers(1, 120, 50)]
    with idx.OpenSearchStream(K, L) as st:
        assert st.info("ring") == 4
        pend = []
        for lo, hi in _cuts(nq, sizes):
            t, out = st.submit(qd[lo:hi])
            pend.append((t, out, lo, hi))
        assert st.info("epochs") == 6
        for t, out, lo, hi in pend:
            assert st.wait(t) == 0
            assert st.wait(t) == 0  # (twice: harmless)
            _check(out, o, lo, hi, "ring")
        assert [t for t, *_ in pend] == list(range(1, 51))
        # the stream's memory does not grow with the batches it has served: full records for the running epoch and the one before it
        from mysteryann_amd.capi import ERR_ARG, RoarError

        st.times(pend[-1][0])
        with pytest.raises(RoarError) as e:
            st.times(pend[0][0])
        assert e.value.code == ERR_ARG and "no longer kept" in str(e.value)
    idx.close()


def test_short_counts_of_tickets_older_than_the_kept_records(built_lib, monkeypatch):
    """wait() answers for every ticket ever handed out: of a ticket whose record was dropped (two epochs back) the non-zero short count
    is still known."""
    import torch

    from mysteryann_amd import index as mi

    monkeypatch.setenv("ROAR_STREAM_CAP", "4")
    n, d = 200, 24
    rng = np.random.default_rng(1)
    base = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((9, d)).astype(np.float32)
    off = np.zeros(n + 1, np.uint64)
    nb = []
    for v in range(n):  # nodes 0..4: a closed clique, nothing else is reached from entry point 0 -> every query ends short of k = 10
        row = [u for u in range(5) if u != v] if v < 5 else [0]
        nb += row
        off[v + 1] = len(nb)
    idx = mi.IndexBipartite(d, n, 0, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(0, off, np.array(nb, np.uint32))
    idx.set_option("calibrate", 0)
    qd = torch.from_numpy(q).cuda()
    with idx.OpenSearchStream(K, 20) as st:
        tickets = [st.submit(qd[: 1 + (i % 9)])[0] for i in range(14)]
        assert st.info("epochs") == 3
        for i, t in enumerate(tickets):
            assert st.wait(t) == 1 + (i % 9), (i, t)
    idx.close()


def test_a_rewritten_query_buffer_is_read_afresh(built_lib):
    """One device buffer, rewritten between batches (after the wait): a workgroup that lives across the batches must not find the
    old rows in its L2 (system-scope loads of the query rows).  Alternates two query sets through one buffer, whole chip."""
    import oracle
    import torch

    from mysteryann_amd import index as mi

    n, d, nq, L = 20000, 200, 2000, 50
    base, q, off, nbrs = _random_case(n, d, 2 * nq, seed=4)
    o = oracle.search_batch(base, off, nbrs, 0, q, K, L, 1, threads=8)
    idx = mi.IndexBipartite(d, n, 1, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(0, off, nbrs)
    idx.set_option("calibrate", 0)
    qa, qb = torch.from_numpy(q[:nq]).cuda(), torch.from_numpy(q[nq:]).cuda()
    buf = torch.empty_like(qa)
    keep = torch.from_numpy(q[:nq]).cuda()  # a second batch that keeps the workgroups alive while `buf` is rewritten
    with idx.OpenSearchStream(K, L) as st:
        for rep in range(6):
            src, lo = (qa, 0) if rep % 2 == 0 else (qb, nq)
            tk, _ = st.submit(keep)
            buf.copy_(src)
            torch.cuda.synchronize()
            t, out = st.submit(buf)
            assert st.wait(t) == 0
            _check(out, o, lo, lo + nq, ("rewritten", rep))
            st.wait(tk)
    idx.close()


def test_several_threads_submit_and_wait(built_lib):
    import oracle
    import torch

    from mysteryann_amd import index as mi

    n, d, nq, L = 8000, 128, 1600, 64
    base, q, off, nbrs = _random_case(n, d, nq, seed=8)
    o = oracle.search_batch(base, off, nbrs, 1, q, K, L, 0, threads=8)
    idx = mi.IndexBipartite(d, n, 0, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(1, off, nbrs)
    idx.set_option("calibrate", 0)
    qd = torch.from_numpy(q).cuda()
    errs = []
    with idx.OpenSearchStream(K, L) as st:
        def worker(w):
            try:
                rng = np.random.default_rng(w)
                for _ in range(25):
                    lo = int(rng.integers(0, nq - 200))
                    hi = lo + int(rng.integers(1, 200))
                    t, out = st.submit(qd[lo:hi])
                    assert st.wait(t) == 0
                    _check(out, o, lo, hi, ("thread", w))
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=worker, args=(w,)) for w in range(4)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs
        assert st.info("batches_completed") == 100
    idx.close()


def test_short_results_and_argument_errors(built_lib):
    """A graph whose searches cannot reach k nodes: wait() reports the short queries (tail ids 0xFFFFFFFF, as the per-launch form);
    visited_mode EXACT / LDS-exact, k > L_pq, a ticket never handed out, a changed index under an open stream: ROAR_ERR_ARG."""
    import torch

    from mysteryann_amd import index as mi
    from mysteryann_amd.capi import ERR_ARG, RoarError

    n, d = 200, 24
    rng = np.random.default_rng(1)
    base = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((9, d)).astype(np.float32)
    # nodes 0..4 form a closed clique, the rest is never reached from entry point 0
    off = np.zeros(n + 1, np.uint64)
    nb = []
    for v in range(n):
        row = [u for u in range(5) if u != v] if v < 5 else [0]
        nb += row
        off[v + 1] = len(nb)
    nbrs = np.array(nb, np.uint32)
    idx = mi.IndexBipartite(d, n, 0, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(0, off, nbrs)
    idx.set_option("calibrate", 0)
    qd = torch.from_numpy(q).cuda()
    # (the reference throws on a short pool, index_bipartite.cpp:2408-2412, and so does the oracle: the checker here is exact L2 over
    # the five reachable nodes, and the per-launch form)
    d2 = ((q[:, None, :].astype(np.float64) - base[None, :5, :].astype(np.float64)) ** 2).sum(-1)
    per_launch = idx.SearchRoarGraphBatchDevice(qd, K, 20)
    torch.cuda.synchronize()
    assert int(per_launch["short"].item()) == 9
    with idx.OpenSearchStream(K, 20) as st:
        t, out = st.submit(qd)
        assert st.wait(t) == 9
        ids = out["ids"].cpu().numpy().view(np.uint32)
        assert np.array_equal(ids[:, :5], np.argsort(d2, axis=1, kind="stable").astype(np.uint32))
        assert (ids[:, 5:] == 0xFFFFFFFF).all()
        assert np.array_equal(ids, per_launch["ids"].cpu().numpy().view(np.uint32))
        assert np.array_equal(out["dists"].cpu().numpy(), per_launch["dists"].cpu().numpy())
        t0, _ = st.submit(qd[:0])
        assert t0 == 0 and st.wait(0) == 0
        with pytest.raises(RoarError) as e:
            st.wait(t + 5)
        assert e.value.code == ERR_ARG
        with pytest.raises(RoarError) as e:  # the index is in use
            idx.SetBaseData(base)
        assert e.value.code == ERR_ARG and "stream" in str(e.value)
    idx.SetBaseData(base)  # (closed: allowed again)
    idx.SetProjectionGraph(0, off, nbrs)
    for mode in (0, 1):
        idx.set_option("visited_mode", mode)
        with pytest.raises(RoarError) as e:
            idx.OpenSearchStream(K, 20)
        assert e.value.code == ERR_ARG
    idx.set_option("visited_mode", 2)
    with pytest.raises(RoarError) as e:
        idx.OpenSearchStream(K, 5)
    assert "L_pq must greater or equal than k" in str(e.value)
    idx.close()


def test_an_idle_stream_holds_no_workgroups_and_restarts(built_lib):
    """After every batch is complete the workgroups leave (nothing spins on the host); the next submit starts new ones."""
    import time

    import oracle
    import torch

    from mysteryann_amd import index as mi

    n, d, nq, L = 6000, 200, 500, 40
    base, q, off, nbrs = _random_case(n, d, nq, seed=14)
    o = oracle.search_batch(base, off, nbrs, 2, q, K, L, 1, threads=8)
    idx = mi.IndexBipartite(d, n, 1, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(2, off, nbrs)
    idx.set_option("calibrate", 0)
    qd = torch.from_numpy(q).cuda()
    with idx.OpenSearchStream(K, L) as st:
        launches = 0
        for rep in range(4):
            t, out = st.submit(qd)
            assert st.wait(t) == 0
            _check(out, o, 0, nq, ("idle", rep))
            t_end = time.time() + 30.0  # (they leave within microseconds; the margin is for a box that is busy with something else)
            while st.info("workgroups_alive") and time.time() < t_end:
                time.sleep(0.001)
            assert st.info("workgroups_alive") == 0, (rep, st.info("workgroups_alive"), st.info("launches"))
            assert st.info("launches") > launches, (rep, st.info("launches"), launches)
            launches = st.info("launches")
            # another kernel gets the whole chip while the stream is open and idle
            x = idx.SearchRoarGraphBatchDevice(qd, K, L)
            torch.cuda.synchronize()
            assert np.array_equal(x["ids"].cpu().numpy().view(np.uint32), o["ids"])
    idx.close()


def test_a_small_first_batch_does_not_make_the_calibration(built_lib):
    """a stream whose first batch is smaller than one residency of the chip must not measure (and cache, for every later
    stream and capped launch of this L_pq band) the launch plan of the largest batches on it: it runs on defaults ("plan_provisional"),
    caches nothing, and plans again -- once, drained -- when a batch arrives that is a sample one can measure on.  Results are the
    oracle's before and after."""
    import oracle
    import torch

    from mysteryann_amd import index as mi

    n, d, nq, L = 30000, 200, 4500, 40
    base, q, off, nbrs = _random_case(n, d, nq, seed=33)
    o = oracle.search_batch(base, off, nbrs, 5, q, K, L, 1, threads=8)
    idx = mi.IndexBipartite(d, n, 1, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(5, off, nbrs)  # (calibrate = 1, the default)
    qd = torch.from_numpy(q).cuda()
    need = idx.info("n_cus") * 12
    assert nq >= need, (nq, need)
    with idx.OpenSearchStream(K, L) as st:
        t, out = st.submit(qd[:5])
        assert st.wait(t) == 0
        _check(out, o, 0, 5, "five queries")
        assert st.info("plan_provisional") == 1 and idx.info("calibrations") == 0
        t, out = st.submit(qd[5:105])
        assert st.wait(t) == 0
        _check(out, o, 5, 105, "a hundred queries")
        assert st.info("plan_provisional") == 1 and idx.info("calibrations") == 0 and st.info("replans") == 0
        pend = [(st.submit(qd[200:300]), 200, 300), (st.submit(qd), 0, nq), (st.submit(qd[100:200]), 100, 200)]  # the second one plans again
        for (t, out), lo, hi in pend:
            assert st.wait(t) == 0
            _check(out, o, lo, hi, ("after the calibration", lo, hi))
        assert st.info("plan_provisional") == 0 and st.info("replans") == 1 and idx.info("calibrations") >= 1
    cals = idx.info("calibrations")
    with idx.OpenSearchStream(K, L) as st:  # a second stream, small batches only: takes what was measured
        t, out = st.submit(qd[:7])
        assert st.wait(t) == 0
        _check(out, o, 0, 7, "second stream")
        assert st.info("plan_provisional") == 0 and idx.info("calibrations") == cals
    # a launch that has the chip to itself is another regime: the same candidates ranked without the preference for resident workgroups,
    # its own entry in the cache
    r = idx.SearchRoarGraphBatchDevice(qd, K, L)
    torch.cuda.synchronize()
    assert idx.info("calibrations") == cals + 1
    assert np.array_equal(r["ids"].cpu().numpy().view(np.uint32), o["ids"])
    idx.close()


def test_closing_a_failed_stream_frees_it_once_its_launches_are_over(built_lib):
    """close() of a failed stream may let the index go (roar_set_base / roar_destroy free what workgroups read) only when
    the stream's launches are known to have ended; then it frees the stream as well and still returns the error."""
    import torch

    from mysteryann_amd import index as mi
    from mysteryann_amd.capi import ERR_ARG, ERR_DEVICE, RoarError

    n, d, nq, L = 3000, 128, 300, 30
    base, q, off, nbrs = _random_case(n, d, nq, seed=5)
    idx = mi.IndexBipartite(d, n, 0, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(1, off, nbrs)
    idx.set_option("calibrate", 0)
    qd = torch.from_numpy(q).cuda()
    ref = idx.SearchRoarGraphBatchDevice(qd, K, L)
    torch.cuda.synchronize()
    st = idx.OpenSearchStream(K, L)
    t, out = st.submit(qd)
    assert st.wait(t) == 0
    assert st.info("debug_break") == 1
    with pytest.raises(RoarError) as e:
        st.submit(qd)
    assert e.value.code == ERR_DEVICE
    with pytest.raises(RoarError) as e:  # (still open: the index is in use)
        idx.SetBaseData(base)
    assert e.value.code == ERR_ARG
    with pytest.raises(RoarError) as e:
        st.close()
    assert e.value.code == ERR_DEVICE and "closed" in str(e.value)
    idx.SetBaseData(base)  # the launches were over: the index is free again
    idx.SetProjectionGraph(1, off, nbrs)
    with idx.OpenSearchStream(K, L) as st2:
        t, out2 = st2.submit(qd)
        assert st2.wait(t) == 0
        assert np.array_equal(out2["ids"].cpu().numpy(), ref["ids"].cpu().numpy())
    idx.close()
