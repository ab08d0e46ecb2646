"""GPU parity: the HIP path (through the C ABI) against the golden fixtures produced by the
reference's own components, and against the CPU oracle on seeded inputs.

Bar: ids bit-exact, distances bit-exact (the device code replays the reference's AVX-512 + FMA
summation order; -0.0 is returned as +0.0), hops exact, cmps exact in EXACT visited mode and in LDS
mode whenever the tag table cannot overflow, otherwise cmps >= reference.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LS = [10, 20, 50, 100, 200]
CASES = {"l2_128": 0, "ip_200": 1, "cos_512": 4}
K = 10


def _index(case, golden_dir, mode):
    from mysteryann_amd import index as mi

    metric = CASES[case]
    from mysteryann_amd import fileio

    base = fileio.read_fbin(os.path.join(golden_dir, case + ".base.fbin"))
    idx = mi.IndexBipartite(base.shape[1], base.shape[0], metric, None)
    idx.LoadSearchNeededData(os.path.join(golden_dir, case + ".base.fbin"))
    idx.LoadProjectionGraph(os.path.join(golden_dir, case + ".roar.index"))
    idx.set_option("visited_mode", mode)
    q = fileio.read_fbin(os.path.join(golden_dir, case + ".query.fbin"))
    if idx.need_normalize:
        q = mi.normalize(q)
    return idx, q


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("mode", [0, 1], ids=["exact", "lds"])
def test_search_matches_reference_golden(case, mode, golden_dir, built_lib):
    exp = np.load(os.path.join(golden_dir, case + ".expected.npz"))
    idx, q = _index(case, golden_dir, mode)
    for L in LS:
        r = idx.SearchRoarGraphBatch(q, K, L)
        if case == "cos_512":
            # base rows are normalised by the loader with a sequential f32 sum, the reference's
            # -Ofast build vectorises that sum: inputs differ in the last bit, so compare sets with
            # the tolerance north_star states (ids equal where distances are not within 1e-4 ties).
            # north_star's criterion, position by position: distances within 1e-4, ids equal WHERE THERE IS NO DISTANCE TIE -- an id
            # that differs must be explained by a near-tie: it stands in the reference's list within 1e-4 of the position it took, or
            # (not in the reference's top k at all) it came in within 1e-4 of the reference's k-th distance.
            e_ids, e_d = exp[f"ids_{L}"], exp[f"dists_{L}"]
            assert np.abs(r["dists"] - e_d).max() <= 1e-4
            for qi, pos in zip(*np.nonzero(r["ids"] != e_ids)):
                where = np.nonzero(e_ids[qi] == r["ids"][qi, pos])[0]
                if where.size:
                    assert abs(float(e_d[qi, where[0]]) - float(e_d[qi, pos])) <= 1e-4, (case, L, qi, pos, "ids swapped without a tie")
                else:
                    assert abs(float(r["dists"][qi, pos]) - float(e_d[qi, -1])) <= 1e-4, (case, L, qi, pos, "a new id without a tie at the k-th")
            assert (r["ids"] == e_ids).mean() >= 0.98, (case, L)
            assert np.array_equal(r["hops"], exp[f"hops_{L}"]) or np.abs(
                r["hops"].astype(int) - exp[f"hops_{L}"].astype(int)).max() <= 2
            continue
        assert np.array_equal(r["ids"], exp[f"ids_{L}"]), (case, L)
        assert np.array_equal(r["dists"], exp[f"dists_{L}"]), (case, L)
        assert np.array_equal(r["hops"], exp[f"hops_{L}"]), (case, L)
        assert np.array_equal(r["scanned"], exp[f"scanned_{L}"]), (case, L)
        assert np.array_equal(r["cmps"], exp[f"cmps_{L}"]), (case, L)
    idx.close()


@pytest.mark.parametrize("case", ["ip_200", "l2_128", "cos_512"])
def test_traffic_counters_of_the_counting_instantiation(case, golden_dir, built_lib):
    """roar_set_traffic_counters_device: the counting instantiation returns the plain kernel's results and, per query,
    {rows on the 8-bit copy, rows on the f32 rows, adjacency words fetched}.  Screen off: every scored neighbour reads its
    f32 row (rows_f32 = cmps + 1 for the entry point) and nothing is coded; screen on: rows_f32 = 1 + cmps - coded + passed
    with 0 <= passed <= coded, and the hop count bounds the adjacency chunks from below."""
    import torch

    from mysteryann_amd import fileio

    from mysteryann_amd.capi import ERR_ARG, RoarError

    idx, q = _index(case, golden_dir, 1)
    qd = torch.from_numpy(fileio.pad_rows(q)).cuda()
    nq = qd.shape[0]
    blk = torch.full((nq + 3, 4), -1, dtype=torch.int32, device="cuda")
    idx.SetTrafficCounters(blk)
    idx.set_option("search_kernel", 2)  # four waves per query: no counting form -> refused, not silently skipped
    with pytest.raises(RoarError) as e:
        idx.SearchRoarGraphBatchDevice(qd, K, 50)
    assert e.value.code == ERR_ARG
    idx.set_option("search_kernel", 1)
    for screen in (1, 2):
        idx.set_option("screen", screen)
        for L in (10, 50, 200):
            idx.SetTrafficCounters(None)
            plain = idx.SearchRoarGraphBatchDevice(qd, K, L)
            torch.cuda.synchronize()
            blk.fill_(-1)
            idx.SetTrafficCounters(blk)
            cnt = idx.SearchRoarGraphBatchDevice(qd, K, L)
            torch.cuda.synchronize()
            for key in ("ids", "dists", "cmps", "hops", "scanned"):
                assert torch.equal(plain[key], cnt[key]), (case, screen, L, key)
            t = blk.cpu().numpy().astype(np.int64)
            assert (t[nq:] == -1).all()  # rows past the batch are not touched
            coded, f32rows, words, spare = t[:nq, 0], t[:nq, 1], t[:nq, 2], t[:nq, 3]
            cmps = cnt["cmps"].cpu().numpy().astype(np.int64)
            hops = cnt["hops"].cpu().numpy().astype(np.int64)
            assert (spare == 0).all() and (words % 64 == 0).all()
            assert (words // 64 >= hops + 1).all()  # at least one chunk per expansion (speculative or not) + the look-ahead of the last
            assert (words // 64 <= 2 * 4 * (hops + 1)).all()  # (hub-bitmap form, L_pq >= 100 without the screen: four chunks per fetch)
            if screen == 1 or not idx.info("last_screen"):
                assert (coded == 0).all() and (f32rows == cmps + 1).all(), (case, screen, L)
            else:
                passed = f32rows - 1 - cmps + coded
                assert (passed >= 0).all() and (passed <= coded).all() and (coded <= cmps).all(), (case, screen, L)
                assert coded.sum() > 0 or L == 200  # (a pool longer than what a query reaches never fills: nothing is screened)
    # a block that is too small for the launch is left alone
    small = torch.full((8, 4), -1, dtype=torch.int32, device="cuda")
    idx.SetTrafficCounters(small)
    idx.SearchRoarGraphBatchDevice(qd, K, 20)
    torch.cuda.synchronize()
    assert (small.cpu().numpy() == -1).all()
    idx.SetTrafficCounters(None)
    idx.close()


def test_cosine_bit_exact_when_inputs_are_prenormalised(golden_dir, built_lib):
    """Same search, but both sides get the SAME normalised rows (IP metric): must be bit-exact."""
    import oracle
    from mysteryann_amd import fileio
    from mysteryann_amd import index as mi

    base = mi.normalize(fileio.read_fbin(os.path.join(golden_dir, "cos_512.base.fbin")))
    q = mi.normalize(fileio.read_fbin(os.path.join(golden_dir, "cos_512.query.fbin")))
    ep, off, nbrs = oracle.graph_load(os.path.join(golden_dir, "cos_512.roar.index"))
    idx = mi.IndexBipartite(512, base.shape[0], mi.INNER_PRODUCT, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(ep, off, nbrs)
    for L in (10, 50, 200):
        o = oracle.search_batch(base, off, nbrs, ep, q, K, L, 1)
        r = idx.SearchRoarGraphBatch(q, K, L)
        for key in ("ids", "dists", "cmps", "hops"):
            assert np.array_equal(r[key], o[key]), (key, L)


@pytest.mark.parametrize("metric,name", [(0, "l2"), (1, "ip")])
def test_distance_kernel_matches_reference_kat(metric, name, golden_dir, built_lib):
    """A2/A3: device distances == the reference's compare() outputs, bit for bit (d % 8 == 0: the
    loader pads rows to a multiple of 8, other lengths never reach the kernel)."""
    from mysteryann_amd import index as mi

    kat = np.load(os.path.join(golden_dir, "distance_kat.npz"))
    for d in (8, 128, 200, 512):
        a, b = kat[f"a_{d}"], kat[f"b_{d}"]
        idx = mi.IndexBipartite(d, a.shape[0], metric, None)
        idx.SetBaseData(a)
        ids = np.arange(a.shape[0], dtype=np.uint32).reshape(-1, 1)
        got = idx.Distances(b, ids)[:, 0]
        assert np.array_equal(got, kat[f"{name}_{d}"]), (name, d, np.abs(got - kat[f"{name}_{d}"]).max())
        idx.close()
    # d = 12, 20: rows are zero padded to 16 / 24 by the loader; compare with the reference run on the
    # padded rows (that is what SearchRoarGraph passes it, index_bipartite.cpp:2337)
    import oracle

    for d in (12, 20):
        a, b = kat[f"a_{d}"], kat[f"b_{d}"]
        pd = (d + 7) // 8 * 8
        ap, bp = np.zeros((a.shape[0], pd), np.float32), np.zeros((a.shape[0], pd), np.float32)
        ap[:, :d], bp[:, :d] = a, b
        want = oracle.dist_many(metric, ap, bp, pd)
        idx = mi.IndexBipartite(d, a.shape[0], metric, None)
        idx.SetBaseData(a)
        got = idx.Distances(b, np.arange(a.shape[0], dtype=np.uint32).reshape(-1, 1))[:, 0]
        assert np.array_equal(got, want), (name, d)
        idx.close()


def _random_case(n, d, nq, metric, seed, maxdeg=40):
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    deg = rng.integers(0, maxdeg, n)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(deg)
    nbrs = rng.integers(0, n, int(off[-1])).astype(np.uint32)
    return base, q, off, nbrs


def test_pool_lengths_across_the_merge_s_boundaries(built_lib):
    """The LDS-pool merge (csrc/roar_wave.h merge_candidates) changes gear with the pool's length: binary search alone up to 128 entries,
    64 splitters + a search inside a (length / 64)-entry segment above (segment length 3 from 129, 4 from 193, 5 from 257 ...), shifts
    in whole 64-entry chunks, four at a time from 256 entries to move, and a masked chunk at either end.  Every L_pq around those
    boundaries on a small graph whose searches keep meeting their own pool (duplicates), forgetful table (duplicates reach the
    merge) and exact tags: ids, distances, hops as the oracle's; cmps as well in the exact form."""
    import oracle
    from mysteryann_amd import index as mi

    n, d, nq = 1500, 24, 24
    base, q, off, nbrs = _random_case(n, d, nq, 0, seed=77, maxdeg=60)
    idx = mi.IndexBipartite(d, n, 0, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(7, off, nbrs)
    idx.set_option("calibrate", 0), idx.set_option("hub_mode", 1)
    Ls = [65, 66, 127, 128, 129, 130, 131, 191, 192, 193, 194, 255, 256, 257, 258, 319, 320, 321, 383, 384, 385, 449, 511, 512, 513,
          640, 641, 1023, 1024, 1025, 1400]
    for L in Ls:
        o = oracle.search_batch(base, off, nbrs, 7, q, K, L, 0, threads=4)
        for mode, ve in ((3, 64), (3, 1024), (0, 0)):
            idx.set_option("visited_mode", mode), idx.set_option("vis_entries", ve)
            r = idx.SearchRoarGraphBatch(q, K, L)
            for key in ("ids", "dists", "hops") + (("cmps",) if mode == 0 else ()):
                assert np.array_equal(r[key], o[key]), (L, mode, ve, key)
    idx.close()


@pytest.mark.parametrize("d", [8, 24, 128, 200, 264, 512])
@pytest.mark.parametrize("metric", [0, 1])
def test_random_graphs_against_oracle(d, metric, built_lib):
    """Random digraphs (self loops, repeated neighbours, zero-degree nodes, ragged dims incl. the
    generic-dim kernel and the 8-float tail), entry point anywhere."""
    import oracle
    from mysteryann_amd import index as mi

    n, nq = 3000, 40
    base, q, off, nbrs = _random_case(n, d, nq, metric, seed=d * 7 + metric)
    ep = int((d * 13) % n)
    for mode in (0, 1):
        idx = mi.IndexBipartite(d, n, metric, None)
        idx.SetBaseData(base)
        idx.SetProjectionGraph(ep, off, nbrs)
        idx.set_option("visited_mode", mode)
        for L in (10, 64, 65, 300):
            o = oracle.search_batch(base, off, nbrs, ep, q, K, L, metric)
            if o["short"]:
                continue
            r = idx.SearchRoarGraphBatch(q, K, L)
            assert np.array_equal(r["ids"], o["ids"]), (d, metric, mode, L)
            assert np.array_equal(r["dists"], o["dists"]), (d, metric, mode, L)
            assert np.array_equal(r["hops"], o["hops"]), (d, metric, mode, L)
            assert np.array_equal(r["cmps"], o["cmps"]), (d, metric, mode, L)
        idx.close()


def test_high_degree_rows_and_large_pool(built_lib):
    """Degrees above 63 (second adjacency chunk), L_pq = 2000 (largest value of the reference's
    sweep, run_roargraph_search_test.sh:13), k = L."""
    import oracle
    from mysteryann_amd import index as mi

    n, d, nq = 6000, 200, 16
    base, q, off, nbrs = _random_case(n, d, nq, 1, seed=77, maxdeg=120)
    idx = mi.IndexBipartite(d, n, 1, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(5, off, nbrs)
    for mode in (0, 1, 3):  # EXACT, LDS (exact: full sets spill to HBM), LDS_FORGETFUL
        idx.set_option("visited_mode", mode)
        # (5000: a pool beyond 4 096 entries takes the pieced second level of the merge's search; the 6 000-node graph
        # never fills it, so this is also the short-of-L_pq pool)
        for L, k in ((2000, 10), (100, 100), (1, 1), (5000, 10), (600, 7)):
            o = oracle.search_batch(base, off, nbrs, 5, q, k, L, 1)
            r = idx.SearchRoarGraphBatch(q, k, L)
            assert np.array_equal(r["ids"], o["ids"]), (mode, L)
            assert np.array_equal(r["dists"], o["dists"]), (mode, L)
            assert np.array_equal(r["hops"], o["hops"]), (mode, L)
            if mode != 3:
                assert np.array_equal(r["cmps"], o["cmps"]), (mode, L)
            else:
                assert (r["cmps"] >= o["cmps"]).all()


def test_pageable_query_block_is_packed_behind_the_launch(built_lib):
    """A pageable host query block is copied into the library's pinned staging rows WHILE the kernel runs (the kernel
    takes queries in row order and waits for the host's rows-written counter).  Many calls of every size, 800-byte rows
    (not a multiple of the 128-byte line: staging rows are line-aligned for that reason) and padded dimensions: ids and
    distances must equal the device-buffer call's, whatever the race between the packing thread and the launch."""
    from mysteryann_amd import index as mi

    rng = np.random.default_rng(3)
    for d, metric, n in ((200, 1, 60000), (100, 0, 30000)):
        base = rng.standard_normal((n, d)).astype(np.float32)
        deg = rng.integers(8, 40, n)
        off = np.zeros(n + 1, np.uint64)
        off[1:] = np.cumsum(deg)
        nbrs = rng.integers(0, n, int(off[-1])).astype(np.uint32)
        idx = mi.IndexBipartite(d, n, metric, None)
        idx.SetBaseData(base)
        idx.SetProjectionGraph(7, off, nbrs)
        dpad = idx.info("dim_padded")
        for it in range(120):
            nq = int(rng.integers(1, 5000)) if it % 4 else int(rng.integers(1, 70))
            q = rng.standard_normal((nq, d)).astype(np.float32)
            idx.set_option("search_kernel", (0, 1, 3, 2)[it % 4] if nq <= 256 else 0)
            r = idx.SearchRoarGraphBatch(q, 10, 30)
            qd = torch_pad(q, dpad)
            o = idx.SearchRoarGraphBatchDevice(qd, 10, 30)
            assert np.array_equal(r["ids"].view(np.int32), o["ids"].cpu().numpy()), (d, it, nq)
            assert np.array_equal(r["dists"], o["dists"].cpu().numpy()), (d, it, nq)
        idx.close()


def torch_pad(q, dpad):
    import torch

    t = torch.zeros((q.shape[0], dpad), dtype=torch.float32, device="cuda")
    t[:, :q.shape[1]] = torch.from_numpy(q).cuda()
    return t


def test_tag_table_of_three_times_a_power_of_two_sets(built_lib):
    """The on-chip table also comes in 3 * 2^k and (round 5) 5 * 2^k sets (12 288 tags is the default for medium pools on a sparse
    graph; 1 280 is what leaves a 1 000-entry pool a twelfth workgroup per CU): set = id mod sets, tag = id div sets must stay a
    bijection -- a hit may never be false.  Tiny tables (80, 96, 160, 192 tags: full all the time), the default-sized ones, and an id
    range that makes the quotient of the division by 3 / by 5 non-trivial."""
    import oracle
    from mysteryann_amd import index as mi

    n, d, nq = 50000, 24, 48
    base, q, off, nbrs = _random_case(n, d, nq, 0, seed=9, maxdeg=40)
    idx = mi.IndexBipartite(d, n, 0, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(3, off, nbrs)
    idx.set_option("visited_mode", 1)
    idx.set_option("hub_mode", 1)
    for L in (40, 150, 400):
        o = oracle.search_batch(base, off, nbrs, 3, q, K, L, 0)
        for ve in (80, 96, 160, 192, 1280, 3072, 5120, 12288, 0):
            idx.set_option("vis_entries", ve)
            r = idx.SearchRoarGraphBatch(q, K, L)
            if ve:
                assert idx.info("last_vis_entries") == ve
            assert np.array_equal(r["ids"], o["ids"]), (L, ve)
            assert np.array_equal(r["dists"], o["dists"]), (L, ve)
            assert np.array_equal(r["hops"], o["hops"]), (L, ve)
            assert np.array_equal(r["cmps"], o["cmps"]), (L, ve)  # a full set spills, it does not forget
    assert idx.info("last_vis_entries") in (1024, 1280, 1536, 2048, 2560, 3072, 4096, 5120, 6144, 8192, 16384)  # auto: one of the sizes the calibration tries
    assert idx.info("last_plan_calibrated") == 1
    idx.set_option("calibrate", 0)
    idx.SearchRoarGraphBatch(q, K, 400)
    assert idx.info("last_vis_entries") == 4096 and idx.info("last_plan_calibrated") == 0  # default with rows under 768 bytes
    with pytest.raises(Exception):
        idx.set_option("vis_entries", 7 * 1024)
    idx.close()


def test_full_sets_spill_or_forget_without_moving_results(built_lib):
    """A deliberately tiny LDS tag table (64 tags) is full all the time.  LDS mode: ids that find their set full go to the
    workgroup's spill list in HBM -- cmps equals the reference's (VisitedList semantics, visited_list_pool.h:8-29), also
    through the 2-wave and 4-wave forms and the instrumented path's siblings.  LDS_FORGETFUL mode: a victim is overwritten --
    ids / dists / hops must not move, cmps may only grow."""
    import oracle
    from mysteryann_amd import index as mi

    n, d, nq = 20000, 128, 32
    base, q, off, nbrs = _random_case(n, d, nq, 0, seed=5, maxdeg=48)
    idx = mi.IndexBipartite(d, n, 0, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(0, off, nbrs)
    idx.set_option("vis_entries", 64)
    idx.set_option("hub_mode", 1)  # (with the hub bitmap every node of a 20 000-point graph has an exact visited bit)
    o = oracle.search_batch(base, off, nbrs, 0, q, K, 150, 0)
    idx.set_option("visited_mode", 3)
    r = idx.SearchRoarGraphBatch(q, K, 150)
    assert idx.info("last_visited_forgetful") == 1
    assert np.array_equal(r["ids"], o["ids"])
    assert np.array_equal(r["dists"], o["dists"])
    assert np.array_equal(r["hops"], o["hops"])
    assert (r["cmps"] >= o["cmps"]).all() and (r["cmps"] > o["cmps"]).any()
    idx.set_option("visited_mode", 1)
    for kern in (1, 3, 2):  # one, two, four waves per query
        idx.set_option("search_kernel", kern)
        for rep in range(2):  # the spill lists' epochs persist across launches
            r = idx.SearchRoarGraphBatch(q, K, 150)
            assert idx.info("last_visited_forgetful") == 0
            for key in ("ids", "dists", "hops", "cmps"):
                assert np.array_equal(r[key], o[key]), (kern, rep, key)
    idx.set_option("search_kernel", 0)
    idx.set_option("hub_mode", 2)  # hub bitmap in front: nothing reaches the table that a hub bit settles
    r = idx.SearchRoarGraphBatch(q, K, 150)
    assert idx.info("last_hub_bitmap") == 1
    assert np.array_equal(r["ids"], o["ids"]) and np.array_equal(r["cmps"], o["cmps"])


def test_pool_members_keep_their_tags_without_moving_results(built_lib):
    """Option "tag_protect" (forgetful on-chip table under a pool in LDS): bit 15 of a tag says "its node is in the pool", the victim
    choice skips such ways, the merge sets the bit for what it inserts and clears it for what falls off the pool's end (the replacement
    policy the last verdict asked for; visited_list_pool.h:8-29 is what the table stands in for).  Tables far smaller than the pool,
    2^k and 3 * 2^k sets, with and without the 8-bit screen, both metrics: ids / distances / hops are the reference's with either
    setting, cmps never below the reference's, and the protected table does not score more rows than the unprotected one.  (What it
    saves is measured where it matters -- f32 rows of revisited pool members behind the 8-bit screen on the benchmark graph:
    profiles/r04_long_pools_lds.txt; on random graphs a revisit is as likely to be of any node as of a pool member.)"""
    import oracle
    import torch
    from mysteryann_amd import index as mi

    # (the third set is small against its pools: a search keeps meeting its own pool again -- where the policy has something to do)
    for metric, d, n, maxdeg in ((1, 200, 30000, 40), (0, 128, 20000, 48), (0, 128, 2500, 48)):
        base, q, off, nbrs = _random_case(n, d, 48, metric, seed=23 + d, maxdeg=maxdeg)
        idx = mi.IndexBipartite(d, n, metric, None)
        idx.SetBaseData(base)
        idx.SetProjectionGraph(1, off, nbrs)
        idx.set_option("calibrate", 0), idx.set_option("hub_mode", 1), idx.set_option("visited_mode", 3)
        qd = torch.from_numpy(q).cuda()
        gains = []
        for L in (70, 300):
            o = oracle.search_batch(base, off, nbrs, 1, q, K, L, metric)
            for screen in (1, 2):
                if screen == 2 and not (metric == 1 and d == 200) and d != 128:
                    continue
                idx.set_option("screen", screen)
                tot = {0: 0, 1: 0}
                for ve in ((64, 160, 192, 256) if L == 70 else (256, 320, 768, 1024)):
                    idx.set_option("vis_entries", ve)
                    for prot in (0, 1):
                        idx.set_option("tag_protect", prot)
                        # (queries on the device: a pageable host block takes the gated kernel form, which has no such instantiation)
                        r = {k_: v.cpu().numpy() for k_, v in idx.SearchRoarGraphBatchDevice(qd, K, L).items()}
                        assert idx.info("last_visited_forgetful") == 1 and idx.info("last_tag_protect") == prot
                        assert idx.info("last_vis_entries") == ve and idx.info("last_register_pool") == 0
                        for key in ("ids", "dists", "hops"):
                            assert np.array_equal(r[key], o[key]), (metric, L, screen, ve, prot, key)
                        assert (r["cmps"] >= o["cmps"]).all()
                        tot[prot] += int(r["cmps"].sum())
                # (which tags go is a heuristic either way: a table much smaller than the pool has nothing but protected ways to evict)
                assert tot[1] <= 1.002 * tot[0], (metric, L, screen, tot)
                gains.append(1.0 - tot[1] / tot[0])
        assert min(gains) > -0.002, gains
        # a pool in registers has no such table
        idx.set_option("tag_protect", 1), idx.set_option("vis_entries", 256)
        idx.SearchRoarGraphBatchDevice(qd, K, 40)
        assert idx.info("last_register_pool") == 1 and idx.info("last_tag_protect") == 0
        r = idx.SearchRoarGraphBatch(q, K, 300)  # a pageable host block: the gated form, plain forgetful table, same results
        assert idx.info("last_tag_protect") == 0
        for key in ("ids", "dists", "hops"):
            assert np.array_equal(r[key], o[key]), key
        idx.close()


def test_spill_list_epoch_wrap(built_lib):
    """More than 65 535 queries through one resident workgroup: the u16 epoch of its spill list wraps and the list is cleared
    (VisitedList::reset, visited_list_pool.h:20-26).  Tiny table, tiny pool, one workgroup."""
    import oracle
    from mysteryann_amd import index as mi

    n, d, nq = 600, 8, 66000
    base, q, off, nbrs = _random_case(n, d, 500, 0, seed=19, maxdeg=24)
    q = np.tile(q, (132, 1))[:nq].copy()
    idx = mi.IndexBipartite(d, n, 0, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(3, off, nbrs)
    idx.set_option("visited_mode", 1)
    idx.set_option("vis_entries", 64)
    idx.set_option("hub_mode", 1)
    idx.set_option("max_grid", 1)
    o = oracle.search_batch(base, off, nbrs, 3, q[:500], K, 12, 0, threads=4)
    r = idx.SearchRoarGraphBatch(q, K, 12)
    for key in ("ids", "hops", "cmps"):
        assert np.array_equal(r[key][:500], o[key]), key
    want = np.tile(o["cmps"], 132)[:nq]
    assert np.array_equal(r["cmps"], want)  # every copy of a query counts the same, before and after the wrap
    assert np.array_equal(r["ids"], np.tile(o["ids"], (132, 1))[:nq])


def test_exact_mode_epoch_wrap(built_lib):
    """More than 255 queries through one resident wave: the u8 epoch wraps and the tag list is
    cleared (VisitedList::reset, visited_list_pool.h:20-26)."""
    import oracle
    from mysteryann_amd import index as mi

    n, d, nq = 2000, 128, 700
    base, q, off, nbrs = _random_case(n, d, nq, 0, seed=9, maxdeg=24)
    idx = mi.IndexBipartite(d, n, 0, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(3, off, nbrs)
    idx.set_option("visited_mode", 0)
    idx.set_option("max_grid", 2)
    idx.InitVisitedListPool(2)
    o = oracle.search_batch(base, off, nbrs, 3, q, K, 30, 0, threads=4)
    for rep in range(3):  # epochs persist across launches
        r = idx.SearchRoarGraphBatch(q, K, 30)
        assert np.array_equal(r["ids"], o["ids"])
        assert np.array_equal(r["cmps"], o["cmps"])


def test_short_result_is_reported_like_the_reference(built_lib):
    """Fewer reachable nodes than k: the reference throws 'not enough results: x, expected: k'."""
    from mysteryann_amd import index as mi

    n, d = 64, 16
    rng = np.random.default_rng(1)
    base = rng.standard_normal((n, d)).astype(np.float32)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = 2  # only node 0 has (two) neighbours
    nbrs = np.array([1, 2], np.uint32)
    idx = mi.IndexBipartite(d, n, 0, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(0, off, nbrs)
    with pytest.raises(mi.RoarError, match="not enough results: 3, expected: 10"):
        idx.SearchRoarGraphBatch(base[:4], 10, 20)
    with pytest.raises(mi.RoarError, match="L_pq must greater or equal than k"):
        idx.SearchRoarGraphBatch(base[:4], 10, 5)


def test_graph_file_round_trip_and_loader_errors(tmp_path, golden_dir, built_lib):
    from mysteryann_amd import index as mi

    idx = mi.IndexBipartite(200, 2000, 1, None)
    src = os.path.join(golden_dir, "ip_200.roar.index")
    idx.LoadProjectionGraph(src)
    out = str(tmp_path / "copy.index")
    idx.SaveProjectionGraph(out)
    assert open(src, "rb").read() == open(out, "rb").read()  # byte-for-byte (A7)
    bad = str(tmp_path / "bad.fbin")
    with open(bad, "wb") as f:
        np.array([10, 200], np.uint32).tofile(f)
        np.zeros(100, np.float32).tofile(f)
    with pytest.raises(mi.RoarError, match="Data file size wrong!"):
        idx.LoadSearchNeededData(bad)
    with pytest.raises(mi.RoarError, match="open file error"):
        idx.LoadSearchNeededData(str(tmp_path / "missing.fbin"))
    with pytest.raises(mi.RoarError, match="cannot open file"):
        idx.LoadProjectionGraph(str(tmp_path / "missing.index"))


def test_device_resident_batch_and_single_query_facade(golden_dir, built_lib):
    import torch

    from mysteryann_amd import index as mi

    exp = np.load(os.path.join(golden_dir, "ip_200.expected.npz"))
    idx, q = _index("ip_200", golden_dir, 1)
    qd = torch.from_numpy(q).cuda()
    out = idx.SearchRoarGraphBatchDevice(qd, K, 50)
    torch.cuda.synchronize()
    assert np.array_equal(out["ids"].cpu().numpy().view(np.uint32), exp["ids_50"])
    assert np.array_equal(out["dists"].cpu().numpy(), exp["dists_50"])
    assert idx.info("last_kernel_ns") > 0
    # per-query facade, SearchRoarGraph(query, k, qid, parameters, indices, res_dists) -> (cmps, hops)
    p = mi.Parameters()
    p.Set("L_pq", 50)
    ind = np.zeros(K, np.uint32)
    dd = np.zeros(K, np.float32)
    cm, hp = idx.SearchRoarGraph(q[7], K, 7, p, ind, dd)
    assert np.array_equal(ind, exp["ids_50"][7]) and np.array_equal(dd, exp["dists_50"][7])
    assert (cm, hp) == (int(exp["cmps_50"][7]), int(exp["hops_50"][7]))


def test_merge_topk_kernel(built_lib):
    import ctypes as C

    import torch

    from mysteryann_amd import capi

    rng = np.random.default_rng(3)
    S, nq, k = 8, 257, 10
    d = np.sort(rng.integers(0, 50, (S, nq, k)).astype(np.float32) * 0.5, axis=2)  # many ties
    ids = rng.permutation(S * nq * k).astype(np.uint32).reshape(S, nq, k)
    for s in range(S):  # per-shard lists sorted by (dist, id)
        for qi in range(nq):
            o = np.lexsort((ids[s, qi], d[s, qi]))
            ids[s, qi], d[s, qi] = ids[s, qi][o], d[s, qi][o]
    ids[3, 5, 4:] = 0xFFFFFFFF  # a short shard list
    want_i = np.zeros((nq, k), np.uint32)
    want_d = np.zeros((nq, k), np.float32)
    for qi in range(nq):
        ci = np.concatenate([ids[s, qi] for s in range(S)])
        cd = np.concatenate([d[s, qi] for s in range(S)])
        keep = ci != 0xFFFFFFFF
        ci, cd = ci[keep], cd[keep]
        o = np.lexsort((ci, cd))[:k]
        want_i[qi], want_d[qi] = ci[o], cd[o]
    ti = torch.from_numpy(ids.view(np.int32)).cuda()
    td = torch.from_numpy(d).cuda()
    oi = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    od = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    capi.check(capi.lib().roar_merge_topk_device(0, S, nq, k, C.c_void_p(ti.data_ptr()), C.c_void_p(td.data_ptr()),
                                                  C.c_void_p(oi.data_ptr()), C.c_void_p(od.data_ptr()),
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert np.array_equal(oi.cpu().numpy().view(np.uint32), want_i)
    assert np.array_equal(od.cpu().numpy(), want_d)


def test_index_sharded_search_equals_merge_of_per_shard_cpu_searches(built_lib):
    """BASELINE configs[4] on one GPU: two shard indexes (own graphs, local ids) searched in turn,
    ids rebased, lists merged by the HIP kernel == the same merge over the oracle's per-shard results."""
    import torch

    import oracle
    from mysteryann_amd import index as mi
    from mysteryann_amd import parallel

    n, d, nq, L = 6000, 200, 50, 40
    rng = np.random.default_rng(8)
    base = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    qd = torch.from_numpy(q).cuda()
    gi, gd, oi, od = [], [], [], []
    for s in range(2):
        lo, hi = parallel.shard_range(n, 2, s)
        m = hi - lo
        srng = np.random.default_rng(50 + s)
        deg = srng.integers(3, 30, m)
        off = np.zeros(m + 1, np.uint64)
        off[1:] = np.cumsum(deg)
        nbrs = srng.integers(0, m, int(off[-1])).astype(np.uint32)
        idx = mi.IndexBipartite(d, m, mi.INNER_PRODUCT, None)
        idx.SetBaseData(base[lo:hi])
        idx.SetProjectionGraph(1, off, nbrs)
        out = idx.SearchRoarGraphBatchDevice(qd, K, L)
        gi.append(parallel.rebase_ids(out["ids"], lo))
        gd.append(out["dists"])
        o = oracle.search_batch(base[lo:hi], off, nbrs, 1, q, K, L, 1)
        oi.append(o["ids"] + np.uint32(lo))
        od.append(o["dists"])
    mi_, md_ = parallel.merge_topk_device(torch.stack(gi), torch.stack(gd))
    torch.cuda.synchronize()
    want_i = np.zeros((nq, K), np.uint32)
    want_d = np.zeros((nq, K), np.float32)
    for qi in range(nq):
        ci = np.concatenate([oi[0][qi], oi[1][qi]])
        cd = np.concatenate([od[0][qi], od[1][qi]])
        o = np.lexsort((ci, cd))[:K]
        want_i[qi], want_d[qi] = ci[o], cd[o]
    assert np.array_equal(mi_.cpu().numpy().view(np.uint32), want_i)
    assert np.array_equal(md_.cpu().numpy(), want_d)


def _device_count():
    import torch

    return torch.cuda.device_count()


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0], "distinct-2", "distinct-3-ragged"])
def test_c_abi_sharded_search_with_rccl_equals_merge_of_per_shard_cpu_searches(built_lib, devices):
    """BASELINE configs[4] through the C ABI (roar_sharded_*, row E2b): shards placed on device 0 form one RCCL rank --
    ncclCommInitAll + ncclAllGather really run -- and the merged global lists equal the merge of the oracle's per-shard
    searches; shard sizes are ragged, cmps / hops are the sums over the shards.  On a box with more than one GPU the
    "distinct" cases put the shards on different devices (several RCCL ranks, the all-gather crosses xGMI; the ragged one
    gives device 0 two shards and device 1 one: blocks padded to the largest shard count per rank); with one GPU they are
    skipped."""
    import oracle
    from mysteryann_amd import index as mi
    from mysteryann_amd import parallel

    want_ranks = 1
    if isinstance(devices, str):
        if _device_count() < 2:
            pytest.skip("one GPU on this box: the shards of this case need distinct devices")
        devices = [0, 1] if devices == "distinct-2" else [0, 0, 1]
        want_ranks = 2
    S = len(devices)
    n, d, nq, L = 6001, 200, 53, 40
    rng = np.random.default_rng(8)
    base = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    sh = mi.ShardedIndex(d, mi.INNER_PRODUCT, devices)
    assert sh.info("shards") == S and sh.info("ranks") == want_ranks
    oi, od, oc, oh = [], [], 0, 0
    for s in range(S):
        lo, hi = parallel.shard_range(n, S, s)
        m = hi - lo
        srng = np.random.default_rng(50 + s)
        deg = srng.integers(3, 30, m)
        off = np.zeros(m + 1, np.uint64)
        off[1:] = np.cumsum(deg)
        nbrs = srng.integers(0, m, int(off[-1])).astype(np.uint32)
        idx = sh.shard(s)
        idx.SetBaseData(base[lo:hi])
        idx.SetProjectionGraph(1, off, nbrs)
        idx.set_option("visited_mode", 0)  # cmps as the reference counts them
        o = oracle.search_batch(base[lo:hi], off, nbrs, 1, q, K, L, 1)
        oi.append(o["ids"] + np.uint32(lo))
        od.append(o["dists"])
        oc, oh = oc + o["cmps"].astype(np.int64), oh + o["hops"].astype(np.int64)
    for _ in range(2):  # the second call reuses communicators and buffers
        r = sh.SearchRoarGraphBatch(q, K, L)
    want_i = np.zeros((nq, K), np.uint32)
    want_d = np.zeros((nq, K), np.float32)
    for qi in range(nq):
        ci = np.concatenate([x[qi] for x in oi])
        cd = np.concatenate([x[qi] for x in od])
        o = np.lexsort((ci, cd))[:K]
        want_i[qi], want_d[qi] = ci[o], cd[o]
    assert np.array_equal(r["ids"], want_i)
    assert np.array_equal(r["dists"], want_d)
    assert np.array_equal(r["cmps"], oc) and np.array_equal(r["hops"], oh)
    assert sh.info("last_search_ns") > 0 and sh.info("last_exchange_ns") > 0
    with pytest.raises(mi.RoarError):
        sh.SearchRoarGraphBatch(q, 10, 5)  # L_pq < k, as the harness refuses it (tests/test_search_roargraph.cpp:192-195)
    sh.close()


def test_c_abi_sharded_refuses_bad_placements(built_lib):
    from mysteryann_amd import index as mi

    with pytest.raises(mi.RoarError):
        mi.ShardedIndex(200, mi.INNER_PRODUCT, [0, 4096])  # no such device
    with pytest.raises(mi.RoarError):
        mi.ShardedIndex(200, mi.INNER_PRODUCT, [0] * 33)
    sh = mi.ShardedIndex(16, mi.L2, [0, 0])
    with pytest.raises(mi.RoarError):  # shards not loaded yet
        sh.SearchRoarGraphBatch(np.zeros((2, 16), np.float32), 1, 4)
    sh.close()


def test_mid_scale_built_graph_many_queries(built_lib):
    """200 k points, graph made by the in-HBM builder, 10 000 queries in one launch (more queries
    than resident waves: exercises the dequeue loop), both visited modes, against the oracle."""
    import torch

    import oracle
    from mysteryann_amd import index as mi
    from mysteryann_amd import synth

    n, nq, d = 200_000, 10_000, 200
    base, queries, learn = synth.t2i_torch(n, nq, d, seed=77, n_learn=50_000)
    idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None)
    idx.SetBaseData(base)
    idx.BuildBootstrapGraph(learn=learn, rounds=2)
    ep, off, nbrs = idx.GetProjectionGraph()
    assert len(off) == n + 1 and nbrs.max() < n
    hb, hq = base.cpu().numpy(), queries.cpu().numpy()
    o = oracle.search_batch(hb, off, nbrs, ep, hq, K, 40, 1, threads=8)
    for mode in (1, 0):
        idx.set_option("visited_mode", mode)
        out = idx.SearchRoarGraphBatchDevice(queries, K, 40)
        torch.cuda.synchronize()
        assert int(out["short"].item()) == 0
        assert np.array_equal(out["ids"].cpu().numpy().view(np.uint32), o["ids"]), mode
        assert np.array_equal(out["dists"].cpu().numpy(), o["dists"]), mode
        assert np.array_equal(out["hops"].cpu().numpy().view(np.uint32), o["hops"]), mode
        cm = out["cmps"].cpu().numpy().view(np.uint32)
        assert np.array_equal(cm, o["cmps"]) if mode == 0 else (cm >= o["cmps"]).all()
    # the saved index is the reference's format: reload it through the file path and search again
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        p = os.path.join(tmp, "built.index")
        idx.SaveProjectionGraph(p)
        ep2, off2, nbrs2 = oracle.graph_load(p)
        assert ep2 == ep and np.array_equal(off2, off) and np.array_equal(nbrs2, nbrs)


@pytest.mark.parametrize("knn_mode", [0, 1], ids=["bf16-screen", "f32-mfma"])
@pytest.mark.parametrize("metric,d,k", [(1, 200, 10), (0, 128, 10), (1, 72, 100), (0, 512, 32), (1, 200, 100)])
def test_exact_knn_generator_matches_brute_force(metric, d, k, knn_mode, built_lib):
    """ids of the MFMA brute-force kNN == numpy float64 brute force (where the k-th and
    (k+1)-th distances are not within rounding), distances within 1e-4 relative of the oracle's.  Both forms: every score by
    f32 MFMA (knn_mode 1), and the bf16 MFMA screen with its bound in front of the exact f32 re-rank (knn_mode 0, the default:
    its distances are the reference's own summation order, bit for bit)."""
    import torch

    import oracle
    from mysteryann_amd import index as mi

    n, nq = 50_000, 300
    rng = np.random.default_rng(d + k)
    base = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    if knn_mode == 0:  # what the bound has to survive: a few rows 30 x longer than the others (it is per row), a zero row, a zero query
        base[:5] *= 30.0
        base[7] = 0.0
        if metric == 0:  # (inner product: every row ties at 0 for a zero query -- no top k to find, the lists overflow in either form)
            q[3] = 0.0
    idx = mi.IndexBipartite(d, n, metric, None)
    idx.SetBaseData(base)
    idx.set_option("knn_mode", knn_mode)
    ids, dists = idx.ExactKnnDevice(torch.from_numpy(q).cuda(), k)
    torch.cuda.synchronize()
    assert idx.info("last_knn_bf16") == (1 if knn_mode == 0 else 0)
    if knn_mode == 0:
        rr = idx.info("last_knn_rerank_rows") / nq
        assert k <= rr <= 8192, rr  # the re-rank sees at least the top k, and a small multiple of it, not the base set
        for qi in (0, 17, nq - 1):  # distances are DistanceInnerProduct / DistanceL2 of the returned ids, bit for bit
            ref = np.array([oracle.dist(metric, base[i], q[qi]) for i in ids.cpu().numpy().view(np.uint32)[qi]], np.float32)
            assert np.array_equal(dists.cpu().numpy()[qi], ref)
    ids = ids.cpu().numpy().view(np.uint32)
    dists = dists.cpu().numpy()
    b64, q64 = base.astype(np.float64), q.astype(np.float64)
    dot = q64 @ b64.T
    full = -dot if metric == 1 else (q64 ** 2).sum(1)[:, None] - 2 * dot + (b64 ** 2).sum(1)[None, :]
    order = np.argsort(full, axis=1, kind="stable")[:, :k + 1]
    want = order[:, :k]
    scale = np.abs(np.take_along_axis(full, want, 1)).max(axis=1)
    gap = np.take_along_axis(full, order[:, k:k + 1], 1)[:, 0] - np.take_along_axis(full, order[:, k - 1:k], 1)[:, 0]
    # queries whose top-k set is unambiguous in f32 (rounding of a distance scales with |q||x| of the rows AROUND the k-th: the few
    # long rows of the bf16 case sit at the top of an inner-product list and would otherwise set the yardstick for everybody)
    kth = np.abs(np.take_along_axis(full, order[:, k - 1:k], 1)[:, 0]) + (q64 ** 2).sum(1) ** 0.5 * np.median((b64 ** 2).sum(1) ** 0.5)
    clear = gap > 1e-5 * np.minimum(scale, kth)
    assert clear.mean() > 0.9
    got_sets = np.sort(ids[clear], axis=1)
    assert np.array_equal(got_sets, np.sort(want[clear], axis=1))
    # distances: sorted ascending, and equal to the reference-order distance of the returned id
    assert (np.diff(dists, axis=1) >= 0).all()
    for qi in (0, 17, nq - 1):
        ref = np.array([oracle.dist(metric, base[i], q[qi]) for i in ids[qi]], np.float32)
        assert np.allclose(dists[qi], ref, rtol=1e-4, atol=1e-4 * scale[qi])


def test_exact_knn_query_block_larger_than_one_launch(built_lib):
    """65 536 queries x 4.3 M rows: (query tiles x base tiles x 256 threads) exceeds the 2^32 - 1 threads one
    launch may have, so a pass is split into runs of base tiles.  The big call must equal 8 192-query calls
    and a torch top-k on a sample."""
    import torch

    from mysteryann_amd import index as mi

    n, d, nq, k = 4_300_000, 64, 65_536, 10
    g = torch.Generator(device="cuda").manual_seed(3)
    base = torch.randn((n, d), device="cuda", generator=g)
    q = torch.randn((nq, d), device="cuda", generator=g)
    idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None)
    idx.SetBaseData(base)
    big, _ = idx.ExactKnnDevice(q, k, want_dists=False)
    small = torch.cat([idx.ExactKnnDevice(q[a:a + 8192], k, want_dists=False)[0] for a in range(0, nq, 8192)])
    torch.cuda.synchronize()
    assert bool((big == small).all())
    ref = torch.topk(q[-512:] @ base.T, k, dim=1).indices
    same = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(ref.cpu(), big[-512:].cpu())) / (512 * k)
    assert same > 0.999
    idx.close()


def test_auto_visited_mode_switches_with_pool_length(golden_dir, built_lib):
    """Default policy without a measurement (option calibrate = 0, plan_search): the on-chip tag table below L_pq = 1536, HBM epoch
    tags beyond; a dense graph (average degree x row bytes >= 24 x 800) takes the hub-bitmap form from L_pq = 100.  With the
    calibration (default) the visited form and the screen are whatever was measured fastest, and the hub rule stands for what it is: a
    rule about launches WITHOUT the screen (with it no graph counts as dense).  All give the reference's results."""
    import oracle
    from mysteryann_amd import index as mi

    n, d, nq = 5000, 128, 20
    for maxdeg, plan in ((80, ((0, ((50, 1, 0), (99, 1, 0), (100, 1, 1), (1535, 1, 1), (1536, 0, 1))),
                               (1, ((50, 1, 0), (1535, 1, 0), (1536, 0, 0))))),      # average degree ~ 39: dense
                         (30, ((0, ((50, 1, 0), (300, 1, 0), (1535, 1, 0), (1536, 0, 0))),),)):  # ~ 14: sparse
        base, q, off, nbrs = _random_case(n, d, nq, 0, seed=21, maxdeg=maxdeg)
        idx = mi.IndexBipartite(d, n, 0, None)
        idx.SetBaseData(base)
        idx.SetProjectionGraph(9, off, nbrs)
        assert idx.info("visited_mode") == 2
        for calibrate in (0, 1):
            idx.set_option("calibrate", calibrate)
            for hub_mode, cases in plan:
                idx.set_option("hub_mode", hub_mode)
                for L, want_mode, want_hub in cases:
                    r = idx.SearchRoarGraphBatch(q, K, L)
                    assert idx.info("last_hub_bitmap") == (0 if idx.info("last_screen") else want_hub), (maxdeg, hub_mode, L)
                    assert idx.info("last_plan_calibrated") == calibrate
                    if not calibrate:
                        assert idx.info("last_screen") == 0  # (rows under 768 bytes: off by default)
                    if not calibrate:
                        assert idx.info("last_visited_mode") == want_mode, (maxdeg, hub_mode, L)
                    o = oracle.search_batch(base, off, nbrs, 9, q, K, L, 0)
                    assert np.array_equal(r["ids"], o["ids"]) and np.array_equal(r["dists"], o["dists"])
                    assert np.array_equal(r["hops"], o["hops"])
        idx.close()


def test_launch_plan_calibration_is_cached_per_band_and_follows_the_options(built_lib):
    """Option "calibrate" (default on): the first search of a (L_pq band, batch band) measures the candidate plans on its own
    queries and keeps the winner; the next one of the same band reuses it; another band, or an option that changes the kernel
    form, measures again; options that fix screen / visited form / table size are obeyed; results are the reference's throughout."""
    import oracle
    from mysteryann_amd import index as mi

    n, d, nq = 20000, 200, 300
    base, q, off, nbrs = _random_case(n, d, nq, 1, seed=33, maxdeg=50)
    idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(7, off, nbrs)
    idx.set_option("hub_mode", 1)
    o = {L: oracle.search_batch(base, off, nbrs, 7, q, K, L, 1) for L in (40, 45, 600)}

    def check(r, L):
        for key in ("ids", "dists", "hops"):
            assert np.array_equal(r[key], o[L][key]), (L, key)

    assert idx.info("calibrations") == 0
    check(idx.SearchRoarGraphBatch(q, K, 40), 40)
    assert idx.info("calibrations") == 1 and idx.info("last_plan_calibrated") == 1
    picked = (idx.info("last_screen"), idx.info("last_visited_mode"), idx.info("last_vis_entries"))
    check(idx.SearchRoarGraphBatch(q, K, 45), 45)  # the same half-octave of L_pq, the same batch band: nothing new is measured
    assert idx.info("calibrations") == 1
    assert picked == (idx.info("last_screen"), idx.info("last_visited_mode"), idx.info("last_vis_entries"))
    check(idx.SearchRoarGraphBatch(q, K, 600), 600)  # another band (and long enough for the HBM tags to be a candidate)
    assert idx.info("calibrations") == 2
    r8 = idx.SearchRoarGraphBatch(q[:8], K, 40)  # a batch of another size band
    assert idx.info("calibrations") == 3
    for key in ("ids", "dists", "hops"):
        assert np.array_equal(r8[key], o[40][key][:8])
    # fixed by option: obeyed, and taken out of the search
    idx.set_option("screen", 1)
    idx.set_option("visited_mode", 0)
    check(idx.SearchRoarGraphBatch(q, K, 40), 40)
    assert idx.info("last_screen") == 0 and idx.info("last_visited_mode") == 0 and idx.info("last_plan_calibrated") == 0
    idx.set_option("screen", 2)
    idx.set_option("visited_mode", 3)
    idx.set_option("vis_entries", 1024)
    check(idx.SearchRoarGraphBatch(q, K, 40), 40)
    assert idx.info("last_screen") == 1 and idx.info("last_vis_entries") == 1024 and idx.info("last_visited_forgetful") == 1
    idx.set_option("screen", 0), idx.set_option("visited_mode", 2), idx.set_option("vis_entries", 0)
    idx.set_option("calibrate", 2)  # forget
    assert idx.info("calibrations") == 0
    idx.set_option("calibrate", 0)
    check(idx.SearchRoarGraphBatch(q, K, 40), 40)
    assert idx.info("calibrations") == 0 and idx.info("last_plan_calibrated") == 0 and idx.info("last_screen") == 1
    idx.set_option("calibrate", 1)
    # a new graph forgets what was measured on the old one
    check(idx.SearchRoarGraphBatch(q, K, 40), 40)
    assert idx.info("calibrations") == 1
    idx.SetProjectionGraph(7, off, nbrs)
    assert idx.info("calibrations") == 0
    # the pageable host path (its rows are packed behind the launch) still calibrates: rows are then packed in front
    check(idx.SearchRoarGraphBatch(np.ascontiguousarray(q), K, 40), 40)
    assert idx.info("calibrations") == 1
    idx.close()


def test_auto_plan_on_a_graph_nobody_tuned_for(built_lib):
    """A random digraph of average degree 64 over 96-d rows (no compile-time row length, no screening kernel; nothing like the
    two graphs rounds 1-3 tuned their plan table on): the calibrated AUTO plan must be within 5 % of the best explicit setting of
    visited form x table size, at a short and at a long pool; results equal the oracle's."""
    import torch

    import oracle
    from mysteryann_amd import index as mi

    n, d, nq = 200_000, 96, 6144
    base, q, off, nbrs = _random_case(n, d, nq, 0, seed=64, maxdeg=128)
    idx = mi.IndexBipartite(d, n, 0, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(17, off, nbrs)
    idx.set_option("hub_mode", 1)
    qd = torch_pad(q, idx.info("dim_padded"))

    def best_ms(L, reps=4):
        best = 1e9
        for _ in range(reps):
            out = idx.SearchRoarGraphBatchDevice(qd, K, L)
            torch.cuda.synchronize()
            best = min(best, idx.info("last_kernel_ns") / 1e6)
        return best, out

    for L in (60, 700):
        times = {}
        idx.set_option("calibrate", 0)
        for mode, ve in ((3, 2048), (3, 4096), (3, 8192), (3, 16384), (0, 0)):
            idx.set_option("visited_mode", mode)
            idx.set_option("vis_entries", ve)
            times[(mode, ve)], _ = best_ms(L)
        idx.set_option("visited_mode", 2)
        idx.set_option("vis_entries", 0)
        idx.set_option("calibrate", 1)
        t_auto, out = best_ms(L)
        assert idx.info("last_plan_calibrated") == 1
        assert t_auto <= 1.05 * min(times.values()) + 0.01, (L, times, t_auto, idx.info("last_visited_mode"), idx.info("last_vis_entries"))
        o = oracle.search_batch(base, off, nbrs, 17, q[:256], K, L, 0, threads=8)
        for key in ("ids", "dists", "hops"):
            assert np.array_equal(out[key][:256].cpu().numpy().view(o[key].dtype), o[key]), (L, key)
    idx.close()


def test_a_cached_plan_keeps_its_residency_at_a_longer_pool_of_its_band(built_lib):
    """Round 6: beyond L_pq = 512 a calibration serves a band of 160 pool lengths (one 1 280-byte LDS block of pool); applied at a
    longer pool of its band the measured table must not cost a resident workgroup per CU -- it steps down (3/4, 5/8, 1/2 ... of itself)
    until the plan keeps what the measurement had, or the table is at its smallest.  Results equal the oracle's at both lengths."""
    import torch

    import oracle
    from mysteryann_amd import index as mi

    n, d, nq = 60_000, 200, 3200
    base, q, off, nbrs = _random_case(n, d, nq, 1, seed=71, maxdeg=40)
    ep = int(np.argmax(np.diff(off.astype(np.int64)) > 10))  # (an entry point that has neighbours: the random degrees start at 0)
    idx = mi.IndexBipartite(d, n, 1, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(ep, off, nbrs)
    qd = torch.from_numpy(q).cuda()
    seen = {}
    for L in (1601, 1755):  # the first and (nearly) the last pool length of one band: 8 (L + 1) bytes = 10 blocks of 1 280
        out = idx.SearchRoarGraphBatchDevice(qd, K, L)
        torch.cuda.synchronize()
        seen[L] = dict(cal=idx.info("calibrations"), wpc=idx.info("last_workgroups_per_cu"), tags=idx.info("last_vis_entries"),
                       mode=idx.info("last_visited_mode"), lds=idx.info("last_lds_bytes"))
        o = oracle.search_batch(base, off, nbrs, ep, q[:64], K, L, 1, threads=8)
        assert int(o["hops"].min()) > L // 2  # (a real search: the pool fills)
        for key in ("ids", "dists", "hops"):
            assert np.array_equal(out[key][:64].cpu().numpy().view(o[key].dtype), o[key]), (L, key)
    a, b = seen[1601], seen[1755]
    assert a["cal"] == 1 and b["cal"] == 1, seen  # one band, one calibration
    if a["mode"] == 1 and a["tags"] > 1024:  # (an on-chip table was picked: the rule applies)
        assert b["wpc"] >= a["wpc"] or b["tags"] <= 1024, seen
        assert b["tags"] <= a["tags"], seen
    idx.close()


def test_auto_screen_decision_follows_the_measured_times(built_lib):
    """Where the 8-bit screen pays is a property of the data -- how many of a hop's neighbours its bound can drop:
      * isotropic rows (t2i-hard, offset 0): scores spread over many slacks of the bound, most rows are dropped;
      * a narrow cone (every row = a common component 30 x the isotropic part + that part): a bound in units of |q||x| would pass
        nearly every row -- the codes are centred on the mean of the rows (round 4), and the pass rate is that of the isotropic part;
      * tight clusters (64 centres, rows within 0.2 % of their centre): a node's neighbours are its cluster, their scores differ by
        less than the bound's slack, the screen passes everything on -- a second dependent round trip for nothing.
    AUTO does not guess from the row length any more: it times both forms on the caller's own queries and takes the faster one --
    checked here against this test's own timing of the two forced settings.  Results equal the oracle's in every setting."""
    import torch

    import oracle
    from mysteryann_amd import index as mi
    from mysteryann_amd import synth

    n, nq, d, L = 400_000, 8192, 200, 50

    def clustered(seed):
        g = torch.Generator(device="cuda").manual_seed(seed)
        centres = torch.randn((64, d), generator=g, device="cuda") * 0.08

        def draw(m, spread):
            c = centres[torch.randint(0, 64, (m,), generator=g, device="cuda")]
            return (c + spread * 0.08 * torch.randn((m, d), generator=g, device="cuda")).contiguous()

        return draw(n, 0.002), draw(nq, 0.02), draw(40_000, 0.02)

    rates = {}
    for name in ("isotropic", "narrow cone", "tight clusters"):
        if name == "tight clusters":
            base, q, learn = clustered(13)
        else:
            base, q, learn = synth.t2i_hard_torch(n, nq, d, seed=11, n_learn=40_000, offset=30.0 if name == "narrow cone" else 0.0)
        idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None)
        idx.SetBaseData(base)
        lists, _ = idx.ExactKnnDevice(learn, 100, want_dists=False)
        prm = mi.Parameters()
        prm.Set("M_sq", 100), prm.Set("M_pjbp", 35), prm.Set("L_pjpq", 500)
        idx.BuildRoarGraphDevice(lists, prm)
        ep, off, nbrs = idx.GetProjectionGraph()
        blk = torch.zeros((nq, 4), dtype=torch.int32, device="cuda")
        idx.set_option("screen", 2)
        idx.SetTrafficCounters(blk)
        forced = idx.SearchRoarGraphBatchDevice(q, K, L)
        torch.cuda.synchronize()
        idx.SetTrafficCounters(None)
        t = blk.cpu().numpy().astype(np.int64).sum(axis=0)
        cm = int(forced["cmps"].cpu().numpy().astype(np.int64).sum())
        rates[name] = (t[1] - nq - (cm - t[0])) / max(t[0], 1)  # rows the screen let through / rows it scored

        def best_ms(reps=5):
            best = 1e9
            for _ in range(reps):
                out = idx.SearchRoarGraphBatchDevice(q, K, L)
                torch.cuda.synchronize()
                best = min(best, idx.info("last_kernel_ns") / 1e6)
            return best, out

        times = {}
        for screen in (1, 2):  # forced off / on, each with its own measured table size
            idx.set_option("screen", screen)
            idx.set_option("calibrate", 2)
            times[screen], _ = best_ms()
        idx.set_option("screen", 0)
        idx.set_option("calibrate", 2)
        t_auto, auto = best_ms()
        assert idx.info("last_plan_calibrated") == 1
        picked = 2 if idx.info("last_screen") else 1
        assert t_auto <= 1.06 * min(times.values()) + 0.01, (name, times, t_auto)
        assert times[picked] <= 1.06 * min(times.values()) + 0.01, (name, times, picked)
        o = oracle.search_batch(base.cpu().numpy(), off, nbrs, ep, q[:512].cpu().numpy(), K, L, 1, threads=8)
        for key in ("ids", "dists", "hops"):
            assert np.array_equal(auto[key][:512].cpu().numpy().view(o[key].dtype), o[key]), (name, key)
            assert np.array_equal(forced[key][:512].cpu().numpy().view(o[key].dtype), o[key]), (name, key)
        idx.close()
    assert rates["isotropic"] < 0.3 and rates["narrow cone"] < 0.3 and rates["tight clusters"] > 0.9, rates


@pytest.mark.parametrize("mode", [0, 1], ids=["exact", "lds"])
def test_latency_kernel_four_waves_per_query(mode, golden_dir, built_lib):
    """search_kernel = 2 (four waves per query) must return exactly what the throughput form and the
    reference return: goldens, plus a random graph with degrees up to 200 (several 128-candidate
    rounds per hop) and L_pq up to 700."""
    import oracle
    from mysteryann_amd import index as mi

    for case in ("ip_200", "l2_128"):
        exp = np.load(os.path.join(golden_dir, case + ".expected.npz"))
        idx, q = _index(case, golden_dir, mode)
        for kernel, waves in ((2, 4), (3, 2)):
            idx.set_option("search_kernel", kernel)
            for L in LS:
                r = idx.SearchRoarGraphBatch(q, K, L)
                assert idx.info("last_waves_per_query") == waves
                for key in ("ids", "dists", "hops", "scanned", "cmps"):
                    assert np.array_equal(r[key], exp[f"{key}_{L}"]), (case, kernel, L, key)
        idx.close()
    n, d, nq = 5000, 200, 24
    base, q, off, nbrs = _random_case(n, d, nq, 1, seed=31, maxdeg=200)
    idx = mi.IndexBipartite(d, n, 1, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(2, off, nbrs)
    idx.set_option("visited_mode", mode)
    for kernel in (2, 3, 1, 0):
        idx.set_option("search_kernel", kernel)
        for L in (10, 130, 700):
            o = oracle.search_batch(base, off, nbrs, 2, q, K, L, 1)
            r = idx.SearchRoarGraphBatch(q, K, L)
            assert np.array_equal(r["ids"], o["ids"]) and np.array_equal(r["dists"], o["dists"]), (kernel, L)
            assert np.array_equal(r["hops"], o["hops"]), (kernel, L)
            if mode == 0:
                assert np.array_equal(r["cmps"], o["cmps"]), (kernel, L)
    idx.set_option("search_kernel", 0)
    idx.set_option("screen", 1)
    idx.SearchRoarGraphBatch(q, K, 50)
    assert idx.info("last_waves_per_query") == 4  # at most one query per CU on a dense graph: the four-wave form
    idx.set_option("screen", 0)
    idx.SearchRoarGraphBatch(q, K, 50)
    # (with the 8-bit screen few rows are left to score: one wave; AUTO takes the screen or not by what it measured)
    assert idx.info("last_waves_per_query") == (1 if idx.info("last_screen") else 4)


@pytest.mark.parametrize("mode", [0, 1], ids=["exact", "lds"])
@pytest.mark.parametrize("pool_form", [0, 1], ids=["regpool", "ldspool"])
def test_every_kernel_form_matches_reference_golden(mode, pool_form, golden_dir, built_lib):
    """One / two / four waves per query x candidate pool in registers (L_pq <= 64) or in LDS x visited mode: all of
    them are the same algorithm and must reproduce the reference's goldens bit for bit (ids, dists, hops, scanned,
    cmps).  pool_form = 1 forces the LDS pool below L_pq = 64 as well."""
    for case in ("ip_200", "l2_128"):
        exp = np.load(os.path.join(golden_dir, case + ".expected.npz"))
        idx, q = _index(case, golden_dir, mode)
        idx.set_option("pool_form", pool_form)
        for kernel, waves in ((1, 1), (3, 2), (2, 4)):
            idx.set_option("search_kernel", kernel)
            for L in LS:
                r = idx.SearchRoarGraphBatch(q, K, L)
                assert idx.info("last_waves_per_query") == waves
                assert idx.info("last_register_pool") == (1 if (L <= 64 and pool_form == 0) else 0)
                for key in ("ids", "dists", "hops", "scanned", "cmps"):
                    assert np.array_equal(r[key], exp[f"{key}_{L}"]), (case, kernel, L, key)
        idx.close()


@pytest.mark.parametrize("mode", [0, 1], ids=["exact", "lds"])
def test_register_pool_boundaries_random_graphs(mode, built_lib):
    """The register-pool form at its edges: L_pq = k = 1, L_pq = 64 (every lane holds an entry), k = L_pq, first hops
    with more than 64 survivors per merge (degrees up to 200), self loops and repeated neighbours, against the oracle."""
    import oracle
    from mysteryann_amd import index as mi

    n, d, nq = 3000, 200, 40
    base, q, off, nbrs = _random_case(n, d, nq, 1, seed=77, maxdeg=200)
    idx = mi.IndexBipartite(d, n, 1, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(5, off, nbrs)
    idx.set_option("visited_mode", mode)
    for kernel in (1, 3, 2):
        idx.set_option("search_kernel", kernel)
        for k, L in ((1, 1), (1, 2), (10, 10), (10, 33), (64, 64), (10, 64), (10, 65)):
            o = oracle.search_batch(base, off, nbrs, 5, q, k, L, 1)
            r = idx.SearchRoarGraphBatch(q, k, L)
            assert np.array_equal(r["ids"], o["ids"]) and np.array_equal(r["dists"], o["dists"]), (kernel, k, L)
            assert np.array_equal(r["hops"], o["hops"]), (kernel, k, L)  # (scanned differs: repeats inside a row are dropped on upload)
            if mode == 0:
                assert np.array_equal(r["cmps"], o["cmps"]), (kernel, k, L)
    idx.close()


@pytest.mark.parametrize("mode", [0, 1], ids=["exact", "lds"])
def test_hub_bitmap_form_matches_goldens_and_oracle(mode, golden_dir, built_lib):
    """Long-pool form: the highest-in-degree nodes keep their visited bits in an exact on-chip bitmap (a hub-annotated
    copy of the adjacency lists them first, with their slots).  Goldens (2 000 points: every node is a hub) in all wave
    forms, and a 70 000-node random graph where only some neighbours are hubs: ids, distances, hops -- and in EXACT mode
    cmps -- equal the reference's / the oracle's, hub form on or off."""
    import oracle
    from mysteryann_amd import index as mi

    for case in ("ip_200", "l2_128"):
        exp = np.load(os.path.join(golden_dir, case + ".expected.npz"))
        idx, q = _index(case, golden_dir, mode)
        idx.set_option("hub_mode", 2)
        for kernel in (1, 3, 2):
            idx.set_option("search_kernel", kernel)
            for L in (100, 200):
                r = idx.SearchRoarGraphBatch(q, K, L)
                assert idx.info("last_hub_bitmap") == 1
                for key in ("ids", "dists", "hops", "scanned", "cmps"):
                    assert np.array_equal(r[key], exp[f"{key}_{L}"]), (case, kernel, L, key)
        idx.close()
    n, d, nq = 70_000, 16, 32
    base, q, off, nbrs = _random_case(n, d, nq, 1, seed=5, maxdeg=60)
    nbrs[: len(nbrs) // 2] = nbrs[: len(nbrs) // 2] % 3000  # a few thousand nodes collect half of the in-edges
    idx = mi.IndexBipartite(d, n, 1, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(11, off, nbrs)
    idx.set_option("visited_mode", mode)
    for L in (70, 130, 600):
        o = oracle.search_batch(base, off, nbrs, 11, q, K, L, 1)
        for hub in (2, 1):
            idx.set_option("hub_mode", hub)
            r = idx.SearchRoarGraphBatch(q, K, L)
            assert idx.info("last_hub_bitmap") == (1 if hub == 2 else 0)
            assert np.array_equal(r["ids"], o["ids"]) and np.array_equal(r["dists"], o["dists"]), (L, hub)
            assert np.array_equal(r["hops"], o["hops"]), (L, hub)
            if mode == 0:
                assert np.array_equal(r["cmps"], o["cmps"]), (L, hub)
    idx.close()


@pytest.mark.parametrize("form", [0, 1], ids=["ldspool", "regpool"])
def test_queue_routines_replay_reference_trace(form, golden_dir, built_lib):
    """A4 on the device: the scripted NeighborPriorityQueue sequence of tests/golden/pq_trace.npz (outputs of the
    reference's neighbor.h: fill, exact duplicate, equal distance / different id, eviction from a full pool, insert
    below the cursor, pops until empty, 200 random ops with ties) replayed by the kernels' own insert / select
    routines -- size, cursor, ids, distances and flags equal after every op."""
    import ctypes as C

    from mysteryann_amd import capi

    t = np.load(os.path.join(golden_dir, "pq_trace.npz"))
    cap = int(t["capacity"])
    ops = np.ascontiguousarray(t["ops"], np.uint8)
    ids = np.ascontiguousarray(t["ids"], np.uint32)
    dists = np.ascontiguousarray(t["dists"], np.float32)
    snap = np.zeros((len(ops), 2 + 3 * cap), np.uint32)
    capi.check(capi.lib().roar_queue_trace(0, cap, len(ops), ops.ctypes.data_as(C.POINTER(C.c_uint8)), capi.u32p(ids),
                                           capi.f32p(dists), form, capi.u32p(snap)))
    bad = np.nonzero((snap != t["snap"]).any(axis=1))[0]
    assert bad.size == 0, (int(bad[0]), snap[bad[0]].tolist(), t["snap"][bad[0]].tolist())


@pytest.mark.parametrize("g", [2, 8, "all-devices"])
def test_one_batch_split_over_g_replicas_equals_single_launch(g, golden_dir, built_lib):
    """BASELINE configs[3] host logic (roar_clone_to_device + roar_search_batch_multi): the index replicated g times
    (all replicas on this box's one device; the "all-devices" case, skipped on a one-GPU box, puts replica r on device r:
    hipMemcpyPeer between real devices, one host thread + stream per device), ONE query block split by query, every replica
    writing its own rows of the caller's arrays.  The result must be the single-launch result, row for row, whatever g is --
    including a block that does not divide by g and one with fewer queries than replicas."""
    from mysteryann_amd import index as mi

    exp = np.load(os.path.join(golden_dir, "ip_200.expected.npz"))
    idx, q = _index("ip_200", golden_dir, 2)
    # (cmps is compared below between the index and its replicas: every replica is an index of its own that would calibrate its own plan
    # on its own dozen queries -- a timing, so a matter of the run -- and a smaller table forgets more; the fixed default plan on all of them)
    idx.set_option("calibrate", 0)
    if g == "all-devices":
        g = _device_count()
        if g < 2:
            pytest.skip("one GPU on this box")
        reps = [idx] + [idx.Clone(r) for r in range(1, g)]
    else:
        reps = [idx] + [idx.Clone(0) for _ in range(g - 1)]
    assert all(r.info("n") == idx.info("n") and r.info("nedges") == idx.info("nedges") for r in reps)
    for L in (10, 50, 200):
        one = idx.SearchRoarGraphBatch(q, K, L)
        many = mi.IndexBipartite.SearchRoarGraphBatchMulti(reps, q, K, L)
        for key in ("ids", "dists", "hops", "cmps", "scanned"):
            assert np.array_equal(one[key], many[key]), (g, L, key)
        assert np.array_equal(many["ids"], exp[f"ids_{L}"]) and np.array_equal(many["dists"], exp[f"dists_{L}"])
    odd = q[: len(q) - 3]  # slices of unequal length
    a = idx.SearchRoarGraphBatch(odd, K, 30)
    b = mi.IndexBipartite.SearchRoarGraphBatchMulti(reps, odd, K, 30)
    assert np.array_equal(a["ids"], b["ids"]) and np.array_equal(a["dists"], b["dists"])
    few = q[:3]  # fewer queries than replicas: some slices are empty
    a = idx.SearchRoarGraphBatch(few, K, 30)
    b = mi.IndexBipartite.SearchRoarGraphBatchMulti(reps, few, K, 30)
    assert np.array_equal(a["ids"], b["ids"]) and np.array_equal(a["hops"], b["hops"])
    # a replica keeps the graph: its saved file equals the original's, byte for byte
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        reps[-1].SaveProjectionGraph(os.path.join(td, "r.index"))
        assert open(os.path.join(td, "r.index"), "rb").read() == open(os.path.join(golden_dir, "ip_200.roar.index"), "rb").read()
    for r in reps[1:]:
        r.close()
    idx.close()


def test_multi_reports_short_results_and_mismatched_replicas(built_lib):
    from mysteryann_amd import index as mi

    n, d = 64, 16
    rng = np.random.default_rng(3)
    base = rng.standard_normal((n, d)).astype(np.float32)
    off = np.zeros(n + 1, np.uint64)  # no edges at all: every query ends with one result
    idx = mi.IndexBipartite(d, n, 0, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(0, off, np.zeros(0, np.uint32))
    rep = idx.Clone(0)
    with pytest.raises(mi.RoarError, match="not enough results: 1, expected: 5"):
        mi.IndexBipartite.SearchRoarGraphBatchMulti([idx, rep], base[:10], 5, 8)
    other = mi.IndexBipartite(d, n - 1, 0, None)
    other.SetBaseData(base[: n - 1])
    with pytest.raises(mi.RoarError, match="does not hold the same index"):
        mi.IndexBipartite.SearchRoarGraphBatchMulti([idx, other], base[:10], 1, 8)
    for x in (idx, rep, other):
        x.close()


def test_pinned_query_block_is_read_in_place(golden_dir, built_lib):
    """roar_host_alloc memory (what the facade's load_data / data_align return) is read by the kernel where it lies;
    results equal those of an ordinary pageable block, and rows strided wider than the dimension work too."""
    import ctypes as C

    from mysteryann_amd import capi

    exp = np.load(os.path.join(golden_dir, "ip_200.expected.npz"))
    idx, q = _index("ip_200", golden_dir, 2)
    lib = capi.lib()
    nq, d = q.shape
    for stride in (d, d + 8):
        ptr = lib.roar_host_alloc(nq * stride * 4)
        assert ptr
        blk = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(nq, stride))
        blk[:] = 7.5  # columns past the dimension must not matter
        blk[:, :d] = q
        ids = np.zeros((nq, K), np.uint32)
        dists = np.zeros((nq, K), np.float32)
        hops = np.zeros(nq, np.uint32)
        capi.check(lib.roar_search_batch(idx._h, C.cast(ptr, C.POINTER(C.c_float)), nq, stride, K, 50, capi.u32p(ids),
                                         capi.f32p(dists), None, capi.u32p(hops), None))
        assert np.array_equal(ids, exp["ids_50"]) and np.array_equal(dists, exp["dists_50"]) and np.array_equal(hops, exp["hops_50"])
        lib.roar_host_free(ptr)
    idx.close()


def test_registered_query_block_is_read_in_place(golden_dir, built_lib):
    """roar_host_register (what the facade's opt-in pin_block calls) on an ordinary malloc'ed block: same results, and the
    block is released by plain free() after roar_host_unregister -- the reference's harness releases its query block with
    delete[] (tests/test_search_roargraph.cpp:246), so the loaders must hand out malloc-compatible memory."""
    import ctypes as C

    from mysteryann_amd import capi

    exp = np.load(os.path.join(golden_dir, "ip_200.expected.npz"))
    idx, q = _index("ip_200", golden_dir, 2)
    lib = capi.lib()
    libc = C.CDLL(None)
    libc.aligned_alloc.restype = C.c_void_p
    libc.aligned_alloc.argtypes = [C.c_size_t, C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    nq, d = q.shape
    nbytes = (nq * d * 4 + 4095) // 4096 * 4096
    ptr = libc.aligned_alloc(4096, nbytes)
    blk = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(nq, d))
    blk[:] = q
    capi.check(lib.roar_host_register(ptr, nbytes))
    ids = np.zeros((nq, K), np.uint32)
    dists = np.zeros((nq, K), np.float32)
    capi.check(lib.roar_search_batch(idx._h, C.cast(ptr, C.POINTER(C.c_float)), nq, d, K, 50, capi.u32p(ids), capi.f32p(dists),
                                     None, None, None))
    assert np.array_equal(ids, exp["ids_50"]) and np.array_equal(dists, exp["dists_50"])
    capi.check(lib.roar_host_unregister(ptr))
    libc.free(ptr)
    idx.close()


def test_pageable_host_path_survives_a_blocking_launch(golden_dir):
    """The pageable host-buffer call launches a kernel that waits for rows the host packs behind the launch.  With
    HIP_LAUNCH_BLOCKING=1 the launch call only returns when the kernel has finished, so the packing must not depend on
    it returning (round 2 packed on the calling thread after the launch: a hang).  Run in a child process with a timeout."""
    import subprocess
    import sys

    code = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from mysteryann_amd import index as mi, fileio\n"
        "g = %r\n"
        "exp = np.load(os.path.join(g, 'ip_200.expected.npz'))\n"
        "q = fileio.read_fbin(os.path.join(g, 'ip_200.query.fbin'))\n"
        "idx = mi.IndexBipartite(200, 0, mi.INNER_PRODUCT, None)\n"
        "idx.LoadSearchNeededData(os.path.join(g, 'ip_200.base.fbin'))\n"
        "idx.LoadProjectionGraph(os.path.join(g, 'ip_200.roar.index'))\n"
        "r = idx.SearchRoarGraphBatch(q, 10, 50)\n"
        "assert np.array_equal(r['ids'], exp['ids_50']) and np.array_equal(r['dists'], exp['dists_50'])\n"
        "print('ok')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), golden_dir)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, HIP_LAUNCH_BLOCKING="1"))
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_device_builder_matches_the_survey_probe_of_the_compiled_reference(golden_dir, built_lib):
    """roar_build_roargraph_device on the survey's probe recipe (50 000 points, canonical parameters) against BOTH the
    figures of the compiled reference builder (the only real-reference builder numbers there are)
    and the restated builder's (tests/golden/builder_probe_50000.json): average degree within 5 %, longest row 70, recall@10
    within 0.02 of the survey's at L_pq = 20 and >= 0.99 at 100, hops within 3 %, distance evaluations within 10 %."""
    import json

    import torch

    import oracle
    from mysteryann_amd import index as mi, synth

    Y = json.load(open(os.path.join(golden_dir, "builder_probe_50000.json")))
    S = Y["survey_probe_of_the_compiled_reference"]
    n, nq = Y["n"], Y["nq"]
    base, q, learn = synth.t2i_numpy(n, nq, 200, seed=Y["seed"], n_learn=Y["n_learn"])
    idx = mi.IndexBipartite(200, n, mi.INNER_PRODUCT, None)
    idx.SetBaseData(base)
    lists, _ = idx.ExactKnnDevice(torch.from_numpy(learn).cuda(), 100, want_dists=False)
    p = mi.Parameters()
    p.Set("M_sq", 100), p.Set("M_pjbp", 35), p.Set("L_pjpq", 500)
    idx.BuildRoarGraphDevice(lists, p)
    assert idx.info("ep") == Y["ep"]
    avg = idx.info("nedges") / n
    assert abs(avg - S["avg_degree"]) <= 0.05 * S["avg_degree"] and abs(avg - Y["avg_degree"]) <= 0.03 * Y["avg_degree"], avg
    assert idx.info("max_degree") == S["max_degree"]
    gt = idx.ExactKnnDevice(torch.from_numpy(q).cuda(), 10, want_dists=False)[0].cpu().numpy().view(np.uint32)
    idx.set_option("visited_mode", 0)
    for L in (10, 20, 50, 100, 200, 500):
        r = idx.SearchRoarGraphBatch(q, 10, L)
        rec, hops, cmps = oracle.recall(r["ids"], gt, 10), float(r["hops"].mean()), float(r["cmps"].mean())
        assert abs(hops - S["avg_hops"][str(L)]) <= 0.03 * S["avg_hops"][str(L)], (L, hops)
        assert abs(cmps - S["avg_cmps"][str(L)]) <= 0.10 * S["avg_cmps"][str(L)], (L, cmps)
        assert abs(cmps - Y["curve"][str(L)]["avg_cmps"]) <= 0.05 * Y["curve"][str(L)]["avg_cmps"], (L, cmps)
        if str(L) in S["recall_at_10"]:
            assert abs(rec - S["recall_at_10"][str(L)]) <= 0.02, (L, rec)
        assert abs(rec - Y["curve"][str(L)]["recall_at_10"]) <= 0.012, (L, rec)
    idx.close()


@pytest.mark.parametrize("shape", ["ip-200", "ip-512", "ip-128", "l2-128", "l2-512"])
@pytest.mark.parametrize("mode", [0, 1, 3], ids=["exact", "lds", "forgetful"])
def test_screening_pass_never_changes_a_result(mode, shape, built_lib):
    """The 8-bit screening copy (inner product: 128-, 200- and 512-d rows; L2: 128- and 512-d rows; option "screen") may only drop
    neighbours whose exact key is certainly above the reject threshold: ids, distances, hops and cmps with it equal those without it
    and the CPU oracle's -- on rows chosen to strain the bound: a wide dynamic range inside a row (one component 1e4 times the
    others: the codes of the others are all zero), rows of zeros, duplicated rows (exact ties), tiny rows, rows scaled by 1e-20 /
    1e+15, rows of small integers (coded without error), constant rows, a zero query, a query with one dominant component, queries
    that are base rows plus a little noise (L2: the three terms of the screened distance cancel)."""
    import oracle
    from mysteryann_amd import index as mi

    ip, d = shape.startswith("ip"), int(shape.split("-")[1])
    n, nq = 6000, 40
    rng = np.random.default_rng(123)
    base = rng.standard_normal((n, d)).astype(np.float32)
    base[100:400, 7] *= 1e4                      # one dominant component
    base[400:420] = 0.0                          # zero rows
    base[500:600] = base[600:700]                # duplicates: equal distances, ids break the tie
    base[700:900] *= np.float32(1e-20)
    base[900:1100] *= np.float32(1e15)
    base[1100:1300] = np.float32(1e-42)          # subnormal components
    base[1300:2300] = np.minimum(np.abs(np.rint(base[1300:2300] * 40)), 255)  # SIFT-like small integers
    base[2300:2320] = np.float32(3.25)           # constant rows
    base[2320:2400] += np.float32(100.0)         # a large common offset
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q[0] = 0.0
    q[1] = 0.0
    q[1, 7] = 3.0
    q[2] *= np.float32(1e-10)
    q[3] *= np.float32(1e10)
    q[4:12] = base[1300:1308] + rng.standard_normal((8, d)).astype(np.float32) * np.float32(0.5)
    q[12:16] = base[2320:2324] + rng.standard_normal((4, d)).astype(np.float32) * np.float32(1e-3)
    q[16:20] = np.minimum(np.abs(np.rint(q[16:20] * 40)), 255)
    deg = rng.integers(20, 70, n)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(deg)
    nbrs = rng.integers(0, n, int(off[-1])).astype(np.uint32)
    idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT if ip else mi.L2, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(5, off, nbrs)
    idx.set_option("visited_mode", mode)
    if mode == 3:  # (the plan picks table size, tag policy and hub bitmap by what a hop gathers: the same form with and without the screen)
        idx.set_option("vis_entries", 4096)
        idx.set_option("tag_protect", 1)
        idx.set_option("hub_mode", 1)
    for L in (10, 50, 64, 65, 300):
        o = oracle.search_batch(base, off, nbrs, 5, q, K, L, 1 if ip else 0)
        idx.set_option("screen", 1)
        a = idx.SearchRoarGraphBatch(q, K, L)
        assert idx.info("last_screen") == 0
        idx.set_option("screen", 2)
        b = idx.SearchRoarGraphBatch(q, K, L)
        assert idx.info("last_screen") == 1
        idx.set_option("screen", 0)
        c = idx.SearchRoarGraphBatch(q, K, L)
        assert idx.info("last_screen") in (0, 1)  # auto: whichever was measured faster on this set
        idx.set_option("calibrate", 0)
        c0 = idx.SearchRoarGraphBatch(q, K, L)
        assert idx.info("last_screen") == (0 if d == 128 else 1)  # no measurement: on for rows of 768 bytes and more
        assert np.array_equal(c0["ids"], o["ids"])
        idx.set_option("calibrate", 1)
        assert np.array_equal(c["ids"], o["ids"])
        for key in ("ids", "dists", "hops") + (("cmps",) if mode != 3 else ()):
            assert np.array_equal(a[key], o[key]), (L, key, "screen off")
            assert np.array_equal(b[key], o[key]), (L, key, "screen on")
        assert np.array_equal(a["cmps"], b["cmps"])
    idx.close()


@pytest.mark.parametrize("shape", ["ip-200", "ip-512", "ip-128", "l2-128", "l2-512"])
def test_screen_never_drops_a_row_at_or_under_the_threshold(shape, built_lib):
    """The property the search relies on, pair by pair (roar_debug_screen_keeps runs the kernels' own screening passes): with the
    threshold set to a row's EXACT distance as the reference computes it (roar_distances: bit-equal to distance.h), the screen must
    keep the row -- for every one of ~1 M (query, row) pairs per shape, ordinary and adversarial (wide dynamic range inside a row,
    zero / constant / integer / tiny / huge / subnormal rows, rows with a common offset, zero and one-component queries, tiny and
    huge queries, queries that are base rows plus noise).  And it is not vacuous: with the threshold a few per cent of |q||x| under
    the exact distance, ordinary rows are dropped."""
    from mysteryann_amd import index as mi

    ip, d = shape.startswith("ip"), int(shape.split("-")[1])
    n, nq, m = 8192, 256, 4096
    rng = np.random.default_rng(77)
    base = rng.standard_normal((n, d)).astype(np.float32)
    base[100:400, 7] *= 1e4
    base[400:420] = 0.0
    base[500:600] = base[600:700]
    base[700:900] *= np.float32(1e-20)
    base[900:1100] *= np.float32(1e15)
    base[1100:1300] = np.float32(1e-42)
    base[1300:2300] = np.minimum(np.abs(np.rint(base[1300:2300] * 40)), 255)
    base[2300:2320] = np.float32(3.25)
    base[2320:2400] += np.float32(100.0)
    base[2400:2600] *= np.float32(1e-12)
    base[2600:2700] = np.float32(1e-25) * rng.standard_normal((100, d)).astype(np.float32)
    base[2700:2800, ::2] = 0.0
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q[0] = 0.0
    q[1] = 0.0
    q[1, 7] = 3.0
    q[2] *= np.float32(1e-10)
    q[3] *= np.float32(1e10)
    q[4:12] = base[1300:1308] + rng.standard_normal((8, d)).astype(np.float32) * np.float32(0.5)
    q[12:16] = base[2320:2324] + rng.standard_normal((4, d)).astype(np.float32) * np.float32(1e-3)
    q[16:20] = np.minimum(np.abs(np.rint(q[16:20] * 40)), 255)
    q[20:24] *= np.float32(1e-20)
    q[24:28] = base[4000:4004]                       # a query that IS a base row
    q[28:32] = -base[4000:4004]
    q[32:36, 1:] *= np.float32(1e-6)                  # one dominant component
    ids = np.stack([rng.permutation(n)[:m] for _ in range(nq)]).astype(np.uint32)
    ids[:, :2800:7] = np.arange(0, 2800, 7, dtype=np.uint32)  # the special rows are in every query's list
    idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT if ip else mi.L2, None)
    idx.SetBaseData(base)
    exact = idx.Distances(q, ids)
    keep = idx.ScreenKeeps(q, ids, exact)
    bad = np.argwhere(keep == 0)
    assert bad.size == 0, (shape, "dropped at threshold = exact distance", bad[:5], [(int(ids[a, b]), float(exact[a, b])) for a, b in bad[:5]])
    # just above the exact distance the verdict must stay "keep" as well (the threshold only got looser)
    up = np.nextafter(exact, np.float32(np.inf))
    assert idx.ScreenKeeps(q, ids, up).all()
    # not vacuous: ordinary queries x ordinary rows, threshold 5 % of |q||x| under the exact distance -> dropped
    oq = np.arange(40, nq)
    ordinary = ids[oq] >= 4100
    qn = np.linalg.norm(q[oq].astype(np.float64), axis=1)[:, None]
    xn = np.linalg.norm(base.astype(np.float64), axis=1)[ids[oq]]
    margin = (0.05 * qn * xn) if ip else (0.1 * qn * xn + 0.0)
    low = (exact[oq].astype(np.float64) - margin).astype(np.float32)
    k2 = idx.ScreenKeeps(q[oq], ids[oq], low)
    assert k2[ordinary].mean() < 0.02, (shape, float(k2[ordinary].mean()))
    idx.close()


def test_device_builder_refuses_lists_that_name_no_base_point(built_lib):
    """roar_build_roargraph_device reads the learn->base lists from the caller's HBM: an id outside the base set must be
    refused like the host form refuses it ("learn base knn file error"), not read or written out of bounds; a float tensor
    is refused by the Python mirror."""
    import torch

    from mysteryann_amd import capi
    from mysteryann_amd import index as mi, synth

    n = 3000
    base, q, learn = synth.t2i_numpy(n, 8, 200, n_learn=500)
    idx = mi.IndexBipartite(200, n, mi.INNER_PRODUCT, None)
    idx.SetBaseData(base)
    lists, _ = idx.ExactKnnDevice(torch.from_numpy(learn).cuda(), 100, want_dists=False)
    p = mi.Parameters()
    p.Set("M_sq", 100), p.Set("M_pjbp", 35), p.Set("L_pjpq", 100)
    bad = lists.clone()
    bad[123, 7] = n + 5
    with pytest.raises(mi.RoarError) as e:
        idx.BuildRoarGraphDevice(bad, p)
    assert e.value.code == capi.ERR_FORMAT and "learn base knn file error" in str(e.value)
    with pytest.raises(AssertionError):
        idx.BuildRoarGraphDevice(lists.float(), p)
    idx.BuildRoarGraphDevice(lists, p)  # the untouched lists still build
    assert idx.info("nedges") > 0
    idx.close()


def test_every_failed_device_allocation_is_reported_and_leaks_nothing(built_lib):
    """Fault injection (roar_debug_fail_device_alloc_at: the n-th device allocation of the library fails as if HBM were
    exhausted), n = 1, 2, ... over a scenario that loads, searches in every visited form (spill lists, tag pools, hub rows
    built on the device), scores explicit ids, runs the exact-kNN generator and makes a replica: every failure comes back as
    ROAR_ERR_DEVICE through the C ABI, and after the objects are destroyed the device's free memory is back where it was
    (hipMemGetInfo) -- round 2 leaked the staging blocks of roar_distances / roar_exact_knn and a half-made replica."""
    import torch

    from mysteryann_amd import capi
    from mysteryann_amd import index as mi

    lib = capi.lib()
    n, d, nq = 4000, 200, 24
    base, q, off, nbrs = _random_case(n, d, nq, 1, seed=31, maxdeg=40)
    ids = np.random.default_rng(1).integers(0, n, (nq, 7)).astype(np.uint32)

    def scenario():
        made = []
        try:
            idx = mi.IndexBipartite(d, n, 1, None)
            made.append(idx)
            idx.SetBaseData(base)
            idx.SetProjectionGraph(5, off, nbrs)
            # (fixed plans: which kernel forms a calibration times depends on the clock, and the first launch of a form with more
            # scratch makes the runtime grow its scratch arena -- device memory that stays, and looks like a leak to this test)
            idx.set_option("calibrate", 0)
            for mode, hub, L in ((1, 1, 50), (0, 1, 50), (1, 2, 120), (3, 1, 300)):
                idx.set_option("visited_mode", mode)
                idx.set_option("hub_mode", hub)
                idx.SearchRoarGraphBatch(q, K, L)
            # one calibrated launch whose candidates are all the same kernel form (only the table size is open): scratch results
            idx.set_option("calibrate", 1), idx.set_option("screen", 2), idx.set_option("hub_mode", 1)
            idx.SearchRoarGraphBatch(q, K, 30)
            idx.Distances(q, ids)
            kid = np.zeros((nq, 5), np.uint32)
            capi.check(lib.roar_exact_knn(idx._h, capi.f32p(q), nq, d, 5, capi.u32p(kid), None))
            idx.set_option("knn_mode", 1)  # (both forms in every pass: the f32 form is also the fallback of the other)
            capi.check(lib.roar_exact_knn(idx._h, capi.f32p(q), nq, d, 5, capi.u32p(kid), None))
            rep = idx.Clone(0)
            made.append(rep)
            mi.IndexBipartite.SearchRoarGraphBatchMulti([idx, rep], q, K, 40)
        finally:
            for x in made:
                x.close()

    scenario()  # warm: code objects, runtime pools
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    failures = absorbed = 0
    for nth in range(1, 400):
        left = 0
        frees = []
        for again in range(2):
            # (twice with the same n: a failure path that leaks does so every time; what the runtime's allocator keeps for itself
            # after a new ORDER of allocations -- the screening copy is made inside the first search when its allocation failed while
            # the base rows were set -- it keeps once: 32 MiB chunks)
            lib.roar_debug_fail_device_alloc_at(nth)
            try:
                scenario()
                absorbed += 1 - again  # e.g. no room for the spill lists: the launch runs the forgetful form instead
            except mi.RoarError as e:
                failures += 1 - again
                assert e.code == capi.ERR_DEVICE, (nth, e)
            finally:
                left = lib.roar_debug_fail_device_alloc_at(0)
            torch.cuda.synchronize()
            frees.append(torch.cuda.mem_get_info()[0])
        # (one-sided: the second run may also find MORE free memory than the first -- the runtime gives a scratch arena back -- a leak cannot)
        assert frees[0] - frees[1] <= (4 << 20), (nth, frees[0] - frees[1])
        assert free0 - frees[1] <= (256 << 20), (nth, free0 - frees[1])
        if left > 0:  # the scenario made fewer than nth allocations: every one of them has been failed once
            absorbed -= 1
            break
    # the scenario makes a few dozen device allocations; absorbed by a documented fallback: no room for the spill lists (forgetful
    # form), for the 8-bit screening copy (every row scored on f32), for a calibration's scratch results (default plan), for the
    # workspaces of the bf16-screened exact kNN (eleven blocks: the f32 form runs instead)
    assert failures >= 20 and absorbed <= 30, (failures, absorbed)
    scenario()  # and the library still works


def _yardsticks():
    import glob

    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return sorted(int(os.path.basename(f)[len("builder_yardstick_"):-len(".json")]) for f in glob.glob(os.path.join(d, "builder_yardstick_*.json")))


@pytest.mark.parametrize("n_points", _yardsticks())
def test_reference_builder_tracks_restated_reference_builder_at_scale(n_points, golden_dir, built_lib):
    """SURVEY 8(f) #1 at sizes where graph quality shows.  tests/golden/builder_yardstick_<n>.json holds the degree
    statistics and the recall@10 / distance-evaluation curve of the CPU restatement of the reference's BuildRoarGraph
    (oracle/roar_builder_oracle.c, single-threaded: 23 minutes at 200 000 points, 2 h 46 min at 1 000 000;
    tests/scale/builder_yardstick.py) on a
    seeded T2I-shape set with the canonical parameters.  roar_build_roargraph_device on the same data and the same exact
    lists must track it: entry point equal, average degree within 3 %, longest row equal, recall within 0.005 (or three
    standard deviations of the 1 000-query estimate, 0.012 at a recall of 0.8) and distance evaluations within 5 % at L_pq =
    20 / 50 / 100 (the reference builder itself is order-dependent, so bytes cannot be
    compared)."""
    import json

    import torch

    from mysteryann_amd import index as mi, synth

    Y = json.load(open(os.path.join(golden_dir, f"builder_yardstick_{n_points}.json")))
    n, nq = Y["n"], Y["nq"]
    base, q, learn = synth.t2i_numpy(n, nq, 200, seed=Y["seed"], n_learn=Y["n_learn"])
    idx = mi.IndexBipartite(200, n, mi.INNER_PRODUCT, None)
    idx.SetBaseData(base)
    lists, _ = idx.ExactKnnDevice(torch.from_numpy(learn).cuda(), 100, want_dists=False)
    p = mi.Parameters()
    p.Set("M_sq", Y["M_sq"]), p.Set("M_pjbp", Y["M_pjbp"]), p.Set("L_pjpq", Y["L_pjpq"])
    idx.BuildRoarGraphDevice(lists, p)
    assert idx.info("ep") == Y["ep"]
    avg = idx.info("nedges") / n
    assert abs(avg - Y["avg_degree"]) <= 0.03 * Y["avg_degree"], (avg, Y["avg_degree"])
    assert idx.info("max_degree") == Y["max_degree"]
    qd = torch.from_numpy(q).cuda()
    gt = idx.ExactKnnDevice(qd, 10, want_dists=False)[0].cpu().numpy().view(np.uint32)
    idx.set_option("visited_mode", 0)  # the reference algorithm's distance-evaluation count
    for L in (20, 50, 100):
        r = idx.SearchRoarGraphBatch(q, 10, L)
        rec = mi.ComputeRecall(r["ids"], gt, 10)
        want = Y["curve"][str(L)]
        # (two different graphs evaluated on nq queries: three standard deviations of a recall of that size, at least 0.005)
        tol = max(0.005, 3.0 * (want["recall_at_10"] * (1.0 - want["recall_at_10"]) / (nq * 10)) ** 0.5)
        assert abs(rec - want["recall_at_10"]) <= tol, (L, rec, want, tol)
        assert abs(r["cmps"].mean() - want["avg_cmps"]) <= 0.05 * want["avg_cmps"], (L, r["cmps"].mean(), want)
    idx.close()


def test_two_builds_of_the_same_data_make_the_same_graph(built_lib):
    """The builder appends no edge through an atomic cursor (proposals are sorted, then applied in order), so the graph
    is a function of the data: two builds of a 200 k-point T2I-shape set have the same 64-bit adjacency hash, rows are
    compacted to the longest row, and a third build with another projection depth hashes differently."""
    import torch

    from mysteryann_amd import index as mi, synth

    n, d = 200_000, 200
    base, q, learn = synth.t2i_torch(n, 256, d, n_learn=n)
    hashes, saved = [], []
    for L_learn in (128, 128, 96):
        idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None)
        idx.SetBaseData(base)
        idx.BuildBootstrapGraph(learn=learn, max_degree=384, row_keep=128, L_learn=L_learn)
        torch.cuda.synchronize()
        hashes.append(idx.info("graph_hash"))
        assert idx.info("adj_stride") == (idx.info("max_degree") + 1 + 31) // 32 * 32 <= 160
        saved.append(idx.SearchRoarGraphBatchDevice(q, 10, 30)["ids"].cpu().numpy())
        idx.close()
    assert hashes[0] == hashes[1] and hashes[0] != hashes[2], [hex(h & (2**64 - 1)) for h in hashes]
    assert np.array_equal(saved[0], saved[1])


def test_cpp_harness_end_to_end_on_files(tmp_path, built_lib):
    """BASELINE configs[0] shape through the file-based drop-in path: SIFT-like 128-d L2 set written as
    .fbin / .ibin / graph index, searched by the C++ harness (mysteryann_amd/csrc/test_search_roargraph,
    own counterpart of tests/test_search_roargraph.cpp) at L_pq = 100; its CSV row
    L_pq,qps,avg_cmps,mean_latency_ms,recall,avg_hops must carry the oracle's avg_cmps / recall / avg_hops."""
    import subprocess

    import oracle
    from mysteryann_amd import fileio, synth
    from tests.golden import small_builder as sb

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "mysteryann_amd", "csrc", "test_search_roargraph")
    assert os.path.exists(exe), "harness not built (python __graft_entry__.py)"
    n, nq, d = 6000, 200, 128
    base, q, learn = synth.sift_numpy(n, nq, d, seed=4, n_learn=3000)
    ep, adj = sb.build_roar_like(base, learn, 0, M_sq=24, M=12)
    gt_i, gt_d = sb.exact_knn(q, base, 100, 0)
    paths = {k: str(tmp_path / k) for k in ("base.fbin", "query.fbin", "gt.ibin", "roar.index", "eval.csv")}
    fileio.write_fbin(paths["base.fbin"], base)
    fileio.write_fbin(paths["query.fbin"], q)
    fileio.write_gt(paths["gt.ibin"], gt_i, gt_d)
    fileio.write_graph(paths["roar.index"], ep, adj)
    # --devices 0,0,0: the index replicated three times (all on this box's device), the query block split by query
    for extra in ([], ["--visited", "exact"], ["--per_query", "-T", "4"], ["--devices", "0,0,0"]):
        out = subprocess.run([exe, "--data_type", "float", "--dist", "l2", "--base_data_path", paths["base.fbin"],
                              "--query_path", paths["query.fbin"], "--gt_path", paths["gt.ibin"],
                              "--projection_index_save_path", paths["roar.index"], "--L_pq", "20", "100", "--k", "10",
                              "--evaluation_save_path", paths["eval.csv"]] + extra,
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr + out.stdout
        rows = [r.split(",") for r in open(paths["eval.csv"]).read().strip().splitlines()]
        assert [int(r[0]) for r in rows] == [20, 100]
        off, nbrs = sb.to_csr(adj)
        for r, L in zip(rows, (20, 100)):
            o = oracle.search_batch(base, off, nbrs, ep, q, 10, L, 0)
            assert abs(float(r[2]) - o["cmps"].mean()) < 2e-4 * o["cmps"].mean() + 1e-2, extra  # 6 printed digits
            assert abs(float(r[5]) - o["hops"].mean()) < 2e-4 * o["hops"].mean() + 1e-3, extra
            assert abs(float(r[4]) - oracle.recall(o["ids"], gt_i, 10)) < 1e-6


def test_cpp_harness_index_sharded_on_files(tmp_path, built_lib):
    """BASELINE configs[4] through the C++ harness: `--shards 0,0`, one base file and one index file per shard (local
    ids), every query on every shard, per-shard lists all-gathered (RCCL, one rank here) and merged.  The CSV row must
    carry the recall of the merged oracle lists and the cmps / hops summed over the shards."""
    import subprocess

    import oracle
    from mysteryann_amd import fileio, synth
    from tests.golden import small_builder as sb

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "mysteryann_amd", "csrc", "test_search_roargraph")
    n, nq, d = 6000, 150, 200
    base, q, learn = synth.t2i_numpy(n, nq, d, seed=6, n_learn=3000)
    gt_i, gt_d = sb.exact_knn(q, base, 100, 1)
    fileio.write_fbin(str(tmp_path / "query.fbin"), q)
    fileio.write_gt(str(tmp_path / "gt.ibin"), gt_i, gt_d)
    bases, graphs, per = [], [], []
    for s_, (lo, hi) in enumerate(((0, 2500), (2500, 6000))):  # ragged shards
        ep, adj = sb.build_roar_like(base[lo:hi], learn, 1, M_sq=24, M=12)
        bases.append(str(tmp_path / f"base{s_}.fbin"))
        graphs.append(str(tmp_path / f"roar{s_}.index"))
        fileio.write_fbin(bases[-1], base[lo:hi])
        fileio.write_graph(graphs[-1], ep, adj)
        per.append((lo, ep) + sb.to_csr(adj))
    csv = str(tmp_path / "eval.csv")
    out = subprocess.run([exe, "--data_type", "float", "--dist", "ip", "--base_data_path", ",".join(bases), "--query_path",
                          str(tmp_path / "query.fbin"), "--gt_path", str(tmp_path / "gt.ibin"), "--projection_index_save_path",
                          ",".join(graphs), "--L_pq", "20", "60", "--k", "10", "--evaluation_save_path", csv, "--shards", "0,0",
                          "--visited", "exact"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "Index sharded 2-way over 1 device(s)" in out.stdout
    rows = [r.split(",") for r in open(csv).read().strip().splitlines()]
    assert [int(r[0]) for r in rows] == [20, 60]
    for r, L in zip(rows, (20, 60)):
        oi, od, cm, hp = [], [], 0, 0
        for lo, ep, off, nbrs in per:
            hi = lo + len(off) - 1
            o = oracle.search_batch(base[lo:hi], off, nbrs, ep, q, 10, L, 1)
            oi.append(o["ids"] + np.uint32(lo)), od.append(o["dists"])
            cm, hp = cm + o["cmps"].astype(np.int64), hp + o["hops"].astype(np.int64)
        merged = np.zeros((nq, 10), np.uint32)
        for qi in range(nq):
            ci, cd = np.concatenate([x[qi] for x in oi]), np.concatenate([x[qi] for x in od])
            merged[qi] = ci[np.lexsort((ci, cd))[:10]]
        assert abs(float(r[4]) - oracle.recall(merged, gt_i, 10)) < 1e-6
        assert abs(float(r[2]) - cm.mean()) < 2e-4 * cm.mean() + 1e-2
        assert abs(float(r[5]) - hp.mean()) < 2e-4 * hp.mean() + 1e-3


@pytest.mark.parametrize("metric", [1, 0])
def test_builder_prune_kernel_matches_restated_prune_rule(metric, built_lib):
    """The builder's occlusion-prune kernel against the CPU restatement of PruneBiSearchBaseGetBase
    (src/index_bipartite.cpp:1612-1694): one exact list projected onto an (almost) empty graph must leave
    exactly the reference rule's kept set in the pivot's row."""
    import torch

    import oracle
    from mysteryann_amd import index as mi
    from tests.golden.small_builder import exact_knn

    n, d, C, Mp = 3000, 200, 48, 24
    rng = np.random.default_rng(17 + metric)
    base = (rng.standard_normal((n, 32)) @ rng.standard_normal((32, d)) * 0.05 + 0.02 * rng.standard_normal((n, d))).astype(
        np.float32)
    q = (rng.standard_normal((12, 32)) @ rng.standard_normal((32, d)) * 0.05).astype(np.float32)
    lists, _ = exact_knn(q, base, C, metric)
    for t in range(lists.shape[0]):
        idx = mi.IndexBipartite(d, n, metric, None)
        idx.SetBaseData(base)
        one = torch.from_numpy(lists[t:t + 1].astype(np.int32)).contiguous().cuda()
        idx.BuildBootstrapGraph(M=1, base_cap=1, max_degree=64, L_build=8, learn=None, M_sq=C, M_pj=Mp, rounds=0,
                                learn_knn=one, supply_k=0, trim_min_proj=0, phase1_cand=0, phase1_l2=(metric == 1))
        ep, off, nbrs = idx.GetProjectionGraph()
        pivot = int(lists[t, 0])
        row = nbrs[int(off[pivot]):int(off[pivot + 1])]
        want = oracle.prune_bisearch(base, metric, pivot, lists[t, 1:], Mp)
        got = set(int(x) for x in row)
        assert set(int(x) for x in want) <= got, (t, sorted(want), sorted(got))
        idx.close()


def test_build_roargraph_reference_shape_against_restated_builder(tmp_path, built_lib):
    """SURVEY 8(f) #1, first version: LoadLearnBaseKNN + BuildRoarGraph on the device against the CPU
    restatement of the reference builder ON THE SAME DATA and the same exact learn->base lists, canonical
    parameters (run_roargraph_test.sh:9).  The reference builder is nondeterministic, so parity is its
    own notion: degree statistics and the recall-vs-cmps curve (here: recall within 0.015 at equal L_pq,
    distance evaluations within 35 % -- the device graph is a little denser and a little more accurate), plus the file path: build CLI -> index file -> search CLI."""
    import subprocess

    import oracle
    from mysteryann_amd import fileio, synth
    from mysteryann_amd import index as mi
    from tests.golden.small_builder import exact_knn

    n, nq = 6000, 400
    base, q, learn = synth.t2i_numpy(n, nq, n_learn=n, seed=77)
    knn, _ = exact_knn(learn, base, 100, 1)
    gt, gtd = exact_knn(q, base, 100, 1)
    ep_r, off_r, nb_r = oracle.build_roargraph(base, knn, 1, 100, 35, 500)
    deg_r = np.diff(off_r.astype(np.int64))
    knn_path = str(tmp_path / "learn.base.knn.ibin")
    fileio.write_knn(knn_path, knn)
    idx = mi.IndexBipartite(200, n, mi.INNER_PRODUCT, None)
    p = mi.Parameters()
    p.Set("M_sq", 100), p.Set("M_pjbp", 35), p.Set("L_pjpq", 500)
    idx.LoadLearnBaseKNN(knn_path)
    idx.BuildRoarGraph(n, learn, n, base, p)
    ep_g, off_g, nb_g = idx.GetProjectionGraph()
    deg_g = np.diff(off_g.astype(np.int64))
    assert ep_g == ep_r  # CalculateProjectionep
    assert deg_g.max() <= 105 and deg_g.min() >= 1
    assert 0.6 * deg_r.mean() <= deg_g.mean() <= 1.4 * deg_r.mean(), (deg_g.mean(), deg_r.mean())
    for L in (30, 50, 100):
        rr = oracle.search_batch(base, off_r, nb_r, ep_r, q, 10, L, 1, threads=4)
        rg = oracle.search_batch(base, off_g, nb_g, ep_g, q, 10, L, 1, threads=4)
        rec_r, rec_g = oracle.recall(rr["ids"], gt, 10), oracle.recall(rg["ids"], gt, 10)
        assert rec_g >= rec_r - 0.015, (L, rec_g, rec_r)
        assert rg["cmps"].mean() <= 1.35 * rr["cmps"].mean(), (L, rg["cmps"].mean(), rr["cmps"].mean())
    # CLI pair on files: test_build_roargraph -> index -> test_search_roargraph
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = {k: str(tmp_path / k) for k in ("base.fbin", "learn.fbin", "query.fbin", "gt.ibin", "built.index", "eval.csv")}
    fileio.write_fbin(paths["base.fbin"], base)
    fileio.write_fbin(paths["learn.fbin"], learn)
    fileio.write_fbin(paths["query.fbin"], q)
    # ground-truth files carry plain inner products for --dist ip (ComputeRderr negates the search's
    # distances before comparing, tests/test_search_roargraph.cpp:47-50)
    fileio.write_gt(paths["gt.ibin"], gt, -gtd)
    b = subprocess.run([os.path.join(root, "mysteryann_amd", "csrc", "test_build_roargraph"), "--data_type", "float",
                        "--dist", "ip", "--base_data_path", paths["base.fbin"], "--sampled_query_data_path",
                        paths["learn.fbin"], "--projection_index_save_path", paths["built.index"],
                        "--learn_base_nn_path", knn_path, "--M_sq", "100", "--M_pjbp", "35", "--L_pjpq", "500", "-T", "4"],
                       capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr + b.stdout
    s = subprocess.run([os.path.join(root, "mysteryann_amd", "csrc", "test_search_roargraph"), "--data_type", "float",
                        "--dist", "ip", "--base_data_path", paths["base.fbin"], "--query_path", paths["query.fbin"],
                        "--gt_path", paths["gt.ibin"], "--projection_index_save_path", paths["built.index"], "--L_pq", "50",
                        "--k", "10", "--evaluation_save_path", paths["eval.csv"], "--rderr"],
                       capture_output=True, text=True, timeout=600)
    assert s.returncode == 0, s.stderr + s.stdout
    row = open(paths["eval.csv"]).read().strip().split(",")
    assert int(row[0]) == 50 and float(row[4]) > 0.97 and len(row) == 7 and abs(float(row[6])) < 0.05


def test_cli_pipeline_from_fbin_files_only(tmp_path, built_lib):
    """The whole reference workflow on files, CLIs only: compute_knn (learn->base lists, ground truth with
    distances) -> test_build_roargraph -> test_search_roargraph (run_roargraph_test.sh,
    run_roargraph_search_test.sh).  Cosine metric, ragged dimension (100 -> padded 104)."""
    import subprocess

    from mysteryann_amd import fileio, synth

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "mysteryann_amd", "csrc")
    n, nq, d = 5000, 300, 100
    base, q, learn = synth.t2i_numpy(n, nq, d, seed=5, n_learn=n)
    P = {k: str(tmp_path / k) for k in ("base.fbin", "learn.fbin", "query.fbin", "knn.ibin", "gt.ibin", "roar.index", "eval.csv")}
    fileio.write_fbin(P["base.fbin"], base)
    fileio.write_fbin(P["learn.fbin"], learn)
    fileio.write_fbin(P["query.fbin"], q)

    def run(*cmd):
        r = subprocess.run(list(cmd), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr + r.stdout
        return r.stdout

    run(os.path.join(csrc, "compute_knn"), "--dist", "cosine", "--base_data_path", P["base.fbin"], "--query_path",
        P["learn.fbin"], "--out_path", P["knn.ibin"], "--K", "100")
    run(os.path.join(csrc, "compute_knn"), "--dist", "cosine", "--base_data_path", P["base.fbin"], "--query_path",
        P["query.fbin"], "--out_path", P["gt.ibin"], "--K", "100", "--with_dists")
    gt_ids, gt_d = fileio.read_gt(P["gt.ibin"])
    bn = base / np.linalg.norm(base, axis=1, keepdims=True)
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)
    want = np.argsort(-(qn.astype(np.float64) @ bn.astype(np.float64).T), axis=1, kind="stable")[:, :10]
    assert (np.sort(gt_ids[:, :10].astype(np.int64), 1) == np.sort(want, 1)).mean() > 0.995
    assert (np.diff(gt_d, axis=1) <= 1e-6).all()  # plain cosines, best first
    run(os.path.join(csrc, "test_build_roargraph"), "--data_type", "float", "--dist", "cosine", "--base_data_path",
        P["base.fbin"], "--sampled_query_data_path", P["learn.fbin"], "--projection_index_save_path", P["roar.index"],
        "--learn_base_nn_path", P["knn.ibin"], "--M_sq", "100", "--M_pjbp", "35", "--L_pjpq", "500")
    run(os.path.join(csrc, "test_search_roargraph"), "--data_type", "float", "--dist", "cosine", "--base_data_path",
        P["base.fbin"], "--query_path", P["query.fbin"], "--gt_path", P["gt.ibin"], "--projection_index_save_path",
        P["roar.index"], "--L_pq", "10", "50", "--k", "10", "--evaluation_save_path", P["eval.csv"])
    rows = [r.split(",") for r in open(P["eval.csv"]).read().strip().splitlines()]
    assert [int(r[0]) for r in rows] == [10, 50]
    assert float(rows[1][4]) > 0.97 and float(rows[1][4]) >= float(rows[0][4])


# ---- bipartite two-hop variant: SearchBipartiteGraph (index_bipartite.cpp:282-356), SURVEY 8(f) row 4 ----
@pytest.mark.parametrize("case,metric", [("ip_200", 1), ("l2_128", 0)])
@pytest.mark.parametrize("mode", [0, 1, 2], ids=["exact", "lds", "auto-8-waves"])
def test_bipartite_search_matches_reference_golden(case, metric, mode, golden_dir, tmp_path, built_lib):
    from mysteryann_amd import fileio
    from mysteryann_amd import index as mi

    exp = np.load(os.path.join(golden_dir, case + ".bipartite.expected.npz"))
    q = fileio.read_fbin(os.path.join(golden_dir, case + ".query.fbin"))
    idx = mi.IndexBipartite(q.shape[1], 0, metric, None)
    idx.LoadSearchNeededData(os.path.join(golden_dir, case + ".base.fbin"))
    src = os.path.join(golden_dir, case + ".bipartite.index")
    idx.Load(src)  # tests/test_search_bipartite.cpp:160
    idx.set_option("visited_mode", mode)
    for L in LS:
        r = idx.SearchBipartiteGraphBatch(q, K, L, exp["init_ids"])
        for key in ("ids", "dists", "hops", "cmps"):  # 2000 base points: the LDS table cannot overflow
            assert np.array_equal(r[key], exp[f"{key}_{L}"]), (case, L, key)
    out = str(tmp_path / "b.index")
    idx.Save(out)
    assert open(src, "rb").read() == open(out, "rb").read()
    idx.close()


@pytest.mark.parametrize("metric", [0, 1])
def test_bipartite_random_graphs_against_oracle(metric, built_lib):
    """Hub rows (> 64 query nodes in one hop), empty rows, ids repeated inside a hop, 1 / 10 / 64 entry
    points with repeats, pools shorter and longer than what is reachable."""
    import oracle
    from mysteryann_amd import index as mi
    from tests.test_oracle_pinning import _random_bipartite

    rng = np.random.default_rng(100 + metric)
    nb, nl, d = 3000, 2000, 72
    base = rng.standard_normal((nb, d)).astype(np.float32)
    q = rng.standard_normal((96, d)).astype(np.float32)
    off, nbrs = _random_bipartite(rng, nb, nl, max_q=8, max_b=40, hub=11)
    for mode in (0, 1, 2):  # 2 = AUTO: the eight-waves-per-query form with test-and-set tags
        idx = mi.IndexBipartite(d, nb, metric, None)
        idx.SetBaseData(base)
        idx.SetBipartiteGraph(off, nbrs)
        idx.set_option("visited_mode", mode)
        assert idx.info("bipartite_nodes") == nb + nl and idx.info("bipartite_edges") == len(nbrs)
        o2, n2 = idx.GetBipartiteGraph()
        assert np.array_equal(o2, off) and np.array_equal(n2, nbrs)
        for n_init in (1, 10, 64):
            init = rng.integers(0, nb, (96, n_init)).astype(np.uint32)
            init[::2, 0] = 11  # half of the queries start on the hub
            if n_init > 1:
                init[:, 1] = init[:, 0]
            else:
                init[:, 0] = 11  # a single entry point on an empty row would end the search short of k
            for L in (5, 33, 200, 700):
                o = oracle.search_bipartite_batch(base, off, nbrs, q, 5, L, metric, init, threads=4)
                assert o["short"] == 0
                r = idx.SearchBipartiteGraphBatch(q, 5, L, init)
                for key in ("ids", "dists", "hops"):
                    assert np.array_equal(r[key], o[key]), (mode, n_init, L, key)
                assert idx.info("last_waves_per_query") == (8 if mode == 2 and L <= 200 else 1)
                if mode != 1:
                    assert np.array_equal(r["cmps"], o["cmps"]), (mode, n_init, L)
                else:
                    assert (r["cmps"] >= o["cmps"]).all()
        idx.close()


def test_bipartite_build_and_generated_entry_points(tmp_path, built_lib):
    """BuildBipartite (qbaseNNbipartite) from kNN lists == the restated builder; entry points drawn by the
    library (init_ids = NULL) are reproducible from the seed; error behaviour of the reference."""
    import oracle
    from mysteryann_amd import fileio, synth
    from mysteryann_amd import index as mi
    from mysteryann_amd.capi import ERR_ARG, ERR_FORMAT, ERR_SHORT, RoarError
    from tests.golden.small_builder import exact_knn

    n, nl = 4000, 3000
    base, q, learn = synth.t2i_numpy(n, 80, 200, seed=5, n_learn=nl)
    knn, _ = exact_knn(learn, base, 50, 1)
    p = str(tmp_path / "learn.nn.ibin")
    fileio.write_knn(p, knn)
    idx = mi.IndexBipartite(200, n + nl, mi.INNER_PRODUCT, None)
    idx.LoadLearnBaseKNN(p)
    par = mi.Parameters()
    par.Set("M_pjbp", 35)
    idx.BuildBipartite(nl, None, n, base, par)  # tests/test_only_build_bipartite.cpp:129
    off, nbrs = idx.GetBipartiteGraph()
    woff, wnbrs = oracle.build_bipartite(knn, n, 35)
    assert np.array_equal(off, woff) and np.array_equal(nbrs, wnbrs)
    sp = str(tmp_path / "bp.index")
    idx.Save(sp)
    wp = str(tmp_path / "bp_want.index")
    oracle.bipartite_save(wp, woff, wnbrs)
    assert open(sp, "rb").read() == open(wp, "rb").read()
    # library-drawn entry points: the stand-in for std::random_device is reproducible from the seed
    idx.set_option("visited_mode", 0)
    idx.set_option("bipartite_seed", 42)
    r = idx.SearchBipartiteGraphBatch(q, K, 100)
    init = mi.IndexBipartite.BipartiteEntryIds(42, len(q), 10, n)
    assert init.max() < n and len(np.unique(init)) > 5 * len(q)
    bp = fileio.pad_rows(base)
    o = oracle.search_bipartite_batch(bp, woff, wnbrs, fileio.pad_rows(q), K, 100, 1, init, threads=4)
    for key in ("ids", "dists", "cmps", "hops"):
        assert np.array_equal(r[key], o[key]), key
    r2 = idx.SearchBipartiteGraphBatch(q, K, 100)  # the seed advances: a fresh draw, as in the reference
    assert not np.array_equal(r2["cmps"], r["cmps"])
    # one query through the reference's signature
    par.Set("L_pq", 100)
    idx.set_option("bipartite_seed", 42)
    ind, dd = np.zeros(K, np.uint32), np.zeros(K, np.float32)
    cmps, hops = idx.SearchBipartiteGraph(q[0], K, 0, par, ind, dd)
    assert (cmps, hops) == (int(o["cmps"][0]), int(o["hops"][0])) and np.array_equal(ind, o["ids"][0])
    # errors
    with pytest.raises(RoarError) as e:
        idx.SearchBipartiteGraphBatch(q, 20, 10, init)
    assert e.value.code == ERR_ARG and "L_pq must greater or equal than k" in str(e.value)
    lonely = np.full((len(q), 1), int(np.setdiff1d(np.arange(n), knn[:, 0])[0]), np.uint32)  # a base node no query targets
    with pytest.raises(RoarError) as e:
        idx.SearchBipartiteGraphBatch(q, 2, 10, lonely)
    assert e.value.code == ERR_SHORT and str(e.value) == "not enough results: 1, expected: 2"
    bad_off = np.array([0, 1, 1], np.uint64)
    with pytest.raises(RoarError) as e:  # a base node linked to a base node
        i2 = mi.IndexBipartite(200, 2, 1, None)
        i2.SetBaseData(base[:1])
        i2.SetBipartiteGraph(bad_off, np.array([0], np.uint32))
    assert e.value.code == ERR_FORMAT
    idx.close()


def test_bipartite_device_resident_batch(golden_dir, built_lib):
    import torch
    from mysteryann_amd import fileio
    from mysteryann_amd import index as mi

    exp = np.load(os.path.join(golden_dir, "ip_200.bipartite.expected.npz"))
    q = fileio.read_fbin(os.path.join(golden_dir, "ip_200.query.fbin"))
    idx = mi.IndexBipartite(200, 0, 1, None)
    idx.LoadSearchNeededData(os.path.join(golden_dir, "ip_200.base.fbin"))
    idx.Load(os.path.join(golden_dir, "ip_200.bipartite.index"))
    qd = torch.from_numpy(fileio.pad_rows(q)).cuda()
    init = torch.from_numpy(exp["init_ids"].astype(np.int32)).cuda()
    # every visited form, the two on-chip table settings included (the two-hop kernels read the table-size field as a plain
    # shift count: the search kernel's form flags must never reach them)
    for mode, ve in ((2, 0), (0, 0), (1, 0), (3, 0), (1, 256), (3, 256)):
        idx.set_option("visited_mode", mode)
        idx.set_option("vis_entries", ve)
        out = idx.SearchBipartiteGraphBatchDevice(qd, K, 50, init)
        torch.cuda.synchronize()
        assert int(out["short"].item()) == 0
        assert np.array_equal(out["ids"].cpu().numpy().astype(np.uint32), exp["ids_50"]), (mode, ve)
        assert np.array_equal(out["dists"].cpu().numpy(), exp["dists_50"]), (mode, ve)
        assert np.array_equal(out["hops"].cpu().numpy().astype(np.uint32), exp["hops_50"]), (mode, ve)
        if ve == 0:  # (2 000 base points: the default table cannot overflow)
            assert np.array_equal(out["cmps"].cpu().numpy().astype(np.uint32), exp["cmps_50"]), (mode, ve)
    idx.close()


def test_bipartite_cli_pipeline_on_files(tmp_path, built_lib):
    """tests/test_only_build_bipartite.cpp + tests/test_search_bipartite.cpp on files, CLIs only: the saved
    index is the restated builder's byte for byte; the CSV has the reference's seven columns
    (L_pq,qps,avg_cmps,latency,recall,rderr,avg_hops) and the values of a replay of the same seeded draws."""
    import subprocess

    import oracle
    from mysteryann_amd import fileio, synth
    from mysteryann_amd import index as mi

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "mysteryann_amd", "csrc")
    n, nq, d = 5000, 200, 200
    base, q, learn = synth.t2i_numpy(n, nq, d, seed=9, n_learn=n)
    P = {k: str(tmp_path / k) for k in ("base.fbin", "learn.fbin", "query.fbin", "knn.ibin", "gt.ibin", "bp.index", "eval.csv")}
    fileio.write_fbin(P["base.fbin"], base)
    fileio.write_fbin(P["learn.fbin"], learn)
    fileio.write_fbin(P["query.fbin"], q)

    def run(*cmd):
        r = subprocess.run(list(cmd), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr + r.stdout
        return r.stdout

    run(os.path.join(csrc, "compute_knn"), "--dist", "ip", "--base_data_path", P["base.fbin"], "--query_path",
        P["learn.fbin"], "--out_path", P["knn.ibin"], "--K", "100")
    run(os.path.join(csrc, "compute_knn"), "--dist", "ip", "--base_data_path", P["base.fbin"], "--query_path",
        P["query.fbin"], "--out_path", P["gt.ibin"], "--K", "100", "--with_dists")
    run(os.path.join(csrc, "test_build_roargraph"), "--data_type", "float", "--dist", "ip", "--base_data_path",
        P["base.fbin"], "--sampled_query_data_path", P["learn.fbin"], "--bipartite_index_save_path", P["bp.index"],
        "--learn_base_nn_path", P["knn.ibin"], "--M_pjbp", "35")
    with open(P["knn.ibin"], "rb") as f:
        nk, kk = np.fromfile(f, np.uint32, 2)
        knn = np.fromfile(f, np.uint32, int(nk) * int(kk)).reshape(int(nk), int(kk))
    woff, wnbrs = oracle.build_bipartite(knn, n, 35)
    want_path = str(tmp_path / "want.index")
    oracle.bipartite_save(want_path, woff, wnbrs)
    assert open(P["bp.index"], "rb").read() == open(want_path, "rb").read()
    for extra in ([], ["--per_query", "-T", "4"]):
        run(os.path.join(csrc, "test_search_roargraph"), "--data_type", "float", "--dist", "ip", "--base_data_path",
            P["base.fbin"], "--query_path", P["query.fbin"], "--gt_path", P["gt.ibin"], "--bipartite_index_save_path",
            P["bp.index"], "--L_pq", "20", "100", "--k", "10", "--evaluation_save_path", P["eval.csv"], "--visited",
            "exact", "--entry_seed", "7", *extra)
        rows = [r.split(",") for r in open(P["eval.csv"]).read().strip().splitlines()]
        assert [int(r[0]) for r in rows] == [20, 100] and all(len(r) == 7 for r in rows)
        assert float(rows[1][4]) >= float(rows[0][4]) > 0.3
    # replay of the batch run (last loop iteration was per-query; redo the batch one): seed 7 is used by the
    # warm-up call of L_pq = 20, then advanced once per call
    run(os.path.join(csrc, "test_search_roargraph"), "--data_type", "float", "--dist", "ip", "--base_data_path",
        P["base.fbin"], "--query_path", P["query.fbin"], "--gt_path", P["gt.ibin"], "--bipartite_index_save_path",
        P["bp.index"], "--L_pq", "20", "--k", "10", "--evaluation_save_path", P["eval.csv"], "--visited", "exact",
        "--entry_seed", "7")
    row = open(P["eval.csv"]).read().strip().split(",")
    idx = mi.IndexBipartite(d, n, 1, None)
    idx.SetBaseData(base)
    idx.SetBipartiteGraph(woff, wnbrs)
    idx.set_option("visited_mode", 0)
    idx.set_option("bipartite_seed", 7)
    idx.SearchBipartiteGraphBatch(q[:100], 10, 20)  # the harness's warm-up draw
    r = idx.SearchBipartiteGraphBatch(q, 10, 20)
    gt_ids, gt_d = fileio.read_gt(P["gt.ibin"])
    assert abs(float(row[2]) - r["cmps"].mean()) < 1e-2 and abs(float(row[6]) - r["hops"].mean()) < 1e-3
    assert abs(float(row[4]) - oracle.recall(r["ids"], gt_ids, 10)) < 1e-5
    assert abs(float(row[5]) - oracle.rderr(gt_d, r["dists"], 10, 1)) < 1e-6 * max(1.0, abs(float(row[5])))
    idx.close()


def test_api_edges_empty_batch_call_order_and_limits(golden_dir, tmp_path, built_lib):
    """Empty query block, calls out of order, a graph that does not match the base set, truncated files,
    out-of-range ids, a pool that cannot fit on chip, k = L_pq = 1 and the smallest row (8 floats)."""
    import oracle
    from mysteryann_amd import index as mi
    from mysteryann_amd.capi import ERR_ARG, ERR_FORMAT, RoarError

    idx = mi.IndexBipartite(200, 0, 1, None)
    q = np.zeros((3, 200), np.float32)
    with pytest.raises(RoarError, match="before base vectors were loaded") as e:
        idx.SearchRoarGraphBatch(q, 1, 1)
    assert e.value.code == ERR_ARG
    idx.LoadSearchNeededData(os.path.join(golden_dir, "ip_200.base.fbin"))
    with pytest.raises(RoarError, match="before the projection graph was loaded"):
        idx.SearchRoarGraphBatch(q, 1, 1)
    with pytest.raises(RoarError, match="before the bipartite graph was loaded"):
        idx.SearchBipartiteGraphBatch(q, 1, 1, np.zeros((3, 1), np.uint32))
    idx.LoadProjectionGraph(os.path.join(golden_dir, "l2_128.roar.index"))  # 2000 nodes too: accepted
    r = idx.SearchRoarGraphBatch(np.zeros((0, 200), np.float32), 10, 10)  # nq = 0 is a no-op
    assert r["ids"].shape == (0, 10)
    with pytest.raises(RoarError, match="bytes of LDS"):
        idx.SearchRoarGraphBatch(q, 10, 30000)
    with pytest.raises(RoarError, match="query stride smaller than the dimension"):
        idx.SearchRoarGraphBatch(np.zeros((3, 64), np.float32), 1, 1)
    # graph / base mismatch and malformed graphs
    small = mi.IndexBipartite(8, 0, 0, None)
    small.SetBaseData(np.eye(8, dtype=np.float32))
    with pytest.raises(RoarError) as e:
        small.SetProjectionGraph(9, np.zeros(9, np.uint64), np.zeros(0, np.uint32))  # entry point outside
    assert e.value.code == ERR_FORMAT
    with pytest.raises(RoarError) as e:
        small.SetProjectionGraph(0, np.array([0, 1] + [1] * 7, np.uint64), np.array([8], np.uint32))  # neighbour outside
    assert e.value.code == ERR_FORMAT
    small.SetProjectionGraph(0, np.zeros(5, np.uint64), np.zeros(0, np.uint32))  # 4 nodes, base has 8 rows
    with pytest.raises(RoarError, match="graph has 4 nodes but base has 8 rows"):
        small.SearchRoarGraphBatch(np.ones((1, 8), np.float32), 1, 1)
    trunc = str(tmp_path / "trunc.index")
    open(trunc, "wb").write(open(os.path.join(golden_dir, "ip_200.roar.index"), "rb").read()[:1001 * 4])
    with pytest.raises(RoarError, match="graph file truncated"):
        small.LoadProjectionGraph(trunc)
    with pytest.raises(RoarError, match="graph file truncated"):
        small.Load(trunc)
    # k = L_pq = 1 on the smallest row: a ring over the 8 unit vectors, the walk ends where the reference's does
    off = np.arange(9, dtype=np.uint64)
    nbrs = ((np.arange(8) + 1) % 8).astype(np.uint32)
    small.SetProjectionGraph(3, off, nbrs)
    qq = np.eye(8, dtype=np.float32)[[5, 0]] * 2
    for mode in (0, 1):
        small.set_option("visited_mode", mode)
        for L in (1, 2, 8):
            o = oracle.search_batch(np.eye(8, dtype=np.float32), off, nbrs, 3, qq, 1, L, 0)
            r = small.SearchRoarGraphBatch(qq, 1, L)
            for key in ("ids", "dists", "cmps", "hops"):
                assert np.array_equal(r[key], o[key]), (mode, L, key)
    idx.close()
    small.close()


def test_concurrent_single_query_calls_are_combined(golden_dir, built_lib):
    """The reference's harness shape: SearchRoarGraph once per query from many host threads
    (tests/test_search_roargraph.cpp:203-209).  Concurrent one-query calls are combined into one launch; every
    caller must still get exactly its own query's reference result, mixed (k, L_pq) included."""
    import threading

    from mysteryann_amd import index as mi

    exp = np.load(os.path.join(golden_dir, "ip_200.expected.npz"))
    idx, q = _index("ip_200", golden_dir, 0)
    nq = q.shape[0]
    got = {L: dict(ids=np.zeros((nq, K), np.uint32), dists=np.zeros((nq, K), np.float32), cmps=np.zeros(nq, np.uint32),
                   hops=np.zeros(nq, np.uint32)) for L in (20, 50)}
    combined = []
    errors = []

    def worker(t, nthreads):
        try:
            for rep in range(3):
                for i in range(t, nq, nthreads):
                    L = 20 if (i + rep) % 2 else 50
                    par = mi.Parameters()
                    par.Set("L_pq", L)
                    ind, dd = np.zeros(K, np.uint32), np.zeros(K, np.float32)
                    c, h = idx.SearchRoarGraph(q[i], K, i, par, ind, dd)
                    g = got[L]
                    g["ids"][i], g["dists"][i], g["cmps"][i], g["hops"][i] = ind, dd, c, h
                    combined.append(idx.info("last_combined"))
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t, 16)) for t in range(16)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for L in (20, 50):
        for key in ("ids", "dists", "cmps", "hops"):
            assert np.array_equal(got[L][key], exp[f"{key}_{L}"]), (L, key)
    assert max(combined) > 1  # at least one launch carried several callers' queries
    # switched off: one launch per call, same results
    idx.set_option("combine_single_queries", 0)
    par = mi.Parameters()
    par.Set("L_pq", 50)
    ind, dd = np.zeros(K, np.uint32), np.zeros(K, np.float32)
    c, h = idx.SearchRoarGraph(q[3], K, 3, par, ind, dd)
    assert np.array_equal(ind, exp["ids_50"][3]) and (c, h) == (int(exp["cmps_50"][3]), int(exp["hops_50"][3]))
    idx.close()


def test_combined_calls_keep_their_own_errors(built_lib):
    """Callers that share a launch still fail or succeed on their own: k = 10 comes up short on a 3-node component
    ('not enough results: 3, expected: 10', index_bipartite.cpp:2408-2412) while concurrent k = 3 callers succeed."""
    import threading

    from mysteryann_amd import index as mi
    from mysteryann_amd.capi import ERR_SHORT

    n, d = 64, 16
    rng = np.random.default_rng(1)
    base = rng.standard_normal((n, d)).astype(np.float32)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = 2  # only node 0 has (two) neighbours
    idx = mi.IndexBipartite(d, n, 0, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(0, off, np.array([1, 2], np.uint32))
    want = idx.SearchRoarGraphBatch(base[:8], 3, 20)
    res = {}

    def worker(t):
        par = mi.Parameters()
        par.Set("L_pq", 20)
        out = []
        for rep in range(20):
            k = 10 if t % 2 else 3
            ind, dd = np.zeros(k, np.uint32), np.zeros(k, np.float32)
            try:
                idx.SearchRoarGraph(base[t], k, t, par, ind, dd)
                out.append(("ok", ind.copy()))
            except mi.RoarError as e:
                out.append((e.code, str(e)))
        res[t] = out

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for t in range(8):
        for tag, val in res[t]:
            if t % 2:
                assert tag == ERR_SHORT and val == "not enough results: 3, expected: 10", (t, tag, val)
            else:
                assert tag == "ok" and np.array_equal(val, want["ids"][t]), (t, tag, val)
    idx.close()


def test_builder_row_keep_cuts_long_rows_to_their_nearest_neighbours(built_lib):
    """row_keep: rows collect proposals up to max_degree and are cut back to their row_keep nearest neighbours
    (index metric), nearest first.  Checked against numpy on every long row; the graph must still answer queries
    with the CPU checker's results."""
    import torch

    import oracle
    from mysteryann_amd import fileio, synth
    from mysteryann_amd import index as mi
    from tests.golden.small_builder import exact_knn

    n, d, keep = 30_000, 200, 48
    base, q, learn = synth.t2i_numpy(n, 200, d, seed=3, n_learn=n)
    idx = mi.IndexBipartite(d, n, mi.INNER_PRODUCT, None)
    idx.SetBaseData(base)
    lt = torch.from_numpy(fileio.pad_rows(learn)).cuda()
    idx.BuildBootstrapGraph(learn=lt, max_degree=160, row_keep=keep)
    ep, off, nbrs = idx.GetProjectionGraph()
    deg = np.diff(off.astype(np.int64))
    # rows may exceed keep only through the orphan rescue / supply of the last phase's tail: none here
    assert deg.max() <= keep and deg.mean() > 15
    full = np.nonzero(deg == keep)[0]
    assert len(full) > 50  # the hubs of this set
    for v in full[:200]:
        row = nbrs[int(off[v]):int(off[v + 1])]
        dots = base[row].astype(np.float64) @ base[v].astype(np.float64)
        assert (np.diff(dots) <= 1e-6 * np.abs(dots).max()).all(), v  # nearest (largest inner product) first
        assert v not in row and len(set(row.tolist())) == len(row)
    gt, _ = exact_knn(q, base, 10, 1)
    bp, qp = fileio.pad_rows(base), fileio.pad_rows(q)
    for L in (20, 60):
        r = idx.SearchRoarGraphBatch(q, K, L)
        o = oracle.search_batch(bp, off, nbrs, ep, qp, K, L, 1, threads=4)
        assert np.array_equal(r["ids"], o["ids"]) and np.array_equal(r["dists"], o["dists"]), L
    assert oracle.recall(r["ids"], gt, 10) > 0.9
    idx.close()


def test_a_dirty_launch_cell_is_detected_not_searched_on(golden_dir, built_lib):
    """A launch whose cell was left dirty by a launch that did not finish (a device fault; more than 256 launches in flight) refuses to
    search -- 0xFFFFFFFF in *short_count_dev, ROAR_ERR_DEVICE from the host-buffer call -- instead of silently skipping the queries
    the dead launch had taken; the host-buffer call puts the cells back, option reset_launch_cells does for the device-buffer form."""
    import torch

    from mysteryann_amd import fileio
    from mysteryann_amd.capi import ERR_DEVICE, RoarError

    exp = np.load(os.path.join(golden_dir, "ip_200.expected.npz"))
    idx, q = _index("ip_200", golden_dir, 2)
    r = idx.SearchRoarGraphBatch(q, K, 50)
    assert np.array_equal(r["ids"], exp["ids_50"])
    # host-buffer call: error, cells reset, next call fine
    idx.set_option("debug_dirty_launch_cell", 17)
    with pytest.raises(RoarError) as e:
        idx.SearchRoarGraphBatch(q, K, 50)
    assert e.value.code == ERR_DEVICE and "launch cell" in str(e.value)
    r = idx.SearchRoarGraphBatch(q, K, 50)
    assert np.array_equal(r["ids"], exp["ids_50"]) and np.array_equal(r["dists"], exp["dists_50"])
    # device-buffer call: the marker, results untouched; after the reset the same buffers are filled
    qd = torch.from_numpy(fileio.pad_rows(q)).cuda()
    out = idx.SearchRoarGraphBatchDevice(qd, K, 50)
    torch.cuda.synchronize()
    out["ids"].fill_(-7)
    idx.set_option("debug_dirty_launch_cell", 3)
    idx.SearchRoarGraphBatchDevice(qd, K, 50, out=out)
    torch.cuda.synchronize()
    assert out["short"].cpu().numpy().view(np.uint32)[0] == 0xFFFFFFFF
    assert (out["ids"].cpu().numpy() == -7).all()
    idx.set_option("reset_launch_cells", 1)
    idx.SearchRoarGraphBatchDevice(qd, K, 50, out=out)
    torch.cuda.synchronize()
    assert out["short"].cpu().numpy()[0] == 0
    assert np.array_equal(out["ids"].cpu().numpy().view(np.uint32), exp["ids_50"])
    idx.close()


@pytest.mark.parametrize("d", [512, 128])
def test_edge_floats_form_changes_no_result(d, built_lib):
    """Option edge_floats (round 5): inner product over rows whose screening codes fill whole lines -- the scale and bounds of every
    neighbour's screening row travel with the adjacency row (RoarSearchParams::adj_meta) instead of being read beside the code rows.
    With it and without it: ids, distances, hops, cmps equal, and equal to the oracle's; per-launch call and batch stream; a graph
    with rows of up to 120 neighbours (second chunk) and zero-degree nodes; pools in registers and in LDS."""
    import oracle
    import torch

    from mysteryann_amd import index as mi

    n, nq = 6000, 200
    rng = np.random.default_rng(d)
    base = rng.standard_normal((n, d)).astype(np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    deg = rng.integers(0, 121, n)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(deg)
    nbrs = rng.integers(0, n, int(off[-1])).astype(np.uint32)
    idx = mi.IndexBipartite(d, n, 1, None)
    idx.SetBaseData(base)
    idx.SetProjectionGraph(9, off, nbrs)
    idx.set_option("calibrate", 0), idx.set_option("screen", 2), idx.set_option("visited_mode", 3)
    qd = torch.from_numpy(q).cuda()
    for L in (20, 64, 65, 300):
        o = oracle.search_batch(base, off, nbrs, 9, q, K, L, 1, threads=8)
        res = {}
        for ef in (1, 0):
            idx.set_option("edge_floats", ef)
            r = idx.SearchRoarGraphBatch(q, K, L)
            assert idx.info("last_edge_floats") == (0 if ef else 1), (d, L, ef)
            assert idx.info("last_screen") == 1
            for key in ("ids", "dists", "hops"):
                assert np.array_equal(r[key], o[key]), (d, L, ef, key)
            res[ef] = r
        assert np.array_equal(res[0]["cmps"], res[1]["cmps"]), (d, L)
        with idx.OpenSearchStream(K, L) as st:  # (edge floats on)
            t, out = st.submit(qd)
            assert st.wait(t) == 0
            assert np.array_equal(out["ids"].cpu().numpy().view(np.uint32), o["ids"]), (d, L, "stream")
            assert np.array_equal(out["dists"].cpu().numpy(), o["dists"]), (d, L, "stream")
    # a new graph and new base rows forget the floats of the old ones
    idx.SetProjectionGraph(3, off, nbrs[::-1].copy())
    o = oracle.search_batch(base, off, nbrs[::-1].copy(), 3, q, K, 50, 1, threads=8)
    r = idx.SearchRoarGraphBatch(q, K, 50)
    assert idx.info("last_edge_floats") == 1 and np.array_equal(r["ids"], o["ids"]) and np.array_equal(r["dists"], o["dists"])
    idx.SetBaseData(base[::-1].copy())
    o = oracle.search_batch(base[::-1].copy(), off, nbrs[::-1].copy(), 3, q, K, 50, 1, threads=8)
    r = idx.SearchRoarGraphBatch(q, K, 50)
    assert idx.info("last_edge_floats") == 1 and np.array_equal(r["ids"], o["ids"]) and np.array_equal(r["dists"], o["dists"])
    idx.close()
