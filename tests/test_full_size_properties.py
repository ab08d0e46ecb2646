"""BASELINE.json's full sizes on the GPU box -- configs[1] the Text-to-Image-10M shape (10 M x 200-d, inner product), configs[2] the WebVid shape
(2.5 M x 512-d, inner product, unit rows), configs[0]'s SIFT-1M shape (1 M x 128-d, L2) on the GPU; 10 000 queries each, k = 10 -- through
the C ABI, checked by what does not depend on size.

The oracle cannot search 10 000 queries over 10 M points in seconds, so at this size the HIP path is held to
  * the oracle itself on a HANDFUL of queries (host copies of the 8 GB base block and of the graph: ids, distances, hops, cmps bit for bit),
  * form invariance: the reference's result does not depend on how the visited set is kept (visited_list_pool.h:8-29 is bookkeeping) --
    HBM tags, the exact on-chip table, the forgetful table and AUTO return the same ids / distances / hops, the exact forms the same cmps,
    the forgetful one at least as many,
  * batch invariance: a query's result does not depend on which batch it arrives in (src/index_bipartite.cpp:2311-2420 has no
    cross-query state) -- a slice alone, the whole block, a stream of ragged batches,
  * what a result row IS: k ids, distinct, ascending by (distance, id) (neighbor.h:29-31), every distance = distance_->compare of that
    id (roar_distances: the reference's arithmetic), nothing below the row's k-th distance missing from it that the search scored --
    checked through recall against an independent exact brute force (roar_exact_knn_device), non-decreasing in L_pq,
  * determinism: the same call twice.
The graphs are the reference's builder on exact lists of n / 5 training queries (T2I: a fifth of bench.py's -- the properties do not
need the benchmark's operating point, the box's minutes do)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NQ, K = 10_000, 10
# shape: points, dimension, metric (distance.h:15), recall@10 floor at L_pq = 100 on this graph
SHAPES = {"t2i-10M": (10_000_000, 200, 1, 0.93), "webvid-2.5M": (2_500_000, 512, 1, 0.85), "sift-1M": (1_000_000, 128, 0, 0.75)}


@pytest.fixture(scope="module", params=list(SHAPES))
def full(request, built_lib):
    """one BASELINE shape at full size: data, the reference-shaped index, exact ground truth (every test below runs on all three)"""
    import torch

    import bench
    from mysteryann_amd import index as mi, synth

    n, d, metric, floor = SHAPES[request.param]
    gen = {"t2i-10M": synth.t2i_torch, "webvid-2.5M": synth.webvid_torch, "sift-1M": synth.sift_torch}[request.param]
    base, queries, learn = gen(n, NQ, d, n_learn=n // 5)
    idx = mi.IndexBipartite(d, n, metric, None)
    idx.SetBaseData(base)
    bench.build_reference_graph(torch, mi, idx, learn, n // 5, 600, lambda m: None)
    del learn
    torch.cuda.synchronize()
    gt = idx.ExactKnnDevice(queries, K, want_dists=False)[0]
    yield dict(idx=idx, base=base, queries=queries, gt=gt, n=n, metric=metric, floor=floor, shape=request.param)
    idx.close()
    del idx, base, queries, gt
    torch.cuda.empty_cache()


def _run(idx, q, L, **opts):
    import torch

    keep = {}
    for key, val in opts.items():
        keep[key] = {"visited_mode": 2, "calibrate": 1, "screen": 0, "vis_entries": 0}[key]
        idx.set_option(key, val)
    out = idx.SearchRoarGraphBatchDevice(q, K, L)
    torch.cuda.synchronize()
    for key, val in keep.items():
        idx.set_option(key, val)
    assert int(out["short"].item()) == 0
    return {k_: v_.cpu().numpy() for k_, v_ in out.items() if k_ != "short"}


def _recall(ids, gt):
    return float((ids[:, :, None] == gt[:, None, :]).any(axis=2).sum()) / gt.size


def test_a_handful_of_queries_against_the_oracle_at_full_size(full):
    import oracle

    idx, queries = full["idx"], full["queries"]
    ep, off, nbrs = idx.GetProjectionGraph()
    hb = full["base"].cpu().numpy()
    hq = queries[:24].cpu().numpy()
    for L in (10, 50, 200):
        o = oracle.search_batch(hb, off, nbrs, ep, hq, K, L, full["metric"], threads=8)
        r = _run(idx, queries[:24].contiguous(), L, visited_mode=0)  # HBM tags: cmps are the reference's as well
        for key in ("ids", "dists", "hops", "cmps"):
            assert np.array_equal(r[key].view(o[key].dtype), o[key]), (L, key)
        a = _run(idx, queries[:24].contiguous(), L)  # AUTO
        for key in ("ids", "dists", "hops"):
            assert np.array_equal(a[key].view(o[key].dtype), o[key]), (L, key, "auto")


def test_visited_forms_agree_at_full_size(full):
    idx, queries = full["idx"], full["queries"]
    for L in (50, 300):
        exact = _run(idx, queries, L, visited_mode=0)
        for name, opts in (("on-chip table, exact", dict(visited_mode=1)), ("forgetful table", dict(visited_mode=3)), ("auto", {}),
                           ("forgetful, 1 024 tags, no screen", dict(visited_mode=3, vis_entries=1024, screen=1, calibrate=0))):
            r = _run(idx, queries, L, **opts)
            for key in ("ids", "dists", "hops", "scanned"):
                assert np.array_equal(r[key], exact[key]), (L, name, key)
            if opts.get("visited_mode") == 1:
                assert np.array_equal(r["cmps"], exact["cmps"]), (L, name)
            else:
                assert (r["cmps"].view(np.uint32) >= exact["cmps"].view(np.uint32)).all(), (L, name)


def test_a_result_row_is_what_the_reference_says_it_is(full):
    idx, queries, gt = full["idx"], full["queries"], full["gt"].cpu().numpy()
    hq = queries.cpu().numpy()
    recalls = []
    for L in (10, 20, 50, 100):
        r = _run(idx, queries, L)
        ids, dists = r["ids"].view(np.uint32), r["dists"]
        assert (ids < full["n"]).all()
        srt = np.sort(ids, axis=1)
        assert (srt[:, 1:] != srt[:, :-1]).all(), (L, "an id twice in a row of results")
        # ascending by (distance, id): neighbor.h:29-31
        d0, d1, i0, i1 = dists[:, :-1], dists[:, 1:], ids[:, :-1], ids[:, 1:]
        assert ((d0 < d1) | ((d0 == d1) & (i0 < i1))).all(), (L, "order")
        # every distance is distance_->compare(base[id], query) in the reference's own summation order
        sub = slice(0, 2000)
        assert np.array_equal(idx.Distances(hq[sub], ids[sub]), dists[sub]), (L, "distances")
        recalls.append(_recall(ids.astype(np.int64), gt.astype(np.int64)))
        assert (r["hops"].view(np.uint32) >= min(L, 2)).all() and (r["cmps"].view(np.uint32) >= r["hops"].view(np.uint32)).all()
    assert all(b >= a - 1e-4 for a, b in zip(recalls, recalls[1:])), recalls  # a longer pool never finds less
    assert recalls[3] >= full["floor"], (full["shape"], recalls)  # (bench.py's operating points: 0.9 at L_pq 50 / 95 / 170)


def test_a_query_s_result_does_not_depend_on_its_batch(full):
    import torch

    idx, queries = full["idx"], full["queries"]
    L = 50
    whole = _run(idx, queries, L, visited_mode=1)  # (exact counting: cmps comparable between launches of any size)
    again = _run(idx, queries, L, visited_mode=1)
    for key in ("ids", "dists", "hops", "cmps", "scanned"):
        assert np.array_equal(whole[key], again[key]), ("the same call twice", key)
    for lo, hi in ((0, 1), (4321, 4358), (9000, 10000)):
        part = _run(idx, queries[lo:hi].contiguous(), L, visited_mode=1)
        for key in ("ids", "dists", "hops", "cmps", "scanned"):
            assert np.array_equal(part[key], whole[key][lo:hi]), (lo, hi, key)
    # the same block as ragged batches through one batch stream (forgetful table: ids / distances / hops)
    cuts = [0, 1, 38, 2500, 2501, 7000, 9999, 10000]
    with idx.OpenSearchStream(K, L) as st:
        pend = [(st.submit(queries[a:b]), a, b) for a, b in zip(cuts, cuts[1:])]
        for (t, out), a, b in pend:
            assert st.wait(t) == 0
            for key in ("ids", "dists", "hops"):
                assert np.array_equal(out[key].cpu().numpy(), whole[key][a:b]), (a, b, key)
    torch.cuda.synchronize()
