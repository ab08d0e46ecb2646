"""world_size-2 gloo tests (CPU) of the N > 1 plumbing: query sharding for the replicated-index
configuration and gather layout + id rebasing for the index-sharded one.  The per-shard searches are
done by the oracle (it is the checker for this path: 'merge of per-shard CPU searches');
the product's merge kernel itself is covered on the GPU (tests/test_gpu_parity.py::test_merge_topk_kernel)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mysteryann_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _numpy_merge(ids_all, d_all, k):
    S, nq, _ = ids_all.shape
    oi = np.zeros((nq, k), np.uint32)
    od = np.zeros((nq, k), np.float32)
    for q in range(nq):
        ci = ids_all[:, q].reshape(-1)
        cd = d_all[:, q].reshape(-1)
        keep = ci != 0xFFFFFFFF
        ci, cd = ci[keep], cd[keep]
        o = np.lexsort((ci, cd))[:k]
        oi[q], od[q] = ci[o], cd[o]
    return oi, od


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle

    assert parallel.ranks_seen() == world  # bench.py's config.rccl_ranks_seen (there over RCCL)
    rng = np.random.default_rng(42)  # same data on every rank
    n, d, nq, k, L = 4000, 64, 24, 10, 60
    base = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    # ---- index-sharded: each rank owns base[lo:hi] with its own random graph (local ids)
    lo, hi = parallel.shard_range(n, world, rank)
    srng = np.random.default_rng(100 + rank)
    m = hi - lo
    deg = srng.integers(4, 24, m)
    off = np.zeros(m + 1, np.uint64)
    off[1:] = np.cumsum(deg)
    nbrs = srng.integers(0, m, int(off[-1])).astype(np.uint32)
    r = oracle.search_batch(base[lo:hi], off, nbrs, 0, q, k, L, 0)
    ids_local = torch.from_numpy(r["ids"].astype(np.int64))
    gids = parallel.rebase_ids(ids_local, lo)
    ids_all, d_all = parallel.allgather_topk(gids, torch.from_numpy(r["dists"]))
    assert ids_all.shape == (world, nq, k)
    assert torch.equal(ids_all[rank], gids)  # own slice in its own slot
    for s in range(world):  # every slot holds ids of that shard's range
        slo, shi = parallel.shard_range(n, world, s)
        assert int(ids_all[s].min()) >= slo and int(ids_all[s].max()) < shi
    mi, md = _numpy_merge(ids_all.numpy().astype(np.uint32), d_all.numpy(), k)
    # merged distances are exact L2 distances of the merged ids, sorted by (dist, id)
    for qi in range(nq):
        want = np.array([oracle.dist(0, base[i], q[qi]) for i in mi[qi]], np.float32)
        assert np.array_equal(want, md[qi])
        assert all((md[qi][j], mi[qi][j]) <= (md[qi][j + 1], mi[qi][j + 1]) for j in range(k - 1))
    # every rank computed the same merged answer
    t = torch.from_numpy(mi.astype(np.int64))
    ref = t.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(t, ref)
    # ---- replicated index, ONE query block split by query (BASELINE configs[3]): rank r searches its slice of the same
    # block on the same graph; the gathered slices must be the single-process answer, row for row
    goff = np.zeros(n + 1, np.uint64)
    grng = np.random.default_rng(7)
    goff[1:] = np.cumsum(grng.integers(4, 24, n))
    gnbrs = grng.integers(0, n, int(goff[-1])).astype(np.uint32)
    nq2 = 25  # not divisible by the world size: slices of 12 and 13 rows
    q2 = rng.standard_normal((nq2, d)).astype(np.float32)
    qlo, qhi = parallel.shard_range(nq2, world, rank)
    mine = oracle.search_batch(base, goff, gnbrs, 3, q2[qlo:qhi], k, L, 0)
    ids_g = parallel.gather_query_slices(torch.from_numpy(mine["ids"].astype(np.int64)), nq2)
    d_g = parallel.gather_query_slices(torch.from_numpy(mine["dists"]), nq2)
    hops_g = parallel.gather_query_slices(torch.from_numpy(mine["hops"].astype(np.int64)), nq2)
    whole = oracle.search_batch(base, goff, gnbrs, 3, q2, k, L, 0)
    assert np.array_equal(ids_g.numpy(), whole["ids"].astype(np.int64))
    assert np.array_equal(d_g.numpy(), whole["dists"])
    assert np.array_equal(hops_g.numpy(), whole["hops"].astype(np.int64))
    cover = torch.zeros(nq2, dtype=torch.int64)
    cover[qlo:qhi] = 1
    dist.all_reduce(cover)
    assert bool((cover == 1).all())
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


def test_world2_gloo_sharding_and_gather(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_rebase_keeps_short_result_filler():
    a = np.array([[1, 2, 0xFFFFFFFF]], np.uint32)
    assert parallel.rebase_ids(a, 10).tolist() == [[11, 12, 0xFFFFFFFF]]
    t = torch.tensor([[1, 2, -1]], dtype=torch.int32)
    assert parallel.rebase_ids(t, 10).tolist() == [[11, 12, -1]]
    b = np.array([[1, 2, -1]], np.int32)  # the same bit patterns in a signed numpy array
    assert parallel.rebase_ids(b, 10).tolist() == [[11, 12, -1]]
    assert parallel.shard_range(10, 4, 0) == (0, 2) and parallel.shard_range(10, 4, 3) == (7, 10)
