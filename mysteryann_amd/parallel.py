"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

Replicated index (BASELINE configs[3]): ONE query block, rank r searches rows `shard_range(nq, world, r)` of it with
no collective in the data path (`gather_query_slices` only assembles the answer for checking).  Index-sharded (configs[4], base set larger than one GPU): every rank searches ALL
queries on its own shard (own graph, local ids), local ids are rebased to global, the per-shard
top-k lists are all-gathered (nq x k x 8 B per rank: latency bound on xGMI) and merged by
(distance, id) with the HIP merge kernel.  Not in the reference; the oracle for it is
the same merge over per-shard CPU searches.
"""
import ctypes as C

INVALID = 0xFFFFFFFF


def shard_range(n, world, rank):
    """Contiguous id range [lo, hi) of shard `rank` (same split for base rows and for queries)."""
    return rank * n // world, (rank + 1) * n // world


def rebase_ids(ids, offset):
    """local -> global ids; the short-result filler 0xFFFFFFFF stays.  torch int32/int64 or numpy."""
    try:
        import torch

        if isinstance(ids, torch.Tensor):
            if ids.dtype == torch.int32:  # u32 bit pattern in an i32 tensor: filler is -1
                return torch.where(ids == -1, ids, ids + int(offset))
            return torch.where(ids == INVALID, ids, ids + int(offset))
    except ImportError:
        pass
    import numpy as np

    ids = np.asarray(ids)
    if ids.dtype == np.int32:  # u32 bit patterns in a signed array (e.g. .cpu().numpy() of the device results): filler is -1
        u = ids.view(np.uint32)
        return np.where(u == INVALID, u, u + np.uint32(offset)).astype(np.uint32).view(np.int32)
    return np.where(ids == INVALID, ids, ids + ids.dtype.type(offset)).astype(ids.dtype)


def ranks_seen(device=None, group=None):
    """How many ranks a collective really spans: a one-element all-reduce of ones (over RCCL on the GPU box: the first collective
    bench.py issues, reported as config.rccl_ranks_seen; over gloo in the CPU tests).  1 without an initialised process group."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return 1
    t = torch.ones(1, dtype=torch.int32, device=device if device is not None else "cpu")
    dist.all_reduce(t, group=group)
    return int(t.item())


def gather_query_slices(local, nq, group=None):
    """Replicated configuration: every rank holds the rows `shard_range(nq, world, rank)` of a per-query result
    (tensor [hi - lo, ...]); returns the whole [nq, ...] tensor on every rank.  Slices differ in length by at most one
    row, so they are padded to the longest for the all-gather.  Checking / reporting only -- the search itself needs no
    exchange."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    longest = max(shard_range(nq, world, r)[1] - shard_range(nq, world, r)[0] for r in range(world))
    pad = torch.zeros((longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = torch.empty((nq,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        lo, hi = shard_range(nq, world, r)
        out[lo:hi] = parts[r][:hi - lo]
    return out


def allgather_topk(ids, dists, group=None):
    """all_gather of per-shard results -> (ids [world, nq, k], dists [world, nq, k]) on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    ids_all = torch.empty((world,) + tuple(ids.shape), dtype=ids.dtype, device=ids.device)
    d_all = torch.empty((world,) + tuple(dists.shape), dtype=dists.dtype, device=dists.device)
    dist.all_gather_into_tensor(ids_all, ids.contiguous(), group=group) if ids.is_cuda else dist.all_gather(
        list(ids_all.unbind(0)), ids.contiguous(), group=group)
    dist.all_gather_into_tensor(d_all, dists.contiguous(), group=group) if dists.is_cuda else dist.all_gather(
        list(d_all.unbind(0)), dists.contiguous(), group=group)
    return ids_all, d_all


def merge_topk_device(ids_all, dists_all, stream=None):
    """HIP k-way merge (roar_merge_topk_device).  ids_all int32 [S, nq, k] (u32 bit patterns), CUDA."""
    import torch

    from . import capi

    assert ids_all.is_cuda and dists_all.is_cuda, "the merge runs on the GPU; there is no CPU path"
    S, nq, k = ids_all.shape
    out_i = torch.empty((nq, k), dtype=ids_all.dtype, device=ids_all.device)
    out_d = torch.empty((nq, k), dtype=torch.float32, device=ids_all.device)
    st = stream if stream is not None else torch.cuda.current_stream(ids_all.device)
    capi.check(capi.lib().roar_merge_topk_device(ids_all.device.index or 0, S, nq, k, C.c_void_p(ids_all.data_ptr()),
                                                 C.c_void_p(dists_all.data_ptr()), C.c_void_p(out_i.data_ptr()),
                                                 C.c_void_p(out_d.data_ptr()), C.c_void_p(st.cuda_stream)))
    return out_i, out_d


def sharded_search(idx, queries, k, L_pq, shard_offset, group=None):
    """configs[4]: search this rank's shard index, rebase, all-gather over RCCL, merge on the GPU."""
    out = idx.SearchRoarGraphBatchDevice(queries, k, L_pq)
    ids = rebase_ids(out["ids"], shard_offset)
    ids_all, d_all = allgather_topk(ids, out["dists"], group)
    return merge_topk_device(ids_all, d_all)
