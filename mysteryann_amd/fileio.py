"""numpy readers / writers for the reference's on-disk formats (host-side helpers for tests,
bench data generation and the Python mirror of the harness).  The product's own loaders are C++
(mysteryann_amd/csrc/roar_capi.cpp); these exist so Python can write inputs for them.

Formats (all little-endian, no magic / padding), /root/reference citations:
  .fbin        u32 n; u32 d; f32[n*d]                         include/efanna2e/util.h:106-127,179-211
  gt .ibin     u32 n; u32 K; u32 ids[n*K]; f32 dists[n*K]     include/efanna2e/util.h:84-105,129-155
  knn .ibin    u32 n; u32 K; u32 ids[n*K]                     src/index_bipartite.cpp:2622-2639
  graph index  u32 ep; u32 npts; npts x {u32 deg; u32 nbr[deg]}  src/index_bipartite.cpp:2097-2117,2606-2619
"""
import numpy as np


def write_fbin(path, arr):
    arr = np.ascontiguousarray(arr, np.float32)
    with open(path, "wb") as f:
        np.array(arr.shape, np.uint32).tofile(f)
        arr.tofile(f)


def read_fbin(path):
    with open(path, "rb") as f:
        n, d = np.fromfile(f, np.uint32, 2)
        a = np.fromfile(f, np.float32, int(n) * int(d))
    if a.size != int(n) * int(d):
        raise RuntimeError("Data file size wrong!")
    return a.reshape(int(n), int(d))


def pad_rows(arr, factor=8):
    """load_data / data_align: row stride rounded up to a multiple of 8 floats, zero filled (util.h:37-75)."""
    n, d = arr.shape
    nd = (d + factor - 1) // factor * factor
    if nd == d:
        return np.ascontiguousarray(arr, np.float32)
    out = np.zeros((n, nd), np.float32)
    out[:, :d] = arr
    return out


def write_gt(path, ids, dists):
    ids = np.ascontiguousarray(ids, np.uint32)
    dists = np.ascontiguousarray(dists, np.float32)
    with open(path, "wb") as f:
        np.array(ids.shape, np.uint32).tofile(f)
        ids.tofile(f)
        dists.tofile(f)


def read_gt(path):
    with open(path, "rb") as f:
        n, k = (int(x) for x in np.fromfile(f, np.uint32, 2))
        ids = np.fromfile(f, np.uint32, n * k)
        dists = np.fromfile(f, np.float32, n * k)
    if ids.size != n * k or dists.size != n * k:
        raise RuntimeError("Data file size wrong!")
    return ids.reshape(n, k), dists.reshape(n, k)


def write_knn(path, ids):
    ids = np.ascontiguousarray(ids, np.uint32)
    with open(path, "wb") as f:
        np.array(ids.shape, np.uint32).tofile(f)
        ids.tofile(f)


def write_graph(path, ep, adj):
    """adj: list of u32 arrays (out-neighbours, order preserved) or (offsets, nbrs) CSR pair."""
    if isinstance(adj, tuple):
        offsets, nbrs = adj
        n = len(offsets) - 1
        degs = np.diff(np.asarray(offsets, np.int64)).astype(np.uint32)
        nbrs = np.asarray(nbrs, np.uint32)
    else:
        n = len(adj)
        degs = np.array([len(a) for a in adj], np.uint32)
        nbrs = np.concatenate([np.asarray(a, np.uint32) for a in adj]) if n else np.zeros(0, np.uint32)
    # interleave {deg, nbrs...} per node without a Python loop
    off = np.zeros(n + 1, np.int64)
    off[1:] = np.cumsum(degs.astype(np.int64))
    out = np.zeros(2 + n + int(off[-1]), np.uint32)
    out[0], out[1] = ep, n
    pos = 2 + np.arange(n, dtype=np.int64) + off[:-1]  # position of each deg word
    out[pos] = degs
    if off[-1]:
        edge_node = np.repeat(np.arange(n, dtype=np.int64), degs.astype(np.int64))
        edge_pos = pos[edge_node] + 1 + (np.arange(off[-1], dtype=np.int64) - off[:-1][edge_node])
        out[edge_pos] = nbrs
    out.tofile(path)


def read_graph(path):
    """-> ep, offsets int64[n+1], nbrs u32."""
    raw = np.fromfile(path, np.uint32)
    ep, n = int(raw[0]), int(raw[1])
    offsets = np.zeros(n + 1, np.int64)
    degs = np.zeros(n, np.int64)
    p = 2
    # sequential walk (deg words are data dependent); vectorised in chunks is not possible, but n is small in tests
    for i in range(n):
        d = int(raw[p])
        degs[i] = d
        p += 1 + d
    if p != raw.size:
        raise RuntimeError("graph file truncated")
    offsets[1:] = np.cumsum(degs)
    pos = 2 + np.arange(n, dtype=np.int64) + offsets[:-1]
    mask = np.ones(raw.size, bool)
    mask[:2] = False
    mask[pos] = False
    return ep, offsets, raw[mask].copy()
