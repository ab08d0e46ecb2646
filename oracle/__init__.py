"""ctypes front-end for the CPU checker (oracle/).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  mysteryann_amd/ never imports this package.

Two implementations behind the same Python functions:
  impl="port"  liboracle.so        plain-C restatement (oracle/roar_oracle.c)
  impl="ref"   _ref/libroar_ref.so the reference's own distance / queue / visited-pool /
                                   loader headers compiled where they lie (oracle/ref_driver.cpp)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PORT = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libroar_ref.so")

METRIC_L2, METRIC_IP, METRIC_COSINE = 0, 1, 4  # include/efanna2e/distance.h:15

_libs = {}


def build(force=False):
    """Compile liboracle.so (always possible) and _ref (only where /root/reference exists)."""
    srcs = [os.path.join(_HERE, f) for f in ("roar_oracle.c", "roar_builder_oracle.c")]
    if force or not os.path.exists(_PORT) or os.path.getmtime(_PORT) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, os.path.join(_HERE, "liboracle.so")],
                              stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/include/efanna2e") and (force or not os.path.exists(_REF)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def _cpu_has_avx512():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    fl = set(line.split(":", 1)[1].split())
                    return {"avx512f", "avx512dq", "avx512bw", "avx512vl", "avx512cd", "fma"} <= fl
    except OSError:
        pass
    return False


def have_ref():
    return os.path.exists(_REF) and _cpu_has_avx512()


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def lib(impl="port"):
    if impl in _libs:
        return _libs[impl]
    if impl == "port":
        build()
        l = C.CDLL(_PORT)
        pre = "oracle_"
    elif impl == "ref":
        if not have_ref():
            raise RuntimeError("oracle/_ref/libroar_ref.so unavailable (not built, or host lacks AVX-512)")
        l = C.CDLL(_REF)
        pre = "ref_"
    else:
        raise ValueError(impl)
    fp, u32, u64 = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    sb = getattr(l, pre + "search_batch")
    sb.restype = C.c_int
    sb.argtypes = [fp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, u64, u32, C.c_uint32, fp, C.c_uint64,
                   C.c_uint64, C.c_uint32, C.c_uint32, u32, fp, u32, u32, u32, C.c_int]
    d = getattr(l, pre + "dist")
    d.restype = C.c_float
    d.argtypes = [C.c_int, fp, fp, C.c_uint]
    tr = getattr(l, pre + "pq_trace")
    tr.restype = None
    tr.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8), u32, fp, u32]
    nm = getattr(l, pre + "normalize")
    nm.restype = None
    nm.argtypes = [fp, C.c_uint64]
    getattr(l, pre + "max_threads").restype = C.c_int
    bs = getattr(l, pre + "search_bipartite_batch")
    bs.restype = C.c_int
    bs.argtypes = [fp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, u64, u32, fp, C.c_uint64, C.c_uint64, C.c_uint32,
                   C.c_uint32, u32, C.c_uint32, u32, fp, u32, u32, C.c_int]
    l._pre = pre
    _libs[impl] = l
    return l


def max_threads(impl="port"):
    l = lib(impl)
    return getattr(l, l._pre + "max_threads")()


def dist(metric, a, b, d=None, impl="port"):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    l = lib(impl)
    return float(getattr(l, l._pre + "dist")(int(metric), _f32p(a), _f32p(b), int(d if d is not None else a.size)))


def dist_many(metric, A, B, d=None, impl="port"):
    A = np.ascontiguousarray(A, np.float32)
    B = np.ascontiguousarray(B, np.float32)
    d = int(d if d is not None else A.shape[1])
    return np.array([dist(metric, A[i], B[i], d, impl) for i in range(A.shape[0])], np.float32)


def pq_trace(capacity, ops, ids, dists, impl="port"):
    ops = np.ascontiguousarray(ops, np.uint8)
    ids = np.ascontiguousarray(ids, np.uint32)
    dists = np.ascontiguousarray(dists, np.float32)
    snap = np.zeros((len(ops), 2 + 3 * capacity), np.uint32)
    l = lib(impl)
    getattr(l, l._pre + "pq_trace")(int(capacity), len(ops), ops.ctypes.data_as(C.POINTER(C.c_uint8)), _u32p(ids),
                                    _f32p(dists), _u32p(snap))
    return snap


def normalize(arr, impl="port"):
    arr = np.array(arr, np.float32, copy=True)
    l = lib(impl)
    getattr(l, l._pre + "normalize")(_f32p(arr), arr.size)
    return arr


def search_batch(base, offsets, nbrs, ep, queries, k, L, metric, threads=1, impl="port", dim=None):
    """base [n, stride] f32, queries [nq, qstride] f32, CSR graph (offsets u64[n+1], nbrs u32).

    Returns dict(ids [nq,k] u32, dists [nq,k] f32, cmps, hops, scanned [nq] u32, short int)."""
    base = np.ascontiguousarray(base, np.float32)
    queries = np.ascontiguousarray(queries, np.float32)
    offsets = np.ascontiguousarray(offsets, np.uint64)
    nbrs = np.ascontiguousarray(nbrs, np.uint32)
    n, stride = base.shape
    nq, qstride = queries.shape
    dim = int(dim if dim is not None else stride)
    ids = np.zeros((nq, k), np.uint32)
    dists = np.zeros((nq, k), np.float32)
    cmps = np.zeros(nq, np.uint32)
    hops = np.zeros(nq, np.uint32)
    scanned = np.zeros(nq, np.uint32)
    l = lib(impl)
    rc = getattr(l, l._pre + "search_batch")(_f32p(base), n, stride, dim, int(metric), _u64p(offsets), _u32p(nbrs),
                                             int(ep), _f32p(queries), qstride, nq, int(k), int(L), _u32p(ids),
                                             _f32p(dists), _u32p(cmps), _u32p(hops), _u32p(scanned), int(threads))
    return dict(ids=ids, dists=dists, cmps=cmps, hops=hops, scanned=scanned, short=rc)


# ---- file formats (port: roar_oracle.c A7; ref: util.h loaders) -------------------------------
def fbin_load_padded(path, impl="port"):
    n, d = C.c_uint32(), C.c_uint32()
    p = path.encode()
    if impl == "port":
        l = lib("port")
        l.oracle_fbin_meta.restype = C.c_int
        rc = l.oracle_fbin_meta(p, C.byref(n), C.byref(d))
        if rc:
            raise RuntimeError("Data file size wrong!" if rc == -2 else "open file error")
        nd = (d.value + 7) // 8 * 8
        out = np.zeros((n.value, nd), np.float32)
        rc = l.oracle_fbin_load_padded(p, n, d, _f32p(out))
        if rc:
            raise RuntimeError("Data file size wrong!")
        return out, d.value
    l = lib("ref")
    if l.ref_load_meta(p, C.byref(n), C.byref(d)):
        raise RuntimeError("Data file size wrong!")
    nd = (d.value + 7) // 8 * 8
    out = np.zeros((n.value, nd), np.float32)
    new_dim = C.c_uint32()
    if l.ref_load_data_aligned(p, n, d, 0, _f32p(out), C.byref(new_dim)):
        raise RuntimeError("Data file size wrong!")
    assert new_dim.value == nd
    return out, d.value


def graph_load(path):
    """-> ep, offsets u64[n+1], nbrs u32 (port reader of the index format)."""
    l = lib("port")
    ep, npts, ne = C.c_uint32(), C.c_uint32(), C.c_uint64()
    p = path.encode()
    rc = l.oracle_graph_load(p, C.byref(ep), C.byref(npts), C.byref(ne), None, None)
    if rc:
        raise RuntimeError("cannot open file" if rc == -1 else "graph file truncated")
    offsets = np.zeros(npts.value + 1, np.uint64)
    nbrs = np.zeros(max(ne.value, 1), np.uint32)
    rc = l.oracle_graph_load(p, C.byref(ep), C.byref(npts), C.byref(ne), _u64p(offsets), _u32p(nbrs))
    if rc:
        raise RuntimeError("graph file truncated")
    return ep.value, offsets, nbrs[:ne.value]


def graph_save(path, ep, offsets, nbrs):
    offsets = np.ascontiguousarray(offsets, np.uint64)
    nbrs = np.ascontiguousarray(nbrs, np.uint32)
    l = lib("port")
    if l.oracle_graph_save(path.encode(), C.c_uint32(ep), C.c_uint32(len(offsets) - 1), _u64p(offsets), _u32p(nbrs)):
        raise RuntimeError("cannot open file")


def recall(res, gt, k):
    res = np.ascontiguousarray(res, np.uint32)
    gt = np.ascontiguousarray(gt, np.uint32)
    l = lib("port")
    l.oracle_recall.restype = C.c_float
    return float(l.oracle_recall(C.c_uint32(res.shape[0]), C.c_uint32(k), C.c_uint32(gt.shape[1]), _u32p(res),
                                 _u32p(gt)))


def build_roargraph(base, learn_knn, metric, M_sq=100, M_pjbp=35, L_pjpq=500, flags=0):
    """CPU restatement of the reference's BuildRoarGraph (oracle/roar_builder_oracle.c), single-threaded.
    base [n, stride] f32, learn_knn [n_learn, K] u32 (nearest base ids per training query, nearest first).
    -> ep, offsets u64[n+1], nbrs u32.  A yardstick for graph quality, not pinned to a reference build."""
    base = np.ascontiguousarray(base, np.float32)
    learn_knn = np.ascontiguousarray(learn_knn, np.uint32)
    n, stride = base.shape
    l = lib("port")
    ep = C.c_uint32()
    offsets = np.zeros(n + 1, np.uint64)
    nbrs = np.zeros(n * 3 * M_pjbp + 16, np.uint32)
    l.oracle_build_roargraph.restype = C.c_int
    rc = l.oracle_build_roargraph(_f32p(base), C.c_uint32(n), C.c_uint64(stride), C.c_uint32(stride), C.c_int(metric),
                                  _u32p(learn_knn), C.c_uint32(learn_knn.shape[0]), C.c_uint32(learn_knn.shape[1]),
                                  C.c_uint32(M_sq), C.c_uint32(M_pjbp), C.c_uint32(L_pjpq), C.byref(ep), _u64p(offsets),
                                  _u32p(nbrs), C.c_int(flags))
    assert rc == 0
    return ep.value, offsets, nbrs[:int(offsets[-1])]


def prune_bisearch(base, metric, tgt, cand, M):
    """PruneBiSearchBaseGetBase (src/index_bipartite.cpp:1612-1694) restated: kept ids, in order."""
    base = np.ascontiguousarray(base, np.float32)
    cand = np.ascontiguousarray(cand, np.uint32)
    res = np.zeros(M + 4, np.uint32)
    l = lib("port")
    l.oracle_prune_bisearch.restype = C.c_uint32
    nr = l.oracle_prune_bisearch(_f32p(base), C.c_uint64(base.shape[1]), C.c_uint32(base.shape[1]), C.c_int(metric),
                                 C.c_uint32(tgt), _u32p(cand), C.c_uint32(len(cand)), C.c_uint32(M), _u32p(res))
    return res[:nr].copy()


# ---- bipartite two-hop variant ------------------------------------------------
def search_bipartite_batch(base, offsets, nbrs, queries, k, L, metric, init_ids, threads=1, impl="port", dim=None):
    """SearchBipartiteGraph (src/index_bipartite.cpp:282-356) with explicit entry points [nq, n_init]."""
    base = np.ascontiguousarray(base, np.float32)
    queries = np.ascontiguousarray(queries, np.float32)
    offsets = np.ascontiguousarray(offsets, np.uint64)
    nbrs = np.ascontiguousarray(nbrs, np.uint32)
    init_ids = np.ascontiguousarray(init_ids, np.uint32)
    n, stride = base.shape
    nq, qstride = queries.shape
    dim = int(dim if dim is not None else stride)
    ids = np.zeros((nq, k), np.uint32)
    dists = np.zeros((nq, k), np.float32)
    cmps = np.zeros(nq, np.uint32)
    hops = np.zeros(nq, np.uint32)
    l = lib(impl)
    rc = getattr(l, l._pre + "search_bipartite_batch")(
        _f32p(base), n, stride, dim, int(metric), _u64p(offsets), _u32p(nbrs), _f32p(queries), qstride, nq, int(k),
        int(L), _u32p(init_ids), init_ids.shape[1], _u32p(ids), _f32p(dists), _u32p(cmps), _u32p(hops), int(threads))
    return dict(ids=ids, dists=dists, cmps=cmps, hops=hops, short=rc)


def build_bipartite(learn_knn, n_base, M_pjbp):
    """qbaseNNbipartite (src/index_bipartite.cpp:235-280) -> offsets u64[n_base + n_learn + 1], nbrs u32."""
    learn_knn = np.ascontiguousarray(learn_knn, np.uint32)
    nl, K = learn_knn.shape
    offsets = np.zeros(n_base + nl + 1, np.uint64)
    nbrs = np.zeros(nl * min(K, M_pjbp) + 1, np.uint32)
    l = lib("port")
    l.oracle_build_bipartite.restype = None
    l.oracle_build_bipartite(_u32p(learn_knn), C.c_uint32(nl), C.c_uint32(K), C.c_uint32(n_base), C.c_uint32(M_pjbp),
                             _u64p(offsets), _u32p(nbrs))
    return offsets, nbrs[:int(offsets[-1])]


def bipartite_save(path, offsets, nbrs):
    offsets = np.ascontiguousarray(offsets, np.uint64)
    nbrs = np.ascontiguousarray(nbrs, np.uint32)
    if lib("port").oracle_bipartite_save(path.encode(), C.c_uint32(len(offsets) - 1), _u64p(offsets), _u32p(nbrs)):
        raise RuntimeError("cannot open file")


def bipartite_load(path):
    l = lib("port")
    npts, ne = C.c_uint32(), C.c_uint64()
    rc = l.oracle_bipartite_load(path.encode(), C.byref(npts), C.byref(ne), None, None)
    if rc:
        raise RuntimeError("cannot open file" if rc == -1 else "graph file truncated")
    offsets = np.zeros(npts.value + 1, np.uint64)
    nbrs = np.zeros(max(ne.value, 1), np.uint32)
    if l.oracle_bipartite_load(path.encode(), C.byref(npts), C.byref(ne), _u64p(offsets), _u32p(nbrs)):
        raise RuntimeError("graph file truncated")
    return offsets, nbrs[:ne.value]


def rderr(gt_dists, res_dists, k, metric):
    """ComputeRderr, tests/test_search_bipartite.cpp:38-62."""
    gt_dists = np.ascontiguousarray(gt_dists, np.float32)
    res_dists = np.ascontiguousarray(res_dists, np.float32)
    l = lib("port")
    l.oracle_rderr.restype = C.c_double
    return float(l.oracle_rderr(_f32p(gt_dists), C.c_uint32(gt_dists.shape[1]), _f32p(res_dists),
                                C.c_uint32(res_dists.shape[0]), C.c_uint32(k), C.c_int(metric)))
