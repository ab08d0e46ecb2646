"""Python mirror of the reference's operator interface for the search path -- same names and
argument meaning as efanna2e::IndexBipartite (include/index_bipartite.h) and efanna2e::Parameters
(include/efanna2e/parameters.h), so that the parity tests read like the reference's harness
(tests/test_search_roargraph.cpp:160-209).  Every method is a thin call into the C ABI.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import METRIC_COSINE, METRIC_IP, METRIC_L2, RoarError  # noqa: F401

L2, INNER_PRODUCT, COSINE = METRIC_L2, METRIC_IP, METRIC_COSINE  # efanna2e::Metric, distance.h:15


class Parameters:
    """efanna2e::Parameters (parameters.h:15-57): string-valued map with typed Get."""

    def __init__(self):
        self._p = {}

    def Set(self, name, value):
        self._p[name] = str(value)

    def Get(self, name, typ=int, default=None):
        if name not in self._p:
            if default is not None:
                return default
            raise ValueError("Invalid parameter name.")  # std::invalid_argument, parameters.h:28
        try:
            return typ(self._p[name])
        except ValueError as e:
            raise RuntimeError(f"Failed to convert value '{self._p[name]}' to type: {typ.__name__}") from e


class IndexBipartite:
    """efanna2e::IndexBipartite restricted to the RoarGraph search path.

    IndexBipartite(dimension, n, metric, initializer)   src/index_bipartite.cpp:30-38
    """

    def __init__(self, dimension, n=0, metric=L2, initializer=None, device=0):
        self._h = C.c_void_p()
        self._lib = capi.lib()
        capi.check(self._lib.roar_create(C.byref(self._h), int(dimension), int(metric), int(device)))
        self.dimension = int(dimension)
        self.metric = int(metric)
        self.device = int(device)
        self.need_normalize = metric == COSINE  # index_bipartite.h:145, set at index_bipartite.cpp:35-37
        self._keep = []  # device tensors adopted by reference

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.roar_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- load ---------------------------------------------------------------------------------
    def LoadSearchNeededData(self, base_file, sampled_query_file=""):
        """index_bipartite.h:62-64 -> LoadVectorData (index_bipartite.cpp:2661-2692)."""
        capi.check(self._lib.roar_load_base_fbin(self._h, str(base_file).encode()))

    def SetBaseData(self, rows):
        """Same as LoadSearchNeededData from memory: numpy [n, dim] (host) or a CUDA/HIP torch tensor
        [n, stride] already zero padded to a multiple of 8 floats (adopted without a copy)."""
        if isinstance(rows, np.ndarray):
            rows = np.ascontiguousarray(rows, np.float32)
            capi.check(self._lib.roar_set_base(self._h, capi.f32p(rows), rows.shape[0], rows.shape[1]))
        else:  # torch tensor on the device
            assert rows.is_cuda and rows.dtype.is_floating_point and rows.element_size() == 4 and rows.is_contiguous()
            self._keep.append(rows)
            capi.check(self._lib.roar_set_base_device(self._h, C.c_void_p(rows.data_ptr()), rows.shape[0],
                                                      rows.shape[1]))

    def LoadProjectionGraph(self, filename):
        """index_bipartite.cpp:2097-2117"""
        capi.check(self._lib.roar_load_graph_file(self._h, str(filename).encode()))

    def SetProjectionGraph(self, ep, offsets, nbrs):
        offsets = np.ascontiguousarray(offsets, np.uint64)
        nbrs = np.ascontiguousarray(nbrs, np.uint32)
        if nbrs.size == 0:
            nbrs = np.zeros(1, np.uint32)
        capi.check(self._lib.roar_set_graph_csr(self._h, int(ep), len(offsets) - 1, capi.u64p(offsets),
                                                capi.u32p(nbrs)))

    def SaveProjectionGraph(self, filename):
        """index_bipartite.cpp:2606-2619"""
        capi.check(self._lib.roar_save_graph_file(self._h, str(filename).encode()))

    def GetProjectionGraph(self):
        ep, n, ne = C.c_uint32(), C.c_uint64(), C.c_uint64()
        capi.check(self._lib.roar_get_graph_csr(self._h, C.byref(ep), C.byref(n), C.byref(ne), None, None))
        offsets = np.zeros(n.value + 1, np.uint64)
        nbrs = np.zeros(max(ne.value, 1), np.uint32)
        capi.check(self._lib.roar_get_graph_csr(self._h, None, None, None, capi.u64p(offsets), capi.u32p(nbrs)))
        return ep.value, offsets, nbrs[:ne.value]

    def BuildBootstrapGraph(self, M=16, base_cap=64, max_degree=128, L_build=48, max_batch=0, learn=None, M_sq=48,
                            M_pj=24, L_learn=128, rounds=2, phase1_l2=True, learn_knn=None, supply_k=12, supply_L=128,
                            L_learn_last=256, trim_min_proj=8, trim_keep_base=0, phase1_cand=48, supply_cand=48,
                            proj_mode=0, supply_mode=0, no_skeleton=False, row_keep=0):
        """In-HBM bootstrap builder (roar_build_graph): NOT the reference's BuildRoarGraph; it exists so
        that full-size synthetic indexes can be made on the GPU box.  `learn`: optional CUDA/HIP torch
        tensor of training queries [n_learn, stride >= dim_padded].  row_keep > 0: max_degree is only the room in
        which a row collects proposals; after every phase a row longer than row_keep keeps its row_keep nearest
        neighbours (index metric), nearest first."""
        bp = capi.BuildParams(M=M, base_cap=base_cap, max_degree=max_degree, L_build=L_build, max_batch=max_batch,
                              learn_dev=None, n_learn=0, learn_stride_floats=0, M_sq=M_sq, M_pj=M_pj,
                              L_learn=L_learn, rounds=rounds, phase1_l2=1 if phase1_l2 else 0, supply_k=supply_k,
                              supply_L=supply_L,
                              L_learn_last=L_learn_last, trim_min_proj=trim_min_proj, trim_keep_base=trim_keep_base,
                              phase1_cand=phase1_cand, supply_cand=supply_cand, proj_mode=proj_mode,
                              supply_mode=supply_mode, no_skeleton=1 if no_skeleton else 0, row_keep=row_keep)
        if learn is not None:
            assert learn.is_cuda and learn.is_contiguous()
            bp.learn_dev, bp.n_learn, bp.learn_stride_floats = learn.data_ptr(), learn.shape[0], learn.shape[1]
        if learn_knn is not None:  # int32/uint32 CUDA tensor [n_learn, M_sq]
            assert learn_knn.is_cuda and learn_knn.is_contiguous() and learn_knn.shape[1] == M_sq
            bp.learn_knn_dev, bp.n_knn = learn_knn.data_ptr(), learn_knn.shape[0]
        capi.check(self._lib.roar_build_graph(self._h, C.byref(bp)))

    def LoadLearnBaseKNN(self, filename):
        """index_bipartite.cpp:2622-2639: u32 n; u32 K; ids[n*K] (anything after is ignored)."""
        with open(filename, "rb") as f:
            n, k = (int(x) for x in np.fromfile(f, np.uint32, 2))
            ids = np.fromfile(f, np.uint32, n * k)
        if ids.size != n * k:
            raise RoarError(capi.ERR_FORMAT, "learn base knn file error")
        self._learn_base_knn = ids.reshape(n, k)

    def BuildRoarGraph(self, n_sq, sq_data, n_bp, bp_data, parameters):
        """index_bipartite.cpp:143-233 (reference shape, roar_build_roargraph).  bp_data: numpy [n_bp, dim] or
        None when the base rows are already loaded; the training queries enter only through the kNN lists
        (LoadLearnBaseKNN), exactly as in the reference."""
        if bp_data is not None:
            self.SetBaseData(np.asarray(bp_data, np.float32)[:n_bp])
        knn = getattr(self, "_learn_base_knn", None)
        if knn is None:
            raise RoarError(capi.ERR_ARG, "learn base knn file error")
        knn = np.ascontiguousarray(knn, np.uint32)
        capi.check(self._lib.roar_build_roargraph(self._h, capi.u32p(knn), knn.shape[0], knn.shape[1],
                                                  parameters.Get("M_sq", int), parameters.Get("M_pjbp", int),
                                                  parameters.Get("L_pjpq", int)))

    def BuildRoarGraphDevice(self, learn_knn, parameters):
        """BuildRoarGraph from learn->base lists that already live in HBM (int32 CUDA/HIP tensor [n_learn, K], e.g. the
        output of ExactKnnDevice): roar_build_roargraph_device."""
        import torch

        assert learn_knn.is_cuda and learn_knn.is_contiguous() and learn_knn.dtype in (torch.int32, torch.uint32), \
            "learn->base lists must be an int32 / uint32 CUDA tensor [n_learn, K]"
        capi.check(self._lib.roar_build_roargraph_device(self._h, C.c_void_p(learn_knn.data_ptr()), learn_knn.shape[0],
                                                         learn_knn.shape[1], parameters.Get("M_sq", int),
                                                         parameters.Get("M_pjbp", int), parameters.Get("L_pjpq", int)))

    def InitVisitedListPool(self, num_threads):
        """index_bipartite.h:133"""
        capi.check(self._lib.roar_init_visited_pool(self._h, int(num_threads)))

    # ---- options --------------------------------------------------------------------------------
    def set_option(self, key, value):
        capi.check(self._lib.roar_set_option(self._h, key.encode(), int(value)))

    def info(self, key):
        v = C.c_int64()
        capi.check(self._lib.roar_get_info(self._h, key.encode(), C.byref(v)))
        return v.value

    def SetTrafficCounters(self, block):
        """roar_set_traffic_counters_device: `block` = int32 CUDA tensor [capacity, 4] (or None to switch it off); launches of at
        most `capacity` queries then write {rows on the 8-bit copy, rows on the f32 rows, adjacency words fetched, 0} per query."""
        if block is None:
            capi.check(self._lib.roar_set_traffic_counters_device(self._h, None, 0))
            self._traffic = None
            return
        assert block.is_cuda and block.is_contiguous() and block.shape[1] == 4 and block.element_size() == 4
        capi.check(self._lib.roar_set_traffic_counters_device(self._h, C.c_void_p(block.data_ptr()), block.shape[0]))
        self._traffic = block  # (keeps the tensor alive while the index writes to it)

    def LastProfile(self, nq):
        """Phase timers of the most recent instrumented launch (option profile_phases): uint64 [nq, 16]
        (layout: roar_last_profile in include/roargraph_amd.h)."""
        out = np.zeros((nq, 16), np.uint64)
        capi.check(self._lib.roar_last_profile(self._h, capi.u64p(out), int(nq)))
        return out

    # ---- search ---------------------------------------------------------------------------------
    def SearchRoarGraph(self, query, k, qid, parameters, indices=None, res_dists=None):
        """index_bipartite.cpp:2311-2420 for ONE query; returns (cmps, hops) like the reference and
        fills indices[k] / res_dists[k] when given.  A batch of one through the same kernel."""
        L = parameters.Get("L_pq", int)
        r = self.SearchRoarGraphBatch(np.asarray(query, np.float32).reshape(1, -1), k, L)
        if indices is not None:
            indices[:k] = r["ids"][0]
        if res_dists is not None:
            res_dists[:k] = r["dists"][0]
        return int(r["cmps"][0]), int(r["hops"][0])

    def SearchRoarGraphBatch(self, queries, k, L_pq):
        """All queries in one launch (host buffers).  -> dict(ids, dists, cmps, hops, scanned)."""
        queries = np.ascontiguousarray(queries, np.float32)
        nq, qs = queries.shape
        ids = np.zeros((nq, k), np.uint32)
        dists = np.zeros((nq, k), np.float32)
        cmps = np.zeros(nq, np.uint32)
        hops = np.zeros(nq, np.uint32)
        scanned = np.zeros(nq, np.uint32)
        rc = self._lib.roar_search_batch(self._h, capi.f32p(queries), nq, qs, int(k), int(L_pq), capi.u32p(ids),
                                         capi.f32p(dists), capi.u32p(cmps), capi.u32p(hops), capi.u32p(scanned))
        if rc == capi.ERR_SHORT:  # the reference throws std::runtime_error here (:2408-2412)
            raise RoarError(rc, self._lib.roar_last_error().decode())
        capi.check(rc)
        return dict(ids=ids, dists=dists, cmps=cmps, hops=hops, scanned=scanned)

    def Clone(self, device):
        """Replica of this index (base rows + projection graph + options) on `device` (roar_clone_to_device)."""
        other = IndexBipartite.__new__(IndexBipartite)
        other._h = C.c_void_p()
        other._lib = self._lib
        capi.check(self._lib.roar_clone_to_device(self._h, int(device), C.byref(other._h)))
        other.dimension, other.metric, other.device = self.dimension, self.metric, int(device)
        other.need_normalize = self.need_normalize
        other._keep = []
        return other

    @staticmethod
    def SearchRoarGraphBatchMulti(replicas, queries, k, L_pq):
        """One query block split over the replicas by query (roar_search_batch_multi): replica r searches rows
        [r * nq / g, (r + 1) * nq / g).  -> dict(ids, dists, cmps, hops, scanned), same as SearchRoarGraphBatch."""
        queries = np.ascontiguousarray(queries, np.float32)
        nq, qs = queries.shape
        ids = np.zeros((nq, k), np.uint32)
        dists = np.zeros((nq, k), np.float32)
        cmps = np.zeros(nq, np.uint32)
        hops = np.zeros(nq, np.uint32)
        scanned = np.zeros(nq, np.uint32)
        hs = (C.c_void_p * len(replicas))(*[r._h for r in replicas])
        lib = replicas[0]._lib
        rc = lib.roar_search_batch_multi(hs, len(replicas), capi.f32p(queries), nq, qs, int(k), int(L_pq), capi.u32p(ids),
                                         capi.f32p(dists), capi.u32p(cmps), capi.u32p(hops), capi.u32p(scanned))
        if rc == capi.ERR_SHORT:
            raise RoarError(rc, lib.roar_last_error().decode())
        capi.check(rc)
        return dict(ids=ids, dists=dists, cmps=cmps, hops=hops, scanned=scanned)

    def SearchRoarGraphBatchDevice(self, queries, k, L_pq, out=None, stream=None):
        """Device-resident form: `queries` is a CUDA/HIP torch tensor [nq, stride>=dim_padded];
        results are torch tensors in HBM; asynchronous on `stream` (torch stream or None)."""
        import torch

        nq = queries.shape[0]
        dev = queries.device
        if out is None:
            out = dict(ids=torch.empty((nq, k), dtype=torch.int32, device=dev),
                       dists=torch.empty((nq, k), dtype=torch.float32, device=dev),
                       cmps=torch.empty(nq, dtype=torch.int32, device=dev),
                       hops=torch.empty(nq, dtype=torch.int32, device=dev),
                       scanned=torch.empty(nq, dtype=torch.int32, device=dev),
                       short=torch.zeros(1, dtype=torch.int32, device=dev))
        st = stream if stream is not None else torch.cuda.current_stream(dev)
        capi.check(self._lib.roar_search_batch_device(
            self._h, C.c_void_p(queries.data_ptr()), nq, queries.shape[1], int(k), int(L_pq),
            C.c_void_p(out["ids"].data_ptr()), C.c_void_p(out["dists"].data_ptr()),
            C.c_void_p(out["cmps"].data_ptr()), C.c_void_p(out["hops"].data_ptr()),
            C.c_void_p(out["scanned"].data_ptr()), C.c_void_p(out["short"].data_ptr()),
            C.c_void_p(st.cuda_stream)))
        return out

    def OpenSearchStream(self, k, L_pq):
        """roar_search_stream_open: a stream of batches served by workgroups that outlive a batch (include/roargraph_amd.h)."""
        return SearchStream(self, k, L_pq)

    # ---- bipartite two-hop variant (index_bipartite.cpp:42-141, 235-356, 2045-2071) ---------------
    def Load(self, filename):
        """IndexBipartite::Load, index_bipartite.cpp:2060-2071"""
        capi.check(self._lib.roar_load_bipartite_file(self._h, str(filename).encode()))

    def Save(self, filename):
        """IndexBipartite::Save, index_bipartite.cpp:2045-2058"""
        capi.check(self._lib.roar_save_bipartite_file(self._h, str(filename).encode()))

    def SetBipartiteGraph(self, offsets, nbrs):
        offsets = np.ascontiguousarray(offsets, np.uint64)
        nbrs = np.ascontiguousarray(nbrs, np.uint32)
        if nbrs.size == 0:
            nbrs = np.zeros(1, np.uint32)
        capi.check(self._lib.roar_set_bipartite_csr(self._h, len(offsets) - 1, capi.u64p(offsets), capi.u32p(nbrs)))

    def GetBipartiteGraph(self):
        n, ne = C.c_uint64(), C.c_uint64()
        capi.check(self._lib.roar_get_bipartite_csr(self._h, C.byref(n), C.byref(ne), None, None))
        offsets = np.zeros(n.value + 1, np.uint64)
        nbrs = np.zeros(max(ne.value, 1), np.uint32)
        capi.check(self._lib.roar_get_bipartite_csr(self._h, None, None, capi.u64p(offsets), capi.u32p(nbrs)))
        return offsets, nbrs[:ne.value]

    def BuildBipartite(self, n_sq, sq_data, n_bp, bp_data, parameters):
        """index_bipartite.cpp:42-141 -> qbaseNNbipartite :235-280; like BuildRoarGraph the training queries
        enter only through the kNN lists (LoadLearnBaseKNN)."""
        if bp_data is not None:
            self.SetBaseData(np.asarray(bp_data, np.float32)[:n_bp])
        knn = getattr(self, "_learn_base_knn", None)
        if knn is None:
            raise RoarError(capi.ERR_ARG, "learn base knn file error")
        knn = np.ascontiguousarray(knn, np.uint32)
        capi.check(self._lib.roar_build_bipartite(self._h, capi.u32p(knn), knn.shape[0], knn.shape[1],
                                                  parameters.Get("M_pjbp", int)))

    @staticmethod
    def BipartiteEntryIds(seed, nq, n_init, n_base):
        """The generator that stands in for std::random_device (index_bipartite.cpp:287-294)."""
        out = np.zeros((nq, n_init), np.uint32)
        capi.check(capi.lib().roar_bipartite_entry_ids(int(seed), int(nq), int(n_init), int(n_base), capi.u32p(out)))
        return out

    def SearchBipartiteGraphBatch(self, queries, k, L_pq, init_ids=None, n_init=10):
        """index_bipartite.cpp:282-356 for a batch.  init_ids [nq, n_init] u32 or None (library draws them)."""
        queries = np.ascontiguousarray(queries, np.float32)
        nq, qs = queries.shape
        if init_ids is not None:
            init_ids = np.ascontiguousarray(init_ids, np.uint32)
            assert init_ids.shape[0] == nq
            n_init = init_ids.shape[1]
        ids = np.zeros((nq, k), np.uint32)
        dists = np.zeros((nq, k), np.float32)
        cmps = np.zeros(nq, np.uint32)
        hops = np.zeros(nq, np.uint32)
        rc = self._lib.roar_search_bipartite_batch(self._h, capi.f32p(queries), nq, qs, int(k), int(L_pq),
                                                   capi.u32p(init_ids) if init_ids is not None else None, int(n_init),
                                                   capi.u32p(ids), capi.f32p(dists), capi.u32p(cmps), capi.u32p(hops))
        if rc == capi.ERR_SHORT:
            raise RoarError(rc, self._lib.roar_last_error().decode())
        capi.check(rc)
        return dict(ids=ids, dists=dists, cmps=cmps, hops=hops)

    def SearchBipartiteGraph(self, query, k, qid, parameters, indices=None, res_dists=None):
        """One query, the reference's signature; returns (cmps, hops)."""
        r = self.SearchBipartiteGraphBatch(np.asarray(query, np.float32).reshape(1, -1), k, parameters.Get("L_pq", int))
        if indices is not None:
            indices[:k] = r["ids"][0]
        if res_dists is not None:
            res_dists[:k] = r["dists"][0]
        return int(r["cmps"][0]), int(r["hops"][0])

    def SearchBipartiteGraphBatchDevice(self, queries, k, L_pq, init_ids, out=None, stream=None):
        """Device-resident form (torch tensors in HBM; init_ids int32 [nq, n_init])."""
        import torch

        nq = queries.shape[0]
        dev = queries.device
        if out is None:
            out = dict(ids=torch.empty((nq, k), dtype=torch.int32, device=dev),
                       dists=torch.empty((nq, k), dtype=torch.float32, device=dev),
                       cmps=torch.empty(nq, dtype=torch.int32, device=dev),
                       hops=torch.empty(nq, dtype=torch.int32, device=dev),
                       scanned=torch.empty(nq, dtype=torch.int32, device=dev),
                       short=torch.zeros(1, dtype=torch.int32, device=dev))
        st = stream if stream is not None else torch.cuda.current_stream(dev)
        capi.check(self._lib.roar_search_bipartite_batch_device(
            self._h, C.c_void_p(queries.data_ptr()), nq, queries.shape[1], int(k), int(L_pq),
            C.c_void_p(init_ids.data_ptr()), init_ids.shape[1],
            C.c_void_p(out["ids"].data_ptr()), C.c_void_p(out["dists"].data_ptr()),
            C.c_void_p(out["cmps"].data_ptr()), C.c_void_p(out["hops"].data_ptr()),
            C.c_void_p(out["scanned"].data_ptr()), C.c_void_p(out["short"].data_ptr()),
            C.c_void_p(st.cuda_stream)))
        return out

    def ExactKnnDevice(self, queries, k, want_dists=True, stream=None):
        """Brute-force top-k of every query among the loaded base rows (roar_exact_knn_device): the
        generator of ground-truth / learn->base kNN files.  queries: CUDA/HIP tensor [nq, stride]."""
        import torch

        nq = queries.shape[0]
        ids = torch.empty((nq, k), dtype=torch.int32, device=queries.device)
        dists = torch.empty((nq, k), dtype=torch.float32, device=queries.device) if want_dists else None
        st = stream if stream is not None else torch.cuda.current_stream(queries.device)
        capi.check(self._lib.roar_exact_knn_device(self._h, C.c_void_p(queries.data_ptr()), nq, queries.shape[1], int(k),
                                                   C.c_void_p(ids.data_ptr()),
                                                   C.c_void_p(dists.data_ptr()) if want_dists else None,
                                                   C.c_void_p(st.cuda_stream)))
        return ids, dists

    def Distances(self, queries, ids):
        """distance_->compare(base[id], query, dim) for explicit ids [nq, m] (distance.h:39-89,179-223)."""
        queries = np.ascontiguousarray(queries, np.float32)
        ids = np.ascontiguousarray(ids, np.uint32)
        out = np.zeros(ids.shape, np.float32)
        capi.check(self._lib.roar_distances(self._h, capi.f32p(queries), queries.shape[0], queries.shape[1],
                                            capi.u32p(ids), ids.shape[1], capi.f32p(out)))
        return out

    def ScreenKeeps(self, queries, ids, thresholds):
        """roar_debug_screen_keeps: would the 8-bit screen pass row ids[q][c] of query q on to the exact scoring when the pool's
        reject threshold is thresholds[q][c]?  -> uint8 [nq, m]."""
        import ctypes as C

        queries = np.ascontiguousarray(queries, np.float32)
        ids = np.ascontiguousarray(ids, np.uint32)
        thresholds = np.ascontiguousarray(thresholds, np.float32)
        assert thresholds.shape == ids.shape
        out = np.zeros(ids.shape, np.uint8)
        capi.check(self._lib.roar_debug_screen_keeps(self._h, capi.f32p(queries), queries.shape[0], queries.shape[1], capi.u32p(ids),
                                                     capi.f32p(thresholds), ids.shape[1], out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out


class ShardedIndex:
    """The index-sharded configuration (roar_sharded_*): shard s is an IndexBipartite over its own
    rows (local ids) on device devices[s]; Search runs every query on every shard and merges the per-shard lists after one
    RCCL all-gather.  shard(s) returns a non-owning IndexBipartite view: load its base rows and graph as usual."""

    def __init__(self, dimension, metric, devices):
        self._lib = capi.lib()
        self._h = C.c_void_p()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        capi.check(self._lib.roar_sharded_create(C.byref(self._h), int(dimension), int(metric), devs, len(devices)))
        self.dimension, self.metric, self.devices = int(dimension), int(metric), [int(d) for d in devices]
        self._views = {}

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            for v in self._views.values():
                v._h = C.c_void_p()  # owned by the roar_sharded
            self._lib.roar_sharded_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shard(self, s):
        if s not in self._views:
            h = self._lib.roar_sharded_shard(self._h, int(s))
            if not h:
                raise RoarError(capi.ERR_ARG, self._lib.roar_last_error().decode())
            v = IndexBipartite.__new__(IndexBipartite)
            v._lib, v._h = self._lib, C.c_void_p(h)
            v.dimension, v.metric, v.device = self.dimension, self.metric, self.devices[s]
            v.need_normalize = self.metric == COSINE
            v._keep = []
            v.close = lambda: None  # non-owning
            self._views[s] = v
        return self._views[s]

    def set_id_base(self, s, first_global_id):
        capi.check(self._lib.roar_sharded_set_id_base(self._h, int(s), int(first_global_id)))

    def info(self, key):
        v = C.c_int64()
        capi.check(self._lib.roar_sharded_info(self._h, key.encode(), C.byref(v)))
        return v.value

    def SearchRoarGraphBatch(self, queries, k, L_pq):
        """-> dict(ids [nq, k] global ids, dists, cmps, hops (summed over the shards)); host buffers."""
        queries = np.ascontiguousarray(queries, np.float32)
        nq, qs = queries.shape
        ids = np.zeros((nq, k), np.uint32)
        dists = np.zeros((nq, k), np.float32)
        cmps = np.zeros(nq, np.uint32)
        hops = np.zeros(nq, np.uint32)
        capi.check(self._lib.roar_search_batch_sharded(self._h, capi.f32p(queries), nq, qs, int(k), int(L_pq), capi.u32p(ids),
                                                       capi.f32p(dists), capi.u32p(cmps), capi.u32p(hops)))
        return dict(ids=ids, dists=dists, cmps=cmps, hops=hops)


def ComputeRecall(res, gt, k):
    """tests/test_search_roargraph.cpp:23-36: |gt[:k] in res[:k]| / (k * nq)."""
    res = np.asarray(res)[:, :k].astype(np.int64)
    gt = np.asarray(gt)[:, :k].astype(np.int64)
    hit = (gt[:, :, None] == res[:, None, :]).any(axis=2).sum()
    return float(hit) / float(k * res.shape[0])


def normalize(arr):
    """util.h:214-225 applied to every row (float32 sequential sum)."""
    arr = np.array(arr, np.float32, copy=True)
    for i in range(arr.shape[0]):
        s = np.float32(0)
        for v in arr[i]:
            s = np.float32(s + v * v)
        arr[i] = arr[i] / np.sqrt(s, dtype=np.float32)
    return arr


_LEAKED_BY_FAILED_STREAMS = []


class SearchStream:
    """ctypes mirror of roar_search_stream_* (device buffers: torch tensors in HBM).  The tensors of a submitted batch are kept alive
    here until its ticket has been waited for."""

    def __init__(self, index, k, L_pq):
        self._lib = index._lib
        self._index = index
        self.k = int(k)
        h = C.c_void_p()
        capi.check(self._lib.roar_search_stream_open(index._h, int(k), int(L_pq), C.byref(h)))
        self._h = h
        self._keep = {}

    def submit(self, queries, out=None):
        """-> (ticket, out).  `queries` must be complete in device memory (synchronise its producer first)."""
        import torch

        nq = queries.shape[0]
        dev = queries.device
        if out is None:
            out = dict(ids=torch.empty((nq, self.k), dtype=torch.int32, device=dev),
                       dists=torch.empty((nq, self.k), dtype=torch.float32, device=dev),
                       cmps=torch.empty(nq, dtype=torch.int32, device=dev),
                       hops=torch.empty(nq, dtype=torch.int32, device=dev),
                       scanned=torch.empty(nq, dtype=torch.int32, device=dev))
        t = C.c_uint64()
        capi.check(self._lib.roar_search_stream_submit(
            self._h, C.c_void_p(queries.data_ptr()), nq, queries.shape[1] if queries.dim() == 2 else 0,
            C.c_void_p(out["ids"].data_ptr()), C.c_void_p(out["dists"].data_ptr()), C.c_void_p(out["cmps"].data_ptr()),
            C.c_void_p(out["hops"].data_ptr()), C.c_void_p(out["scanned"].data_ptr()), C.byref(t)))
        if t.value:
            self._keep[t.value] = (queries, out)
            if len(self._keep) > 1024:  # tickets nobody waits for: their tensors are let go once the stream says their batches are complete
                done = self.info("batches_completed")
                for old in [x for x in self._keep if x <= done]:
                    del self._keep[old]
        return t.value, out

    def wait(self, ticket):
        """-> queries of the batch that ended with fewer than k results"""
        n = C.c_uint32()
        capi.check(self._lib.roar_search_stream_wait(self._h, int(ticket), C.byref(n)))
        self._keep.pop(int(ticket), None)
        return n.value

    def times(self, ticket):
        a, b = C.c_uint64(), C.c_uint64()
        capi.check(self._lib.roar_search_stream_times(self._h, int(ticket), C.byref(a), C.byref(b)))
        return a.value, b.value

    def info(self, key):
        v = C.c_int64()
        capi.check(self._lib.roar_search_stream_info(self._h, key.encode(), C.byref(v)))
        return v.value

    def close(self):
        if self._h is not None:
            h, self._h = self._h, None
            try:
                capi.check(self._lib.roar_search_stream_close(h))
            except capi.RoarError:
                # a failed stream whose launches may still be running keeps its memory and pins the index (include/roargraph_amd.h):
                # the tensors of its batches must outlive them too
                _LEAKED_BY_FAILED_STREAMS.append((self._index, self._keep))
                self._keep = {}
                raise
            self._keep.clear()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

