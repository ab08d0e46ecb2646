// roar_bipartite.cpp -- the bipartite two-hop variant behind the C ABI
// (round 5: roar_capi.cpp was one 3 203-line translation unit; it is now cut by concern.  Shared declarations: roar_internal.h.)
#include "roar_internal.h"

using namespace roar_impl;

namespace roar_impl {

// Every row of a base node must hold training-query nodes and every row of a training-query node base
// nodes: SearchBipartiteGraph scores graph[graph[cur]] entries as base rows (index_bipartite.cpp:330-340),
// anything else reads outside data_bp_ in the reference.
int check_bipartite(roar_index *idx) {
    if (idx->bp_checked) return ROAR_OK;
    const uint64_t nb = idx->n, nt = idx->bp_n;
    if (nt < nb)
        return fail(ROAR_ERR_FORMAT, "bipartite graph has %llu nodes, fewer than the %llu base rows",
                    (unsigned long long)nt, (unsigned long long)nb);
    for (uint64_t v = 0; v < nt; ++v)
        for (uint64_t j = idx->hb_off[v]; j < idx->hb_off[v + 1]; ++j) {
            const uint32_t t = idx->hb_nbrs[j];
            if (v < nb ? (t < nb || t >= nt) : (t >= nb))
                return fail(ROAR_ERR_FORMAT, "bipartite graph: node %llu links to %u on its own side",
                            (unsigned long long)v, t);
        }
    idx->bp_checked = true;
    return ROAR_OK;
}

inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
}  // namespace roar_impl

extern "C" {

int roar_set_bipartite_csr(roar_index *idx, uint64_t n_total, const uint64_t *offsets, const uint32_t *nbrs) {
    DeviceRestore keep_device_;
    if (!idx || !offsets || (n_total && !nbrs && offsets[n_total] != 0)) return fail(ROAR_ERR_ARG, "bad graph arguments");
    if (n_total == 0 || n_total >= (1ull << 32)) return fail(ROAR_ERR_ARG, "graph must have between 1 and 2^32 - 1 nodes");
    for (uint64_t i = 0; i < n_total; ++i)
        if (offsets[i + 1] < offsets[i]) return fail(ROAR_ERR_FORMAT, "CSR offsets are not monotone");
    const uint64_t ne = offsets[n_total];
    for (uint64_t j = 0; j < ne; ++j)
        if (nbrs[j] >= n_total) return fail(ROAR_ERR_FORMAT, "neighbour id %u outside the graph", nbrs[j]);
    std::lock_guard<std::mutex> guard(idx->mu);
    int e = use_device(idx);
    if (e) return e;
    free_bipartite(idx);
    idx->hb_off.assign(offsets, offsets + n_total + 1);
    idx->hb_nbrs.assign(nbrs, nbrs + ne);
    idx->bp_n = n_total;
    idx->bp_edges = ne;
    if (idx->n && (e = check_bipartite(idx))) {
        free_bipartite(idx);
        return e;
    }
    HIP_TRY(roar_dev_malloc((void **)&idx->bp_off, (size_t)(n_total + 1) * sizeof(uint64_t)));
    HIP_TRY(roar_dev_malloc((void **)&idx->bp_nbrs, (size_t)(ne + 64) * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(idx->bp_off, offsets, (size_t)(n_total + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    if (ne) HIP_TRY(hipMemcpy(idx->bp_nbrs, nbrs, (size_t)ne * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(idx->bp_nbrs + ne, 0, 64 * sizeof(uint32_t)));
    return ROAR_OK;
}

int roar_load_bipartite_file(roar_index *idx, const char *path) {
    if (!idx || !path) return fail(ROAR_ERR_ARG, "bad arguments");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(ROAR_ERR_IO, "cannot open file");
    fseek(f, 0, SEEK_END);
    const uint64_t fsize = (uint64_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    if (fsize < 4 || (fsize & 3)) {
        fclose(f);
        return fail(ROAR_ERR_FORMAT, "graph file truncated");
    }
    std::vector<uint32_t> raw((size_t)(fsize / 4));
    if (fread(raw.data(), 4, raw.size(), f) != raw.size()) {
        fclose(f);
        return fail(ROAR_ERR_FORMAT, "graph file truncated");
    }
    fclose(f);
    const uint32_t npts = raw[0];  // src/index_bipartite.cpp:2065-2067: no entry point word in this format
    std::vector<uint64_t> off((size_t)npts + 1);
    std::vector<uint32_t> nb;
    nb.reserve(raw.size());
    size_t p = 1;
    for (uint32_t i = 0; i < npts; ++i) {
        if (p >= raw.size()) return fail(ROAR_ERR_FORMAT, "graph file truncated");
        uint32_t deg = raw[p++];
        if (p + deg > raw.size()) return fail(ROAR_ERR_FORMAT, "graph file truncated");
        off[i] = nb.size();
        nb.insert(nb.end(), raw.begin() + p, raw.begin() + p + deg);
        p += deg;
    }
    off[npts] = nb.size();
    static const uint32_t dummy = 0;
    return roar_set_bipartite_csr(idx, npts, off.data(), nb.empty() ? &dummy : nb.data());
}

int roar_save_bipartite_file(roar_index *idx, const char *path) {
    if (!idx || !path) return fail(ROAR_ERR_ARG, "bad arguments");
    if (idx->bp_n == 0) return fail(ROAR_ERR_ARG, "no bipartite graph loaded");
    FILE *f = fopen(path, "wb");
    if (!f) return fail(ROAR_ERR_IO, "cannot open file");
    uint32_t npts = (uint32_t)idx->bp_n;  // src/index_bipartite.cpp:2049-2051
    fwrite(&npts, 4, 1, f);
    for (uint64_t i = 0; i < idx->bp_n; ++i) {
        uint32_t deg = (uint32_t)(idx->hb_off[i + 1] - idx->hb_off[i]);
        fwrite(&deg, 4, 1, f);
        fwrite(idx->hb_nbrs.data() + idx->hb_off[i], 4, deg, f);
    }
    fclose(f);
    return ROAR_OK;
}

int roar_get_bipartite_csr(roar_index *idx, uint64_t *n_total, uint64_t *nedges, uint64_t *offsets, uint32_t *nbrs) {
    if (!idx) return fail(ROAR_ERR_ARG, "bad arguments");
    if (idx->bp_n == 0) return fail(ROAR_ERR_ARG, "no bipartite graph loaded");
    if (n_total) *n_total = idx->bp_n;
    if (nedges) *nedges = idx->bp_edges;
    if (offsets) memcpy(offsets, idx->hb_off.data(), (size_t)(idx->bp_n + 1) * sizeof(uint64_t));
    if (nbrs && idx->bp_edges) memcpy(nbrs, idx->hb_nbrs.data(), (size_t)idx->bp_edges * sizeof(uint32_t));
    return ROAR_OK;
}

// qbaseNNbipartite (src/index_bipartite.cpp:235-280): training query sq (node n + sq) keeps its first
// M_pjbp nearest base points; the nearest is its target, the others its row; the target's row gains
// n + sq.  The reference appends to the target rows under per-node locks from an OpenMP loop, so their
// order is run-dependent there; here it is ascending sq.  (Within a hop the search is order-free.)
int roar_build_bipartite(roar_index *idx, const uint32_t *learn_knn, uint64_t n_learn, uint32_t K, uint32_t M_pjbp) {
    if (!idx || !learn_knn) return fail(ROAR_ERR_ARG, "bad arguments");
    if (!idx->base || idx->n == 0) return fail(ROAR_ERR_ARG, "build before base vectors were loaded");
    if (n_learn == 0 || K == 0 || M_pjbp == 0) return fail(ROAR_ERR_ARG, "n_learn, K and M_pjbp must be positive");
    const uint64_t nb = idx->n, nt = nb + n_learn;
    if (nt >= (1ull << 32)) return fail(ROAR_ERR_ARG, "base + training queries exceed 2^32 - 1 nodes");
    const uint32_t keep = std::min(K, M_pjbp);  // :256-259
    std::vector<uint64_t> off((size_t)nt + 1, 0);
    for (uint64_t sq = 0; sq < n_learn; ++sq) {
        const uint32_t *nn = learn_knn + sq * K;
        const uint32_t tgt = nn[0];
        for (uint32_t i = 0; i < keep; ++i)
            if (nn[i] >= nb)
                return fail(ROAR_ERR_FORMAT, "kNN list %llu names base id %u of %llu", (unsigned long long)sq, nn[i],
                            (unsigned long long)nb);
        uint32_t len = 0;
        for (uint32_t i = 0; i < keep; ++i) len += nn[i] != tgt;  // :264-268
        off[nb + sq + 1] = len;
        off[tgt + 1] += 1;
    }
    for (uint64_t v = 0; v < nt; ++v) off[v + 1] += off[v];
    std::vector<uint32_t> nbrs((size_t)off[nt] + 1);
    std::vector<uint64_t> fill(off.begin(), off.begin() + nb);
    for (uint64_t sq = 0; sq < n_learn; ++sq) {
        const uint32_t *nn = learn_knn + sq * K;
        const uint32_t tgt = nn[0];
        uint64_t w = off[nb + sq];
        for (uint32_t i = 0; i < keep; ++i)
            if (nn[i] != tgt) nbrs[w++] = nn[i];
        nbrs[fill[tgt]++] = (uint32_t)(nb + sq);  // :269-272
    }
    return roar_set_bipartite_csr(idx, nt, off.data(), nbrs.data());
}

// Entry points for SearchBipartiteGraph: the reference draws 10 uniform base ids per query from
// std::random_device (src/index_bipartite.cpp:287-294); a counter-based generator takes its place so
// that a search can be replayed against the CPU checker.  out[q * n_init + i].
int roar_bipartite_entry_ids(uint64_t seed, uint64_t nq, uint32_t n_init, uint64_t n_base, uint32_t *out) {
    if (!out || n_base == 0 || n_base >= (1ull << 32)) return fail(ROAR_ERR_ARG, "bad arguments");
    for (uint64_t q = 0; q < nq; ++q)
        for (uint32_t i = 0; i < n_init; ++i)
            out[q * n_init + i] = (uint32_t)(splitmix64(seed ^ splitmix64(q * 64 + i)) % n_base);
    return ROAR_OK;
}

int roar_search_bipartite_batch_device(roar_index *idx, const float *queries_dev, uint64_t nq, uint32_t qstride_floats,
                                       uint32_t k, uint32_t L_pq, const uint32_t *init_ids_dev, uint32_t n_init,
                                       uint32_t *ids_dev, float *dists_dev, uint32_t *cmps_dev, uint32_t *hops_dev,
                                       uint32_t *scanned_dev, uint32_t *short_count_dev, void *hip_stream) {
    DeviceRestore keep_device_;
    if (!idx || !queries_dev || !ids_dev || !dists_dev || !init_ids_dev) return fail(ROAR_ERR_ARG, "bad search arguments");
    if (nq == 0) return ROAR_OK;
    if (n_init == 0 || n_init > 64) return fail(ROAR_ERR_ARG, "between 1 and 64 entry points per query");
    if (qstride_floats < idx->dim_pad || (qstride_floats & 3))
        return fail(ROAR_ERR_ARG, "device queries must be padded to %u floats (stride multiple of 4)", idx->dim_pad);
    int e = use_device(idx);
    if (e) return e;
    if (idx->bp_n && idx->n && (e = check_bipartite(idx))) return e;
    LaunchPlan plan;
    if ((e = plan_search(idx, nq, k, L_pq, &plan, true))) return e;
    RoarSearchParams &p = plan.p;
    p.queries = queries_dev;
    p.qstride = qstride_floats;
    p.init_ids = init_ids_dev;
    p.n_init = n_init;
    p.out_ids = ids_dev;
    p.out_dists = dists_dev;
    p.out_cmps = cmps_dev;
    p.out_hops = hops_dev;
    p.out_scanned = scanned_dev;
    p.short_count = short_count_dev;
    hipStream_t st = (hipStream_t)hip_stream;
    HIP_TRY(hipMemsetAsync(idx->counters, 0, 2 * sizeof(uint32_t), st));
    if (short_count_dev) HIP_TRY(hipMemsetAsync(short_count_dev, 0, sizeof(uint32_t), st));
    HIP_TRY(hipEventRecord(idx->ev0, st));
    int rc = roar_launch_search(&p, &plan.sel, plan.grid, st);
    if (rc != 0) return fail(ROAR_ERR_DEVICE, "search kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    HIP_TRY(hipEventRecord(idx->ev1, st));
    idx->ev_valid = true;
    return ROAR_OK;
}

int roar_search_bipartite_batch(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats, uint32_t k,
                                uint32_t L_pq, const uint32_t *init_ids, uint32_t n_init, uint32_t *ids, float *dists,
                                uint32_t *cmps, uint32_t *hops) {
    DeviceRestore keep_device_;
    if (!idx || !ids || !dists || (nq && !queries)) return fail(ROAR_ERR_ARG, "bad search arguments");
    if (nq == 0) return ROAR_OK;
    if (qstride_floats < idx->dim) return fail(ROAR_ERR_ARG, "query stride smaller than the dimension");
    if (n_init == 0 || n_init > 64) return fail(ROAR_ERR_ARG, "between 1 and 64 entry points per query");
    if (!idx->base || idx->n == 0) return fail(ROAR_ERR_ARG, "search before base vectors were loaded");
    std::lock_guard<std::mutex> guard(idx->mu);
    int e = use_device(idx);
    if (e) return e;
    std::vector<uint32_t> gen;
    if (!init_ids) {  // std::random_device in the reference (:287-294): a fresh draw per call
        gen.resize((size_t)nq * n_init);
        if ((e = roar_bipartite_entry_ids(idx->bp_seed, nq, n_init, idx->n, gen.data()))) return e;
        idx->bp_seed = splitmix64(idx->bp_seed);
        init_ids = gen.data();
    } else {
        for (uint64_t i = 0; i < nq * n_init; ++i)
            if (init_ids[i] >= idx->n) return fail(ROAR_ERR_ARG, "entry point %u outside the base set", init_ids[i]);
    }
    if (idx->init_cap < nq * n_init) {
        if (idx->init_dev) (void)hipFree(idx->init_dev);
        idx->init_dev = nullptr;
        HIP_TRY(roar_dev_malloc((void **)&idx->init_dev, (size_t)nq * n_init * sizeof(uint32_t)));
        idx->init_cap = nq * n_init;
    }
    HIP_TRY(hipMemcpyAsync(idx->init_dev, init_ids, (size_t)nq * n_init * sizeof(uint32_t), hipMemcpyHostToDevice, idx->stream));
    HIP_TRY(hipStreamSynchronize(idx->stream));  // init_ids may be a temporary of this call
    const float *qdev = nullptr;
    uint32_t qs = 0;
    bool deferred = false;
    if ((e = stage_queries(idx, queries, nq, qstride_floats, k, &qdev, &qs, &deferred))) return e;
    if (deferred) pack_queries(idx, queries, nq, qstride_floats);  // (the two-hop kernels do not wait for a packing thread)
    uint32_t *d_ids = idx->out_dev;
    float *d_dists = reinterpret_cast<float *>(d_ids + nq * k);
    uint32_t *d_cmps = d_ids + 2 * nq * k;
    uint32_t *d_hops = d_cmps + nq;
    uint32_t *d_scan = d_hops + nq;
    uint32_t *d_short = d_scan + nq;
    e = roar_search_bipartite_batch_device(idx, qdev, nq, qs, k, L_pq, idx->init_dev, n_init, d_ids, d_dists, d_cmps,
                                           d_hops, d_scan, d_short, idx->stream);
    if (e) return e;
    return fetch_results(idx, nq, k, ids, dists, cmps, hops, nullptr);
}
}  // extern "C"
