// roar_misc.cpp -- small entry points: roar_distances, the screen's known-answer surface, top-k merge, id rebase, queue trace, allocation-failure hook, phase timers, traffic counters, options and info keys
// (round 5: roar_capi.cpp was one 3 203-line translation unit; it is now cut by concern.  Shared declarations: roar_internal.h.)
#include "roar_internal.h"

using namespace roar_impl;

extern "C" {

int roar_distances(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats, const uint32_t *ids,
                   uint32_t m, float *out) {
    DeviceRestore keep_device_;
    if (!idx || !queries || !ids || !out) return fail(ROAR_ERR_ARG, "bad arguments");
    if (!idx->base) return fail(ROAR_ERR_ARG, "distances before base vectors were loaded");
    if (nq == 0 || m == 0) return ROAR_OK;
    if (qstride_floats < idx->dim) return fail(ROAR_ERR_ARG, "query stride smaller than the dimension");
    for (uint64_t i = 0; i < nq * m; ++i)
        if (ids[i] >= idx->n) return fail(ROAR_ERR_ARG, "id %u outside the base set", ids[i]);
    std::lock_guard<std::mutex> guard(idx->mu);
    int e = use_device(idx);
    if (e) return e;
    float *q_dev = nullptr, *o_dev = nullptr;
    uint32_t *i_dev = nullptr;
    std::vector<float> buf;
    try {
        buf.assign((size_t)nq * idx->dim_pad, 0.0f);
    } catch (...) {
        return fail(ROAR_ERR_DEVICE, "out of host memory");
    }
    for (uint64_t i = 0; i < nq; ++i)
        memcpy(buf.data() + i * idx->dim_pad, queries + i * qstride_floats, idx->dim * sizeof(float));
    struct Free {  // every exit below releases the three device blocks
        float *&q, *&o;
        uint32_t *&i;
        ~Free() { (void)hipFree(q), (void)hipFree(o), (void)hipFree(i); }
    } free_all{q_dev, o_dev, i_dev};
    HIP_TRY(roar_dev_malloc((void **)&q_dev, buf.size() * sizeof(float)));
    HIP_TRY(roar_dev_malloc((void **)&i_dev, (size_t)nq * m * 4));
    HIP_TRY(roar_dev_malloc((void **)&o_dev, (size_t)nq * m * 4));
    HIP_TRY(hipMemcpy(q_dev, buf.data(), buf.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(i_dev, ids, (size_t)nq * m * 4, hipMemcpyHostToDevice));
    RoarDistParams p;
    memset(&p, 0, sizeof p);
    p.base = idx->base;
    p.ld = idx->ld;
    p.dim = idx->dim_pad;
    p.queries = q_dev;
    p.qstride = idx->dim_pad;
    p.nq = (uint32_t)nq;
    p.ids = i_dev;
    p.m = m;
    p.out = o_dev;
    p.off_ck = 0;
    p.off_cid = 8 * 64;
    p.off_q = p.off_cid + 4 * 64;
    p.lds_bytes = p.off_q + round_up(4 * idx->dim_pad, 16);
    int rc = roar_launch_distances(&p, idx->ip ? 1 : 0, nullptr);
    if (rc != 0) return fail(ROAR_ERR_DEVICE, "distance kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, o_dev, (size_t)nq * m * 4, hipMemcpyDeviceToHost));
    return ROAR_OK;
}

int roar_debug_screen_keeps(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats, const uint32_t *ids,
                            const float *thresholds, uint32_t m, uint8_t *keeps) {
    DeviceRestore keep_device_;
    if (!idx || !queries || !ids || !thresholds || !keeps) return fail(ROAR_ERR_ARG, "bad arguments");
    if (!idx->base) return fail(ROAR_ERR_ARG, "screen check before base vectors were loaded");
    if (nq == 0 || m == 0) return ROAR_OK;
    if (qstride_floats < idx->dim) return fail(ROAR_ERR_ARG, "query stride smaller than the dimension");
    if (!(idx->dim_pad == 128 || idx->dim_pad == 512 || (idx->ip && idx->dim_pad == 200)))
        return fail(ROAR_ERR_ARG, "no screening kernel for rows of %u floats with this metric", idx->dim_pad);
    for (uint64_t i = 0; i < nq * m; ++i)
        if (ids[i] >= idx->n) return fail(ROAR_ERR_ARG, "id %u outside the base set", ids[i]);
    std::lock_guard<std::mutex> guard(idx->mu);
    int e = use_device(idx);
    if (e) return e;
    ensure_screen(idx);
    if (!idx->screen) return fail(ROAR_ERR_DEVICE, "the screening copy could not be made");
    float *q_dev = nullptr, *c_dev = nullptr;
    uint32_t *i_dev = nullptr;
    std::vector<float> buf;
    try {
        buf.assign((size_t)nq * idx->dim_pad, 0.0f);
    } catch (...) {
        return fail(ROAR_ERR_DEVICE, "out of host memory");
    }
    for (uint64_t i = 0; i < nq; ++i)
        memcpy(buf.data() + i * idx->dim_pad, queries + i * qstride_floats, idx->dim * sizeof(float));
    struct Free {  // every exit below releases the three device blocks
        float *&q, *&c;
        uint32_t *&i;
        ~Free() { (void)hipFree(q), (void)hipFree(c), (void)hipFree(i); }
    } free_all{q_dev, c_dev, i_dev};
    HIP_TRY(roar_dev_malloc((void **)&q_dev, buf.size() * sizeof(float)));
    HIP_TRY(roar_dev_malloc((void **)&i_dev, (size_t)nq * m * 4));
    HIP_TRY(roar_dev_malloc((void **)&c_dev, (size_t)nq * m * 4));
    HIP_TRY(hipMemcpy(q_dev, buf.data(), buf.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(i_dev, ids, (size_t)nq * m * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c_dev, thresholds, (size_t)nq * m * 4, hipMemcpyHostToDevice));
    int rc = idx->ip ? roar_launch_screen_check_1(idx->screen, idx->screen_mu, idx->dim_pad, q_dev, idx->dim_pad, nq, i_dev, c_dev, m, idx->n, nullptr)
                     : roar_launch_screen_check_0(idx->screen, nullptr, idx->dim_pad, q_dev, idx->dim_pad, nq, i_dev, c_dev, m, idx->n, nullptr);
    if (rc != 0) return fail(ROAR_ERR_DEVICE, "screen check launch failed: %s", hipGetErrorString((hipError_t)rc));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<float> verdict;
    try {
        verdict.resize((size_t)nq * m);
    } catch (...) {
        return fail(ROAR_ERR_DEVICE, "out of host memory");
    }
    HIP_TRY(hipMemcpy(verdict.data(), c_dev, (size_t)nq * m * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < verdict.size(); ++i) keeps[i] = verdict[i] != 0.0f ? 1 : 0;
    return ROAR_OK;
}
int roar_merge_topk_device(int device, uint32_t nshards, uint64_t nq, uint32_t k, const uint32_t *ids_dev,
                           const float *dists_dev, uint32_t *out_ids_dev, float *out_dists_dev, void *hip_stream) {
    DeviceRestore keep_device_;
    if (nshards == 0 || nshards > 32 || k == 0) return fail(ROAR_ERR_ARG, "merge needs 1..32 shards and k > 0");
    if (!ids_dev || !dists_dev || !out_ids_dev || !out_dists_dev) return fail(ROAR_ERR_ARG, "bad merge arguments");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(ROAR_ERR_DEVICE, "no HIP device available; this library has no CPU path");
    HIP_TRY(hipSetDevice(device));
    int rc = roar_launch_merge_topk(nshards, nq, k, ids_dev, dists_dev, out_ids_dev, out_dists_dev, hip_stream);
    if (rc != 0) return fail(ROAR_ERR_DEVICE, "merge kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return ROAR_OK;
}

int roar_rebase_ids_device(int device, uint32_t *ids_dev, uint64_t count, uint32_t base, void *hip_stream) {
    DeviceRestore keep_device_;
    if (!ids_dev && count) return fail(ROAR_ERR_ARG, "bad rebase arguments");
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0)
        return fail(ROAR_ERR_DEVICE, "no HIP device available; this library has no CPU path");
    HIP_TRY(hipSetDevice(device));
    int rc = roar_launch_rebase_ids(ids_dev, count, base, hip_stream);
    if (rc != 0) return fail(ROAR_ERR_DEVICE, "rebase kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return ROAR_OK;
}

int roar_queue_trace(int device, uint32_t capacity, uint32_t n_ops, const uint8_t *ops, const uint32_t *ids, const float *dists,
                     int pool_in_registers, uint32_t *snap) {
    DeviceRestore keep_device_;
    if (!ops || !ids || !dists || !snap || capacity == 0 || capacity > 64)
        return fail(ROAR_ERR_ARG, "queue trace: capacity 1..64 and non-null arrays");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(ROAR_ERR_DEVICE, "no HIP device available; this library has no CPU path");
    HIP_TRY(hipSetDevice(device));
    if (n_ops == 0) return ROAR_OK;
    const size_t row = 2 + 3 * (size_t)capacity;
    uint8_t *d_ops = nullptr;
    uint32_t *d_ids = nullptr, *d_snap = nullptr;
    float *d_d = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_ops); (void)hipFree(d_ids); (void)hipFree(d_d); (void)hipFree(d_snap); };
    hipError_t e = roar_dev_malloc((void **)&d_ops, n_ops);
    if (e == hipSuccess) e = roar_dev_malloc((void **)&d_ids, (size_t)n_ops * 4);
    if (e == hipSuccess) e = roar_dev_malloc((void **)&d_d, (size_t)n_ops * 4);
    if (e == hipSuccess) e = roar_dev_malloc((void **)&d_snap, (size_t)n_ops * row * 4);
    if (e == hipSuccess) e = hipMemcpy(d_ops, ops, n_ops, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_ids, ids, (size_t)n_ops * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_d, dists, (size_t)n_ops * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = (hipError_t)roar_launch_queue_trace(capacity, n_ops, d_ops, d_ids, d_d, pool_in_registers ? 1 : 0, d_snap, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(snap, d_snap, (size_t)n_ops * row * 4, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail(ROAR_ERR_DEVICE, "queue trace failed: %s", hipGetErrorString(e));
    return ROAR_OK;
}

int roar_debug_fail_device_alloc_at(int64_t n) {
    const int64_t left = g_fail_alloc_in.exchange(n > 0 ? n : 0);
    return (int)std::min<int64_t>(std::max<int64_t>(left, 0), 0x7FFFFFFF);  // > 0: it was armed and has not fired
}

int roar_last_profile(roar_index *idx, uint64_t *out, uint64_t nq) {
    DeviceRestore keep_device_;
    if (!idx || !out) return fail(ROAR_ERR_ARG, "bad arguments");
    if (!idx->prof_dev || idx->prof_nq == 0) return fail(ROAR_ERR_ARG, "no instrumented search has run (option profile_phases)");
    if (nq > idx->prof_nq) return fail(ROAR_ERR_ARG, "the last instrumented launch had %llu queries", (unsigned long long)idx->prof_nq);
    int e = use_device(idx);
    if (e) return e;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, idx->prof_dev, (size_t)nq * 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return ROAR_OK;
}

int roar_set_traffic_counters_device(roar_index *idx, uint32_t *counters_dev, uint64_t capacity_queries) {
    if (!idx) return fail(ROAR_ERR_ARG, "bad arguments");
    if (counters_dev && ((uintptr_t)counters_dev & 15u)) return fail(ROAR_ERR_ARG, "the counter block must be 16-byte aligned");
    std::lock_guard<std::mutex> guard(idx->mu);
    idx->traffic_dev = counters_dev;
    idx->traffic_cap = counters_dev ? capacity_queries : 0;
    return ROAR_OK;
}

int roar_set_option(roar_index *idx, const char *key, int64_t value) {
    if (!idx || !key) return fail(ROAR_ERR_ARG, "bad arguments");
    std::string k(key);
    if (k == "visited_mode") {
        if (value != ROAR_VISITED_EXACT && value != ROAR_VISITED_LDS && value != ROAR_VISITED_AUTO && value != ROAR_VISITED_LDS_FORGETFUL)
            return fail(ROAR_ERR_ARG, "bad visited_mode");
        idx->visited_mode = (int)value;
    } else if (k == "waves_per_cu") {
        if (value < 0 || value > 32) return fail(ROAR_ERR_ARG, "waves_per_cu must be 0..32");
        idx->waves_per_cu = (int)value;
    } else if (k == "vis_entries") {
        {
            int64_t v = value;
            while (v > 5 && !(v & 1)) v >>= 1;
            if (value < 0 || (value > 0 && v != 1 && v != 2 && v != 3 && v != 4 && v != 5))
                return fail(ROAR_ERR_ARG, "vis_entries must be 0, 2^k, 3 * 2^k or 5 * 2^k");
        }
        idx->vis_entries = (int)value;
    } else if (k == "tag_protect") {
        if (value < -1 || value > 1) return fail(ROAR_ERR_ARG, "tag_protect must be -1 (measured), 0 or 1");
        if (idx->tag_protect != (int)value) idx->plan_cache.clear();  // (table sizes were measured with the other setting)
        idx->tag_protect = (int)value;
    } else if (k == "search_kernel") {
        if (value < 0 || value > 3) return fail(ROAR_ERR_ARG, "search_kernel must be 0 (auto), 1, 2 or 3");
        if (idx->search_kernel != (int)value) idx->plan_cache.clear();  // (another kernel form: measured choices no longer apply)
        idx->search_kernel = (int)value;
    } else if (k == "hub_mode") {
        if (value < 0 || value > 2) return fail(ROAR_ERR_ARG, "hub_mode must be 0 (auto), 1 (off) or 2 (on)");
        if (idx->hub_mode != (int)value) idx->plan_cache.clear();  // (another kernel form: measured choices no longer apply)
        idx->hub_mode = (int)value;
    } else if (k == "launch_events") {
        idx->launch_events = value ? 1 : 0;
    } else if (k == "knn_mode") {
        if (value != 0 && value != 1) return fail(ROAR_ERR_ARG, "knn_mode must be 0 (auto: bf16 screen + exact re-rank) or 1 (f32 MFMA only)");
        idx->knn_mode = (int)value;
    } else if (k == "calibrate") {
        if (value < 0 || value > 2) return fail(ROAR_ERR_ARG, "calibrate must be 0 (defaults), 1 (measure, default) or 2 (forget what was measured)");
        if (value == 2) idx->plan_cache.clear();
        else idx->calibrate = (int)value;
    } else if (k == "screen") {
        if (value < 0 || value > 2) return fail(ROAR_ERR_ARG, "screen must be 0 (auto), 1 (off) or 2 (on)");
        idx->screen_mode = (int)value;
    } else if (k == "pool_form") {
        if (value != 0 && value != 1) return fail(ROAR_ERR_ARG, "pool_form must be 0 (auto) or 1 (LDS pool)");
        if (idx->pool_form != (int)value) idx->plan_cache.clear();  // (another kernel form: measured choices no longer apply)
        idx->pool_form = (int)value;
    } else if (k == "profile_phases") {
        idx->profile = value ? 1 : 0;
    } else if (k == "combine_single_queries") {
        idx->combine = value ? 1 : 0;
    } else if (k == "bipartite_seed") {
        idx->bp_seed = (uint64_t)value;
    } else if (k == "edge_floats") {
        if (value != 0 && value != 1) return fail(ROAR_ERR_ARG, "edge_floats must be 0 (auto) or 1 (off)");
        if (idx->edge_floats_mode != (int)value) idx->plan_cache.clear();  // (another kernel form: measured choices no longer apply)
        idx->edge_floats_mode = (int)value;
    } else if (k == "reset_launch_cells") {
        // after a launch of the device-buffer form reported 0xFFFFFFFF in *short_count_dev (its cell was left dirty by a launch that did
        // not finish): with no launch of this index in flight, put every cell back
        DeviceRestore keep_device_;
        int e = use_device(idx);
        if (e) return e;
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemset(idx->ring, 0, LAUNCH_RING * 5 * sizeof(uint32_t)));
        HIP_TRY(hipDeviceSynchronize());
        idx->launch_seq.store(0);
    } else if (k == "debug_dirty_launch_cell") {
        // tests: leave the NEXT launch's cell as a launch that died would have (word 0 advanced, a stale stamp)
        DeviceRestore keep_device_;
        int e = use_device(idx);
        if (e) return e;
        const uint32_t seq = idx->launch_seq.load();
        const uint32_t dirty[4] = {(uint32_t)value, 0u, 0u, 0u}, stale = seq / LAUNCH_RING + 7u;
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(idx->ring + 4 * (size_t)(seq % LAUNCH_RING), dirty, sizeof dirty, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(idx->ring + 4 * (size_t)LAUNCH_RING + (seq % LAUNCH_RING), &stale, sizeof stale, hipMemcpyHostToDevice));
    } else if (k == "max_grid") {
        if (value < 0) return fail(ROAR_ERR_ARG, "max_grid must be >= 0");
        idx->max_grid = (int)value;
    } else {
        return fail(ROAR_ERR_ARG, "Invalid parameter name.");  // parameters.h:28
    }
    return ROAR_OK;
}

int roar_get_info(roar_index *idx, const char *key, int64_t *value) {
    DeviceRestore keep_device_;
    if (!idx || !key || !value) return fail(ROAR_ERR_ARG, "bad arguments");
    std::string k(key);
    if (k == "n") *value = (int64_t)idx->n;
    else if (k == "dim") *value = idx->dim;
    else if (k == "dim_padded") *value = idx->dim_pad;
    else if (k == "ep") *value = idx->ep;
    else if (k == "max_degree") *value = idx->max_deg;
    else if (k == "nedges") *value = (int64_t)idx->nedges;
    else if (k == "adj_stride") *value = idx->adj_stride;
    else if (k == "row_stride") *value = idx->ld;
    else if (k == "n_cus") *value = idx->n_cus;
    else if (k == "need_normalize") *value = idx->need_normalize ? 1 : 0;
    else if (k == "visited_mode") *value = idx->visited_mode;
    else if (k == "last_visited_mode") *value = idx->last_mode;
    else if (k == "last_visited_forgetful") *value = idx->last_forgetful;
    else if (k == "last_waves_per_query")
        *value = idx->last_sel.form == ROAR_FORM_BIPARTITE_WG ? 8
                 : idx->last_sel.form == ROAR_FORM_SEARCH ? idx->last_sel.waves : 1;
    else if (k == "last_register_pool") *value = idx->last_sel.regpool;
    else if (k == "last_hub_bitmap") *value = idx->last_sel.hub;
    else if (k == "last_screen") *value = idx->last_screen;
    else if (k == "last_tag_protect") *value = idx->last_tag_protect;
    else if (k == "last_plan_calibrated") *value = idx->last_calibrated;
    else if (k == "last_knn_bf16") *value = idx->last_knn_bf16;
    else if (k == "last_knn_rerank_rows") *value = idx->last_knn_rerank_rows;
    else if (k == "calibrations") *value = (int64_t)idx->plan_cache.size();
    else if (k == "last_edge_floats") *value = idx->last_ann;
    else if (k == "hw_queues_env") {  // GPU_MAX_HW_QUEUES as the environment holds it (0: unset) -- what the HIP runtime read if it started after this library was loaded
        const char *v = getenv("GPU_MAX_HW_QUEUES");
        *value = v ? atoll(v) : 0;
    } else if (k == "screen_row_bytes") *value = roar_screen8_row_bytes(idx->dim_pad);
    else if (k == "launch_events_on") *value = idx->launch_events;
    else if (k == "last_combined") *value = idx->last_combined;
    else if (k == "bipartite_nodes") *value = (int64_t)idx->bp_n;
    else if (k == "bipartite_edges") *value = (int64_t)idx->bp_edges;
    else if (k == "graph_hash") {
        if (!idx->adj || idx->gn == 0) return fail(ROAR_ERR_ARG, "no graph loaded");
        if (!idx->graph_hash_valid) {
            HIP_TRY(hipSetDevice(idx->device));
            unsigned long long *h_dev = nullptr, h = 0;
            HIP_TRY(roar_dev_malloc((void **)&h_dev, 8));
            int rc = roar_bk_graph_hash(idx->adj, idx->adj_stride, idx->gn, idx->ep, h_dev, nullptr);
            hipError_t he = rc == 0 ? hipMemcpy(&h, h_dev, 8, hipMemcpyDeviceToHost) : (hipError_t)rc;
            (void)hipFree(h_dev);
            if (he != hipSuccess) return fail(ROAR_ERR_DEVICE, "graph hash failed: %s", hipGetErrorString(he));
            idx->graph_hash = h;
            idx->graph_hash_valid = true;
        }
        *value = (int64_t)idx->graph_hash;
    }
    else if (k == "last_grid") *value = idx->last_grid;
    else if (k == "last_workgroups_per_cu") *value = idx->last_wpc;
    else if (k == "last_lds_bytes") *value = idx->last_lds;
    else if (k == "last_vis_entries") *value = idx->last_vis_entries;
    else if (k == "last_kernel_ns") {
        if (!idx->ev_valid) return fail(ROAR_ERR_ARG, "no search has been launched yet");
        HIP_TRY(hipSetDevice(idx->device));
        HIP_TRY(hipEventSynchronize(idx->ev1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, idx->ev0, idx->ev1));
        *value = (int64_t)((double)ms * 1e6);
    } else {
        return fail(ROAR_ERR_ARG, "Invalid parameter name.");
    }
    return ROAR_OK;
}
}  // extern "C"
