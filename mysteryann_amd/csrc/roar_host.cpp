// roar_host.cpp -- the host-buffer search calls: query staging / packing behind a running launch, result fetch, combining of concurrent one-query calls, replicas (roar_clone_to_device, roar_search_batch_multi), pinned host memory
// (round 5: roar_capi.cpp was one 3 203-line translation unit; it is now cut by concern.  Shared declarations: roar_internal.h.)
#include "roar_internal.h"

using namespace roar_impl;

namespace roar_impl {

// Host-buffer queries -> something the kernel can read.  Rows are padded to dim_pad with zeros (data_align of the
// query block, tests/test_search_roargraph.cpp:131).  Two ways, neither with a host-to-device copy in front of the
// kernel: (1) the caller's block is already pinned (roar_host_alloc, hipHostMalloc, hipHostRegister), strided at
// >= dim_pad floats with nothing to zero -> the kernel reads it in place over PCIe (each query is read once, 4 * dim
// bytes, at its own start: 8 MB per 10 000 queries, hidden under the search); (2) otherwise the rows are packed into
// the index's pinned staging block by a few host threads and read from there.  *qdev / *qstride: what to launch with.
int stage_queries(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats, uint32_t k, const float **qdev,
                  uint32_t *qstride, bool *deferred) {
    *deferred = false;
    const uint64_t owords = nq * (2ull * k + 3) + 1;
    if (idx->out_cap < owords) {
        if (idx->out_dev) (void)hipFree(idx->out_dev);
        if (idx->out_pin) (void)hipHostFree(idx->out_pin);
        idx->out_dev = idx->out_pin = nullptr;
        idx->out_cap = 0;
        HIP_TRY(roar_dev_malloc((void **)&idx->out_dev, (size_t)owords * sizeof(uint32_t)));
        HIP_TRY(hipHostMalloc((void **)&idx->out_pin, (size_t)owords * sizeof(uint32_t), hipHostMallocDefault));
        idx->out_cap = owords;
    }
    if (idx->dim == idx->dim_pad && qstride_floats >= idx->dim_pad && (qstride_floats & 3) == 0) {
        hipPointerAttribute_t attr;
        memset(&attr, 0, sizeof attr);
        if (hipPointerGetAttributes(&attr, queries) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer) {
            *qdev = static_cast<const float *>(attr.devicePointer);
            *qstride = qstride_floats;
            return ROAR_OK;
        }
        (void)hipGetLastError();  // an ordinary (pageable) pointer is reported as an error by some runtimes
    }
    // staging rows start on 128-byte lines: the kernel reads rows while later ones are still being written
    // (pack_queries), and no line it fetches may hold bytes of a row that is not there yet
    const uint32_t sp = round_up(idx->dim_pad, 32);
    const uint64_t qfloats = nq * sp;
    if (idx->q_cap < qfloats) {
        if (idx->q_pin) (void)hipHostFree(idx->q_pin);
        idx->q_pin = idx->q_pin_dev = nullptr;
        idx->q_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&idx->q_pin, (size_t)qfloats * sizeof(float), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void **)&idx->q_pin_dev, idx->q_pin, 0));
        idx->q_cap = qfloats;
    }
    if (!idx->ready_pin) {
        HIP_TRY(hipHostMalloc((void **)&idx->ready_pin, 64, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void **)&idx->ready_dev, idx->ready_pin, 0));
    }
    *idx->ready_pin = 0;  // nothing packed yet: pack_queries() runs behind the launch
    *qdev = idx->q_pin_dev;
    *qstride = sp;
    *deferred = true;
    return ROAR_OK;
}

// The packing itself, on the calling thread, while the kernel is already running: the kernel takes queries in row order
// and waits for the counter (RoarSearchParams::ready), so the 8 MB copy of a 10 000-query block costs the launch nothing
// but the first few rows (two helper threads claiming chunks in order changed nothing: 2.51 vs 2.54 M QPS on the T2I-10M
// bench).  Plain memcpy / memset: nothing here can throw or fail, the counter always reaches nq.
void pack_queries(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats) {
    const uint32_t dp = round_up(idx->dim_pad, 32), d = idx->dim;  // (the stride stage_queries chose)
    volatile uint32_t *ready = idx->ready_pin;
    for (uint64_t r0 = 0; r0 < nq;) {
        const uint64_t r1 = std::min<uint64_t>(nq, r0 + (r0 < 4096 ? 256 : 1024));  // small steps first: the launch is waiting
        for (uint64_t i = r0; i < r1; ++i) {
            float *dst = idx->q_pin + i * dp;
            memcpy(dst, queries + i * qstride_floats, d * sizeof(float));
            if (dp > d) memset(dst + d, 0, (dp - d) * sizeof(float));  // anything past dim is padding and must read as zero
        }
        std::atomic_thread_fence(std::memory_order_release);
        *ready = (uint32_t)r1;
        r0 = r1;
    }
}

// results: one asynchronous copy of the block [ids | dists | cmps | hops | scanned | short] into pinned memory,
// then plain copies into the caller's arrays
int fetch_results(roar_index *idx, uint64_t nq, uint32_t k, uint32_t *ids, float *dists, uint32_t *cmps, uint32_t *hops,
                  uint32_t *scanned) {
    const uint64_t words = nq * (2ull * k + 3) + 1;
    HIP_TRY(hipMemcpyAsync(idx->out_pin, idx->out_dev, (size_t)words * 4, hipMemcpyDeviceToHost, idx->stream));
    HIP_TRY(hipStreamSynchronize(idx->stream));
    const uint32_t *blk = idx->out_pin;
    memcpy(ids, blk, (size_t)nq * k * 4);
    memcpy(dists, blk + nq * k, (size_t)nq * k * 4);
    if (cmps) memcpy(cmps, blk + 2 * nq * k, (size_t)nq * 4);
    if (hops) memcpy(hops, blk + 2 * nq * k + nq, (size_t)nq * 4);
    if (scanned) memcpy(scanned, blk + 2 * nq * k + 2 * nq, (size_t)nq * 4);
    const uint32_t nshort = blk[(size_t)words - 1];
    if (nshort == 0xFFFFFFFFu) {
        // the launch found its cell as a launch that did not finish left it (RoarSearchParams::counter) and did not search.  This
        // call's stream is idle now (synchronised above): every cell is put back, so the next call works -- if the device still does.
        (void)hipMemset(idx->ring, 0, LAUNCH_RING * 5 * sizeof(uint32_t));
        (void)hipDeviceSynchronize();
        idx->launch_seq.store(0);
        return fail(ROAR_ERR_DEVICE, "the search did not run: its launch cell was left dirty by an earlier launch that did not finish (a device fault, or "
                                     "more than %u launches of this index in flight); the cells have been reset", LAUNCH_RING);
    }
    if (nshort) {
        // message of the first short query, as the reference would have thrown it (:2408-2412)
        for (uint64_t i = 0; i < nq; ++i) {
            uint32_t have = 0;
            while (have < k && ids[i * k + have] != 0xFFFFFFFFu) have++;
            if (have < k) return fail(ROAR_ERR_SHORT, "not enough results: %u, expected: %u", have, k);
        }
        return fail(ROAR_ERR_SHORT, "not enough results");
    }
    return ROAR_OK;
}
}  // namespace roar_impl

namespace roar_impl {

// The reference's harness calls SearchRoarGraph once per query from num_threads OpenMP threads
// (tests/test_search_roargraph.cpp:203-209).  One-query launches would leave the GPU idle, so concurrent
// one-query calls are combined: a caller that finds no leader becomes the leader, takes every request queued with
// the same (k, L_pq) as the oldest one -- those that arrived while the previous launch was in flight -- runs them as
// one batch and hands each caller its own rows.  A leader serves exactly one batch: when its own request is done it
// returns, and a caller still waiting takes over (so no caller's latency grows with the load of the others).  Nothing
// thrown inside a leader's turn escapes the C ABI or leaves the queue without a leader: the batch is failed instead.
// Results are those of the batch path, i.e. of the reference, query by query.
int search_one(roar_index *idx, roar_index::OneQuery &req) {
    std::unique_lock<std::mutex> lk(idx->cq_mu);
    try {
        idx->cq.push_back(&req);
    } catch (...) {
        return fail(ROAR_ERR_DEVICE, "out of host memory");
    }
    while (!req.done) {
        if (idx->cq_leader) {
            idx->cq_cv.wait(lk, [&] { return req.done || !idx->cq_leader; });
            continue;
        }
        idx->cq_leader = true;
        std::vector<roar_index::OneQuery *> batch;
        int rc = ROAR_OK;
        std::string err;
        try {
            std::vector<roar_index::OneQuery *> rest;
            const roar_index::OneQuery *head = idx->cq.front();
            for (roar_index::OneQuery *r : idx->cq)
                ((r->k == head->k && r->L == head->L && batch.size() < 4096) ? batch : rest).push_back(r);
            idx->cq.swap(rest);
            lk.unlock();
            const uint64_t m = batch.size();
            const uint32_t k = head->k, dp = idx->dim_pad;
            std::vector<float> qbuf((size_t)m * dp, 0.0f);
            for (uint64_t i = 0; i < m; ++i)
                memcpy(&qbuf[i * dp], batch[i]->q, std::min(batch[i]->qstride, idx->dim) * sizeof(float));
            std::vector<uint32_t> r_ids((size_t)m * k), r_cmps(m), r_hops(m), r_scan(m);
            std::vector<float> r_d((size_t)m * k);
            rc = search_host(idx, qbuf.data(), m, dp, k, head->L, r_ids.data(), r_d.data(), r_cmps.data(), r_hops.data(),
                             r_scan.data());
            err = g_err;
            for (uint64_t i = 0; i < m; ++i) {
                roar_index::OneQuery *r = batch[i];
                r->rc = rc;
                if (rc == ROAR_OK || rc == ROAR_ERR_SHORT) {
                    memcpy(r->ids, &r_ids[i * k], k * sizeof(uint32_t));
                    memcpy(r->dists, &r_d[i * k], k * sizeof(float));
                    if (r->cmps) *r->cmps = r_cmps[i];
                    if (r->hops) *r->hops = r_hops[i];
                    if (r->scanned) *r->scanned = r_scan[i];
                    uint32_t have = 0;
                    while (have < k && r_ids[i * k + have] != 0xFFFFFFFFu) have++;
                    r->rc = ROAR_OK;
                    if (have < k) {  // this query's own shortfall, worded as the reference throws it (:2408-2412)
                        char msg[96];
                        snprintf(msg, sizeof msg, "not enough results: %u, expected: %u", have, k);
                        r->rc = ROAR_ERR_SHORT;
                        r->err = msg;
                    }
                } else {
                    r->err = err;
                }
            }
        } catch (...) {  // std::bad_alloc in the staging vectors: fail this batch, keep the queue alive
            for (roar_index::OneQuery *r : batch) {
                r->rc = ROAR_ERR_DEVICE;
                try {
                    r->err = "out of host memory while combining one-query calls";
                } catch (...) {
                }
            }
        }
        if (!lk.owns_lock()) lk.lock();
        idx->last_combined = (int64_t)batch.size();
        for (roar_index::OneQuery *r : batch) r->done = true;
        idx->cq_leader = false;
        idx->cq_cv.notify_all();
    }
    if (req.rc != ROAR_OK) g_err = req.err;
    return req.rc;
}
}  // namespace roar_impl

namespace roar_impl {

int search_host(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats, uint32_t k, uint32_t L_pq,
                uint32_t *ids, float *dists, uint32_t *cmps, uint32_t *hops, uint32_t *scanned) {
    std::lock_guard<std::mutex> guard(idx->mu);
    int e = use_device(idx);
    if (e) return e;
    const float *qdev = nullptr;
    uint32_t qs = 0;
    bool deferred = false;
    if ((e = stage_queries(idx, queries, nq, qstride_floats, k, &qdev, &qs, &deferred))) return e;
    LaunchPlan probe;
    bool have_plan = false;
    if (deferred) {
        // the kernel forms that can wait for the packing thread: one wave per query, no hub bitmap, not instrumented
        // (the plan made here is the one the launch uses: planning can be expensive -- tag pools, hub rows)
        bool need_queries = false;
        if ((e = resolve_plan(idx, nullptr, nq, qs, k, L_pq, idx->stream, &probe, &need_queries))) return e;
        if (need_queries) {  // a launch plan calibration is due: it reads the queries, so they are packed in front of it
            pack_queries(idx, queries, nq, qstride_floats);
            deferred = false;
        } else {
            have_plan = true;
            const bool gate = probe.sel.form == ROAR_FORM_SEARCH && probe.sel.waves == 1 && !probe.sel.hub && !probe.sel.prof &&
                              !(idx->traffic_dev && nq <= idx->traffic_cap);
            if (!gate) {  // pack in front of the launch
                pack_queries(idx, queries, nq, qstride_floats);
                deferred = false;
            }
        }
    }
    uint32_t *d_ids = idx->out_dev;
    float *d_dists = reinterpret_cast<float *>(d_ids + nq * k);
    uint32_t *d_cmps = d_ids + 2 * nq * k;
    uint32_t *d_hops = d_cmps + nq;
    uint32_t *d_scan = d_hops + nq;
    uint32_t *d_short = d_scan + nq;
    // The gated launch waits for rows that are packed behind it -- by a helper thread that is started BEFORE the launch,
    // so that a launch call that only returns when the kernel has finished (HIP_LAUNCH_BLOCKING, AMD_SERIALIZE_KERNEL,
    // a debugger or a tool that serialises dispatches) still finds its rows arriving.  No helper thread, no gate.
    std::thread packer;
    if (deferred) {
        try {
            packer = std::thread(pack_queries, idx, queries, nq, qstride_floats);
        } catch (...) {
            pack_queries(idx, queries, nq, qstride_floats);
            deferred = false;
        }
    }
    e = search_device(idx, qdev, nq, qs, k, L_pq, d_ids, d_dists, d_cmps, d_hops, d_scan, d_short, idx->stream,
                      deferred ? idx->ready_dev : nullptr, have_plan ? &probe : nullptr);
    if (packer.joinable()) packer.join();  // (plain memcpy: it always runs to the end, launch or no launch)
    if (e) return e;
    return fetch_results(idx, nq, k, ids, dists, cmps, hops, scanned);
}
}  // namespace roar_impl

extern "C" {

int roar_search_batch(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats, uint32_t k,
                      uint32_t L_pq, uint32_t *ids, float *dists, uint32_t *cmps, uint32_t *hops,
                      uint32_t *scanned) {
    DeviceRestore keep_device_;
    if (!idx || !ids || !dists || (nq && !queries)) return fail(ROAR_ERR_ARG, "bad search arguments");
    if (nq == 0) return ROAR_OK;
    if (qstride_floats < idx->dim) return fail(ROAR_ERR_ARG, "query stride smaller than the dimension");
    if (nq == 1 && idx->combine && k > 0 && k <= 4096) {
        roar_index::OneQuery req;
        req.q = queries, req.qstride = qstride_floats, req.k = k, req.L = L_pq;
        req.ids = ids, req.dists = dists, req.cmps = cmps, req.hops = hops, req.scanned = scanned;
        return search_one(idx, req);
    }
    return search_host(idx, queries, nq, qstride_floats, k, L_pq, ids, dists, cmps, hops, scanned);
}
// ---- replicated index, one query block split over g devices (BASELINE configs[3]) --------------------

int roar_clone_to_device(roar_index *src, int device, roar_index **out) {
    DeviceRestore keep_device_;
    if (!src || !out) return fail(ROAR_ERR_ARG, "bad arguments");
    *out = nullptr;
    if (!src->base || src->n == 0) return fail(ROAR_ERR_ARG, "clone before base vectors were loaded");
    roar_index *dst = nullptr;
    int e = roar_create(&dst, src->dim, src->metric, device);
    if (e) return e;
    auto bail = [&](int code) {
        const std::string keep = g_err;
        roar_destroy(dst);
        g_err = keep;
        return code;
    };
    std::lock_guard<std::mutex> guard(src->mu);
    dst->visited_mode = src->visited_mode;
    dst->waves_per_cu = src->waves_per_cu;
    dst->vis_entries = src->vis_entries;
    dst->max_grid = src->max_grid;
    dst->edge_floats_mode = src->edge_floats_mode;
    dst->search_kernel = src->search_kernel;
    dst->pool_form = src->pool_form;
    dst->combine = src->combine;
    dst->hub_mode = src->hub_mode;
    dst->profile = src->profile;
    dst->screen_mode = src->screen_mode;
    dst->calibrate = src->calibrate;
    if ((e = alloc_base(dst, src->n))) return bail(e);
    // rows are re-strided to the padded dimension if the source adopted a wider block
    hipError_t he = hipSuccess;
    if (src->ld == dst->ld) {
        he = hipMemcpyPeer(dst->base, dst->device, src->base, src->device, (size_t)src->n * src->ld * sizeof(float));
    } else {
        he = hipMemcpy2D(dst->base, (size_t)dst->ld * 4, src->base, (size_t)src->ld * 4, (size_t)dst->ld * 4, (size_t)src->n,
                         hipMemcpyDefault);
    }
    if (he != hipSuccess) return bail(fail(ROAR_ERR_DEVICE, "copy of the base rows to device %d failed: %s", device, hipGetErrorString(he)));
    if (src->adj) {
        const size_t words = (size_t)src->gn * src->adj_stride + ADJ_SLACK_WORDS;
        if (roar_dev_malloc((void **)&dst->adj, words * sizeof(uint32_t)) != hipSuccess)
            return bail(fail(ROAR_ERR_DEVICE, "out of device memory for the graph replica"));
        he = hipMemcpyPeer(dst->adj, dst->device, src->adj, src->device, words * sizeof(uint32_t));
        if (he != hipSuccess) return bail(fail(ROAR_ERR_DEVICE, "copy of the graph to device %d failed: %s", device, hipGetErrorString(he)));
        dst->adj_stride = src->adj_stride;
        dst->ep = src->ep;
        dst->max_deg = src->max_deg;
        dst->gn = src->gn;
        dst->nedges = src->nedges;  // the host CSR copy is rebuilt from the device rows if it is ever asked for
    }
    he = hipDeviceSynchronize();
    if (he != hipSuccess) return bail(fail(ROAR_ERR_DEVICE, "replica on device %d: %s", device, hipGetErrorString(he)));
    *out = dst;
    return ROAR_OK;
}

int roar_search_batch_multi(roar_index *const *replicas, uint32_t g, const float *queries, uint64_t nq, uint32_t qstride_floats,
                            uint32_t k, uint32_t L_pq, uint32_t *ids, float *dists, uint32_t *cmps, uint32_t *hops,
                            uint32_t *scanned) {
    DeviceRestore keep_device_;
    if (!replicas || g == 0 || g > 64) return fail(ROAR_ERR_ARG, "between 1 and 64 replicas");
    if (!ids || !dists || (nq && !queries)) return fail(ROAR_ERR_ARG, "bad search arguments");
    for (uint32_t r = 0; r < g; ++r) {
        if (!replicas[r]) return fail(ROAR_ERR_ARG, "replica %u is null", r);
        if (replicas[r]->dim != replicas[0]->dim || replicas[r]->n != replicas[0]->n || replicas[r]->gn != replicas[0]->gn ||
            replicas[r]->metric != replicas[0]->metric)
            return fail(ROAR_ERR_ARG, "replica %u does not hold the same index as replica 0", r);
    }
    if (nq == 0) return ROAR_OK;
    if (qstride_floats < replicas[0]->dim) return fail(ROAR_ERR_ARG, "query stride smaller than the dimension");
    // replica r takes queries [r * nq / g, (r + 1) * nq / g) and writes the same rows of the caller's arrays
    std::vector<int> rc(g, ROAR_OK);
    std::vector<std::string> msg(g);
    auto work = [&](uint32_t r) {
        const uint64_t lo = nq * r / g, hi = nq * (r + 1) / g;
        if (hi == lo) return;
        try {
            rc[r] = search_host(replicas[r], queries + lo * qstride_floats, hi - lo, qstride_floats, k, L_pq, ids + lo * k,
                                dists + lo * k, cmps ? cmps + lo : nullptr, hops ? hops + lo : nullptr,
                                scanned ? scanned + lo : nullptr);
            if (rc[r] != ROAR_OK) msg[r] = g_err;
        } catch (...) {
            rc[r] = ROAR_ERR_DEVICE;
            msg[r] = "out of host memory in a replica's search";
        }
    };
    {
        std::vector<std::thread> th;
        try {
            for (uint32_t r = 1; r < g; ++r) th.emplace_back(work, r);
        } catch (...) {
            for (auto &x : th) x.join();
            return fail(ROAR_ERR_DEVICE, "cannot start the per-device host threads");
        }
        work(0);
        for (auto &x : th) x.join();
    }
    int first_short = -1;
    for (uint32_t r = 0; r < g; ++r) {
        if (rc[r] == ROAR_ERR_SHORT) {
            if (first_short < 0) first_short = (int)r;
        } else if (rc[r] != ROAR_OK) {
            g_err = msg[r];
            return rc[r];
        }
    }
    if (first_short >= 0) {
        g_err = msg[first_short];
        return ROAR_ERR_SHORT;
    }
    return ROAR_OK;
}

// ---- pinned host memory for query / result blocks ----------------------------------------------------------------------

void *roar_host_alloc(uint64_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, (size_t)(bytes ? bytes : 1), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        fail(ROAR_ERR_DEVICE, "pinned host allocation of %llu bytes failed", (unsigned long long)bytes);
        return nullptr;
    }
    return p;
}

void roar_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int roar_host_register(void *p, uint64_t bytes) {
    if (!p || bytes == 0) return fail(ROAR_ERR_ARG, "bad arguments");
    const hipError_t e = hipHostRegister(p, (size_t)bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(ROAR_ERR_DEVICE, "hipHostRegister of %llu bytes failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
    }
    return ROAR_OK;
}

int roar_host_unregister(void *p) {
    if (!p) return fail(ROAR_ERR_ARG, "bad arguments");
    const hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(ROAR_ERR_DEVICE, "hipHostUnregister failed: %s", hipGetErrorString(e));
    }
    return ROAR_OK;
}
}  // extern "C"
