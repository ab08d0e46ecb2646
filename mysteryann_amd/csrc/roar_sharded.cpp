// roar_sharded.cpp -- the index-sharded configuration behind the C ABI (include/roargraph_amd.h, "index-sharded search";
// SURVEY 8(e) config 5, BASELINE configs[4]): the base set is cut into contiguous id ranges, every shard has its own
// RoarGraph over its rows (local ids, own entry point), every query is searched on every shard with the same L_pq, and
// the per-shard top-k lists meet in ONE exchange step -- an RCCL all-gather over xGMI -- before a k-way merge by
// (distance, id).  Not in the reference (its one index lives in one address space); the checker is the same merge over
// the CPU searches of the shards.
//
// One process.  A "rank" is a distinct device: the shards placed on it are searched back to back on the rank's stream
// (so 8 shards on 8 devices, 8 shards on 4 devices or -- the one-GPU test -- 2 shards on one device are the same code),
// then every rank contributes its block [local shard][nq][k] of ids and of distances to ncclAllGather and merges what it
// receives; rank 0's merged lists go back to the caller.  Host side of a call: one thread per rank stages the query block
// on its device and enqueues the rank's searches; the collective and the merges are then enqueued for all ranks from the
// calling thread inside one ncclGroupStart / ncclGroupEnd (the single-process form of the collective), so a rank that
// failed before the exchange can never leave the others waiting inside it.
//
// Built on the public C ABI only (roar_create, roar_search_batch_device, roar_merge_topk_device, roar_rebase_ids_device):
// nothing here reaches into an index object.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/roargraph_amd.h"

extern "C" void roar_set_last_error(const char *msg);  // roar_capi.cpp: the thread-local message of roar_last_error()

namespace {

int sfail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    roar_set_last_error(buf);
    return code;
}

struct DeviceRestore {  // the caller's current device is put back on every exit
    int prev = -1;
    DeviceRestore() {
        if (hipGetDevice(&prev) != hipSuccess) {
            (void)hipGetLastError();
            prev = -1;
        }
    }
    ~DeviceRestore() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

struct Rank {
    int device = 0;
    std::vector<uint32_t> shards;  // shard numbers placed on this device, ascending
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    hipEvent_t ev_start = nullptr, ev_searched = nullptr, ev_done = nullptr;
    // per-call device buffers (grown on demand)
    float *q_dev = nullptr;
    uint32_t *send_ids = nullptr, *recv_ids = nullptr, *out_ids = nullptr, *counts = nullptr;  // counts: [local][2][nq] cmps, hops
    float *send_d = nullptr, *recv_d = nullptr, *out_d = nullptr;
    uint64_t cap_q = 0, cap_res = 0, cap_cnt = 0;
};

}  // namespace

struct roar_sharded {
    uint32_t dim = 0, dim_pad = 0;
    int metric = 0;
    std::vector<roar_index *> shard;   // owned
    std::vector<int> shard_rank;       // which rank holds shard s
    std::vector<uint32_t> shard_slot;  // its position among that rank's shards
    std::vector<uint64_t> id_base;     // global id of the shard's row 0
    std::vector<bool> id_base_set;
    std::vector<Rank> ranks;
    uint32_t max_local = 0;
    bool comms_ready = false;
    std::mutex mu;
    // pinned staging of the padded query block and of the results
    float *q_pin = nullptr;
    uint64_t q_pin_cap = 0;
    uint32_t *res_pin = nullptr;
    uint64_t res_pin_cap = 0;
    int64_t last_search_ns = 0, last_exchange_ns = 0;
};

namespace {

#define S_HIP(expr)                                                                                \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return sfail(ROAR_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define S_NCCL(expr)                                                                                  \
    do {                                                                                              \
        ncclResult_t r_ = (expr);                                                                     \
        if (r_ != ncclSuccess) return sfail(ROAR_ERR_DEVICE, "%s: %s", #expr, ncclGetErrorString(r_)); \
    } while (0)

void free_rank_buffers(Rank &r) {
    (void)hipSetDevice(r.device);
    (void)hipFree(r.q_dev), (void)hipFree(r.send_ids), (void)hipFree(r.recv_ids), (void)hipFree(r.out_ids), (void)hipFree(r.counts);
    (void)hipFree(r.send_d), (void)hipFree(r.recv_d), (void)hipFree(r.out_d);
    r.q_dev = nullptr, r.send_ids = r.recv_ids = r.out_ids = r.counts = nullptr, r.send_d = r.recv_d = r.out_d = nullptr;
    r.cap_q = r.cap_res = r.cap_cnt = 0;
}

// Communicators: one per rank, made on first use (ncclCommInitAll, the single-process initialiser).
int ensure_comms(roar_sharded *sh) {
    if (sh->comms_ready) return ROAR_OK;
    std::vector<int> devs;
    for (const Rank &r : sh->ranks) devs.push_back(r.device);
    std::vector<ncclComm_t> comms(devs.size());
    S_NCCL(ncclCommInitAll(comms.data(), (int)devs.size(), devs.data()));
    for (size_t i = 0; i < devs.size(); ++i) sh->ranks[i].comm = comms[i];
    sh->comms_ready = true;
    return ROAR_OK;
}

int ensure_buffers(roar_sharded *sh, Rank &r, uint64_t nq, uint32_t k) {
    S_HIP(hipSetDevice(r.device));
    const uint64_t qf = nq * sh->dim_pad, res = nq * k;
    if (r.cap_q < qf) {
        (void)hipFree(r.q_dev);
        r.q_dev = nullptr, r.cap_q = 0;
        S_HIP(hipMalloc((void **)&r.q_dev, (size_t)qf * sizeof(float)));
        r.cap_q = qf;
    }
    if (r.cap_res < res) {
        (void)hipFree(r.send_ids), (void)hipFree(r.recv_ids), (void)hipFree(r.out_ids);
        (void)hipFree(r.send_d), (void)hipFree(r.recv_d), (void)hipFree(r.out_d);
        r.send_ids = r.recv_ids = r.out_ids = nullptr, r.send_d = r.recv_d = r.out_d = nullptr, r.cap_res = 0;
        const size_t blk = (size_t)sh->max_local * res, all = blk * sh->ranks.size();
        S_HIP(hipMalloc((void **)&r.send_ids, blk * 4));
        S_HIP(hipMalloc((void **)&r.send_d, blk * 4));
        S_HIP(hipMalloc((void **)&r.recv_ids, all * 4));
        S_HIP(hipMalloc((void **)&r.recv_d, all * 4));
        S_HIP(hipMalloc((void **)&r.out_ids, (size_t)res * 4));
        S_HIP(hipMalloc((void **)&r.out_d, (size_t)res * 4));
        r.cap_res = res;
    }
    if (r.cap_cnt < nq) {  // (the counters do not depend on k)
        (void)hipFree(r.counts);
        r.counts = nullptr, r.cap_cnt = 0;
        S_HIP(hipMalloc((void **)&r.counts, (size_t)sh->max_local * 2 * nq * 4));
        r.cap_cnt = nq;
    }
    return ROAR_OK;
}

}  // namespace

extern "C" {

int roar_sharded_create(roar_sharded **out, uint32_t dim, int metric, const int *devices, uint32_t n_shards) {
    if (!out) return sfail(ROAR_ERR_ARG, "null output pointer");
    *out = nullptr;
    if (!devices || n_shards == 0 || n_shards > 32) return sfail(ROAR_ERR_ARG, "between 1 and 32 shards");
    DeviceRestore keep_device_;
    roar_sharded *sh = new (std::nothrow) roar_sharded();
    if (!sh) return sfail(ROAR_ERR_DEVICE, "out of host memory");
    try {
    sh->dim = dim;
    sh->dim_pad = (dim + 7) / 8 * 8;
    sh->metric = metric;
    for (uint32_t s = 0; s < n_shards; ++s) {
        roar_index *idx = nullptr;
        const int e = roar_create(&idx, dim, metric, devices[s]);  // (checks the device ordinal and the metric)
        if (e) {
            const std::string keep = roar_last_error();
            roar_sharded_destroy(sh);
            roar_set_last_error(keep.c_str());
            return e;
        }
        sh->shard.push_back(idx);
        size_t r = 0;
        while (r < sh->ranks.size() && sh->ranks[r].device != devices[s]) ++r;
        if (r == sh->ranks.size()) {
            sh->ranks.emplace_back();
            sh->ranks.back().device = devices[s];
        }
        sh->shard_rank.push_back((int)r);
        sh->shard_slot.push_back((uint32_t)sh->ranks[r].shards.size());
        sh->ranks[r].shards.push_back(s);
        sh->max_local = std::max<uint32_t>(sh->max_local, (uint32_t)sh->ranks[r].shards.size());
    }
    if (sh->ranks.size() * sh->max_local > 32) {  // (the merge kernel walks at most 32 lists per query)
        roar_sharded_destroy(sh);
        return sfail(ROAR_ERR_ARG, "devices x shards-per-device must not exceed 32");
    }
    sh->id_base.assign(n_shards, 0);
    sh->id_base_set.assign(n_shards, false);
    for (Rank &r : sh->ranks) {
        hipError_t e = hipSetDevice(r.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreate(&r.ev_start);
        if (e == hipSuccess) e = hipEventCreate(&r.ev_searched);
        if (e == hipSuccess) e = hipEventCreate(&r.ev_done);
        if (e != hipSuccess) {
            roar_sharded_destroy(sh);
            return sfail(ROAR_ERR_DEVICE, "stream / events on device %d: %s", r.device, hipGetErrorString(e));
        }
    }
    } catch (...) {  // (std::bad_alloc from the bookkeeping vectors: nothing may cross the C ABI)
        roar_sharded_destroy(sh);
        return sfail(ROAR_ERR_DEVICE, "out of host memory");
    }
    *out = sh;
    return ROAR_OK;
}

void roar_sharded_destroy(roar_sharded *sh) {
    if (!sh) return;
    DeviceRestore keep_device_;
    for (Rank &r : sh->ranks) {
        (void)hipSetDevice(r.device);
        if (r.stream) (void)hipStreamSynchronize(r.stream);
        if (r.comm) (void)ncclCommDestroy(r.comm);
        free_rank_buffers(r);
        if (r.stream) (void)hipStreamDestroy(r.stream);
        if (r.ev_start) (void)hipEventDestroy(r.ev_start);
        if (r.ev_searched) (void)hipEventDestroy(r.ev_searched);
        if (r.ev_done) (void)hipEventDestroy(r.ev_done);
    }
    for (roar_index *idx : sh->shard) roar_destroy(idx);
    if (sh->q_pin) (void)hipHostFree(sh->q_pin);
    if (sh->res_pin) (void)hipHostFree(sh->res_pin);
    delete sh;
}

roar_index *roar_sharded_shard(roar_sharded *sh, uint32_t s) {
    if (!sh || s >= sh->shard.size()) {
        sfail(ROAR_ERR_ARG, "no such shard");
        return nullptr;
    }
    return sh->shard[s];
}

int roar_sharded_set_id_base(roar_sharded *sh, uint32_t s, uint64_t first_global_id) {
    if (!sh || s >= sh->shard.size()) return sfail(ROAR_ERR_ARG, "no such shard");
    sh->id_base[s] = first_global_id;
    sh->id_base_set[s] = true;
    return ROAR_OK;
}

int roar_sharded_info(roar_sharded *sh, const char *key, int64_t *value) {
    if (!sh || !key || !value) return sfail(ROAR_ERR_ARG, "bad arguments");
    const std::string k(key);
    if (k == "shards") *value = (int64_t)sh->shard.size();
    else if (k == "ranks") *value = (int64_t)sh->ranks.size();
    else if (k == "last_search_ns") *value = sh->last_search_ns;
    else if (k == "last_exchange_ns") *value = sh->last_exchange_ns;
    else return sfail(ROAR_ERR_ARG, "Invalid parameter name.");
    return ROAR_OK;
}

int roar_search_batch_sharded(roar_sharded *sh, const float *queries, uint64_t nq, uint32_t qstride_floats, uint32_t k,
                              uint32_t L_pq, uint32_t *ids, float *dists, uint32_t *cmps, uint32_t *hops) {
    if (!sh || !ids || !dists || (nq && !queries)) return sfail(ROAR_ERR_ARG, "bad search arguments");
    if (nq == 0) return ROAR_OK;
    if (qstride_floats < sh->dim) return sfail(ROAR_ERR_ARG, "query stride smaller than the dimension");
    if (k == 0 || k > L_pq) return sfail(ROAR_ERR_ARG, "L_pq must greater or equal than k");
    if (nq * (uint64_t)k >= (1ull << 31)) return sfail(ROAR_ERR_ARG, "too many results in one batch");
    std::lock_guard<std::mutex> guard(sh->mu);
    DeviceRestore keep_device_;
    const uint32_t S = (uint32_t)sh->shard.size(), R = (uint32_t)sh->ranks.size();
    // global id of every shard's row 0: as set, else the running sum of the shard sizes
    {
        uint64_t run = 0;
        for (uint32_t s = 0; s < S; ++s) {
            int64_t n = 0;
            int e = roar_get_info(sh->shard[s], "n", &n);
            if (e) return e;
            if (n == 0) return sfail(ROAR_ERR_ARG, "shard %u has no base vectors", s);
            if (!sh->id_base_set[s]) sh->id_base[s] = run;
            run = sh->id_base[s] + (uint64_t)n;
            if (run > 0xFFFFFFFFull) return sfail(ROAR_ERR_ARG, "global ids exceed 32 bits");
        }
    }
    int e = ensure_comms(sh);
    if (e) return e;
    // the padded query block, once, in pinned memory (data_align of the query block, tests/test_search_roargraph.cpp:131)
    const uint64_t qf = nq * sh->dim_pad;
    if (sh->q_pin_cap < qf) {
        if (sh->q_pin) (void)hipHostFree(sh->q_pin);
        sh->q_pin = nullptr, sh->q_pin_cap = 0;
        S_HIP(hipHostMalloc((void **)&sh->q_pin, (size_t)qf * sizeof(float), hipHostMallocPortable));
        sh->q_pin_cap = qf;
    }
    const uint64_t rwords = nq * (2ull * k) + 2ull * S * nq;
    if (sh->res_pin_cap < rwords) {
        if (sh->res_pin) (void)hipHostFree(sh->res_pin);
        sh->res_pin = nullptr, sh->res_pin_cap = 0;
        S_HIP(hipHostMalloc((void **)&sh->res_pin, (size_t)rwords * 4, hipHostMallocPortable));
        sh->res_pin_cap = rwords;
    }
    for (uint64_t i = 0; i < nq; ++i) {
        float *dst = sh->q_pin + i * sh->dim_pad;
        memcpy(dst, queries + i * qstride_floats, sh->dim * sizeof(float));
        if (sh->dim_pad > sh->dim) memset(dst + sh->dim, 0, (sh->dim_pad - sh->dim) * sizeof(float));
    }
    // phase 1, one host thread per rank: query block to the device, the rank's shards searched back to back on its stream,
    // local ids rebased to global ids
    std::vector<int> rc(R, ROAR_OK);
    std::vector<std::string> msg(R);
    const size_t res = (size_t)nq * k;
    auto phase1 = [&](uint32_t ri) {
        Rank &r = sh->ranks[ri];
        auto body = [&]() -> int {
            int e1 = ensure_buffers(sh, r, nq, k);
            if (e1) return e1;
            S_HIP(hipEventRecord(r.ev_start, r.stream));
            S_HIP(hipMemcpyAsync(r.q_dev, sh->q_pin, (size_t)qf * sizeof(float), hipMemcpyHostToDevice, r.stream));
            if (r.shards.size() < sh->max_local) {  // unused block slots read as "no result": id 0xFFFFFFFF, distance NaN-free +inf
                S_HIP(hipMemsetAsync(r.send_ids, 0xFF, (size_t)sh->max_local * res * 4, r.stream));
                // (what goes over the wire is then the same on every run; the merge skips the slot by its id)
                S_HIP(hipMemsetD32Async((hipDeviceptr_t)r.send_d, (int)0x7F800000, (size_t)sh->max_local * res, r.stream));
            }
            for (size_t j = 0; j < r.shards.size(); ++j) {
                const uint32_t s = r.shards[j];
                uint32_t *cm = r.counts + (size_t)j * 2 * nq;
                e1 = roar_search_batch_device(sh->shard[s], r.q_dev, nq, sh->dim_pad, k, L_pq, r.send_ids + j * res, r.send_d + j * res,
                                              cm, cm + nq, nullptr, nullptr, r.stream);
                if (e1) return e1;
                e1 = roar_rebase_ids_device(r.device, r.send_ids + j * res, res, (uint32_t)sh->id_base[s], r.stream);
                if (e1) return e1;
            }
            S_HIP(hipEventRecord(r.ev_searched, r.stream));
            return ROAR_OK;
        };
        try {
            rc[ri] = body();
        } catch (...) {
            rc[ri] = sfail(ROAR_ERR_DEVICE, "out of host memory in a rank's thread");
        }
        if (rc[ri]) msg[ri] = roar_last_error();
    };
    {
        std::vector<std::thread> th;
        try {
            for (uint32_t ri = 1; ri < R; ++ri) th.emplace_back(phase1, ri);
        } catch (...) {
            for (auto &x : th) x.join();
            return sfail(ROAR_ERR_DEVICE, "cannot start the per-device host threads");
        }
        phase1(0);
        for (auto &x : th) x.join();
    }
    for (uint32_t ri = 0; ri < R; ++ri)
        if (rc[ri]) {  // nothing has entered the collective yet: drain what was queued and report
            for (Rank &r : sh->ranks) {
                (void)hipSetDevice(r.device);
                (void)hipStreamSynchronize(r.stream);
            }
            roar_set_last_error(msg[ri].c_str());
            return rc[ri];
        }
    auto drain = [&]() {  // after a failure: nothing of this call may still be running when the caller gets its buffers back
        for (Rank &r : sh->ranks) {
            (void)hipSetDevice(r.device);
            (void)hipStreamSynchronize(r.stream);
        }
    };
    // phase 2: the exchange -- every rank's [local shard][nq][k] block to every rank (ids, then distances) -- and the merge
    const size_t blk = (size_t)sh->max_local * res;
    auto phase2 = [&]() -> int {
    S_NCCL(ncclGroupStart());
    for (Rank &r : sh->ranks) {
        ncclResult_t a = ncclAllGather(r.send_ids, r.recv_ids, blk, ncclUint32, r.comm, r.stream);
        ncclResult_t b = a == ncclSuccess ? ncclAllGather(r.send_d, r.recv_d, blk, ncclFloat32, r.comm, r.stream) : a;
        if (b != ncclSuccess) {
            (void)ncclGroupEnd();
            return sfail(ROAR_ERR_DEVICE, "ncclAllGather on device %d: %s", r.device, ncclGetErrorString(b));
        }
    }
    S_NCCL(ncclGroupEnd());
    for (Rank &r : sh->ranks) {
        e = roar_merge_topk_device(r.device, R * sh->max_local, nq, k, r.recv_ids, r.recv_d, r.out_ids, r.out_d, r.stream);
        if (e) return e;
        S_HIP(hipSetDevice(r.device));
        S_HIP(hipEventRecord(r.ev_done, r.stream));
    }
    // results: rank 0's merged lists; per-shard counters from every rank
    {
        Rank &r0 = sh->ranks[0];
        S_HIP(hipSetDevice(r0.device));
        S_HIP(hipMemcpyAsync(sh->res_pin, r0.out_ids, res * 4, hipMemcpyDeviceToHost, r0.stream));
        S_HIP(hipMemcpyAsync(sh->res_pin + res, r0.out_d, res * 4, hipMemcpyDeviceToHost, r0.stream));
    }
    for (Rank &r : sh->ranks) {
        S_HIP(hipSetDevice(r.device));
        for (size_t j = 0; j < r.shards.size(); ++j)
            S_HIP(hipMemcpyAsync(sh->res_pin + 2 * res + (size_t)r.shards[j] * 2 * nq, r.counts + j * 2 * nq, (size_t)2 * nq * 4,
                                 hipMemcpyDeviceToHost, r.stream));
    }
    int64_t search_ns = 0, exch_ns = 0;
    for (Rank &r : sh->ranks) {
        S_HIP(hipSetDevice(r.device));
        S_HIP(hipStreamSynchronize(r.stream));
        float a = 0.f, b = 0.f;
        S_HIP(hipEventElapsedTime(&a, r.ev_start, r.ev_searched));
        S_HIP(hipEventElapsedTime(&b, r.ev_searched, r.ev_done));
        search_ns = std::max<int64_t>(search_ns, (int64_t)((double)a * 1e6));
        exch_ns = std::max<int64_t>(exch_ns, (int64_t)((double)b * 1e6));
    }
    sh->last_search_ns = search_ns;
    sh->last_exchange_ns = exch_ns;
    return ROAR_OK;
    };
    if ((e = phase2())) {
        const std::string keep = roar_last_error();
        drain();
        roar_set_last_error(keep.c_str());
        return e;
    }
    memcpy(ids, sh->res_pin, res * 4);
    memcpy(dists, sh->res_pin + res, res * 4);
    if (cmps || hops)
        for (uint64_t i = 0; i < nq; ++i) {  // distance evaluations / expansions of a query, summed over the shards
            uint64_t c = 0, h = 0;
            for (uint32_t s = 0; s < S; ++s) {
                c += sh->res_pin[2 * res + (size_t)s * 2 * nq + i];
                h += sh->res_pin[2 * res + (size_t)s * 2 * nq + nq + i];
            }
            if (cmps) cmps[i] = (uint32_t)std::min<uint64_t>(c, 0xFFFFFFFFull);
            if (hops) hops[i] = (uint32_t)std::min<uint64_t>(h, 0xFFFFFFFFull);
        }
    // a query is short only if all shards together hold fewer than k points (the reference throws, :2408-2412)
    for (uint64_t i = 0; i < nq; ++i)
        if (ids[i * k + k - 1] == 0xFFFFFFFFu) {
            uint32_t have = 0;
            while (have < k && ids[i * k + have] != 0xFFFFFFFFu) have++;
            return sfail(ROAR_ERR_SHORT, "not enough results: %u, expected: %u", have, k);
        }
    return ROAR_OK;
}

}  // extern "C"
