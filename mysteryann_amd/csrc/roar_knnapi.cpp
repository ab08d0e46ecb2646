// roar_knnapi.cpp -- the exact kNN generator behind the C ABI (roar_exact_knn(_device)): bf16 MFMA screen + f32 re-rank, f32 MFMA form
// (round 5: roar_capi.cpp was one 3 203-line translation unit; it is now cut by concern.  Shared declarations: roar_internal.h.)
#include "roar_internal.h"

using namespace roar_impl;

namespace roar_impl {

// The f32 form (roar_knn.hip): every score by v_mfma_f32_32x32x2_f32.
int exact_knn_f32(roar_index *idx, const float *queries_dev, uint64_t nq, uint32_t qstride_floats, uint32_t k, uint32_t *ids_dev, float *dists_dev,
                  void *hip_stream) {
    uint32_t cap = 2048;
    while (cap < 64u * k && cap < 8192) cap <<= 1;
    const uint64_t chunk = std::min<uint64_t>(nq, 65536);
    void *cand = nullptr, *cursor = nullptr, *tau = nullptr, *bnorm = nullptr, *ovf = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(cand); (void)hipFree(cursor); (void)hipFree(tau); (void)hipFree(bnorm); (void)hipFree(ovf);
    };
    {
        hipError_t he = roar_dev_malloc(&cursor, chunk * 4);
        if (he == hipSuccess) he = roar_dev_malloc(&tau, chunk * 4);
        if (he == hipSuccess) he = roar_dev_malloc(&ovf, 4);
        if (he == hipSuccess && !idx->ip) he = roar_dev_malloc(&bnorm, idx->n * 4);
        if (he != hipSuccess) {
            cleanup();
            return fail(ROAR_ERR_DEVICE, "out of device memory for the exact kNN workspaces: %s", hipGetErrorString(he));
        }
    }
    for (;;) {
        if (roar_dev_malloc(&cand, chunk * cap * 8) != hipSuccess) {
            cleanup();
            return fail(ROAR_ERR_DEVICE, "out of device memory for the kNN candidate lists");
        }
        bool retry = false;
        for (uint64_t q0 = 0; q0 < nq && !retry; q0 += chunk) {
            const uint32_t cnt = (uint32_t)std::min<uint64_t>(chunk, nq - q0);
            uint32_t overflowed = 0;
            int rc = roar_knn_run(queries_dev + q0 * qstride_floats, qstride_floats, cnt, idx->base, idx->ld, idx->n,
                                  idx->dim_pad, idx->ip ? 1 : 0, k, cap, cand, cursor, tau, bnorm, ovf, ids_dev + q0 * k,
                                  dists_dev ? dists_dev + q0 * k : nullptr, &overflowed, hip_stream);
            if (rc != 0) {
                cleanup();
                return fail(ROAR_ERR_DEVICE, "exact kNN kernels failed: %s", hipGetErrorString((hipError_t)rc));
            }
            if (overflowed) retry = true;
        }
        if (!retry) break;
        (void)hipFree(cand);
        cand = nullptr;
        if (cap >= 8192) {
            cleanup();
            return fail(ROAR_ERR_DEVICE, "exact kNN candidate lists overflowed at the largest capacity");
        }
        cap <<= 1;
    }
    cleanup();
    return ROAR_OK;
}

// The bf16-screened form (roar_knn_bf16.hip): scores by v_mfma_f32_32x32x16_bf16 with a rigorous bound, the rows that can be among the
// top k re-ranked on the f32 rows in the reference's summation order.  *done = false: not applicable here (rows too long, a norm that is
// not finite, no memory for the bf16 copy) -- the caller runs the f32 form.
int exact_knn_bf16(roar_index *idx, const float *queries_dev, uint64_t nq, uint32_t qstride_floats, uint32_t k, uint32_t *ids_dev, float *dists_dev,
                   void *hip_stream, bool *done) {
    *done = false;
    if (idx->dim_pad > 512) return ROAR_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    const uint32_t KQ = roar_knn_bf16_row_quads(idx->dim_pad);
    const uint64_t chunk = std::min<uint64_t>(nq, 65536), chunk_pad = (chunk + 511) / 512 * 512;
    void *bb = nullptr, *bnorm = nullptr, *xmax = nullptr, *qb = nullptr, *cand = nullptr, *cursor = nullptr, *tau = nullptr, *slack = nullptr,
         *thr = nullptr, *ovf = nullptr, *rr = nullptr;
    auto cleanup = [&]() {
        for (void *p_ : {bb, bnorm, xmax, qb, cand, cursor, tau, slack, thr, ovf, rr}) (void)hipFree(p_);
    };
    hipError_t he = roar_dev_malloc(&bb, (size_t)idx->n * KQ * 16);
    if (he == hipSuccess) he = roar_dev_malloc(&bnorm, (size_t)idx->n * 4);
    if (he == hipSuccess) he = roar_dev_malloc(&xmax, 4);
    if (he == hipSuccess) he = roar_dev_malloc(&qb, (size_t)chunk_pad * KQ * 16);
    for (void **w : {&cursor, &tau, &slack, &thr})
        if (he == hipSuccess) he = roar_dev_malloc(w, chunk * 4);
    if (he == hipSuccess) he = roar_dev_malloc(&ovf, 4);
    if (he == hipSuccess) he = roar_dev_malloc(&rr, 4);
    if (he != hipSuccess) {  // (the f32 form needs far less memory: let it try)
        (void)hipGetLastError();
        cleanup();
        return ROAR_OK;
    }
    uint32_t xbits = 0;
    he = hipMemsetAsync(xmax, 0, 4, st);
    if (he == hipSuccess) he = hipMemsetAsync(rr, 0, 4, st);
    if (he == hipSuccess)
        he = (hipError_t)roar_knn_bf16_convert(idx->base, idx->ld, idx->dim_pad, idx->n, idx->n, bb, (float *)bnorm, (uint32_t *)xmax, idx->ip ? 1 : 2, st);
    if (he == hipSuccess) he = hipMemcpyAsync(&xbits, xmax, 4, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    if (he != hipSuccess) {
        cleanup();
        return fail(ROAR_ERR_DEVICE, "exact kNN: bf16 copy of the base rows: %s", hipGetErrorString(he));
    }
    float xmax2;
    memcpy(&xmax2, &xbits, 4);
    if (!(xmax2 <= 1.0e36f)) {  // a row that is not finite (or whose norm overflows): no bound -- the f32 form scores everything
        cleanup();
        return ROAR_OK;
    }
    uint32_t cap = 2048;
    while (cap < 64u * k && cap < 8192) cap <<= 1;
    for (;;) {
        if (roar_dev_malloc(&cand, chunk * cap * 8) != hipSuccess) {
            cleanup();
            return fail(ROAR_ERR_DEVICE, "out of device memory for the kNN candidate lists");
        }
        bool retry = false;
        for (uint64_t q0 = 0; q0 < nq && !retry; q0 += chunk) {
            const uint32_t cnt = (uint32_t)std::min<uint64_t>(chunk, nq - q0);
            uint32_t overflowed = 0;
            int rc = roar_knn_run_bf16(queries_dev + q0 * qstride_floats, qstride_floats, cnt, idx->base, idx->ld, idx->n, idx->dim_pad,
                                       idx->ip ? 1 : 0, k, cap, bb, (const float *)bnorm, qb, cand, cursor, tau, slack, thr, ovf,
                                       (uint32_t *)rr, ids_dev + q0 * k, dists_dev ? dists_dev + q0 * k : nullptr, &overflowed, hip_stream);
            if (rc != 0) {
                cleanup();
                return fail(ROAR_ERR_DEVICE, "exact kNN (bf16 screen) kernels failed: %s", hipGetErrorString((hipError_t)rc));
            }
            if (overflowed) retry = true;
        }
        if (!retry) break;
        (void)hipFree(cand);
        cand = nullptr;
        if (cap >= 16384) {  // (the selection and the re-rank keep a list in LDS: 8 bytes per entry)
            cleanup();
            return ROAR_OK;  // the bound admits more than fits: the f32 form
        }
        cap <<= 1;
        (void)hipMemsetAsync(rr, 0, 4, st);
    }
    uint32_t rows = 0;
    (void)hipMemcpy(&rows, rr, 4, hipMemcpyDeviceToHost);
    idx->last_knn_rerank_rows = rows;
    cleanup();
    *done = true;
    return ROAR_OK;
}
}  // namespace roar_impl

extern "C" {

int roar_exact_knn_device(roar_index *idx, const float *queries_dev, uint64_t nq, uint32_t qstride_floats, uint32_t k,
                          uint32_t *ids_dev, float *dists_dev, void *hip_stream) {
    DeviceRestore keep_device_;
    if (!idx || !queries_dev || !ids_dev) return fail(ROAR_ERR_ARG, "bad kNN arguments");
    if (!idx->base || idx->n == 0) return fail(ROAR_ERR_ARG, "exact kNN before base vectors were loaded");
    if (k == 0 || k > 128 || k > idx->n) return fail(ROAR_ERR_ARG, "k must be in 1..min(128, n)");
    if (qstride_floats < idx->dim_pad || (qstride_floats & 3))
        return fail(ROAR_ERR_ARG, "device queries need a stride >= %u floats that is a multiple of 4", idx->dim_pad);
    if (nq == 0) return ROAR_OK;
    int e = use_device(idx);
    if (e) return e;
    idx->last_knn_bf16 = 0;
    idx->last_knn_rerank_rows = 0;
    if (idx->knn_mode != 1) {
        bool done = false;
        if ((e = exact_knn_bf16(idx, queries_dev, nq, qstride_floats, k, ids_dev, dists_dev, hip_stream, &done))) return e;
        if (done) {
            idx->last_knn_bf16 = 1;
            return ROAR_OK;
        }
    }
    return exact_knn_f32(idx, queries_dev, nq, qstride_floats, k, ids_dev, dists_dev, hip_stream);
}

int roar_exact_knn(roar_index *idx, const float *queries, uint64_t nq, uint32_t qstride_floats, uint32_t k, uint32_t *ids,
                   float *dists) {
    DeviceRestore keep_device_;
    if (!idx || !queries || !ids) return fail(ROAR_ERR_ARG, "bad kNN arguments");
    if (qstride_floats < idx->dim) return fail(ROAR_ERR_ARG, "query stride smaller than the dimension");
    if (nq == 0) return ROAR_OK;
    int e = use_device(idx);
    if (e) return e;
    const uint64_t chunk = std::min<uint64_t>(nq, 1u << 18);
    float *q_dev = nullptr, *d_dev = nullptr;
    uint32_t *i_dev = nullptr;
    std::vector<float> buf;
    try {
        buf.resize((size_t)chunk * idx->dim_pad);
    } catch (...) {
        return fail(ROAR_ERR_DEVICE, "out of host memory");
    }
    auto cleanup = [&]() { (void)hipFree(q_dev); (void)hipFree(i_dev); (void)hipFree(d_dev); };
    {
        hipError_t he = roar_dev_malloc((void **)&q_dev, buf.size() * sizeof(float));
        if (he == hipSuccess) he = roar_dev_malloc((void **)&i_dev, (size_t)chunk * k * 4);
        if (he == hipSuccess) he = roar_dev_malloc((void **)&d_dev, (size_t)chunk * k * 4);
        if (he != hipSuccess) {
            cleanup();
            return fail(ROAR_ERR_DEVICE, "out of device memory for the exact kNN staging blocks: %s", hipGetErrorString(he));
        }
    }
    for (uint64_t q0 = 0; q0 < nq; q0 += chunk) {
        const uint64_t cnt = std::min<uint64_t>(chunk, nq - q0);
        std::fill(buf.begin(), buf.begin() + cnt * idx->dim_pad, 0.0f);
        for (uint64_t i = 0; i < cnt; ++i)
            memcpy(buf.data() + i * idx->dim_pad, queries + (q0 + i) * qstride_floats, idx->dim * sizeof(float));
        hipError_t he = hipMemcpy(q_dev, buf.data(), (size_t)cnt * idx->dim_pad * sizeof(float), hipMemcpyHostToDevice);
        if (he == hipSuccess) {
            e = roar_exact_knn_device(idx, q_dev, cnt, idx->dim_pad, k, i_dev, d_dev, nullptr);
            if (e) { cleanup(); return e; }
            he = hipMemcpy(ids + q0 * k, i_dev, (size_t)cnt * k * 4, hipMemcpyDeviceToHost);
        }
        if (he == hipSuccess && dists) he = hipMemcpy(dists + q0 * k, d_dev, (size_t)cnt * k * 4, hipMemcpyDeviceToHost);
        if (he != hipSuccess) { cleanup(); return fail(ROAR_ERR_DEVICE, "exact kNN copy failed: %s", hipGetErrorString(he)); }
    }
    cleanup();
    return ROAR_OK;
}
}  // extern "C"
