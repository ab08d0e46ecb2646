// roar_knn.hip -- exact brute-force kNN (ground truth / learn->base kNN lists), gfx950.
//
// the reference consumes two exact-kNN files it cannot produce itself -- the
// query->base ground truth (include/efanna2e/util.h:84-155) and the learn->base top-M_sq lists
// (src/index_bipartite.cpp:2622-2639); their producer lived in the empty thirdparty/DiskANN
// submodule.  A dense nq x n x D contraction followed by a top-k is the one place on this path where
// the matrix cores are the right roofline, so the scores go through v_mfma_f32_32x32x2_f32
// (exact f32, k-ordered fma chain).
//
// Structure: a few passes over nested strided samples of the base set with a per-query
// admission threshold tau (distance of the k-th best seen in the previous, smaller pass):
//   pass p: scores of all queries x sample_p rows; every (query, row) with dist <= tau[q] is
//           appended to the query's candidate list (atomic cursor);  then a selection kernel sorts
//           the list, keeps the best k and sets tau[q] to the k-th distance.
// sample_0 is small enough to be admitted whole (tau = +inf), each next sample is `ratio` times denser
// and contains the previous one (strides are powers of `ratio`), the last one is every row.
// Expected admissions per query stay ~k * ratio; a list that overflows its capacity is reported and
// the caller retries with a larger capacity.
//
// GEMM tile: 128 queries x 128 base rows per workgroup (4 waves, each 64 x 64 = 2 x 2 MFMA tiles,
// 64 accumulator VGPRs), K walked in chunks of 32 floats staged through LDS (row stride 33 words:
// conflict-free ds_read_b32 for the one-float-per-lane MFMA operands).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

namespace {

typedef unsigned long long u64;
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TQ = 128, TB = 128, KC = 32, LDS_LD = KC + 1;

__device__ __forceinline__ uint32_t f2ord_k(float f) {
    if (f == 0.0f) f = 0.0f;
    uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float ord2f_k(uint32_t o) { return __uint_as_float((o >> 31) ? (o ^ 0x80000000u) : ~o); }

struct KnnPassParams {
    const float *q;      // [nq][ldq]
    const float *b;      // [n][ldb]
    const float *bnorm;  // [n] squared norms (L2 only)
    uint64_t ldq, ldb;
    uint32_t nq, dim;    // dim % 8 == 0 (zero padded)
    uint64_t n;
    uint64_t first, stride, count;  // rows first + i*stride, i < count
    const float *tau;    // [nq] admission threshold (reference-signed distance without the |q|^2 term)
    u64 *cand;           // [nq][cap] keys (ord(dist) << 32 | id)
    uint32_t *cursor;    // [nq]
    uint32_t cap;
    int ip;
};

// scores for a 128 x 128 tile, then threshold-append
__global__ void __launch_bounds__(256) knn_pass_kernel(const KnnPassParams p) {
    __shared__ float sq[2][TQ * LDS_LD];
    __shared__ float sb[2][TB * LDS_LD];
    __shared__ float stau[TQ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;  // wave tile origin: queries wr*64, rows wc*64
    // 1-D grid, query tile fastest: neighbouring workgroups share one base tile (L2), the query block
    // (a few MB) stays cache resident, so every base row leaves HBM about once per pass
    const uint32_t nqt = (p.nq + TQ - 1) / TQ;
    const uint32_t q0 = (blockIdx.x % nqt) * TQ;
    const u64 r0 = (u64)(blockIdx.x / nqt) * TB;  // index into the strided sample
    if (tid < TQ) stau[tid] = (q0 + tid < p.nq) ? p.tau[q0 + tid] : -__builtin_inff();

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    // staging: 256 threads move 128 rows x 32 floats of Q and of B per chunk: thread -> (row = tid / 2,
    // 16 floats starting at (tid & 1) * 16), four float4 each
    const int srow = tid >> 1, scol = (tid & 1) * 16;
    const uint32_t gq = q0 + srow;
    const u64 gs = r0 + srow;
    const bool qok = gq < p.nq, bok = gs < p.count;
    const float *qsrc = p.q + (u64)(qok ? gq : 0) * p.ldq + scol;
    const float *bsrc = p.b + (p.first + (bok ? gs : 0) * p.stride) * p.ldb + scol;
    const uint32_t nchunks = p.dim / KC + ((p.dim % KC) ? 1 : 0);

    float4 rq[4], rb[4];
    auto load_regs = [&](uint32_t kc) {
        const uint32_t k0 = kc * KC;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const uint32_t k = k0 + scol + 4 * v;
            rq[v] = make_float4(0.f, 0.f, 0.f, 0.f);
            rb[v] = rq[v];
            if (k < p.dim) {  // dim % 4 == 0
                if (qok) rq[v] = *reinterpret_cast<const float4 *>(qsrc + k0 + 4 * v);
                if (bok) rb[v] = *reinterpret_cast<const float4 *>(bsrc + k0 + 4 * v);
            }
        }
    };
    auto write_lds = [&](int buf) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float *dq = &sq[buf][srow * LDS_LD + scol + 4 * v];
            float *db = &sb[buf][srow * LDS_LD + scol + 4 * v];
            dq[0] = rq[v].x; dq[1] = rq[v].y; dq[2] = rq[v].z; dq[3] = rq[v].w;
            db[0] = rb[v].x; db[1] = rb[v].y; db[2] = rb[v].z; db[3] = rb[v].w;
        }
    };

    load_regs(0);
    write_lds(0);
    __syncthreads();
    for (uint32_t kc = 0; kc < nchunks; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nchunks) load_regs(kc + 1);  // in flight under the MFMAs of this chunk
        const float *aq = &sq[buf][(wr * 64 + (lane & 31)) * LDS_LD + (lane >> 5)];
        const float *ab = &sb[buf][(wc * 64 + (lane & 31)) * LDS_LD + (lane >> 5)];
        auto mfma_step = [&](int s) {
            const float a0 = aq[2 * s], a1 = aq[32 * LDS_LD + 2 * s];
            const float b0 = ab[2 * s], b1 = ab[32 * LDS_LD + 2 * s];
            // D = A(32 x 2, queries) * B(2 x 32, base rows) + C
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        };
        const uint32_t kleft = p.dim - kc * KC;
        if (kleft >= (uint32_t)KC) {
#pragma unroll
            for (int s = 0; s < KC / 2; ++s) mfma_step(s);
        } else {  // last chunk of a row length that is not a multiple of 32 (D = 200: 8 floats): skip the zero columns
            for (int s = 0; s < (int)(kleft / 2); ++s) mfma_step(s);
        }
        if (kc + 1 < nchunks) write_lds(buf ^ 1);
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32: col (base row) = lane & 31, row (query) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const u64 sidx = r0 + wc * 64 + c * 32 + (lane & 31);
        if (sidx >= p.count) continue;
        const u64 row = p.first + sidx * p.stride;
        const float bn = p.ip ? 0.f : p.bnorm[row];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const uint32_t qg = q0 + ql;
                if (qg >= p.nq) continue;
                const float dot = acc[a][c][r];
                const float dist = p.ip ? -dot : __builtin_fmaf(-2.f, dot, bn);  // |q|^2 added by the selection kernel
                if (dist <= stau[ql]) {
                    const uint32_t pos = atomicAdd(&p.cursor[qg], 1u);
                    if (pos < p.cap) p.cand[(u64)qg * p.cap + pos] = ((u64)f2ord_k(dist) << 32) | (uint32_t)row;
                }
            }
        }
    }
}

// per query: sort the candidate list by (dist, id), keep the best k at the front, tau = k-th distance.
// One workgroup (256 threads) per query; rank by counting (lists are a few hundred to a few thousand long).
__global__ void __launch_bounds__(256) knn_select_kernel(u64 *cand, uint32_t *cursor, uint32_t cap, uint32_t nq, uint32_t k,
                                                        float *tau, uint32_t *overflow) {
    extern __shared__ u64 skeys[];
    const uint32_t q = blockIdx.x;
    if (q >= nq) return;
    uint32_t m = cursor[q];
    if (m > cap) {
        if (threadIdx.x == 0) atomicAdd(overflow, 1u);
        m = cap;
    }
    u64 *list = cand + (u64)q * cap;
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) skeys[i] = list[i];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
        const u64 key = skeys[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < m; ++j) {
            const u64 o = skeys[j];
            rank += (o < key || (o == key && j < i)) ? 1u : 0u;
        }
        if (rank < k) list[rank] = key;
        if (rank == k - 1) tau[q] = ord2f_k((uint32_t)(key >> 32));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        cursor[q] = m < k ? m : k;
        if (m < k) tau[q] = __builtin_inff();
    }
}

__global__ void row_norms_kernel(const float *__restrict__ b, u64 ld, uint32_t dim, u64 n, float *out) {
    const int lane = threadIdx.x & 63;
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u64 nw = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 r = wave; r < n; r += nw) {
        float s = 0.f;
        for (uint32_t c = lane; c < dim; c += 64) {
            const float v = b[r * ld + c];
            s = __builtin_fmaf(v, v, s);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) out[r] = s;
    }
}

__global__ void fill_f32_kernel(float *p, uint32_t n, float v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// final output: ids + reference-signed distances (L2 gets its |q|^2 back)
__global__ void knn_emit_kernel(const u64 *cand, const uint32_t *cursor, uint32_t cap, uint32_t nq, uint32_t k,
                                const float *__restrict__ q, u64 ldq, uint32_t dim, int ip, uint32_t *ids, float *dists) {
    const uint32_t qi = blockIdx.x;
    if (qi >= nq) return;
    float qn = 0.f;
    if (!ip) {
        float s = 0.f;
        for (uint32_t c = threadIdx.x; c < dim; c += 64) {
            const float v = q[(u64)qi * ldq + c];
            s = __builtin_fmaf(v, v, s);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        qn = s;
    }
    const uint32_t m = cursor[qi];
    for (uint32_t i = threadIdx.x; i < k; i += 64) {
        uint32_t id = 0xFFFFFFFFu;
        float d = __builtin_inff();
        if (i < m) {
            const u64 key = cand[(u64)qi * cap + i];
            id = (uint32_t)key;
            d = ord2f_k((uint32_t)(key >> 32)) + qn;
        }
        ids[(u64)qi * k + i] = id;
        if (dists) dists[(u64)qi * k + i] = d;
    }
}

}  // namespace

extern "C" {

// Workspace sizes: cand nq*cap*8 B, cursor nq*4 B, tau nq*4 B, bnorm n*4 B (L2), overflow 4 B.
// Returns a hipError_t (0 ok); *overflowed (host) > 0 means some list exceeded `cap`: retry larger.
int roar_knn_run(const float *q, uint64_t ldq, uint32_t nq, const float *b, uint64_t ldb, uint64_t n, uint32_t dim,
                 int ip, uint32_t k, uint32_t cap, void *cand, void *cursor, void *tau, void *bnorm, void *overflow_dev,
                 uint32_t *ids, float *dists, uint32_t *overflowed, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    if (!ip) hipLaunchKernelGGL(row_norms_kernel, dim3(2048), dim3(256), 0, st, b, (u64)ldb, dim, (u64)n, (float *)bnorm);
    hipLaunchKernelGGL(fill_f32_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, (float *)tau, nq, __builtin_inff());
    if ((e = hipMemsetAsync(overflow_dev, 0, 4, st)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(cursor, 0, (size_t)nq * 4, st)) != hipSuccess) return (int)e;
    // nested strided samples S_0 c S_1 c ... c S_{P-1} = all rows, S_p = multiples of ratio^(P-1-p): the
    // k-th best of S_{p-1} bounds the k-th best of S_p from above, so admitting dist <= tau loses
    // nothing; expected admissions per query ~ k * ratio
    uint64_t ratio = cap / (4ull * k);
    if (ratio < 4) ratio = 4;
    if (ratio > 64) ratio = 64;
    int P = 1;
    uint64_t stride0 = 1;
    while ((n + stride0 - 1) / stride0 > cap / 2) {
        stride0 *= ratio;
        ++P;
    }
    if ((e = hipFuncSetAttribute((const void *)knn_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)((size_t)cap * 8))) != hipSuccess)
        return (int)e;
    uint64_t stride = stride0;
    for (int pass = 0; pass < P; ++pass, stride /= ratio) {
        KnnPassParams p;
        p.q = q; p.b = b; p.bnorm = (const float *)bnorm; p.ldq = ldq; p.ldb = ldb; p.nq = nq; p.dim = dim; p.n = n;
        p.first = 0; p.stride = stride; p.count = (n + stride - 1) / stride;
        p.tau = (const float *)tau; p.cand = (u64 *)cand; p.cursor = (uint32_t *)cursor; p.cap = cap; p.ip = ip;
        if ((e = hipMemsetAsync(cursor, 0, (size_t)nq * 4, st)) != hipSuccess) return (int)e;
        // a launch may not exceed 2^32 - 1 threads in x: split the sample of this pass into runs of base tiles
        const uint64_t nqt = (nq + TQ - 1) / TQ;
        const uint64_t total_tiles = ((n + stride - 1) / stride + TB - 1) / TB;
        const uint64_t max_tiles = std::max<uint64_t>(1, (0xFFFFFFFFull / 256) / nqt);
        for (uint64_t t0 = 0; t0 < total_tiles; t0 += max_tiles) {
            const uint64_t s0 = t0 * TB;  // first sample index of this run
            const uint64_t tiles = std::min(max_tiles, total_tiles - t0);
            p.first = s0 * stride;
            p.count = std::min<uint64_t>((n + stride - 1) / stride - s0, tiles * TB);
            hipLaunchKernelGGL(knn_pass_kernel, dim3((uint32_t)(tiles * nqt)), dim3(256), 0, st, p);
            if ((e = hipGetLastError()) != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(knn_select_kernel, dim3(nq), dim3(256), (size_t)cap * 8, st, (u64 *)cand, (uint32_t *)cursor, cap,
                           nq, k, (float *)tau, (uint32_t *)overflow_dev);
        if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(knn_emit_kernel, dim3(nq), dim3(64), 0, st, (const u64 *)cand, (const uint32_t *)cursor, cap, nq, k,
                       q, (u64)ldq, dim, ip, ids, dists);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    if (overflowed) {
        if ((e = hipMemcpyAsync(overflowed, overflow_dev, 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return (int)e;
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return (int)e;
    }
    return 0;
}
}
