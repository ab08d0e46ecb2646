// roar_knn_bf16.hip -- exact brute-force kNN, second form: a bf16 MFMA screen with a rigorous bound in front of an exact f32
// re-rank (gfx950).
//
// the reference's builder consumes the EXACT top-M_sq base neighbours of every training query
// (LoadLearnBaseKNN, src/index_bipartite.cpp:2622-2639; run_roargraph_test.sh:5-9: as many training queries as base points), a
// dense nq x n x D contraction.  In f32 (roar_knn.hip, v_mfma_f32_32x32x2_f32) 10 M x 10 M x 200 is 4e16 flop against a 157 TFLOP/s
// peak: 255 s at best.  The matrix cores run bf16 sixteen times faster, and "which rows are the top k" needs the exact distance only
// for the few rows near the k-th -- the same idea as the search kernel's 8-bit screen:
//
//   1. both sides rounded to bf16 (round to nearest even: |x^ - x| <= 2^-8 |x| component by component), copies with rows of an ODD
//      number of 16-byte quads (conflict-free ds_read_b128 of MFMA operands);
//   2. passes over nested strided samples as in roar_knn.hip, scores by v_mfma_f32_32x32x16_bf16 (products of bf16 values are exact in
//      f32, the sum is f32).  For every (query q, row r):  |d^ - d| <= eps(q, r) = C |q| |x_r|  (inner product) with
//      C = 2.02 * 2^-8 + (K + D / 16 + 8) * 1.25 * 2^-24 -- Cauchy-Schwarz on the two roundings, the f32 accumulation of the MFMA and
//      of the exact pass's own sum; for L2 (d = |x|^2 - 2 q.x + |q|^2 against the reference's sum of squared differences) twice that
//      plus kappa (|x_r|^2 + |q|^2).  The bound is per ROW (its own norm, kept in f32): a few long rows do not loosen it for the others;
//   3. every row gets an interval [lo, hi] = d^ -+ eps around its exact distance.  If U is the k-th smallest hi over any set of rows, k
//      rows have exact distances <= U, so the exact k-th distance t <= U, and every row of the exact top k has lo <= t <= U: a pass
//      admits the rows with lo <= U_prev (U of the previous, sparser sample: its k rows are in this one too, so U <= U_prev and nothing
//      with hi <= U is missed), keyed by hi; the selection finds U among them; after the last pass the rows with lo <= U go on;
//   4. those (~1.3-2 k rows per query) are scored on the f32 rows by the search kernels' own distance routine -- the reference's
//      summation order (roar_wave.h row_distance = distance.h:39-89,179-223), bit-identical to DistanceInnerProduct / DistanceL2 -- and
//      sorted by (distance, id).  The result is the exact top k under the reference's own arithmetic, whatever the bf16 pass did.
//
// GEMM tile: a workgroup of 8 waves owns 256 queries (wave w: 32 of them, its A fragments -- 13 x 4 VGPRs at D = 200 -- stay in
// REGISTERS for the whole slice) and walks a slice of the sample in tiles of 64 base rows, double-buffered in LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers); per tile and wave 2 x K/16 MFMAs of 32x32x16, then 32 compares against the
// per-query admission thresholds (16 registers) and a rare append.  1-D grid, query tile fastest: the workgroups that run together
// stream the same slice (L2 / Infinity Cache), every base row leaves HBM about once per pass.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "roar_wave.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int QT = 256;  // queries per workgroup: 8 waves x 32
constexpr int BT = 64;   // base rows per LDS tile

__device__ __forceinline__ uint32_t f2ord_b(float f) {
    if (f == 0.0f) f = 0.0f;
    uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float ord2f_b(uint32_t o) { return __uint_as_float((o >> 31) ? (o ^ 0x80000000u) : ~o); }

// f32 rows -> bf16 rows of KQ quads (8 values each; everything past `dim` is zero), rows [0, n) from src, rows [n, n_pad) zero
__global__ void knn_to_bf16_kernel(const float *__restrict__ src, u64 ld, uint32_t dim, u64 n, u64 n_pad, uint32_t KQ, uint4 *__restrict__ dst) {
    const u64 total = n_pad * KQ;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (u64)gridDim.x * blockDim.x) {
        const u64 row = i / KQ;
        const uint32_t c0 = (uint32_t)(i % KQ) * 8u;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (row < n && c0 < dim) {  // dim % 8 == 0
            const float4 a = *reinterpret_cast<const float4 *>(src + row * ld + c0), b = *reinterpret_cast<const float4 *>(src + row * ld + c0 + 4);
            v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)v[j];  // v_cvt_pk_bf16_f32: round to nearest even
        dst[i] = __builtin_bit_cast(uint4, o);
    }
}

// squared norms of f32 rows (one wave per row) and the largest of them (bits of a non-negative float order like the float)
__global__ void knn_sqnorm_kernel(const float *__restrict__ b, u64 ld, uint32_t dim, u64 n, float *out, uint32_t *max_bits) {
    const int lane = threadIdx.x & 63;
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((u64)gridDim.x * blockDim.x) >> 6;
    float mx = 0.f;
    for (u64 r = wave; r < n; r += nw) {
        float s = 0.f;
        for (uint32_t c = lane; c < dim; c += 64) {
            const float v = b[r * ld + c];
            s = __builtin_fmaf(v, v, s);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (out && lane == 0) out[r] = s;
        mx = __builtin_fmaxf(mx, s);
    }
    if (max_bits && lane == 0) {
        if (!(mx == mx) || mx > 3.0e38f) mx = __builtin_inff();  // a NaN / overflowing norm: no bound can be claimed
        atomicMax(max_bits, __float_as_uint(mx));
    }
}

// per query: the query's side of the bound (header, step 2): ea = C |q| (inner product) or 2 C |q| (L2), kq = kappa |q|^2 (L2; 0 otherwise)
__global__ void knn_slack_kernel(const float *__restrict__ q, u64 ldq, uint32_t dim, uint32_t nq, int ip, float C, float kappa, float *ea,
                                 float *kq) {
    const uint32_t qi = blockIdx.x;
    if (qi >= nq) return;
    float s = 0.f;
    for (uint32_t c = threadIdx.x; c < dim; c += 64) {
        const float v = q[(u64)qi * ldq + c];
        s = __builtin_fmaf(v, v, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) {
        float e = (ip ? 1.f : 2.f) * C * __builtin_sqrtf(s) * 1.0002f;
        if (!(e == e)) e = __builtin_inff();  // a query that is not finite: every row is admitted, the lists overflow, the f32 form takes over
        ea[qi] = e;
        kq[qi] = ip ? 0.f : kappa * s * 1.0002f;
    }
}

// eps(q, r) from the two sides: ea[q] * |x_r| (+ kappa |x_r|^2 + kq[q] for L2)
__device__ __forceinline__ float knn_eps(float ea, float kq, float bn, float kappa, int ip) {
    const float xn = __builtin_sqrtf(bn) * 1.0002f;
    return ip ? ea * xn : __builtin_fmaf(ea, xn, __builtin_fmaf(kappa, bn, kq));
}

struct KnnBfParams {
    const uint4 *qb;  // [nq padded to QT][KQ]
    const uint4 *bb;  // [n][KQ]
    uint32_t KQ, ksteps;
    uint32_t nq;
    u64 first, stride, count;  // rows first + i * stride, i < count
    uint32_t rows_per_slice;   // multiple of BT
    const float *U;            // [nq] k-th smallest hi of the previous sample (+inf in the first pass)
    const float *ea, *kq;      // [nq] the query's side of the bound
    float kappa;
    const float *bnorm;        // [n] squared row norms in f32
    u64 *cand;                 // [nq][cap]  ord(hi) << 32 | row
    uint32_t *cursor;
    uint32_t cap;
    int ip;
};

// KS = K steps of 16 the A fragments are sized for (dim <= 16 KS); p.ksteps <= KS are real, the others are skipped
template <int KS>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) knn_pass_bf16_kernel(const KnnBfParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const uint32_t KQ = p.KQ;
    const uint32_t tile_quads = BT * KQ, J = (tile_quads + 511u) / 512u;  // DMA instructions per thread and tile
    uint4 *buf0 = reinterpret_cast<uint4 *>(smem), *buf1 = buf0 + (size_t)J * 512;
    const uint32_t nqt = (p.nq + QT - 1) / QT;
    const uint32_t q0 = (blockIdx.x % nqt) * QT;
    const u64 s0 = (u64)(blockIdx.x / nqt) * p.rows_per_slice;
    const u64 s1 = std::min<u64>(p.count, s0 + p.rows_per_slice);

    // A fragments: query row q0 + 32 wave + r, k = 16 ks + 8 h .. + 7  (the copy is padded to whole tiles with zero rows)
    bf16x8 af[KS];
    {
        const uint4 *qrow = p.qb + (u64)(q0 + 32 * wave + r) * KQ + h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if ((uint32_t)ks < p.ksteps) v = qrow[2 * ks];
            af[ks] = __builtin_bit_cast(bf16x8, v);
        }
    }
    // this lane's 16 query rows (D row = (reg & 3) + 8 (reg >> 2) + 4 h): admission bound U (+ the query's constant part of eps) and
    // the query's factor of the row-dependent part
    float thr[16], eq[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const uint32_t qi = q0 + 32 * wave + (g & 3) + 8 * (g >> 2) + 4 * h;
        thr[g] = qi < p.nq ? p.U[qi] + p.kq[qi] : -__builtin_inff();
        eq[g] = qi < p.nq ? p.ea[qi] : 0.f;
    }
    // where this thread's DMA slots of a tile come from: slot = j * 512 + tid -> (row of the tile, quad of the row)
    uint32_t srow[9], scq[9];  // J <= 9 (KQ <= 65)
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const uint32_t slot = (uint32_t)j * 512u + (uint32_t)tid;
        srow[j] = slot / KQ;
        scq[j] = slot % KQ;
    }
    auto dma = [&](uint4 *buf, u64 t0) {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            if ((uint32_t)j < J && srow[j] < (uint32_t)BT) {
                u64 si = t0 + srow[j];
                si = si < p.count ? si : p.count - 1;  // (a tile past the end re-reads the last row: masked in the epilogue)
                const uint4 *src = p.bb + (p.first + si * p.stride) * KQ + scq[j];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(buf + (size_t)j * 512 + wave * 64), 16, 0, 0);
            }
        }
    };
    if (s0 >= s1) return;
    dma(buf0, s0);
    int cur = 0;
    for (u64 t0 = s0; t0 < s1; t0 += BT, cur ^= 1) {
        uint4 *bufc = cur ? buf1 : buf0, *bufn = cur ? buf0 : buf1;
        __syncthreads();  // (drains this wave's DMA of tile t0 -- vmcnt(0) -- and everybody has left the other buffer)
        if (t0 + BT < s1) dma(bufn, t0 + BT);
        f32x16 acc0, acc1;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc0[g] = 0.f, acc1[g] = 0.f;
        const uint4 *b0 = bufc + (size_t)r * KQ + h, *b1 = bufc + (size_t)(32 + r) * KQ + h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if ((uint32_t)ks < p.ksteps) {
                const bf16x8 f0 = __builtin_bit_cast(bf16x8, b0[2 * ks]), f1 = __builtin_bit_cast(bf16x8, b1[2 * ks]);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], f0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], f1, acc1, 0, 0, 0);
            }
        }
        // epilogue: column (base row) = lane & 31 of each 32-wide half, row (query) in the register index
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const u64 si = t0 + 32 * t + r;
            const bool cv = si < s1;
            const u64 row = p.first + (cv ? si : s1 - 1) * p.stride;
            const float bn = p.bnorm[row];
            const float xn = __builtin_sqrtf(bn) * 1.0002f;
            const float bl = p.ip ? 0.f : __builtin_fmaf(-p.kappa, bn, bn);  // L2: |x|^2 less its own part of eps
            uint32_t hits = 0;
#pragma unroll
            for (int g = 0; g < 16; ++g) {  // lo = d^ - eps <= U  (kq, the query's constant part of eps, sits in thr)
                const float dot = t ? acc1[g] : acc0[g];
                const float lo = __builtin_fmaf(-eq[g], xn, p.ip ? -dot : __builtin_fmaf(-2.f, dot, bl));
                hits |= (cv && lo <= thr[g]) ? (1u << g) : 0u;
            }
            if (__ballot(hits != 0)) {  // rare: ~k x ratio admissions per query and pass
                for (; hits; hits &= hits - 1) {
                    const int g = __ffs((int)hits) - 1;
                    const float dot = t ? acc1[g] : acc0[g];
                    const uint32_t qi = q0 + 32 * wave + (g & 3) + 8 * (g >> 2) + 4 * h;
                    const float dist = p.ip ? -dot : __builtin_fmaf(-2.f, dot, bn);
                    const float hi = dist + knn_eps(eq[g], p.kq[qi], bn, p.kappa, p.ip);
                    const uint32_t pos = atomicAdd(&p.cursor[qi], 1u);
                    if (pos < p.cap) p.cand[(u64)qi * p.cap + pos] = ((u64)f2ord_b(hi) << 32) | (uint32_t)row;
                }
            }
        }
    }
}

// per query (one workgroup): sort the admitted list by (hi, id), keep the best k in front (every row after the last pass), tau = U = the k-th hi
__global__ void __launch_bounds__(256) knn_bf_select_kernel(u64 *cand, uint32_t *cursor, uint32_t cap, uint32_t nq, uint32_t k, float *tau,
                                                           uint32_t *overflow, int keep_all) {
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
        if (rank < k || keep_all) list[rank] = key;  // (last pass: the whole list stays, sorted, for the re-rank)
        if (rank == k - 1) tau[q] = ord2f_b((uint32_t)(key >> 32));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!keep_all) cursor[q] = m < k ? m : k;
        else cursor[q] = m;
        if (m < k) tau[q] = __builtin_inff();
    }
}

// Exact re-rank (header, step 4).  One workgroup of four waves per query: the admitted rows whose lo = hi - 2 eps is <= U are scored on
// the f32 rows in the reference's summation order (16 rows per wave and pass, four lanes per row), keyed ord(distance) << 32 | id, ranked
// by counting; the best k leave as ids (+ reference-signed distances).
template <bool IP>
__global__ void __launch_bounds__(256) knn_rerank_kernel(const u64 *__restrict__ cand, const uint32_t *__restrict__ cursor, uint32_t cap, uint32_t nq,
                                                         uint32_t k, const float *__restrict__ tau, const float *__restrict__ ea,
                                                         const float *__restrict__ kq, float kappa, const float *__restrict__ bnorm,
                                                         const float *__restrict__ q, u64 ldq, const float *__restrict__ base, u64 ldb, uint32_t dim,
                                                         uint32_t *ids, float *dists, uint32_t *rerank_rows) {
    extern __shared__ __align__(16) unsigned char sm[];
    const uint32_t qi = blockIdx.x;
    if (qi >= nq) return;
    float *lq = reinterpret_cast<float *>(sm);                             // [dim]
    u64 *keys = reinterpret_cast<u64 *>(sm + ((dim * 4u + 15u) & ~15u));  // [cap]: first the row ids to score, then their exact keys
    __shared__ uint32_t s_m;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, t = lane & 3, g = lane >> 2;
    for (uint32_t c = tid; c < dim; c += 256) lq[c] = q[(u64)qi * ldq + c];
    const uint32_t m = std::min(cursor[qi], cap);
    const float U = tau[qi], e_q = ea[qi], k_q = kq[qi];
    const u64 *list = cand + (u64)qi * cap;
    if (tid == 0) s_m = 0;
    __syncthreads();
    for (uint32_t i = tid; i < m; i += 256) {  // (sorted by hi: the first k are the rows U comes from -- always in)
        const u64 key = list[i];
        const uint32_t id = (uint32_t)key;
        const float hi = ord2f_b((uint32_t)(key >> 32));
        const float lo = hi - 2.0002f * knn_eps(e_q, k_q, bnorm[id], kappa, IP ? 1 : 0);
        if (lo <= U || i < k) keys[atomicAdd(&s_m, 1u)] = id;
    }
    __syncthreads();
    const uint32_t m2 = s_m;
    if (rerank_rows && tid == 0) atomicAdd(rerank_rows, m2);
    for (uint32_t p0 = 64u * wave; p0 < m2; p0 += 256u) {  // each wave: 16 rows per pass, 4 passes cover its 64
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const uint32_t c = p0 + 16u * pass + g;
            if (c < m2) {
                const uint32_t id = (uint32_t)keys[c];
                const float d = row_distance<IP, 0>(base + (u64)id * ldb, lq, dim, t);
                wave_sync();
                if (t == 0) keys[c] = ((u64)f2ord_b(d) << 32) | id;
            }
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < m2; i += 256) {
        const u64 key = keys[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < m2; ++j) rank += (keys[j] < key) ? 1u : 0u;  // (ids are distinct: no equal keys)
        if (rank < k) {
            ids[(u64)qi * k + rank] = (uint32_t)key;
            if (dists) dists[(u64)qi * k + rank] = ord2f_b((uint32_t)(key >> 32));
        }
    }
    for (uint32_t i = m2 + tid; i < k; i += 256) {  // fewer rows than k (n < k cannot happen; a short list would be a bug): "no result"
        ids[(u64)qi * k + i] = 0xFFFFFFFFu;
        if (dists) dists[(u64)qi * k + i] = __builtin_inff();
    }
}

template <int KS>
hipError_t launch_pass(const KnnBfParams &p, uint32_t grid, size_t lds, hipStream_t st) {
    hipError_t e = hipFuncSetAttribute((const void *)knn_pass_bf16_kernel<KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(knn_pass_bf16_kernel<KS>, dim3(grid), dim3(512), lds, st, p);
    return hipGetLastError();
}

}  // namespace

extern "C" {

// quads per row of the bf16 copies: dim rounded up to whole K steps of 16, an odd number of quads
uint32_t roar_knn_bf16_row_quads(uint32_t dim) {
    const uint32_t kq = ((dim + 15u) / 16u) * 2u;
    return kq | 1u;
}

// f32 rows -> bf16 copy ([n_pad][KQ] quads; rows past n zero); sqnorm / max_sq_bits may be null.  *max_sq_bits must be zeroed by the caller.
int roar_knn_bf16_convert(const float *src, uint64_t ld, uint32_t dim, uint64_t n, uint64_t n_pad, void *dst, float *sqnorm, uint32_t *max_sq_bits,
                          void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const uint32_t KQ = roar_knn_bf16_row_quads(dim);
    hipLaunchKernelGGL(knn_to_bf16_kernel, dim3(4096), dim3(256), 0, st, src, (u64)ld, dim, (u64)n, (u64)n_pad, KQ, (uint4 *)dst);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (sqnorm || max_sq_bits) {
        hipLaunchKernelGGL(knn_sqnorm_kernel, dim3(2048), dim3(256), 0, st, src, (u64)ld, dim, (u64)n, sqnorm, max_sq_bits);
        e = hipGetLastError();
    }
    return (int)e;
}

// One block of queries (nq <= 65 536) against the whole base set.  Workspaces: qb ((nq rounded up to 256) x KQ quads), cand nq * cap * 8 B,
// cursor / tau / ea / kq nq * 4 B each, overflow_dev 4 B, rerank_rows_dev 4 B (may be null: rows that reached the exact re-rank, summed).
// bb = bf16 copy of the base rows, bnorm = their squared norms (f32; both metrics: the bound is per row).
// Returns a hipError_t (0 ok); *overflowed (host) > 0: some list exceeded `cap`, retry larger.
int roar_knn_run_bf16(const float *q, uint64_t ldq, uint32_t nq, const float *b, uint64_t ldb, uint64_t n, uint32_t dim, int ip, uint32_t k,
                      uint32_t cap, const void *bb, const float *bnorm, void *qb, void *cand, void *cursor, void *tau, void *ea, void *kq,
                      void *overflow_dev, uint32_t *rerank_rows_dev, uint32_t *ids, float *dists, uint32_t *overflowed, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    const uint32_t KQ = roar_knn_bf16_row_quads(dim), ksteps = (dim + 15u) / 16u;
    if (ksteps > 32) return (int)hipErrorInvalidValue;
    const uint32_t nq_pad = (nq + QT - 1) / QT * QT;
    if ((e = (hipError_t)roar_knn_bf16_convert(q, ldq, dim, nq, nq_pad, qb, nullptr, nullptr, stream)) != hipSuccess) return (int)e;
    // the bound (header): C for the two bf16 roundings (2^-8 each, Cauchy-Schwarz), the MFMA's f32 accumulation over K and the exact pass's own
    // f32 summation (distance.h's tree: D / 16 + 6 additions per partial sum)
    const double K = 16.0 * ksteps;
    const float C = (float)(2.02 / 256.0 + (K + dim / 16.0 + 8.0) * 1.25 / 16777216.0);
    const float kappa = (float)((dim + 2.0 * (dim / 16.0 + 8.0)) * 1.5 / 16777216.0);
    hipLaunchKernelGGL(knn_slack_kernel, dim3(nq), dim3(64), 0, st, q, (u64)ldq, dim, nq, ip, C, kappa, (float *)ea, (float *)kq);
    {
        // tau = +inf
        if ((e = hipMemsetD32Async((hipDeviceptr_t)tau, (int)0x7F800000, nq, st)) != hipSuccess) return (int)e;
    }
    if ((e = hipMemsetAsync(overflow_dev, 0, 4, st)) != hipSuccess) return (int)e;
    uint64_t ratio = cap / (4ull * k);
    if (ratio < 4) ratio = 4;
    if (ratio > 64) ratio = 64;
    int P = 1;
    uint64_t stride0 = 1;
    while ((n + stride0 - 1) / stride0 > cap / 2) {
        stride0 *= ratio;
        ++P;
    }
    if ((e = hipFuncSetAttribute((const void *)knn_bf_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)cap * 8))) != hipSuccess)
        return (int)e;
    const uint32_t J = (BT * KQ + 511u) / 512u;
    const size_t lds = (size_t)2 * J * 512 * 16;
    uint64_t stride = stride0;
    for (int pass = 0; pass < P; ++pass, stride /= ratio) {
        const int last = pass == P - 1;
        if ((e = hipMemsetAsync(cursor, 0, (size_t)nq * 4, st)) != hipSuccess) return (int)e;
        KnnBfParams p;
        p.qb = (const uint4 *)qb, p.bb = (const uint4 *)bb, p.KQ = KQ, p.ksteps = ksteps, p.nq = nq;
        p.first = 0, p.stride = stride, p.count = (n + stride - 1) / stride;
        // slices: ~64 MB of bf16 rows (they sit in the Infinity Cache while the workgroups of one slice stream them), whole tiles
        uint64_t rps = std::max<uint64_t>(BT, ((64ull << 20) / (KQ * 16ull)) / BT * BT);
        p.rows_per_slice = (uint32_t)rps;
        p.U = (const float *)tau, p.ea = (const float *)ea, p.kq = (const float *)kq, p.kappa = kappa;
        p.bnorm = bnorm, p.cand = (u64 *)cand, p.cursor = (uint32_t *)cursor, p.cap = cap, p.ip = ip;
        const uint64_t nqt = nq_pad / QT, nslices = (p.count + rps - 1) / rps;
        const uint64_t grid = nqt * nslices;
        if (grid > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
        if (ksteps <= 8) e = launch_pass<8>(p, (uint32_t)grid, lds, st);
        else if (ksteps <= 13) e = launch_pass<13>(p, (uint32_t)grid, lds, st);
        else e = launch_pass<32>(p, (uint32_t)grid, lds, st);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(knn_bf_select_kernel, dim3(nq), dim3(256), (size_t)cap * 8, st, (u64 *)cand, (uint32_t *)cursor, cap, nq, k, (float *)tau,
                           (uint32_t *)overflow_dev, last);
        if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    }
    const size_t rl = ((dim * 4u + 15u) & ~15u) + (size_t)cap * 8;
    const void *rk = ip ? (const void *)knn_rerank_kernel<true> : (const void *)knn_rerank_kernel<false>;
    if ((e = hipFuncSetAttribute(rk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl)) != hipSuccess) return (int)e;
    if (ip)
        hipLaunchKernelGGL(knn_rerank_kernel<true>, dim3(nq), dim3(256), rl, st, (const u64 *)cand, (const uint32_t *)cursor, cap, nq, k, (const float *)tau,
                           (const float *)ea, (const float *)kq, kappa, bnorm, q, (u64)ldq, b, (u64)ldb, dim, ids, dists, rerank_rows_dev);
    else
        hipLaunchKernelGGL(knn_rerank_kernel<false>, dim3(nq), dim3(256), rl, st, (const u64 *)cand, (const uint32_t *)cursor, cap, nq, k, (const float *)tau,
                           (const float *)ea, (const float *)kq, kappa, bnorm, q, (u64)ldq, b, (u64)ldb, dim, ids, dists, rerank_rows_dev);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    if (overflowed) {
        if ((e = hipMemcpyAsync(overflowed, overflow_dev, 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return (int)e;
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return (int)e;
    }
    return 0;
}
}
