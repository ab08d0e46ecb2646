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
//      with hi <= U is missed), keyed by hi; the selection finds U among them -- also BETWEEN the launches of one pass (round 5: the
//      k-th smallest hi of what the pass has admitted so far is such a bound too; the smaller of the two is kept) --; after the last
//      pass the rows with lo <= U go on;
//   4. those (~1.3-2 k rows per query) are scored on the f32 rows by the search kernels' own distance routine -- the reference's
//      summation order (roar_wave.h row_distance = distance.h:39-89,179-223), bit-identical to DistanceInnerProduct / DistanceL2 -- and
//      sorted by (distance, id).  The result is the exact top k under the reference's own arithmetic, whatever the bf16 pass did.
//
// GEMM tile: a workgroup of 8 waves owns 256 queries (wave w: 32 of them, its A fragments -- 13 x 4 VGPRs at D = 200 -- stay in
// REGISTERS for the whole slice) and walks a slice of the sample in tiles of 64 base rows, NBUF tiles deep in LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, uniform tile base + 32-bit lane offset; issued one at a time between the MFMAs).  Per
// tile and wave two blocks of 32 rows: K/16 MFMAs of 32x32x16 each, then ONE more MFMA that forms the admission test itself (the
// verdict columns: knn_pass_bf16_kernel, "The verdict as one more MFMA"), and between the next block's MFMAs the maximum of the sixteen
// verdict registers and one compare; admitted elements are staged in LDS.  Two tiles between two barriers.  1-D grid, query tile
// fastest: the workgroups that run together stream the same slice (L2 / Infinity Cache), every base row leaves HBM about once per
// pass.  A pass over many slices is up to eight launches with a selection after each: the bound tightens while the pass runs.
// Measured: round 4 (profiles/r04_knn_kernel_stats.txt, r04_knn_counters.txt, r04_mfma_probe.txt) last pass 281 ms = 0.39 of the bf16
// peak, one 65 536-query block of the builder's lists 0.340 s; round 5 (profiles/r05_knn_forms.txt, r05_knn_kernel_stats.csv):
// 0.281 s per block (512-d, 2.5 M rows: 0.241 -> 0.186 s; 128-d, 1 M rows: 0.071 -> 0.058 s).  What the round learned about the rest:
// with nothing admitted, no loads and no barrier the loop still takes 0.6 of its time at the nominal peak -- the shader clock, by the
// instrumented kernel's cycle counts against wall time, falls from ~2.45 GHz to ~2.1 GHz as the matrix cores fill.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "roar_wave.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BT = 64;   // base rows per LDS tile

__device__ __forceinline__ uint32_t f2ord_b(float f) {
    if (f == 0.0f) f = 0.0f;
    uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float ord2f_b(uint32_t o) { return __uint_as_float((o >> 31) ? (o ^ 0x80000000u) : ~o); }

// bf16 images of an f32 value: the nearest (ties to even) and the nearest that is NOT BELOW it; as f32 again
__device__ __forceinline__ float bf16_rne(float v) {
    uint32_t u = __float_as_uint(v);
    if ((u & 0x7F800000u) == 0x7F800000u) return __uint_as_float(u & 0xFFFF0000u);  // inf stays inf
    u += 0x7FFFu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xFFFF0000u);
}
__device__ __forceinline__ float bf16_up(float v) {
    const uint32_t u = __float_as_uint(v), t = u & 0xFFFF0000u;
    if ((u & 0x7F800000u) == 0x7F800000u || (u & 0xFFFFu) == 0u) return __uint_as_float(t);
    return (u >> 31) ? __uint_as_float(t) : __uint_as_float(t + 0x10000u);  // negative: cutting the low bits rounds up
}
// v as a sum of three bf16 values t1 + t2 + t3 >= v (the two differences are exact in f32; the last part is rounded up): a value of
// f32 precision as three columns of a bf16 MFMA whose other side holds 1.0.  +-inf: the first part alone.
__device__ __forceinline__ void bf16_split3_up(float v, float &t1, float &t2, float &t3) {
    t1 = bf16_rne(v);
    if ((__float_as_uint(v) & 0x7F800000u) == 0x7F800000u) {
        t2 = t3 = 0.f;
        return;
    }
    const float r1 = v - t1;
    t2 = bf16_rne(r1);
    t3 = bf16_up(r1 - t2);
}
// what the verdict columns leave to the sums' own rounding (knn_pass_bf16_kernel, "The verdict as one more MFMA"): relative to the
// threshold / the row's |x|^2 they carry
constexpr float KNN_VERDICT_MARGIN = 4.0e-5f;

// f32 rows -> bf16 rows of KQ quads (8 values each; everything past `dim` is zero), rows [0, n) from src, rows [n, n_pad) zero.
// aug != 0 (the BASE rows; sqnorm = their squared norms, f32): quad dim / 8 -- the first past the data, always there: KQ is odd -- holds
// the row's side of the pass kernel's verdict columns: 1 (inner product): xn^, 1, 1, 1;  2 (L2): xn^, 1, 1, 1, b1, b2, b3 with
// xn^ = |x_r| rounded up, b1 + b2 + b3 >= -(|x_r|^2 less its own part of eps) / 2.
__global__ void knn_to_bf16_kernel(const float *__restrict__ src, u64 ld, uint32_t dim, u64 n, u64 n_pad, uint32_t KQ, uint4 *__restrict__ dst,
                                   int aug, const float *__restrict__ sqnorm, float kappa) {
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
        uint4 q = __builtin_bit_cast(uint4, o);
        if (aug && row < n && c0 == dim) {
            const float bn = sqnorm[row];
            const float xn = bf16_up(__builtin_sqrtf(__builtin_fmaxf(bn, 1e-30f)) * 1.0003f);  // >= the pass kernel's own xn (hardware sqrt x 1.0002)
            const uint32_t one = 0x3F80u;
            q.x = (__float_as_uint(xn) >> 16) | (one << 16);
            q.y = one | (one << 16);
            if (aug == 2) {
                const float bl = __builtin_fmaf(-kappa, bn, bn);
                float b1, b2, b3;
                bf16_split3_up(-0.5f * (bl - KNN_VERDICT_MARGIN * bl), b1, b2, b3);
                q.z = (__float_as_uint(b1) >> 16) | (__float_as_uint(b2) & 0xFFFF0000u);
                q.w = __float_as_uint(b3) >> 16;
            }
        }
        dst[i] = q;
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

// LDS accesses of the pass kernel's main loop are written as instructions: the compiler cannot tell that a ds_read does not touch the
// buffer an LDS-DMA in flight is writing (one LDS array, run-time buffer index) and would wait vmcnt(0) in front of every one of them --
// draining the look-ahead loads the pipeline lives on.  The waits these need (lgkmcnt) are written out with them.
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(size_t)(__attribute__((address_space(3))) const void *)p; }
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// LDS-DMA with the address split the way the hardware takes it: uniform 64-bit base in scalar registers + 32-bit per-lane offset; the
// LDS side is M0 (uniform base; the lanes' 16 / 4 bytes follow each other from there).  Every LDS-DMA of the pass kernel goes through
// these four; each tells the compiler that it overwrites M0.
__device__ __forceinline__ void dma_b128(const void *base, uint32_t off, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void dma_b32(const void *base, uint32_t off, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void dma_b128_at(const void *src, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void dma_b32_at(const void *src, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(lds) : "memory", "m0");
}
template <int OFF>
__device__ __forceinline__ void lds_read_b128(uint32_t a, u32x4 &f) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(f) : "v"(a), "n"(OFF) : "memory");
}
template <int LEFT>
__device__ __forceinline__ void lds_wait1(u32x4 &f) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(LEFT) : "memory");
}
__device__ __forceinline__ void lds_read2_f32_nowait(uint32_t a, float &x0, float &x1) {
    asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:128" : "=&v"(x0), "=&v"(x1) : "v"(a) : "memory");
}
__device__ __forceinline__ uint32_t lds_add_rtn(uint32_t a, uint32_t v) {
    uint32_t old;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(old) : "v"(a), "v"(v) : "memory");
    return old;
}
__device__ __forceinline__ uint32_t lds_read_b32(uint32_t a) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
    return v;
}
__device__ __forceinline__ void lds_write_b32(uint32_t a, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_write_b64(uint32_t a, u64 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_write_b16(uint32_t a, uint32_t v) { asm volatile("ds_write_b16 %0, %1" ::"v"(a), "v"(v) : "memory"); }

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>());
        static_for<I + 1, N>(f);
    }
}

struct KnnBfParams {
    const uint4 *qb;  // [nq padded to QT][KQ]
    const uint4 *bb;  // [n][KQ]
    uint32_t KQ;
    uint32_t dim8;  // quads of data per row (dim / 8): the quad behind them holds the base rows' verdict columns
    uint32_t nq;
    u64 first, stride, count;  // rows first + i * stride, i < count
    uint32_t slice0;           // this launch's first slice of the sample (a pass may be several launches, see roar_knn_run_bf16)
    uint32_t rows_per_slice;   // multiple of BT
    const float *U;            // [nq] k-th smallest hi of the previous sample (+inf in the first pass)
    const float *ea, *kq;      // [nq] the query's side of the bound
    float kappa;
    const float *bnorm;        // [n] squared row norms in f32
    u64 *cand;                 // [nq][cap]  ord(hi) << 32 | row
    uint32_t *cursor;
    uint32_t cap;
    int ip;
    unsigned long long *prof;  // debug (ROAR_KNN_PROF): cycles per phase, summed over the waves that ran the instrumented form
    int debug;                 // debug (ROAR_KNN_DEBUG): bit 0 nothing is admitted, bit 1 no periodic flush -- timing experiments only
};

// KS = K steps of 16 the A fragments are sized for (dim <= 16 KS; p.ksteps <= KS are real, the others are skipped); JT = LDS-DMA
// instructions per thread and tile for the longest row of that bucket; NBUF = tile buffers (NBUF - 1 tiles in flight); QH = blocks of
// 32 queries per wave (their A fragments stay in registers), NW = waves per workgroup: a workgroup owns 32 QH NW queries.
//
// Round 5 -- why QH = 2: with 32 queries per wave every B fragment read from LDS (64 lanes x 16 bytes = 1 KB) feeds ONE MFMA of 32 cycles;
// four SIMDs doing that ask the LDS for 4 KB per 32 cycles = its whole 128 bytes per cycle, before the tile's own LDS-DMA writes: the
// matrix cores and the LDS were both the bound, and behind a per-tile barrier neither was ever busy all the time (0.39 of the bf16 peak,
// profiles/r04_knn_kernel_stats.txt).  With 64 queries per wave a fragment feeds two MFMAs and the LDS runs at half its rate.
//
// Pipeline (round 4's, still the 512-d bucket's; up to 208-d an iteration is a PAIR of tiles now: TPB at the main loop).  One iteration =
// one tile of 64 base rows: a COUNTED wait -- s_waitcnt vmcnt((NBUF - 2)(JT + 1)): this thread's loads of the
// tile about to be read have landed, those of the NBUF - 2 tiles behind it stay in flight -- then a raw s_barrier (everybody's have, and
// everybody has left the buffer read last), the loads of tile t + NBUF - 1 into that buffer, the MFMAs, the epilogue.  A
// __syncthreads() would drain every load in flight (the compiler's fence waits vmcnt(0) for a pending LDS write), and one tile ahead
// is not enough: a tile is ~0.35 us of MFMA work, a load from the L2 / Infinity Cache ~1-2 us away (first form of this kernel: 267
// TFLOP/s).  For the count to be the same in every wave each thread issues exactly JT + 1 loads per tile (slots past the tile and tiles
// past the slice re-read a valid row into spare LDS), the rows' squared norms travel by LDS-DMA too, and nothing else in the loop is a
// vector-memory operation: admitted rows are staged in LDS (ds_add_rtn + ds_write) and leave for the candidate lists every 32 tiles.
template <int KS, int JT, int NBUF, int HCAP, bool IP, int QH, int NW, int WPE, bool PROF = false>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) knn_pass_bf16_kernel(const KnnBfParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NT = 64 * NW;       // threads
    constexpr int QT = 32 * QH * NW;  // queries per workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    constexpr uint32_t KQ = 2 * KS + 1;  // (= p.KQ: roar_knn_bf16_row_quads)
    static_assert((uint32_t)JT * NT >= (uint32_t)BT * KQ, "JT: a tile's quads must fit the workgroup's DMA slots");
    uint4 *bufs = reinterpret_cast<uint4 *>(smem);                                      // [NBUF][JT * NT] quads
    float *nrm = reinterpret_cast<float *>(smem + (size_t)NBUF * JT * NT * 16);          // [NBUF][64] squared norms of the tile's rows
    u64 *skey = reinterpret_cast<u64 *>(smem + (size_t)NBUF * JT * NT * 16 + NBUF * 256);  // [HCAP] bits(d^) << 32 | row
    uint16_t *sql = reinterpret_cast<uint16_t *>(skey + HCAP);                           // [HCAP] query of the entry, relative to q0
    uint32_t *scnt = reinterpret_cast<uint32_t *>(sql + HCAP);
    uint32_t *sflag = scnt + 1;  // "flush at the next opportunity" (see the main loop)
    const uint32_t nqt = (p.nq + QT - 1) / QT;
    const uint32_t q0 = (blockIdx.x % nqt) * QT;
    const u64 s0 = (u64)(p.slice0 + blockIdx.x / nqt) * p.rows_per_slice;
    const u64 s1 = std::min<u64>(p.count, s0 + p.rows_per_slice);
    if (s0 >= s1) return;
    if (tid == 0) scnt[0] = 0, scnt[1] = 0;  // (word 1: the flush flag)

    // A fragments: query row q0 + 32 (QH wave + qh) + r, k = 16 ks + 8 h .. + 7  (the copy is padded to whole tiles with zero rows)
    bf16x8 af[QH][KS];
#pragma unroll
    for (int qh = 0; qh < QH; ++qh) {
        const uint4 *qrow = p.qb + (u64)(q0 + 32 * (QH * wave + qh) + r) * KQ + h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[qh][ks] = __builtin_bit_cast(bf16x8, qrow[2 * ks]);  // (rows are zero beyond the dimension)
    }
    // The verdict as one more MFMA (round 5).  "Admit (q, r) iff lo = d^ - eps(q, r) <= U_q" was an fma and a compare per element, 32 vector
    // instructions per block of 13 MFMAs -- and a vector instruction and an MFMA of one SIMD do not overlap (profiles/r04_mfma_probe.txt:
    // 0.92 -> 0.72 of the peak).  Both sides of the test are bilinear in (something of q) x (something of r), so the matrix core can
    // form it: the quad behind the data of every base row (knn_to_bf16_kernel) holds xn^ >= |x_r| and ones, the matching A fragment
    // (a_aug: this lane's query, built here) holds e^ >= the query's factor of eps and the bound U (+ margin) split into three bf16
    // parts, and  S = mfma(a_aug, that quad, acc)  =  acc + e^ xn^ + U'  (inner product: d^ = -acc, so lo <= U  <=>  acc + eps + U >= 0;
    // L2: acc + (e/2) xn + (U + kq) / 2 - bl / 2 >= 0 with the row's bl in three parts against ones on the query side).  Every rounding
    // goes the admitting way (parts rounded up, KNN_VERDICT_MARGIN on U and bl, 0.1 % on e for the seven extra terms of the f32 sum), so
    // {S >= 0} is a superset of {lo <= U}: all the pass needs.  The clean accumulator stays as it is for the admitted elements' d^.
    // Verdicts of a block: max over its 16 registers (v_max3 x 8) and ONE compare, under the next block's MFMAs.
    const uint32_t AQ = p.dim8;  // quad of the verdict columns = the first behind the data
    bf16x8 a_aug[QH];
#pragma unroll
    for (int qh = 0; qh < QH; ++qh) {
        const uint32_t qi = q0 + 32 * (QH * wave + qh) + r;
        float e = 0.f, t1 = -__builtin_inff(), t2 = 0.f, t3 = 0.f;  // (a query past the end: never admitted)
        if (qi < p.nq && !(p.debug & 1)) {
            e = bf16_up(p.ea[qi] * (IP ? 1.001f : 0.5005f));
            const float thr = p.U[qi] + p.kq[qi];
            const float thr_m = thr + __builtin_fabsf(thr) * KNN_VERDICT_MARGIN + 1.0e-30f;
            bf16_split3_up(IP ? thr_m : 0.5f * thr_m, t1, t2, t3);
        }
        u32x4 w;
        w[0] = (__float_as_uint(e) >> 16) | (__float_as_uint(t1) & 0xFFFF0000u);
        w[1] = (__float_as_uint(t2) >> 16) | (__float_as_uint(t3) & 0xFFFF0000u);
        w[2] = IP ? 0u : 0x3F803F80u;  // L2: ones against the row's b1, b2
        w[3] = IP ? 0u : 0x00003F80u;  //     and b3
        if (h != 0) w[0] = w[1] = w[2] = w[3] = 0u;  // (the fragment's other half: its lanes read the same quad, against zeros)
        a_aug[qh] = __builtin_bit_cast(bf16x8, w);
    }
    // where this thread's DMA slots of a tile come from: slot = j * NT + tid -> (row of the tile, quad of the row); slots past the tile
    // re-read its last row (into the spare end of the buffer).  Kept as ONE 32-bit byte offset from the tile's first row: inside the slice
    // a source address is a UNIFORM tile base (scalar registers, scalar arithmetic) + that offset (the first form multiplied
    // row x stride x KQ in 64 bits per lane and load: five quarter-rate multiplies per load, ~600 cycles per wave and tile).
    // (the 512-d bucket, its A fragments half the register file, recomputes it per load: a spilled offset comes back by a scratch load, and
    // the wait for THAT drains the LDS-DMA pipeline)
    constexpr bool KEEP = KS <= 16;
    const uint32_t rowstep = (uint32_t)(p.stride * KQ);
    auto slot_off = [&](int j, uint32_t tid_) {
        const uint32_t slot = (uint32_t)j * (uint32_t)NT + tid_;
        return (std::min<uint32_t>(slot / KQ, BT - 1) * rowstep + slot % KQ) * 16u;
    };
    uint32_t soff[KEEP ? JT : 1];
    if constexpr (KEEP) {
#pragma unroll
        for (int j = 0; j < JT; ++j) soff[j] = slot_off(j, (uint32_t)tid);
    }
    const uint32_t noff = (uint32_t)((u64)lane * p.stride) * 4u;
    const bool near = (u64)BT * p.stride * KQ * 16u < (1ull << 31);  // (uniform; the offsets above fit 32 bits)
    // everything loaded so far is in its registers before the pipeline starts: inside the loop the only vector-memory operations are the
    // LDS-DMA loads, and the compiler has no reason to wait for anything there (it would wait for ALL of them)
#pragma unroll
    for (int qh = 0; qh < QH; ++qh) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(af[qh][ks]));
        asm volatile("" ::"v"(a_aug[qh]));
    }
    // The JT + 1 loads of a tile (always that many).  Inside the slice (`fast`) they are issued ONE AT A TIME between the MFMAs of the
    // tile's first block (dma_one): the eight waves of a workgroup leave the tile's barrier together, and forty 1 KB LDS-DMA instructions
    // through the CU's one address path at once cost the last wave ~500 cycles before its first MFMA (round 5, phase counters of the
    // instrumented form); behind an MFMA the same wait costs nothing.
    struct DmaTile {
        const unsigned char *tb, *nb;
        uint32_t lb, ln;
        bool fast;
    };
    auto dma_prepare = [&](uint32_t b, u64 t0) {
        DmaTile d;
        d.lb = __builtin_amdgcn_readfirstlane(lds_addr(bufs + (size_t)b * JT * NT + wave * 64));
        d.ln = lds_addr(nrm + b * 64);
        d.fast = near && t0 + BT <= p.count;
        // (uniform, and SAID so: the LDS-DMA takes its base from scalar registers, and the register allocator has been seen to keep such a
        // pointer in vector registers and hand them to the instruction)
        auto uniform_ptr = [](const void *q) {
            const u64 v = (u64)q;
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)), lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
            return reinterpret_cast<const unsigned char *>(((u64)hi << 32) | (u64)lo);  // (readfirstlane returns an int: no sign extension)
        };
        d.tb = uniform_ptr(p.bb + (p.first + t0 * p.stride) * KQ);
        d.nb = uniform_ptr(p.bnorm + (p.first + t0 * p.stride));
        return d;
    };
    auto dma_one = [&](const DmaTile &d, int j) {  // j < JT: quads; j == JT: the rows' squared norms
        uint32_t tid_ = (uint32_t)tid;
        if constexpr (!KEEP) asm volatile("" : "+v"(tid_));  // (recomputed HERE, every time: hoisted out of the loop it would be spilled)
        if (j < JT) dma_b128(d.tb, KEEP ? soff[KEEP ? j : 0] : slot_off(j, tid_), d.lb + (uint32_t)j * (uint32_t)(NT * 16));
        else dma_b32(d.nb, KEEP ? noff : (tid_ & 63u) * (uint32_t)p.stride * 4u, d.ln);
    };
    auto dma_slow = [&](const DmaTile &d, u64 t0) {
        // the slice's last tiles (and strides past 32-bit offsets): rows past the end re-read the last one (masked in the epilogue)
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            const uint32_t slot = (uint32_t)j * (uint32_t)NT + (uint32_t)tid;
            u64 si = t0 + std::min<uint32_t>(slot / KQ, BT - 1);
            si = si < p.count ? si : p.count - 1;
            dma_b128_at(p.bb + (p.first + si * p.stride) * KQ + slot % KQ, d.lb + (uint32_t)j * (uint32_t)(NT * 16));
        }
        u64 si = t0 + (uint32_t)lane;
        si = si < p.count ? si : p.count - 1;
        dma_b32_at(p.bnorm + (p.first + si * p.stride), d.ln);  // (every wave: the same 256 bytes)
    };
    auto dma = [&](uint32_t b, u64 t0) {  // all at once: the pipeline's first tiles
        const DmaTile d = dma_prepare(b, t0);
        if (d.fast) {
#pragma unroll
            for (int j = 0; j <= JT; ++j) dma_one(d, j);
        } else {
            dma_slow(d, t0);
        }
    };
    // Staged admissions -> candidate lists (hi = d^ + eps), by the whole workgroup.  (Round 5, measured and dropped: a staging list per wave
    // with its count in a scalar register -- no ds_add_rtn round trip per admission -- made the pass 14 % SLOWER, 323 against 284 ms: the
    // scan of an admitting block's sixteen registers got four instructions longer per register, and blocks that admit are every third.)
    auto flush = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const uint32_t have = *reinterpret_cast<volatile uint32_t *>(scnt);
        const uint32_t cnt = have < (uint32_t)HCAP ? have : (uint32_t)HCAP;
        for (uint32_t i = tid; i < cnt; i += NT) {
            const u64 e = skey[i];
            const uint32_t row = (uint32_t)e, qi = q0 + sql[i];
            const float dist = __uint_as_float((uint32_t)(e >> 32));
            const float hi = dist + knn_eps(p.ea[qi], p.kq[qi], p.bnorm[row], p.kappa, IP ? 1 : 0);
            const uint32_t pos = atomicAdd(&p.cursor[qi], 1u);
            if (pos < p.cap) p.cand[(u64)qi * p.cap + pos] = ((u64)f2ord_b(hi) << 32) | row;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (stores may overtake loads in the count: nothing of this may be left)
        __builtin_amdgcn_s_barrier();
        if (tid == 0) *scnt = 0;
    };

    const uint32_t ntiles = (uint32_t)((s1 - s0 + BT - 1) / BT);
#pragma unroll
    for (int i = 0; i < ((NBUF == 4 && KS <= 16) ? 2 : NBUF - 1); ++i) dma((uint32_t)i, s0 + (u64)i * BT);  // (see TPB at the main loop)

    // (a quarter of the blocks: ~k x ratio admissions per query and pass): the admitted elements of one block -> the staging list.
    // S = the block's verdict sums (above), acc its clean scores.
    // gm[0..4] = the maxima of the block's registers 0-2, 3-5, ..., 12-14 (the verdict's max tree left them): a group whose maximum is
    // negative in every lane is not looked at -- an admitting block (every third) costs ~9 compare-and-branch pairs instead of 16, and
    // every test is a scalar branch on a lane mask (written otherwise, the compiler turned the sixteen tests into predicated code: ten
    // dependent vector / scalar instructions per register, +25 % on the whole pass).
    auto nonzero = [](u64 m) { return m != 0; };  // (of a lane mask: a scalar compare)
    auto admit = [&](const f32x16 &acc, const f32x16 &S, const float (&gm)[5], int qh, u64 colmask, u64 tile0, int half, float bn) {
        const u64 si = tile0 + 32 * half + r;
        const bool cv = si < s1;
        const uint32_t row = (uint32_t)(p.first + (cv ? si : s1 - 1) * p.stride);
        const u64 cols = colmask & __ballot(cv);  // (rows past the slice are nobody's)
        bool spilled = false;
        auto one = [&](auto cg) {
            constexpr int g = decltype(cg)::value;
            const u64 hm = __ballot(S[g] >= 0.f) & cols;
            if (!nonzero(hm)) return;  // uniform
            asm volatile("" ::: "memory");  // (something that runs under the UNIFORM condition only: the branch above stays a scalar branch)
            if ((hm >> lane) & 1ull) {
                const float dot = acc[g];
                const uint32_t ql = 32 * (QH * wave + qh) + (g & 3) + 8 * (g >> 2) + 4 * h;
                const float dist = IP ? -dot : __builtin_fmaf(-2.f, dot, bn);
                const uint32_t at = lds_add_rtn(lds_addr(scnt), 1u);
                if (at < (uint32_t)HCAP) {
                    lds_write_b64(lds_addr(skey + at), ((u64)__float_as_uint(dist) << 32) | row);
                    lds_write_b16(lds_addr(sql + at), ql);
                } else {  // the staging block is full (the first pass admits every row): straight to the list
                    uint32_t q0v = q0;
                    asm volatile("" : "+v"(q0v));  // (worked out HERE: hoisted out of the loop -- as the compiler did since round 4 -- the sixteen list addresses of this rare path are 32 registers, spilled, and the 512-d bucket reloaded some inside the loop)
                    const uint32_t qi = q0v + ql;
                    const float hi = dist + knn_eps(p.ea[qi], p.kq[qi], bn, p.kappa, IP ? 1 : 0);
                    const uint32_t pos = atomicAdd(&p.cursor[qi], 1u);
                    if (pos < p.cap) p.cand[(u64)qi * p.cap + pos] = ((u64)f2ord_b(hi) << 32) | row;
                    spilled = true;
                }
            }
        };
        static_for<0, 5>([&](auto cgr) {
            constexpr int gr = decltype(cgr)::value;
            if (nonzero(__ballot(gm[gr] >= 0.f) & cols)) {
                asm volatile("" ::: "memory");
                one(std::integral_constant<int, 3 * gr>());
                one(std::integral_constant<int, 3 * gr + 1>());
                one(std::integral_constant<int, 3 * gr + 2>());
            }
        });
        one(std::integral_constant<int, 15>());
        if (__ballot(spilled)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (see flush)
    };
    // KS (K / 16) steps of one block of 32 base rows: a B fragment read LD - 1 steps ahead from LDS address `a` (quad 2 ks + h of the lane's
    // row) feeds QH MFMAs; then the verdict MFMA (S = acc + the verdict columns, their quad read first of all); and BETWEEN the MFMAs the
    // verdict of the block finished before (`Sold`): max of its 16 registers, one compare.  An MFMA occupies the matrix core for 32 cycles
    // and the wave's issue slot for 4, the verdict is independent of it -- written in this order it runs under the MFMAs instead of after
    // them (second form of the loop: all MFMAs of a tile, then all verdicts, both waves of a SIMD in step behind the tile's barrier: 36 %
    // MFMA busy, profiles/r04_knn_counters.txt)
    constexpr int LD = KS <= 16 ? 3 : 2;
    auto block = [&](f32x16 (&acc)[QH], f32x16 (&S)[QH], uint32_t a, const f32x16 (&Sold)[QH], u64 (&any)[QH], float (&gm)[QH][5], auto &&between) {
        u32x4 f[LD], fa;
        lds_read_b128<0>(a + (AQ - (uint32_t)h) * 16u, fa);  // (every lane its row's quad AQ: `a` points at quad h)
        static_for<0, LD - 1>([&](auto c) {
            constexpr int i = decltype(c)::value;
            if constexpr (i < KS) lds_read_b128<i * 32>(a, f[i]);
        });
        float m[QH][7];  // (0..4: the groups' maxima, kept for admit)
        static_for<0, KS>([&](auto c) {
            constexpr int ks = decltype(c)::value;
            constexpr int nx = ks + LD - 1;
            if constexpr (nx < KS) lds_read_b128<(nx < KS ? nx : 0) * 32>(a, f[nx % LD]);
            constexpr int later = (KS - 1 - ks) < (LD - 1) ? (KS - 1 - ks) : (LD - 1);  // reads issued after step ks's
            lds_wait1<later>(f[ks % LD]);
            constexpr int fi = ks % LD;
            static_for<0, QH>([&](auto cq) {
                constexpr int qh = decltype(cq)::value;
                if constexpr (ks == 0) {
                    f32x16 z;
#pragma unroll
                    for (int g = 0; g < 16; ++g) z[g] = 0.f;
                    acc[qh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[qh][ks], __builtin_bit_cast(bf16x8, f[fi]), z, 0, 0, 0);
                } else {
                    acc[qh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[qh][ks], __builtin_bit_cast(bf16x8, f[fi]), acc[qh], 0, 0, 0);
                }
                // the old block's verdict, a step of the max tree behind each of the first eight MFMAs (KS >= 8), the compare behind the ninth
                // (or the last)
                const f32x16 &o = Sold[qh];
                constexpr int step = ks;  // (QH > 1: the trees of the blocks of queries run side by side)
                if constexpr (step == 0) m[qh][0] = __builtin_fmaxf(__builtin_fmaxf(o[0], o[1]), o[2]);
                if constexpr (step == 1) m[qh][1] = __builtin_fmaxf(__builtin_fmaxf(o[3], o[4]), o[5]);
                if constexpr (step == 2) m[qh][2] = __builtin_fmaxf(__builtin_fmaxf(o[6], o[7]), o[8]);
                if constexpr (step == 3) m[qh][3] = __builtin_fmaxf(__builtin_fmaxf(o[9], o[10]), o[11]);
                if constexpr (step == 4) m[qh][4] = __builtin_fmaxf(__builtin_fmaxf(o[12], o[13]), o[14]);
                if constexpr (step == 5) m[qh][5] = __builtin_fmaxf(__builtin_fmaxf(m[qh][0], m[qh][1]), o[15]);
                if constexpr (step == 6) m[qh][6] = __builtin_fmaxf(__builtin_fmaxf(m[qh][2], m[qh][3]), m[qh][4]);
                if constexpr (step == 7) any[qh] = __ballot(__builtin_fmaxf(m[qh][5], m[qh][6]) >= 0.f);
            });
            between(c);
        });
        static_assert(KS >= 8, "the verdict's max tree is spread over the first eight steps");
        lds_wait1<0>(fa);
        static_for<0, QH>([&](auto cq) {
            constexpr int qh = decltype(cq)::value;
            S[qh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_aug[qh], __builtin_bit_cast(bf16x8, fa), acc[qh], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 5; ++i) gm[qh][i] = m[qh][i];
        });
    };
    f32x16 acc0[QH], acc1[QH], S0[QH], S1[QH];
#pragma unroll
    for (int qh = 0; qh < QH; ++qh)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc0[qh][g] = 0.f, acc1[qh][g] = 0.f, S0[qh][g] = -1.f, S1[qh][g] = -1.f;
    float bn1p = 0.f;  // the previous tile's second block: its rows' squared norms
    [[maybe_unused]] long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c0 = 0;
    auto tick = [&](int i) {
        // (a scheduling fence between the phases of an iteration, instrumented or not)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PROF) {
            const long long c1 = (long long)__builtin_readcyclecounter();
            pc[i] += c1 - c0;
            c0 = c1;
        }
    };
    if constexpr (PROF) c0 = (long long)__builtin_readcyclecounter();
    // TPB tiles between two barriers.  TPB = 1: NBUF - 1 tiles in flight, a counted wait per tile (round 4).  TPB = 2 (NBUF = 4): the loop
    // body is a PAIR of tiles -- one wait (everything this thread loaded: the pair), one barrier, then the pair's four blocks with the next
    // pair's loads between the MFMAs of its two first blocks.  From a barrier to the first MFMA behind it the matrix cores of the CU idle
    // (both waves of every SIMD stand there: ~1 000 of an iteration's ~3 000 cycles, phase counters of the instrumented form), and this
    // halves how often that happens.
    constexpr int TPB = (NBUF == 4 && KS <= 16) ? 2 : 1;
    const uint32_t ngroups = (ntiles + TPB - 1) / TPB;
    for (uint32_t tg = 0; tg < ngroups; ++tg) {
        // Everything the tiles need that does not wait for them -- where their fragments are, where the next loads go -- is worked out
        // BEFORE the barrier, under the last MFMAs of the tile before.
        DmaTile dt[TPB];
        uint32_t a0[TPB], a1[TPB], an[TPB];
#pragma unroll
        for (int u = 0; u < TPB; ++u) {
            const uint32_t ti = tg * TPB + u, b = ti % NBUF;
            const uint32_t tn = TPB == 1 ? ti + NBUF - 1 : ti + TPB;  // the tile whose loads go out under tile ti's first block
            dt[u] = dma_prepare(tn % NBUF, s0 + (u64)tn * BT);
            const uint4 *bufc = bufs + (size_t)b * JT * NT;
            a0[u] = lds_addr(bufc + (size_t)r * KQ + h), a1[u] = lds_addr(bufc + (size_t)(32 + r) * KQ + h);
            an[u] = lds_addr(nrm + b * 64 + r);
            asm volatile("" ::"s"(dt[u].tb), "s"(dt[u].nb), "s"(dt[u].lb), "v"(a0[u]), "v"(a1[u]), "v"(an[u]));  // (in their registers here)
        }
        if constexpr (TPB == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * (JT + 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tick(0);
        if (!(p.debug & 8)) __builtin_amdgcn_s_barrier();
        tick(1);
#pragma unroll
        for (int u = 0; u < TPB; ++u) {
            const uint32_t ti = tg * TPB + u;
            const u64 t0 = s0 + (u64)ti * BT;
            const u64 tnext = s0 + (u64)(TPB == 1 ? ti + NBUF - 1 : ti + TPB) * BT;
            if (!dt[u].fast && !(p.debug & 4)) dma_slow(dt[u], tnext);
            u64 any[QH];
            float gm[QH][5];
            tick(2);
            // rows 0..31 of this tile, under them the verdict of rows 32..63 of the tile before (a whole tile: no column mask)
            // (the next loads: load j behind step j * KS / (JT + 1) + 1 of the first block)
            float bn0, bn1;  // issued behind the block's first fragment reads, first used a block later: LDS operations complete in order, the block's counted waits cover them
            block(acc0, S0, a0[u], S1, any, gm, [&](auto c) {
                constexpr int ks = decltype(c)::value;
                if constexpr (ks == 0) lds_read2_f32_nowait(an[u], bn0, bn1);
                static_for<0, JT + 1>([&](auto cj) {
                    constexpr int j = decltype(cj)::value;
                    if constexpr ((j * KS) / (JT + 1) + 1 == ks || (ks == KS - 1 && (j * KS) / (JT + 1) + 1 > ks)) {
                        if (dt[u].fast && !(p.debug & 4)) dma_one(dt[u], j);
                    }
                });
            });
            tick(3);
#pragma unroll
            for (int qh = 0; qh < QH; ++qh)
                if (nonzero(any[qh])) admit(acc1[qh], S1[qh], gm[qh], qh, ~0ull, t0 - BT, 1, bn1p);  // (the first tile: S1 is -1 everywhere)
            tick(4);
            // rows 32..63, under them the verdict of rows 0..31
            asm volatile("" : "+v"(bn0), "+v"(bn1));  // (usable from here: behind the first block's waits)
            block(acc1, S1, a1[u], S0, any, gm, [](auto) {});
            tick(3);
            {
                u64 cm = ~0ull;
                if (t0 + 32 > s1) cm = __ballot(t0 + r < s1);
#pragma unroll
                for (int qh = 0; qh < QH; ++qh)
                    if (nonzero(any[qh] & cm)) admit(acc0[qh], S0[qh], gm[qh], qh, cm, t0, 0, bn0);
            }
            tick(4);
            bn1p = bn1;
        }
        // The staged admissions leave when the list is half full -- decided for everybody by one thread, an iteration ahead, through LDS: a
        // flush waits for every load in flight (stores and loads leave the counter out of order), i.e. it drains the LDS-DMA pipeline;
        // every 32 tiles, as it was, that cost 170 cycles per tile (phase counters of the instrumented form).
        if ((tg & 15u) == 14u && tid == 0) lds_write_b32(lds_addr(sflag), lds_read_b32(lds_addr(scnt)) > (uint32_t)(HCAP / 2) ? 1u : 0u);
        if ((tg & 15u) == 15u && !(p.debug & 2)) {
            if (__builtin_amdgcn_readfirstlane(lds_read_b32(lds_addr(sflag)))) flush();
        }
        tick(5);
    }
    if constexpr (PROF) {
        if (lane == 0 && p.prof) {
#pragma unroll
            for (int i = 0; i < 6; ++i) atomicAdd(p.prof + i, (unsigned long long)pc[i]);
            atomicAdd(p.prof + 6, (unsigned long long)ntiles);
        }
    }
    {  // the last tile's second block (TPB = 2, an odd number of tiles: the loop went one tile past the slice -- nothing of it is anybody's)
        const u64 t0 = s0 + (u64)(ngroups * TPB - 1) * BT;
        const u64 cm = __ballot(t0 + 32 + r < s1);
#pragma unroll
        for (int qh = 0; qh < QH; ++qh) {
            float gmt[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) gmt[i] = __builtin_fmaxf(__builtin_fmaxf(S1[qh][3 * i], S1[qh][3 * i + 1]), S1[qh][3 * i + 2]);
            admit(acc1[qh], S1[qh], gmt, qh, cm, t0, 1, bn1p);
        }
        flush();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the look-ahead loads past the slice must land before the LDS goes away)
}

// per query (one workgroup): U = the k-th smallest hi of the admitted list -- a SELECTION, not a sort: four rounds of an 8-bit digit
// histogram over the order-preserving 32-bit image of hi (the first form ranked every entry against every other: 2 700^2 compares per
// query, 43 % of the generator's time).  The list itself stays as it is: the next pass starts an empty one (its sample contains this
// one's), and the re-rank takes every row with lo <= U wherever it sits.
__global__ void __launch_bounds__(256) knn_bf_select_kernel(const u64 *__restrict__ cand, uint32_t *cursor, uint32_t cap, uint32_t nq, uint32_t k,
                                                           float *tau, uint32_t *overflow) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_prefix, s_want;
    const uint32_t q = blockIdx.x;
    if (q >= nq) return;
    uint32_t m = cursor[q];
    if (m > cap) {
        if (threadIdx.x == 0) {
            atomicAdd(overflow, 1u);
            cursor[q] = cap;
        }
        m = cap;
    }
    if (m < k) return;  // (fewer rows than k so far: the bound stays what it was -- +inf in the first pass: everything is admitted)
    const u64 *list = cand + (u64)q * cap;
    if (threadIdx.x == 0) s_prefix = 0, s_want = k - 1;  // rank (0-based) of the wanted key among those that share the prefix
    for (int shift = 24; shift >= 0; shift -= 8) {
        hist[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix, want = s_want;
        const uint32_t himask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (uint32_t i = threadIdx.x; i < m; i += 256) {
            const uint32_t o = (uint32_t)(list[i] >> 32);
            if ((o & himask) == prefix) atomicAdd(&hist[(o >> shift) & 0xFFu], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) {  // one wave: the bin that holds rank `want` (exclusive prefix sums of four bins per lane)
            const uint32_t l = threadIdx.x;
            const uint32_t c0 = hist[4 * l], c1 = hist[4 * l + 1], c2 = hist[4 * l + 2], c3 = hist[4 * l + 3];
            uint32_t incl = c0 + c1 + c2 + c3;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = __shfl_up(incl, o, 64);
                if ((int)l >= o) incl += v;
            }
            uint32_t base = incl - (c0 + c1 + c2 + c3);
            const uint32_t cs[4] = {c0, c1, c2, c3};
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (want >= base && want < base + cs[b]) {  // exactly one (lane, b)
                    s_prefix = prefix | ((4 * l + b) << shift);
                    s_want = want - base;
                }
                base += cs[b];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) tau[q] = __builtin_fminf(tau[q], ord2f_b(s_prefix));  // (both bound the k-th distance: roar_knn_run_bf16, launches of a pass)
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
    for (uint32_t i = tid; i < m; i += 256) {  // (the k rows U comes from have hi <= U, hence lo <= U: at least k rows go on)
        const u64 key = list[i];
        const uint32_t id = (uint32_t)key;
        const float hi = ord2f_b((uint32_t)(key >> 32));
        const float lo = hi - 2.0002f * knn_eps(e_q, k_q, bnorm[id], kappa, IP ? 1 : 0);
        if (lo <= U) keys[atomicAdd(&s_m, 1u)] = id;
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

template <int KS, int JT, int NBUF, int HCAP, bool IP, int QH, int NW, int WPE>
hipError_t launch_pass_m(const KnnBfParams &p, uint32_t grid, hipStream_t st) {
    const size_t lds = (size_t)NBUF * JT * (64 * NW) * 16 + NBUF * 256 + (size_t)HCAP * 10 + 16;
    hipError_t e = hipFuncSetAttribute((const void *)knn_pass_bf16_kernel<KS, JT, NBUF, HCAP, IP, QH, NW, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if constexpr (KS == 13 && IP && QH == 1 && NW == 8) {
        if (p.prof) {
            e = hipFuncSetAttribute((const void *)knn_pass_bf16_kernel<KS, JT, NBUF, HCAP, IP, QH, NW, WPE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((knn_pass_bf16_kernel<KS, JT, NBUF, HCAP, IP, QH, NW, WPE, true>), dim3(grid), dim3(64 * NW), lds, st, p);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((knn_pass_bf16_kernel<KS, JT, NBUF, HCAP, IP, QH, NW, WPE>), dim3(grid), dim3(64 * NW), lds, st, p);
    return hipGetLastError();
}
template <int KS, int JT, int NBUF, int HCAP, int QH, int NW, int WPE = NW / 4>
hipError_t launch_pass(const KnnBfParams &p, uint32_t grid, hipStream_t st) {
    return p.ip ? launch_pass_m<KS, JT, NBUF, HCAP, true, QH, NW, WPE>(p, grid, st) : launch_pass_m<KS, JT, NBUF, HCAP, false, QH, NW, WPE>(p, grid, st);
}

// L2: the part of eps that goes with |x|^2 + |q|^2 (f32 summation of the exact pass and of the bound's own terms)
inline float knn_kappa(uint32_t dim) { return (float)((dim + 2.0 * (dim / 16.0 + 8.0)) * 1.5 / 16777216.0); }

}  // namespace

extern "C" {

// quads per row of the bf16 copies: the K steps of the kernel instantiation that serves this row length (8, 13 or 32 steps of 16: rows
// are zero beyond the dimension, the kernels walk every step), two quads per step + one: an odd number
uint32_t roar_knn_bf16_row_quads(uint32_t dim) {
    const uint32_t ks = (dim + 15u) / 16u;
    return (ks <= 8 ? 8u : ks <= 13 ? 13u : 32u) * 2u + 1u;
}

// f32 rows -> bf16 copy ([n_pad][KQ] quads; rows past n zero); sqnorm / max_sq_bits may be null.  *max_sq_bits must be zeroed by the caller.
// aug: 0 the query side (nothing behind the data), 1 / 2 the base rows of an inner-product / L2 set: their side of the verdict columns
// (knn_to_bf16_kernel; needs sqnorm).
int roar_knn_bf16_convert(const float *src, uint64_t ld, uint32_t dim, uint64_t n, uint64_t n_pad, void *dst, float *sqnorm, uint32_t *max_sq_bits,
                          int aug, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const uint32_t KQ = roar_knn_bf16_row_quads(dim);
    hipError_t e = hipSuccess;
    if (aug && !sqnorm) return (int)hipErrorInvalidValue;
    if (sqnorm || max_sq_bits) {
        hipLaunchKernelGGL(knn_sqnorm_kernel, dim3(2048), dim3(256), 0, st, src, (u64)ld, dim, (u64)n, sqnorm, max_sq_bits);
        if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(knn_to_bf16_kernel, dim3(4096), dim3(256), 0, st, src, (u64)ld, dim, (u64)n, (u64)n_pad, KQ, (uint4 *)dst, aug,
                       (const float *)sqnorm, knn_kappa(dim));
    return (int)hipGetLastError();
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
    const uint32_t QT = 256;  // queries per workgroup of the pass kernel
    const uint32_t nq_pad = (nq + QT - 1) / QT * QT;  // (the caller's qb holds nq rounded up to 512 rows)
    if ((e = (hipError_t)roar_knn_bf16_convert(q, ldq, dim, nq, nq_pad, qb, nullptr, nullptr, 0, stream)) != hipSuccess) return (int)e;
    // the bound (header): C for the two bf16 roundings (2^-8 each, Cauchy-Schwarz), the MFMA's f32 accumulation over K and the exact pass's own
    // f32 summation (distance.h's tree: D / 16 + 6 additions per partial sum)
    const double K = 16.0 * ksteps;
    const float C = (float)(2.02 / 256.0 + (K + dim / 16.0 + 8.0) * 1.25 / 16777216.0);
    const float kappa = knn_kappa(dim);
    hipLaunchKernelGGL(knn_slack_kernel, dim3(nq), dim3(64), 0, st, q, (u64)ldq, dim, nq, ip, C, kappa, (float *)ea, (float *)kq);
    {
        // tau = +inf
        if ((e = hipMemsetD32Async((hipDeviceptr_t)tau, (int)0x7F800000, nq, st)) != hipSuccess) return (int)e;
    }
    if ((e = hipMemsetAsync(overflow_dev, 0, 4, st)) != hipSuccess) return (int)e;
    uint64_t ratio = cap / (4ull * k);
    if (ratio < 4) ratio = 4;
    if (ratio > 64) ratio = 64;
#ifdef ROAR_KNN_EXPERIMENTS
    {
        static const int forced = [] {
            const char *e = getenv("ROAR_KNN_RATIO");
            return e ? atoi(e) : 0;
        }();
        if (forced >= 2) ratio = (uint64_t)forced;
    }
#endif
    int P = 1;
    uint64_t stride0 = 1;
    while ((n + stride0 - 1) / stride0 > cap / 2) {
        stride0 *= ratio;
        ++P;
    }
    uint64_t stride = stride0;
    // The timing experiments of scripts/knn_forms.sh (ROAR_KNN_DEBUG: nothing admitted, the lists come out WRONG; _RATIO, _SPLIT, _PROF, _TIME)
    // exist only in a library built with -DROAR_KNN_EXPERIMENTS (scripts/build_variant.sh): the shipped library reads none of them, so
    // a variable left exported cannot turn the builder's lists into a bad index.
#ifdef ROAR_KNN_EXPERIMENTS
    static const int debug = [] {
        const char *e = getenv("ROAR_KNN_DEBUG");
        const int v = e ? atoi(e) : 0;
        if (v) fprintf(stderr, "roargraph_amd: ROAR_KNN_DEBUG=%d -- a TIMING experiment of the kNN pass kernel (scripts/knn_forms.sh): the lists this process returns are WRONG\n", v);
        return v;
    }();
    static unsigned long long *prof_dev = [] {
        unsigned long long *d = nullptr;
        if (getenv("ROAR_KNN_PROF") && hipMalloc((void **)&d, 64) == hipSuccess) (void)hipMemset(d, 0, 64);
        return d;
    }();
#else
    constexpr int debug = 0;
    unsigned long long *const prof_dev = nullptr;
#endif
    for (int pass = 0; pass < P; ++pass, stride /= ratio) {
        if ((e = hipMemsetAsync(cursor, 0, (size_t)nq * 4, st)) != hipSuccess) return (int)e;
        KnnBfParams p;
        p.qb = (const uint4 *)qb, p.bb = (const uint4 *)bb, p.KQ = KQ, p.dim8 = dim / 8u, p.nq = nq;
        p.first = 0, p.stride = stride, p.count = (n + stride - 1) / stride;
        // slices: ~64 MB of bf16 rows (they sit in the Infinity Cache while the workgroups of one slice stream them), whole tiles
        uint64_t rps = std::max<uint64_t>(BT, ((64ull << 20) / (KQ * 16ull)) / BT * BT);
        p.rows_per_slice = (uint32_t)rps;
        p.U = (const float *)tau, p.ea = (const float *)ea, p.kq = (const float *)kq, p.kappa = kappa;
        p.bnorm = bnorm, p.cand = (u64 *)cand, p.cursor = (uint32_t *)cursor, p.cap = cap, p.ip = ip;
        p.prof = (pass == P - 1) ? prof_dev : nullptr, p.debug = (pass == P - 1) ? debug : 0;
        const uint64_t nqt = nq_pad / QT, nslices = (p.count + rps - 1) / rps;
        if (nqt * nslices > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
        // A pass over many slices goes in G launches with a selection after each: U, read by a workgroup when it starts, is then the k-th
        // smallest hi of everything admitted SO FAR in this pass as well (any k rows bound the k-th distance; the selection keeps the
        // smaller of the old and the new bound) -- after the first quarter of the rows it is the bound of a sample five times denser than
        // the previous pass's, and the rest of the pass admits accordingly fewer rows: T2I-10M, k = 100, last pass: 3 060 rows per query in one
        // launch, 1 061 in four, 799 in eight -- and an admitting block is the expensive thing in the pass kernel (round 5,
        // profiles/r05_knn_forms.txt: 0.340 -> 0.281 s per 65 536-query block with everything of the round; 512-d 0.241 -> 0.186).
#ifdef ROAR_KNN_EXPERIMENTS
        static const int split_env = [] {
            const char *e = getenv("ROAR_KNN_SPLIT");
            return e ? atoi(e) : 0;
        }();
        const bool timed = pass == P - 1 && getenv("ROAR_KNN_TIME");  // the last pass alone, by the device's clock
#else
        constexpr int split_env = 0;
        constexpr bool timed = false;
#endif
        const uint64_t G = std::min<uint64_t>(split_env > 0 ? (uint64_t)split_env : 8, nslices);
        static hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (timed) {
            if (!ev0) (void)hipEventCreate(&ev0), (void)hipEventCreate(&ev1);
            (void)hipEventRecord(ev0, st);
        }
        for (uint64_t g = 0; g < G; ++g) {
            const uint64_t sl0 = nslices * g / G, sl1 = nslices * (g + 1) / G;
            if (sl1 == sl0) continue;
            p.slice0 = (uint32_t)sl0;
            const uint32_t grid = (uint32_t)(nqt * (sl1 - sl0));
            // (K steps, DMA instructions per thread and tile for the longest row of the bucket, tile buffers, staged admissions; 32 queries per
            // wave, eight waves.  Tried in round 5 and dropped -- profiles/r05_knn_forms.txt: 64 queries per wave on four waves, one per SIMD;
            // four-wave workgroups, two per CU; a staging list per wave)
            if (ksteps <= 8) e = launch_pass<8, 3, 4, 2048, 1, 8>(p, grid, st);
            else if (ksteps <= 13) e = launch_pass<13, 4, 4, 2048, 1, 8>(p, grid, st);
            else e = launch_pass<32, 9, 2, 1024, 1, 8>(p, grid, st);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(knn_bf_select_kernel, dim3(nq), dim3(256), 0, st, (const u64 *)cand, (uint32_t *)cursor, cap, nq, k, (float *)tau,
                               (uint32_t *)overflow_dev);
            if ((e = hipGetLastError()) != hipSuccess) return (int)e;
        }
        if (timed) {
            (void)hipEventRecord(ev1, st);
            (void)hipEventSynchronize(ev1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev0, ev1);
            fprintf(stderr, "knn, last pass (%llu rows x %u queries, %llu launches + selections): %.2f ms\n", (unsigned long long)p.count, nq,
                    (unsigned long long)G, ms);
        }
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
    if (prof_dev) {
        unsigned long long h[8];
        if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h, prof_dev, 64, hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "knn pass kernel, last pass, cycles per wave and tile: wait for the tile's loads %.0f, barrier %.0f, dma issue %.0f, blocks (2) %.0f, admits %.0f, flush %.0f (waves x tiles %llu)\n",
                    (double)h[0] / h[6], (double)h[1] / h[6], (double)h[2] / h[6], (double)h[3] / h[6], (double)h[4] / h[6], (double)h[5] / h[6], h[6]);
            (void)hipMemset(prof_dev, 0, 64);
            std::vector<uint32_t> cur(nq);
            if (hipMemcpy(cur.data(), cursor, (size_t)nq * 4, hipMemcpyDeviceToHost) == hipSuccess) {
                double sum = 0;
                uint32_t mx = 0;
                for (uint32_t v : cur) sum += v, mx = std::max(mx, v);
                fprintf(stderr, "knn pass kernel, last pass: admitted rows per query: mean %.0f, max %u (cap %u, ratio %llu, passes %d)\n", sum / nq, mx, cap, (unsigned long long)ratio, P);
            }
        }
    }
    if (overflowed) {
        if ((e = hipMemcpyAsync(overflowed, overflow_dev, 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return (int)e;
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return (int)e;
    }
    return 0;
}
}
