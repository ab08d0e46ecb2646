// roar_wave.h -- wave64 building blocks shared by the gfx950 kernels of the RoarGraph beam search
// (roar_search_v2.h, roar_kernels.hip): cross-lane helpers, the order-preserving key, the reference's
// distance reduction tree, the sorted-pool merge and the visited-set filters.
//
//   reference (CPU)                                   here (per wave)
//   ------------------------------------------------  ---------------------------------------------
//   NeighborPriorityQueue  neighbor.h:138-223         sorted u64 keys  (ord(dist)<<32 | id<<1 | expanded)
//   VisitedList            visited_list_pool.h:8-29   LDS 4-way u16 tag table (+ u16 epoch spill list in HBM: exact; or forgetful), or u8 epoch tags in HBM (mode EXACT)
//   distance_->compare     distance.h:39-89,179-223   4 lanes per row, 64-B pieces, fmaf chains, quad-DPP reduction tree
//
// Distances are bit-identical to the reference's AVX-512 code as gcc builds it (-Ofast contracts
// every mul+add to vfmadd231ps): lane t of a 4-lane group holds SIMD lanes 4t..4t+3 of the
// reference's zmm accumulator and walks the row in the same 16-float steps, then the
// 16->8->4->hadd reduction tree is replayed with quad DPP moves.  Bit-identical distances make the
// (distance, id) order, hence every expansion and every returned id, identical to the CPU path.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "roar_device.h"

namespace {

typedef unsigned long long u64;

// LDS traffic between lanes of one wave: the LDS executes a wave's instructions in order, the
// fences only stop the compiler from reordering across the hand-off.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t lane_u32(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ u64 lane_u64(u64 v, int l) {
    uint32_t lo = lane_u32((uint32_t)v, l), hi = lane_u32((uint32_t)(v >> 32), l);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t lanes_below(u64 mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// float <-> order-preserving u32 (so that one u64 compare is Neighbor::operator<, neighbor.h:29-31)
__device__ __forceinline__ uint32_t f2ord(float f) {
    if (f == 0.0f) f = 0.0f;  // -0.0 == +0.0 for operator<
    uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    return __uint_as_float((o >> 31) ? (o ^ 0x80000000u) : ~o);
}
__device__ __forceinline__ u64 make_key(float d, uint32_t id) { return ((u64)f2ord(d) << 32) | ((u64)id << 1); }

// quad DPP: value of lane (l ^ 2) / (l ^ 1) inside each 4-lane group
__device__ __forceinline__ float quad_xor2(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}

template <bool IP>
__device__ __forceinline__ void acc16(float4 &acc, const float4 x, const float4 y) {
    if (IP) {  // distance.h:187  msum0 += mx * my  (fma)
        acc.x = __builtin_fmaf(x.x, y.x, acc.x);
        acc.y = __builtin_fmaf(x.y, y.y, acc.y);
        acc.z = __builtin_fmaf(x.z, y.z, acc.z);
        acc.w = __builtin_fmaf(x.w, y.w, acc.w);
    } else {  // distance.h:47-48  d = mx - my; msum0 += d * d
        float a = x.x - y.x, b = x.y - y.y, c = x.z - y.z, d = x.w - y.w;
        acc.x = __builtin_fmaf(a, a, acc.x);
        acc.y = __builtin_fmaf(b, b, acc.y);
        acc.z = __builtin_fmaf(c, c, acc.z);
        acc.w = __builtin_fmaf(d, d, acc.w);
    }
}

// One base row against the query held in LDS, by the 4 lanes of a group (t = lane & 3).
// NCH > 0: number of full 16-float steps known at compile time (all loads issued up front);
// NCH == 0: runtime dim, 8 steps in flight at a time.
template <bool IP, int NCH>
__device__ __forceinline__ float row_distance(const float *__restrict__ rowp, const float *lq, uint32_t dim, int t) {
    const float4 *row = reinterpret_cast<const float4 *>(rowp) + t;
    const float4 *q4 = reinterpret_cast<const float4 *>(lq) + t;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t done;
    if (NCH > 0) {
        float4 x[NCH > 0 ? NCH : 1];
#pragma unroll
        for (int i = 0; i < NCH; ++i) x[i] = row[4 * i];
#pragma unroll
        for (int i = 0; i < NCH; ++i) acc16<IP>(acc, x[i], q4[4 * i]);
        done = NCH;
    } else {
        // Run-time row length: the full 16-float steps eight at a time (every load of a group before its first FMA), then the last
        // 1..7 steps as 4 + 2 + 1 -- each piece straight-line loads, then its FMAs in row order (round 4 walked them one load + one FMA
        // at a time: the six steps of a 96-d row were six dependent round trips; a switch over the count with one register array was
        // tried first and is worse than either: the array goes to scratch)
        const uint32_t nfull = dim >> 4;
        uint32_t c = 0;
        for (; c + 8 <= nfull; c += 8) {
            float4 x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = row[4 * (c + i)];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc16<IP>(acc, x[i], q4[4 * (c + i)]);
        }
        const uint32_t rem = nfull - c;  // (the same in every lane)
        if (rem & 4u) {
            const float4 x0 = row[4 * c], x1 = row[4 * (c + 1)], x2 = row[4 * (c + 2)], x3 = row[4 * (c + 3)];
            acc16<IP>(acc, x0, q4[4 * c]), acc16<IP>(acc, x1, q4[4 * (c + 1)]), acc16<IP>(acc, x2, q4[4 * (c + 2)]), acc16<IP>(acc, x3, q4[4 * (c + 3)]);
            c += 4;
        }
        if (rem & 2u) {
            const float4 x0 = row[4 * c], x1 = row[4 * (c + 1)];
            acc16<IP>(acc, x0, q4[4 * c]), acc16<IP>(acc, x1, q4[4 * (c + 1)]);
            c += 2;
        }
        if (rem & 1u) acc16<IP>(acc, row[4 * c], q4[4 * c]);
        done = nfull;
    }
    // 16 -> 8 (distance.h:52-53 / :191-192): lanes t and t^2 now both hold s1[4*(t&1) .. +3]
    float4 s1;
    s1.x = quad_xor2(acc.x) + acc.x;
    s1.y = quad_xor2(acc.y) + acc.y;
    s1.z = quad_xor2(acc.z) + acc.z;
    s1.w = quad_xor2(acc.w) + acc.w;
    if (dim & 8) {  // one 8-float step (distance.h:55-63 / :194-201)
        const float4 *rt = reinterpret_cast<const float4 *>(rowp) + 4 * done + (t & 1);
        const float4 *qt = reinterpret_cast<const float4 *>(lq) + 4 * done + (t & 1);
        acc16<IP>(s1, *rt, *qt);
    }
    // 8 -> 4 (:65-66 / :203-204), then the two hadd (:85-86 / :221-222)
    float a = quad_xor1(s1.x) + s1.x;
    float b = quad_xor1(s1.y) + s1.y;
    float c2 = quad_xor1(s1.z) + s1.z;
    float d = quad_xor1(s1.w) + s1.w;
    float r = (a + b) + (c2 + d);
    return IP ? -r : r;  // distance.h:223
}

// Tail of row_distance shared by the one-row and two-row forms: 16 -> 8 -> (8-float step) -> 4 -> hadd.
template <bool IP>
__device__ __forceinline__ float reduce_row(float4 acc, const float *__restrict__ rowp, const float *lq, uint32_t dim,
                                            uint32_t done, int t) {
    float4 s1;
    s1.x = quad_xor2(acc.x) + acc.x;
    s1.y = quad_xor2(acc.y) + acc.y;
    s1.z = quad_xor2(acc.z) + acc.z;
    s1.w = quad_xor2(acc.w) + acc.w;
    if (dim & 8) {
        const float4 *rt = reinterpret_cast<const float4 *>(rowp) + 4 * done + (t & 1);
        const float4 *qt = reinterpret_cast<const float4 *>(lq) + 4 * done + (t & 1);
        acc16<IP>(s1, *rt, *qt);
    }
    float a = quad_xor1(s1.x) + s1.x;
    float b = quad_xor1(s1.y) + s1.y;
    float c2 = quad_xor1(s1.z) + s1.z;
    float d = quad_xor1(s1.w) + s1.w;
    float r = (a + b) + (c2 + d);
    return IP ? -r : r;
}

// Score cid[0..m) against the query; ck[i] = key(distance, id).  16 rows per pass; with a
// compile-time row length two passes (32 rows) are in flight at once: every load of both rows is
// issued before the first FMA, the query fragments are read from LDS once for the pair.
template <bool IP, int NCH>
__device__ __forceinline__ void score_candidates(const float *__restrict__ base, uint64_t ld, uint32_t dim,
                                                 const float *lq, const uint32_t *cid, u64 *ck, uint32_t m,
                                                 int lane) {
    const int t = lane & 3, g = lane >> 2;
    uint32_t p0 = 0;
    if (NCH > 0) {
        for (; p0 + 16 < m; p0 += 32) {
            const uint32_t c0 = p0 + g, c1 = p0 + 16 + g;  // c0 < m always here
            const bool has1 = c1 < m;
            const uint32_t id0 = cid[c0];
            const uint32_t id1 = has1 ? cid[c1] : id0;
            const float *r0 = base + (u64)id0 * ld, *r1 = base + (u64)id1 * ld;
            const float4 *row0 = reinterpret_cast<const float4 *>(r0) + t;
            const float4 *row1 = reinterpret_cast<const float4 *>(r1) + t;
            const float4 *q4 = reinterpret_cast<const float4 *>(lq) + t;
            float4 x0[NCH > 0 ? NCH : 1], x1[NCH > 0 ? NCH : 1];
#pragma unroll
            for (int i = 0; i < NCH; ++i) x0[i] = row0[4 * i];
            if (has1) {
#pragma unroll
                for (int i = 0; i < NCH; ++i) x1[i] = row1[4 * i];
            }
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const float4 y = q4[4 * i];
                acc16<IP>(a0, x0[i], y);
                if (has1) acc16<IP>(a1, x1[i], y);
            }
            const float d0 = reduce_row<IP>(a0, r0, lq, dim, NCH, t);
            if (t == 0) ck[c0] = make_key(d0, id0);
            if (has1) {
                const float d1 = reduce_row<IP>(a1, r1, lq, dim, NCH, t);
                if (t == 0) ck[c1] = make_key(d1, id1);
            }
        }
    }
    for (; p0 < m; p0 += 16) {
        const uint32_t c = p0 + g;
        if (c < m) {  // whole 4-lane groups are active or idle together
            const uint32_t id = cid[c];
            const float d = row_distance<IP, NCH>(base + (u64)id * ld, lq, dim, t);
            if (t == 0) ck[c] = make_key(d, id);
        }
    }
}

// "Never forget a node that is in the pool" (forgetful on-chip tag table, LDS pool): bit 15 of a u16 tag says its node is a pool member,
// and the filter's victim choice skips such ways.  A revisited pool member always passes the 8-bit screen (its key is under the pool's
// threshold), so forgetting one costs an 8-bit row AND an f32 row and a merge candidate that the duplicate test throws away; a
// forgotten node outside the pool costs its 8-bit row.  The merge sets the bit for what it inserts and clears it for what falls off
// the pool's end.  Needs every tag below 2^15 (the launch plan checks: base points / sets < 32 767); `vis` == nullptr: off.
struct TagProtect {
    uint16_t *vis = nullptr;
    uint32_t sets_mask = 0, sets_log2 = 0;
    uint32_t mult = 1;  // sets = mult << sets_log2 (1, 3 or 5)
};
__device__ __forceinline__ void tag_protect(const TagProtect &tp, uint32_t id, bool on) {
    uint32_t set = id & tp.sets_mask, hi = id >> tp.sets_log2;
    if (tp.mult == 3u) {
        const uint32_t q3 = __umulhi(hi, 0xAAAAAAABu) >> 1;
        set |= (hi - 3u * q3) << tp.sets_log2;
        hi = q3;
    } else if (tp.mult == 5u) {
        const uint32_t q5 = __umulhi(hi, 0xCCCCCCCDu) >> 2;
        set |= (hi - 5u * q5) << tp.sets_log2;
        hi = q5;
    }
    const uint32_t tag = hi + 1u;
    uint32_t *w = reinterpret_cast<uint32_t *>(tp.vis + 4 * set);
    const uint2 ww = *reinterpret_cast<const uint2 *>(w);
    const bool h0 = (ww.x & 0x7FFFu) == tag, h1 = ((ww.x >> 16) & 0x7FFFu) == tag, h2 = (ww.y & 0x7FFFu) == tag,
               h3 = ((ww.y >> 16) & 0x7FFFu) == tag;
    if (!(h0 | h1 | h2 | h3)) return;  // (its tag went to another neighbour of the same hop: it stays forgotten)
    uint32_t *word = w + ((h0 | h1) ? 0 : 1);
    const uint32_t bit = (h0 | h2) ? 0x8000u : 0x80000000u;
    if (on) atomicOr(word, bit);
    else atomicAnd(word, ~bit);
}

// NeighborPriorityQueue::insert (neighbor.h:150-183) for a batch of m <= 64 candidates at once.
// Sequential inserts keep the top-L of everything seen; so does this merge, entry for entry.
// Duplicate (distance, id) pairs are dropped exactly as the binary-search probe does (:161-162).
//
// A hop of a long pool spends a third of its time here (T2I-10M, L_pq = 1 000: 6 000 of 17 500 cycles, profiles/r04_long_pools_phases.txt),
// and that time is INSTRUCTIONS, not LDS latency: a wave-wide instruction costs its four cycles whether it moves one survivor or
// sixty-four.  So: (1) the lower bounds start from 64 splitters read once (every (psize / 64)-th entry; a ballot per survivor names its
// segment) and finish with log2(psize / 64) per-lane probes instead of log2(psize); (2) the pool is shifted SEGMENT by segment -- the
// entries between two survivors' landing points all move by the same amount, so a 64-entry chunk is a read, an add and a write
// (the first form recomputed every entry's shift from ballots over the survivors: ~75 instructions per chunk).
// front / ub (optional): the caller's "everything below `front` is expanded, except at most `ub` entries" (SearchRoarGraph kernel, select):
// survivors that land below the front move it up and are counted.
__device__ __forceinline__ void merge_candidates(u64 *pool, const u64 *ck, uint32_t m_, uint32_t L_, uint32_t &psize,
                                                 uint32_t &cur, int lane, const TagProtect tp = TagProtect(), uint32_t *front = nullptr,
                                                 uint32_t *ub = nullptr) {
    const u64 FLAG = 1ull;
    // Everything that steers a loop or a branch here is the same in every lane; said so explicitly (readfirstlane), it lives in scalar
    // registers and the loops are scalar loops -- left to the compiler's divergence analysis, pool sizes that passed through LDS or
    // through per-lane arithmetic were vector values and every `if` below a save / mask / restore of EXEC.
    const uint32_t m = uni(m_), L = uni(L_), P = uni(psize), Pm1 = P ? P - 1u : 0u;
    bool valid = (uint32_t)lane < m;
    u64 key = ck[(uint32_t)lane < m ? lane : 0];
    key = valid ? key : ~0ull;
    if (P == L) {  // neighbor.h:151-153
        const u64 last = pool[L - 1] & ~FLAG;
        valid = valid & (key < last);
    }
    const u64 vm0 = __ballot(valid);
    if (vm0 == 0) return;
    // lower bound of key among the (flag-masked) pool keys
    uint32_t lo = 0, len = P, span = P;  // (span: the longest range any lane still has to search)
    if (P > 128u) {
        // segment l = entries [l step, (l + 1) step) cut at P; its splitter = its last entry (segments past the end repeat the pool's last)
        const uint32_t step = (P + 63u) >> 6;
        const uint32_t e_l = ((uint32_t)(lane + 1) * step < P ? (uint32_t)(lane + 1) * step : P) - 1u;
        const u64 spl = pool[e_l] & ~FLAG;
        uint32_t seg = 0;
        for (u64 it = vm0; it; it &= it - 1) {
            const int j = __ffsll((long long)it) - 1;
            const u64 kj = lane_u64(key, j);
            const uint32_t below = (uint32_t)__popcll(__ballot(spl < kj));  // segments that lie wholly below kj
            seg = (lane == j) ? below : seg;
        }
        lo = seg * step < P ? seg * step : P;
        len = P - lo < step ? P - lo : step;
        span = step;
    }
    {
        const int iters = span ? 32 - __builtin_clz(span) : 0;
        for (int it = 0; it < iters; ++it) {  // (branch-free: a lane that is done probes its own `lo` again, cut to the pool, and ignores it)
            const uint32_t half = len >> 1, mid = lo + half;
            const u64 e = pool[mid < P ? mid : Pm1] & ~FLAG;
            const bool less = (len > 0u) & (e < key);
            lo = less ? mid + 1u : lo;
            len = less ? len - half - 1u : half;
        }
    }
    {
        const u64 e = pool[lo < P ? lo : Pm1] & ~FLAG;
        valid = valid & !((lo < P) & (e == key));  // already in the pool
    }
    // rank among the surviving candidates; a later lane with an equal key is a duplicate
    uint32_t r = 0;
    for (u64 it = __ballot(valid); it; it &= it - 1) {
        const int j = __ffsll((long long)it) - 1;
        if (!((__ballot(valid) >> j) & 1ull)) continue;  // (uniform: j was a copy of an earlier key)
        const u64 kj = lane_u64(key, j);
        const bool other = valid & (lane != j);
        r += (other & (kj < key)) ? 1u : 0u;
        valid = valid & !(other & (kj == key) & (j < lane));
    }
    const u64 vm = __ballot(valid);
    if (vm == 0) return;
    const uint32_t s = (uint32_t)__popcll(vm);
    if (front) {
        const uint32_t below = (uint32_t)__popcll(__ballot(valid & (lo < *front)));
        const uint32_t nf = *front + below;
        *front = nf < L ? nf : L;
        *ub += below;
    }
    // Shift, top segment first: the entries from the rank-t survivor's lower bound up to the next survivor's (the pool's end for the
    // last one) move up by t + 1.  64-entry chunks, top chunk first, up to four chunks read before the first is written back: the LDS
    // executes a wave's instructions in order, a lower chunk's reads never overlap an upper chunk's (higher) write range, and a lower
    // chunk's writes come after the upper chunk's reads in program order -- so no wait is needed in between.
    uint32_t hi = P;
    for (uint32_t t = s; t-- > 0;) {
        const int l = __ffsll((long long)__ballot(valid & (r == t))) - 1;
        const uint32_t seg_lo = lane_u32(lo, l), sh = t + 1u;
        for (uint32_t top = hi; top > seg_lo;) {
            const uint32_t left = top - seg_lo;
            if (left >= 256u && top + sh <= L) {  // four whole chunks that stay inside the pool: no lane has anything to decide
                u64 e[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) e[u] = pool[top - 64u * (uint32_t)(u + 1) + (uint32_t)lane];
#pragma unroll
                for (int u = 0; u < 4; ++u) pool[top - 64u * (uint32_t)(u + 1) + (uint32_t)lane + sh] = e[u];
                top -= 256u;
            } else if (left >= 64u && top + sh <= L) {
                const u64 e = pool[top - 64u + (uint32_t)lane];
                pool[top - 64u + (uint32_t)lane + sh] = e;
                top -= 64u;
            } else {  // the chunk at the pool's end when it is full (entries pushed past L_pq are dropped) and the segment's last, partial one
                const uint32_t lo_c = left > 64u ? top - 64u : seg_lo;
                const uint32_t kidx = lo_c + (uint32_t)lane;
                const bool a = kidx < top;
                const u64 e = pool[a ? kidx : lo_c];
                if (a & (kidx + sh < L)) pool[kidx + sh] = e;
                if (tp.vis) {
                    if (a & !(kidx + sh < L)) tag_protect(tp, (uint32_t)e >> 1, false);  // fell off the pool's end
                }
                top = lo_c;
            }
        }
        hi = seg_lo;
    }
    const uint32_t pmin = hi;  // the rank-0 survivor's lower bound
    wave_sync();
    if (valid && lo + r < L) {
        pool[lo + r] = key;
        if (tp.vis) tag_protect(tp, (uint32_t)key >> 1, true);
    }
    psize = (P + s < L) ? P + s : L;
    const uint32_t c = uni(cur);
    cur = c < pmin ? c : pmin;  // neighbor.h:180-182
    wave_sync();
}

// NeighborPriorityQueue::insert (neighbor.h:150-183) for a batch of cnt <= 64 candidates when the pool (L_pq <= 64)
// lives in a register pair of the wave: lane i holds entry i (`pk`; lanes >= psize hold ~0), candidates come from
// ck[0..cnt).  One pass over the candidates a full pool does not reject (:151-153): a ballot of "pool entry < key"
// is the key's lower bound (its popcount) -- no binary search; a ballot of "pool entry == key" is the reference's
// duplicate test (:161-162); the same broadcast key ranks the other candidates and tells every pool entry whether it
// moves up.  Entry i ends at i + #{candidates landing at or before it}, candidate c at lower_bound(c) + rank(c);
// the scatter goes through `scratch` (the LDS pool region) and is read back one entry per lane.
__device__ __forceinline__ void merge_regpool(u64 &pk, uint32_t &psize, const u64 *ck, uint32_t cnt, uint32_t L, u64 *scratch,
                                              int lane) {
    const u64 FLAG = 1ull;
    const u64 ci = (uint32_t)lane < cnt ? ck[lane] : ~0ull;
    bool ok = (uint32_t)lane < cnt;
    if (psize == L) ok = ok && ci < (lane_u64(pk, (int)L - 1) & ~FLAG);
    const u64 todo = __ballot(ok);
    if (todo == 0) return;
    // lanes past the pool hold ~0: their masked key is above every candidate's (ids are below 2^31 - 1), so the
    // compares below need no "is a pool entry" term.  Everything indexed by j is wave-uniform and stays in SGPRs.
    const u64 pkm = pk & ~FLAG;
    u64 keep = todo;
    uint32_t r = 0, shift = 0, lo_mine = 0;
    // (two candidates per step -- two independent chains of readlane / ballot / popcount interleaved -- measured in round 4: merge 2 435
    // against 2 476 cycles per hop for a lone query, 210.3 against 212.6 us: the chain is not what the merge's time is; dropped)
    for (u64 it = todo; it; it &= it - 1) {
        const int j = __ffsll((long long)it) - 1;
        const u64 kj = lane_u64(ci, j);
        const u64 rep = __ballot(ci == kj) & (0ull - (2ull << j));  // the same key again further up (rows are deduplicated: never)
        keep &= ~rep;
        it &= ~rep;
        const bool dup = __ballot(pkm == kj) != 0;  // already in the pool (neighbor.h:161-162)
        const uint32_t lo_j = (uint32_t)__popcll(__ballot(pkm < kj));
        keep &= dup ? ~(1ull << j) : ~0ull;
        const u64 kje = dup ? ~0ull : kj;             // a dropped candidate ranks nobody ...
        const uint32_t loe = dup ? 0xFFFFFFFFu : lo_j;  // ... and moves nobody
        lo_mine = (lane == j) ? lo_j : lo_mine;
        r += (kje < ci) ? 1u : 0u;
        shift += (loe <= (uint32_t)lane) ? 1u : 0u;
    }
    const bool valid = (keep >> lane) & 1ull;
    const uint32_t s = (uint32_t)__popcll(keep);
    if ((uint32_t)lane < psize && (uint32_t)lane + shift < L) scratch[(uint32_t)lane + shift] = pk;
    if (valid && lo_mine + r < L) scratch[lo_mine + r] = ci;
    psize = (psize + s < L) ? psize + s : L;
    wave_sync();
    pk = (uint32_t)lane < psize ? scratch[lane] : ~0ull;
    wave_sync();
}

constexpr int VIS_EXACT = 0;
constexpr int VIS_LDS = 1;

// visited test-and-set for one adjacency chunk (lane = one neighbour); returns "not visited before".
template <int MODE>
__device__ __forceinline__ bool visited_filter(bool isnbr, uint32_t nb, uint16_t *vis, uint32_t sets_mask,
                                               uint32_t sets_log2, uint8_t *tags, uint32_t epoch) {
    bool fresh;
    if (MODE == VIS_LDS) {
        // 4-way set of u16 tags; (set, tag) <-> id is a bijection, so a hit is never false.
        // The table may forget (way overwritten): the node is then scored again and the
        // duplicate key is dropped by the merge, results are unchanged.
        const uint32_t set = nb & sets_mask;
        const uint32_t tag = (nb >> sets_log2) + 1u;
        const uint2 ww = *reinterpret_cast<const uint2 *>(vis + 4 * set);
        const uint32_t t0 = ww.x & 0xFFFFu, t1 = ww.x >> 16, t2 = ww.y & 0xFFFFu, t3 = ww.y >> 16;
        const bool hit = (t0 == tag) | (t1 == tag) | (t2 == tag) | (t3 == tag);
        fresh = isnbr && !hit;
        // Insert.  Lanes of this instruction that share a set race for the same empty
        // way, so a write into an empty way is verified on the next round and retried
        // by the loser; a full set overwrites a victim way without verification (that is
        // the only place the table forgets).
        bool pending = fresh;
        uint32_t a0 = t0, a1 = t1, a2 = t2, a3 = t3;
        while (__ballot(pending)) {
            if (pending) {
                const bool present = (a0 == tag) | (a1 == tag) | (a2 == tag) | (a3 == tag);
                const bool full = (a0 != 0) & (a1 != 0) & (a2 != 0) & (a3 != 0);
                if (present) {
                    pending = false;
                } else {
                    const uint32_t way = a0 == 0 ? 0u : a1 == 0 ? 1u : a2 == 0 ? 2u : a3 == 0 ? 3u : (tag & 3u);
                    vis[4 * set + way] = (uint16_t)tag;
                    if (full) pending = false;
                }
            }
            wave_sync();
            if (pending) {
                const uint2 w2 = *reinterpret_cast<const uint2 *>(vis + 4 * set);
                a0 = w2.x & 0xFFFFu, a1 = w2.x >> 16, a2 = w2.y & 0xFFFFu, a3 = w2.y >> 16;
            }
            wave_sync();
        }
    } else {  // visited_array[nbr] != tag ... = tag  (index_bipartite.cpp:2378,2385)
        uint32_t tg = 0;
        if (isnbr) tg = __builtin_nontemporal_load(tags + nb);
        fresh = isnbr && tg != epoch;
        if (fresh) __builtin_nontemporal_store((uint8_t)epoch, tags + nb);
    }
    return fresh;
}


}  // namespace
