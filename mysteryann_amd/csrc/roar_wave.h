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
// A hop of a long pool spends a third of its time here, and a launch of long pools is bound by its waves' own chains, not by the memory
// system (round 6, profiles/r06_hop_pad_probe.txt: 4 096 idle cycles more per hop cost a 10 000-query launch at L_pq = 1 000 20 % --
// 0.7 of what a purely latency-bound launch would pay).  That time is INSTRUCTIONS and dependent LDS round trips, not LDS bandwidth: a
// wave-wide instruction costs its four cycles whether it moves one survivor or sixty-four.  Three steps, each a short chain:
//   (1) lower bounds: 64 splitters read once (every (P / 64)-th entry; a ballot per survivor names its segment), then the segments
//       THEMSELVES -- up to four survivors' segments side by side in one wave-wide read, a ballot of "entry < key" per group is the offset
//       in the segment and a ballot of "entry == key" the reference's duplicate test: two dependent reads where rounds 4-5 made seven
//       (five binary-search probes and the duplicate probe behind the splitters);
//   (2) ranks among the survivors: one compare per survivor;
//   (3) the shift, CHUNK by chunk from the top: 64-entry chunks on absolute 64-entry boundaries, eight read before the first is written
//       back (the LDS executes a wave's instructions in order; a chunk's writes go to higher addresses than every chunk read after it).
//       A chunk that no survivor lands in moves by one uniform amount; in a chunk with landing points a lane subtracts one per landing
//       point above its entry.  (Rounds 4-5 shifted segment by segment between landing points: a read-wait-write per survivor for the
//       segment's partial chunk on top of the whole chunks' -- 2 440 of the merge's 5 440 cycles per hop at L_pq = 1 000,
//       profiles/r06_merge_phases.txt.)
// front / ub (optional): the caller's "everything below `front` is expanded, except at most `ub` entries" (SearchRoarGraph kernel, select):
// survivors that land below the front move it up and are counted.
__device__ __forceinline__ void merge_candidates(u64 *pool, const u64 *ck, uint32_t m_, uint32_t L_, uint32_t &psize,
                                                 uint32_t &cur, int lane, const TagProtect tp = TagProtect(), uint32_t *front = nullptr,
                                                 uint32_t *ub = nullptr, long long *mprof = nullptr) {
    const u64 FLAG = 1ull;
    // (mprof, instrumented instantiation only: cycles of {the lower bounds, ranks and duplicates, the shift}; the rest of the caller's
    // merge time is the keys' own stores and the tag protection)
    long long mt0 = mprof ? (long long)__builtin_readcyclecounter() : 0;
    auto lap = [&](int i) {
        if (mprof) {
            const long long t = (long long)__builtin_readcyclecounter();
            mprof[i] += t - mt0;
            mt0 = t;
        }
    };
    // Everything that steers a loop or a branch here is the same in every lane; said so explicitly (readfirstlane), it lives in scalar
    // registers and the loops are scalar loops -- left to the compiler's divergence analysis, pool sizes that passed through LDS or
    // through per-lane arithmetic were vector values and every `if` below a save / mask / restore of EXEC.
    const uint32_t m = uni(m_), L = uni(L_), P = uni(psize), Pm1 = P ? P - 1u : 0u;
    bool valid = (uint32_t)lane < m;
    // (the three reads nothing here depends on go out together: the candidates' keys, the pool's last entry -- read whether the pool is
    // full or not, used if it is -- and the splitters: one LDS round trip in front of the lower bounds instead of three)
    // segment l = entries [l step, (l + 1) step) cut at P; its splitter = its last entry (segments past the end repeat the pool's last)
    const uint32_t step = (P + 63u) >> 6;
    const uint32_t e_l = P ? ((uint32_t)(lane + 1) * step < P ? (uint32_t)(lane + 1) * step : P) - 1u : 0u;
    u64 key = ck[(uint32_t)lane < m ? lane : 0];
    const u64 last = pool[L - 1] & ~FLAG;
    const u64 spl = pool[e_l] & ~FLAG;
    key = valid ? key : ~0ull;
    if (P == L) valid = valid & (key < last);  // neighbor.h:151-153
    const u64 vm0 = __ballot(valid);
    if (vm0 == 0) return;
    // ---- (1) lower bound of key among the (flag-masked) pool keys, and "already in the pool"
    uint32_t lo = 0, len = P, span = P;  // (span: the longest range any lane still has to search)
    if (P > 128u) {
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
    lap(0);
    // ---- (2) rank among the surviving candidates; a later lane with an equal key is a duplicate (never among the neighbours of one
    // row -- their ids are distinct -- but among the bipartite variant's random entry points)
    uint32_t r = 0;
    {
        bool dupc = false;
        for (u64 it = __ballot(valid); it; it &= it - 1) {
            const int j = __ffsll((long long)it) - 1;
            const u64 kj = lane_u64(key, j);
            r += (kj < key) ? 1u : 0u;
            dupc = dupc | ((kj == key) & (j < lane));
        }
        if (__ballot(valid & dupc)) {  // (the copies go; ranks once more over what is left)
            valid = valid & !dupc;
            r = 0;
            for (u64 it = __ballot(valid); it; it &= it - 1) {
                const int j = __ffsll((long long)it) - 1;
                r += (lane_u64(key, j) < key) ? 1u : 0u;
            }
        }
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
    lap(1);
    // ---- (3) shift.  The entry at index i moves up by the number of survivors whose lower bound is <= i; the rank-t survivor goes to its
    // lower bound + t.  Chunks of 64 entries on absolute boundaries, top chunk first.  What a chunk costs is its INSTRUCTIONS (a first form
    // of this walk ran one general body per chunk -- forty instructions each -- and took as long as the segment-wise shift it replaced:
    // 2 400 cycles per hop at L_pq = 1 000), so: a chunk that no survivor lands in and that stays inside the pool is a read and a write
    // with one uniform amount, four such chunks are read before the first is written back; only the chunks with a landing point, and the
    // top chunk(s) when entries fall off the pool's end, run the general body (a lane subtracts one per landing point above its entry).
    auto landing = [&](uint32_t t) -> uint32_t {  // lower bound of the rank-t survivor (nondecreasing in t)
        const int l = __ffsll((long long)__ballot(valid & (r == t))) - 1;
        return lane_u32(lo, l);
    };
    const uint32_t pmin = landing(0);
    if (P > pmin) {
        uint32_t t_rem = s;                            // survivors landing below the end of the chunk in hand (the others are taken off)
        uint32_t lp = s > 1u ? landing(s - 1u) : pmin;  // lower bound of survivor t_rem - 1
        int c = (int)((P - 1u) >> 6);
        while (t_rem > 0u) {
            const uint32_t base = (uint32_t)c << 6;
            if (lp >= base || base + 64u + t_rem > L || base + 64u > P) {
                // general body: landing points inside, or entries that fall off the pool's end, or the pool's partial top chunk
                const uint32_t i = base + (uint32_t)lane;
                const u64 e = pool[i < P ? i : Pm1];
                uint32_t shl = t_rem;
                while (t_rem > 0u && lp >= base) {  // entries below a landing point do not move for it
                    shl -= (i < lp) ? 1u : 0u;
                    --t_rem;
                    if (t_rem > 0u) lp = landing(t_rem - 1u);
                }
                const bool a = (i < P) & (shl > 0u);
                const uint32_t dest = i + shl;
                if (a & (dest < L)) pool[dest] = e;
                if (tp.vis) {
                    if (a & !(dest < L)) tag_protect(tp, (uint32_t)e >> 1, false);  // fell off the pool's end
                }
                --c;
            } else {
                // whole chunks down to (not including) the chunk of the next landing point, all by t_rem
                uint32_t n = (uint32_t)c - (lp >> 6);
                const uint32_t amount = t_rem;
                u64 *src = pool + base + (uint32_t)lane;
                for (; n >= 4u; n -= 4u, src -= 256) {
                    u64 e4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) e4[u] = src[-64 * u];
#pragma unroll
                    for (int u = 0; u < 4; ++u) src[-64 * u + (int)amount] = e4[u];
                }
                for (; n > 0u; --n, src -= 64) {
                    const u64 e1 = src[0];
                    src[amount] = e1;
                }
                c = (int)(lp >> 6);
            }
        }
    }
    wave_sync();
    lap(2);
    if (valid && lo + r < L) {
        pool[lo + r] = key;
        if (tp.vis) tag_protect(tp, (uint32_t)key >> 1, true);
    }
    psize = (P + s < L) ? P + s : L;
    const uint32_t c0 = uni(cur);
    cur = c0 < pmin ? c0 : pmin;  // neighbor.h:180-182
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
