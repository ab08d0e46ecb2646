// roar_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the RoarGraph beam search.
//
// One wavefront owns one query at a time (single-wave workgroups, persistent, dynamic dequeue),
// so every piece of per-query state of the reference lives on-chip:
//
//   reference (CPU)                                   here (per wave)
//   ------------------------------------------------  ---------------------------------------------
//   NeighborPriorityQueue  neighbor.h:138-223         sorted u64 keys in LDS  (ord(dist)<<32 | id<<1 | expanded)
//   VisitedList            visited_list_pool.h:8-29   LDS 4-way u16 tag table (mode LDS) or u8 epoch tags in HBM (mode EXACT)
//   projection_graph_[v]   index_bipartite.h:150      one coalesced read of a fixed-stride adjacency row
//   distance_->compare     distance.h:39-89,179-223   16 rows per pass, 4 lanes per row, 64-B pieces, fmaf chains
//   SearchRoarGraph loop   index_bipartite.cpp:2356-2405  select -> adjacency -> visited filter -> score -> merge
//
// Distances are bit-identical to the reference's AVX-512 code as gcc builds it (-Ofast contracts
// every mul+add to vfmadd231ps): lane t of a 4-lane group holds SIMD lanes 4t..4t+3 of the
// reference's zmm accumulator and walks the row in the same 16-float steps, then the
// 16->8->4->hadd reduction tree is replayed with quad DPP moves.  Bit-identical distances make the
// (distance, id) order, hence every expansion and every returned id, identical to the CPU path.
//
// The kernel is HBM-gather bound (0.5 flop/byte): nothing here goes through MFMA on purpose.
#include <hip/hip_runtime.h>
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
        const uint32_t nfull = dim >> 4;
        uint32_t c = 0;
        for (; c + 8 <= nfull; c += 8) {
            float4 x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = row[4 * (c + i)];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc16<IP>(acc, x[i], q4[4 * (c + i)]);
        }
        for (; c < nfull; ++c) acc16<IP>(acc, row[4 * c], q4[4 * c]);
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

// NeighborPriorityQueue::insert (neighbor.h:150-183) for a batch of m <= 64 candidates at once.
// Sequential inserts keep the top-L of everything seen; so does this merge, entry for entry.
// Duplicate (distance, id) pairs are dropped exactly as the binary-search probe does (:161-162).
__device__ __forceinline__ void merge_candidates(u64 *pool, const u64 *ck, uint32_t m, uint32_t L, uint32_t &psize,
                                                 uint32_t &cur, int lane) {
    const u64 FLAG = 1ull;
    bool valid = (uint32_t)lane < m;
    u64 key = valid ? ck[lane] : ~0ull;
    if (psize == L) {  // neighbor.h:151-153
        const u64 last = pool[L - 1] & ~FLAG;
        valid = valid && key < last;
    }
    if (__ballot(valid) == 0) return;
    // lower bound of key among the (flag-masked) pool keys
    uint32_t lo = 0;
    {
        uint32_t len = psize;
        const int iters = 32 - __builtin_clz(psize);  // psize >= 1
        for (int it = 0; it < iters; ++it) {
            if (len > 0) {
                const uint32_t half = len >> 1, mid = lo + half;
                const bool less = (pool[mid] & ~FLAG) < key;
                lo = less ? mid + 1 : lo;
                len = less ? len - half - 1 : half;
            }
        }
    }
    if (valid && lo < psize && (pool[lo] & ~FLAG) == key) valid = false;  // already in the pool
    // rank among the surviving candidates; a later lane with an equal key is a duplicate
    uint32_t r = 0;
    for (u64 it = __ballot(valid); it;) {
        const int j = __ffsll((long long)it) - 1;
        it &= it - 1;
        if (!((__ballot(valid) >> j) & 1)) continue;
        const u64 kj = lane_u64(key, j);
        if (valid && j != lane) {
            if (kj < key) r++;
            else if (kj == key && j < lane) valid = false;
        }
    }
    const u64 vm = __ballot(valid);
    if (vm == 0) return;
    const uint32_t s = (uint32_t)__popcll(vm);
    const int l0 = __ffsll((long long)__ballot(valid && r == 0)) - 1;
    const uint32_t pmin = lane_u32(lo, l0);
    // Shift pool[pmin, psize) right: the entry at index i moves up by the number of survivors whose lower bound
    // is <= i.  64-entry chunks, top chunk first, four chunks read before the first is written back: the LDS executes
    // a wave's instructions in order, a lower chunk's reads never overlap an upper chunk's (higher) write range, and
    // a lower chunk's writes come after the upper chunk's reads in program order -- so no wait is needed in between.
    const uint32_t slo = valid ? lo : 0xFFFFFFFFu;
    for (uint32_t hi = psize; hi > pmin;) {
        u64 e[4];
        uint32_t dst[4];
        bool act[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            act[u] = false;
            e[u] = 0ull;
            dst[u] = 0u;
            if (hi > pmin) {  // uniform
                const uint32_t lo_c = (hi - pmin > 64u) ? hi - 64u : pmin;
                const uint32_t kidx = lo_c + (uint32_t)lane;
                const bool a = kidx < hi;
                if (a) e[u] = pool[kidx];
                uint32_t shift = (uint32_t)__popcll(__ballot(slo <= lo_c));
                for (u64 it = __ballot(slo > lo_c && slo < hi); it;) {  // survivors landing inside this chunk
                    const int j = __ffsll((long long)it) - 1;
                    it &= it - 1;
                    shift += (lane_u32(slo, j) <= kidx) ? 1u : 0u;
                }
                dst[u] = kidx + shift;
                act[u] = a && dst[u] < L;
                hi = lo_c;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (act[u]) pool[dst[u]] = e[u];
    }
    wave_sync();
    if (valid && lo + r < L) pool[lo + r] = key;
    psize = (psize + s < L) ? psize + s : L;
    cur = cur < pmin ? cur : pmin;  // neighbor.h:180-182
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


template <int MODE, bool IP, int NCH>
__device__ __forceinline__ void search_one_wave(const RoarSearchParams &p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    u64 *pool = reinterpret_cast<u64 *>(smem + p.off_pool);
    u64 *ck = reinterpret_cast<u64 *>(smem + p.off_ck);
    uint32_t *cid = reinterpret_cast<uint32_t *>(smem + p.off_cid);
    float *lq = reinterpret_cast<float *>(smem + p.off_q);
    uint16_t *vis = reinterpret_cast<uint16_t *>(smem + p.off_vis);
    const uint32_t L = p.L;
    const uint32_t sets_mask = (1u << p.vis_sets_log2) - 1u;
    uint8_t *tags = (MODE == VIS_EXACT) ? p.tags + (u64)blockIdx.x * p.tag_stride : nullptr;
    uint32_t epoch = (MODE == VIS_EXACT) ? p.epochs[blockIdx.x] : 0u;

    for (;;) {
        uint32_t q = 0;
        if (lane == 0) q = atomicAdd(p.counter, 1u);
        q = uni(q);
        if (q >= p.nq) break;

        // query -> LDS (tests/test_search_roargraph.cpp:205 passes aligned_query_data + i * q_dim)
        const float *qp = p.queries + (u64)q * p.qstride;
        for (uint32_t i = lane * 4; i < p.dim; i += 256)
            *reinterpret_cast<float4 *>(lq + i) = *reinterpret_cast<const float4 *>(qp + i);
        if (MODE == VIS_LDS) {  // fresh visited table
            const uint32_t n16 = (4u << p.vis_sets_log2) >> 3;  // sets * 4 ways * 2 B / 16 B
            for (uint32_t i = lane; i < n16; i += 64) reinterpret_cast<uint4 *>(vis)[i] = make_uint4(0, 0, 0, 0);
        } else {  // VisitedList::reset (visited_list_pool.h:20-26) with a u8 epoch
            epoch++;
            if (epoch == 256u) {
                for (u64 i = (u64)lane * 16; i < p.tag_stride; i += 1024)
                    *reinterpret_cast<uint4 *>(tags + i) = make_uint4(0, 0, 0, 0);
                __threadfence();
                epoch = 1;
            }
        }
        // entry point: scored, inserted, NOT marked visited (index_bipartite.cpp:2337-2353)
        if (lane == 0) cid[0] = p.ep;
        wave_sync();
        score_candidates<IP, NCH>(p.base, p.ld, p.dim, lq, cid, ck, 1, lane);
        wave_sync();
        if (lane == 0) pool[0] = ck[0];
        wave_sync();
        uint32_t psize = 1, cur = 0, cmps = 0, hops = 0, scanned = 0;

        while (cur < psize) {  // has_unexpanded_node, :2356
            // closest_unexpanded (neighbor.h:185-192): flag the entry, move the cursor past flagged ones
            const u64 e = pool[cur];
            const uint32_t v = (uint32_t)e >> 1;
            if (lane == 0) pool[cur] = e | 1ull;
            wave_sync();
            {
                uint32_t found = psize;
                for (uint32_t c = cur + 1; c < psize; c += 64) {
                    const uint32_t kidx = c + (uint32_t)lane;
                    const bool unf = kidx < psize && !(pool[kidx] & 1ull);
                    const u64 b = __ballot(unf);
                    if (b) {
                        found = c + (uint32_t)__ffsll((long long)b) - 1u;
                        break;
                    }
                }
                cur = found;
            }
            ++hops;
            // adjacency row: word 0 = degree, word w (>= 1) = neighbour w - 1; 64 words per chunk
            const uint32_t *arow = p.adj + (u64)v * p.adj_stride;
            uint32_t word = arow[lane];
            const uint32_t deg = uni(word);
            scanned += deg;
            for (uint32_t w0 = 0; w0 <= deg; w0 += 64) {
                if (w0) word = arow[w0 + lane];
                const uint32_t w = w0 + (uint32_t)lane;
                const bool isnbr = w >= 1 && w <= deg;
                const uint32_t nb = isnbr ? word : 0u;
                const bool fresh = visited_filter<MODE>(isnbr, nb, vis, sets_mask, p.vis_sets_log2, tags, epoch);
                const u64 fm = __ballot(fresh);
                const uint32_t m = (uint32_t)__popcll(fm);
                if (m == 0) continue;
                if (fresh) cid[lanes_below(fm)] = nb;
                cmps += m;  // ++cmps per scored neighbour, :2397
                wave_sync();
                score_candidates<IP, NCH>(p.base, p.ld, p.dim, lq, cid, ck, m, lane);
                wave_sync();
                merge_candidates(pool, ck, m, L, psize, cur, lane);
            }
        }

        // results (:2408-2418); a short pool is reported, not thrown
        for (uint32_t i = lane; i < p.k; i += 64) {
            uint32_t id = 0xFFFFFFFFu;
            float d = __builtin_inff();
            if (i < psize) {
                const u64 e = pool[i];
                id = (uint32_t)e >> 1;
                d = ord2f((uint32_t)(e >> 32));
            }
            p.out_ids[(u64)q * p.k + i] = id;
            p.out_dists[(u64)q * p.k + i] = d;
        }
        if (lane == 0) {
            if (p.out_cmps) p.out_cmps[q] = cmps;
            if (p.out_hops) p.out_hops[q] = hops;
            if (p.out_scanned) p.out_scanned[q] = scanned;
            if (psize < p.k && p.short_count) atomicAdd(p.short_count, 1u);
        }
        wave_sync();
    }
    if (MODE == VIS_EXACT && lane == 0) p.epochs[blockIdx.x] = epoch;
}

template <int MODE, bool IP, int NCH>
__global__ void __launch_bounds__(64, 3) roar_search_kernel(const RoarSearchParams p) {
    search_one_wave<MODE, IP, NCH>(p);
}

// Latency form of the same search: one workgroup of W waves (4 or 2) per query.  Wave 0 runs the
// sequential part exactly as roar_search_kernel does (select, adjacency, visited filter, merge); the
// row gathers of a hop -- up to 128 candidates -- are spread over the W waves (32 rows each with both
// 16-row passes in flight for W = 4, 64 rows each for W = 2), so a hop costs one adjacency round trip + one
// or two row round trips instead of one per 16 rows.  Used when the whole batch fits the chip that way
// (W = 4: small batches and the per-query facade; W = 2: the batch sizes between that and the one-wave
// throughput form); results are the same by construction (same routines, same merge).
template <int MODE, bool IP, int NCH, int W>
__global__ void __launch_bounds__(64 * W) roar_search_kernel_mw(const RoarSearchParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    u64 *pool = reinterpret_cast<u64 *>(smem + p.off_pool);
    u64 *ck = reinterpret_cast<u64 *>(smem + p.off_ck);
    uint32_t *cid = reinterpret_cast<uint32_t *>(smem + p.off_cid);
    float *lq = reinterpret_cast<float *>(smem + p.off_q);
    uint16_t *vis = reinterpret_cast<uint16_t *>(smem + p.off_vis);
    volatile uint32_t *ctrl = reinterpret_cast<volatile uint32_t *>(smem + p.off_ctrl);
    const uint32_t L = p.L;
    const uint32_t sets_mask = (1u << p.vis_sets_log2) - 1u;
    const uint32_t DONE = 0xFFFFFFFFu;
    uint8_t *tags = (MODE == VIS_EXACT) ? p.tags + (u64)blockIdx.x * p.tag_stride : nullptr;
    uint32_t epoch = (MODE == VIS_EXACT) ? p.epochs[blockIdx.x] : 0u;

    for (;;) {
        if (threadIdx.x == 0) ctrl[0] = atomicAdd(p.counter, 1u);
        __syncthreads();
        const uint32_t q = ctrl[0];
        if (q >= p.nq) break;
        const float *qp = p.queries + (u64)q * p.qstride;
        for (uint32_t i = threadIdx.x * 4; i < p.dim; i += 256 * W)
            *reinterpret_cast<float4 *>(lq + i) = *reinterpret_cast<const float4 *>(qp + i);
        if (MODE == VIS_LDS) {
            const uint32_t n16 = (4u << p.vis_sets_log2) >> 3;
            for (uint32_t i = threadIdx.x; i < n16; i += 64 * W) reinterpret_cast<uint4 *>(vis)[i] = make_uint4(0, 0, 0, 0);
        } else {
            epoch++;
            if (epoch == 256u) {
                for (u64 i = (u64)threadIdx.x * 16; i < p.tag_stride; i += 1024 * W)
                    *reinterpret_cast<uint4 *>(tags + i) = make_uint4(0, 0, 0, 0);
                __threadfence();
                epoch = 1;
            }
        }
        __syncthreads();
        uint32_t psize = 1, cur = 0, cmps = 0, hops = 0, scanned = 0;  // live in wave 0
        uint32_t wpos = 1, deg = 0, word0 = 0;
        const uint32_t *arow = p.adj;
        if (wid == 0) {
            if (lane == 0) cid[0] = p.ep;
            wave_sync();
            score_candidates<IP, NCH>(p.base, p.ld, p.dim, lq, cid, ck, 1, lane);
            wave_sync();
            if (lane == 0) pool[0] = ck[0];
            wave_sync();
        }
        for (;;) {
            if (wid == 0) {
                uint32_t m = 0;
                bool done = false;
                if (wpos > deg) {  // current node exhausted: expand the next one
                    if (cur >= psize) {
                        done = true;
                    } else {
                        const u64 e = pool[cur];
                        const uint32_t v = (uint32_t)e >> 1;
                        if (lane == 0) pool[cur] = e | 1ull;
                        wave_sync();
                        uint32_t found = psize;
                        for (uint32_t c = cur + 1; c < psize; c += 64) {
                            const uint32_t kidx = c + (uint32_t)lane;
                            const bool unf = kidx < psize && !(pool[kidx] & 1ull);
                            const u64 b = __ballot(unf);
                            if (b) {
                                found = c + (uint32_t)__ffsll((long long)b) - 1u;
                                break;
                            }
                        }
                        cur = found;
                        ++hops;
                        arow = p.adj + (u64)v * p.adj_stride;
                        word0 = arow[lane];
                        deg = uni(word0);
                        scanned += deg;
                        wpos = 0;
                    }
                }
                if (!done) {
                    for (int c = 0; c < 2 && wpos <= deg; ++c, wpos += 64) {
                        const uint32_t word = wpos ? arow[wpos + lane] : word0;
                        const uint32_t w = wpos + (uint32_t)lane;
                        const bool isnbr = w >= 1 && w <= deg;
                        const uint32_t nb = isnbr ? word : 0u;
                        const bool fresh = visited_filter<MODE>(isnbr, nb, vis, sets_mask, p.vis_sets_log2, tags, epoch);
                        const u64 fm = __ballot(fresh);
                        if (fresh) cid[m + lanes_below(fm)] = nb;
                        m += (uint32_t)__popcll(fm);
                    }
                    cmps += m;
                }
                if (lane == 0) ctrl[1] = done ? DONE : m;
            }
            __syncthreads();
            const uint32_t m = ctrl[1];
            if (m == DONE) break;
            constexpr uint32_t PER = 128u / W;  // rows per wave and hop
            const uint32_t lo = (uint32_t)wid * PER;
            if (lo < m)
                score_candidates<IP, NCH>(p.base, p.ld, p.dim, lq, cid + lo, ck + lo, (m - lo < PER) ? m - lo : PER, lane);
            __syncthreads();
            if (wid == 0 && m) {
                merge_candidates(pool, ck, m < 64u ? m : 64u, L, psize, cur, lane);
                if (m > 64u) merge_candidates(pool, ck + 64, m - 64u, L, psize, cur, lane);
            }
        }
        if (wid == 0) {
            for (uint32_t i = lane; i < p.k; i += 64) {
                uint32_t id = 0xFFFFFFFFu;
                float d = __builtin_inff();
                if (i < psize) {
                    const u64 e = pool[i];
                    id = (uint32_t)e >> 1;
                    d = ord2f((uint32_t)(e >> 32));
                }
                p.out_ids[(u64)q * p.k + i] = id;
                p.out_dists[(u64)q * p.k + i] = d;
            }
            if (lane == 0) {
                if (p.out_cmps) p.out_cmps[q] = cmps;
                if (p.out_hops) p.out_hops[q] = hops;
                if (p.out_scanned) p.out_scanned[q] = scanned;
                if (psize < p.k && p.short_count) atomicAdd(p.short_count, 1u);
            }
        }
        __syncthreads();
    }
    if (MODE == VIS_EXACT && threadIdx.x == 0) p.epochs[blockIdx.x] = epoch;
}

// ---------------------------------------------------------------------------------------------------
// Second-generation search kernel.  Same algorithm, same routines for the queue, the visited set and the
// reduction tree; what changes is how many dependent memory round trips a hop costs:
//   * the whole adjacency row (up to 3 x 64 words) is fetched in ONE round trip, not one per 64-word chunk;
//   * while the rows of a hop are being scored, the adjacency row of the LIKELY next expansion (the
//     closest unexpanded pool entry as it stands before the hop's merges) is already in flight; it is used
//     if the merges leave that entry in front, dropped otherwise (results cannot depend on it);
//   * the visited filter runs over the whole row first (EXACT mode: the tag loads of all chunks are in flight
//     together), the survivors of the whole hop are packed, scored R x 16 rows per pass and merged 64 at a time;
//   * the 8-float tail of a row (D % 16 == 8, e.g. D = 200) is loaded with the row, not after its reduction;
//   * W waves (1, 2 or 4) share a query: wave 0 owns the sequential part, the rows of a hop are split evenly.
// Within a hop the order of the inserts cannot change the pool (top-L of a set, duplicate keys dropped), so
// packing a hop's candidates is exact; cmps / hops / scanned are counted as before.
// ---------------------------------------------------------------------------------------------------

// LDS visited table, test-and-set of up to three ids per lane at once (one adjacency round trip).  Same table as
// visited_filter<VIS_LDS> -- sets x 4 ways of u16 tags, (set, tag) <-> id a bijection, so a hit is never false -- but an
// empty way is claimed with a 32-bit compare-and-swap on the half of the set that holds it: no write / barrier /
// re-read round per insertion, a lane that loses a race sees the winner's word in the CAS result and tries the next
// way.  A full set overwrites a victim way; that is the only place the table forgets (the node is then scored again
// and its key is dropped by the merge as a duplicate or as worse than the pool's last entry).
__device__ __forceinline__ void visited_lds3(const bool (&isn)[3], const uint32_t (&nb)[3], uint16_t *vis, uint32_t sets_mask,
                                             uint32_t sets_log2, bool (&fresh)[3]) {
    uint32_t *w[3];
    uint32_t tag[3], x[3], y[3];
    bool pend[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint32_t set = nb[c] & sets_mask;
        tag[c] = (nb[c] >> sets_log2) + 1u;
        w[c] = reinterpret_cast<uint32_t *>(vis + 4 * set);
        const uint2 ww = *reinterpret_cast<const uint2 *>(w[c]);
        x[c] = ww.x;
        y[c] = ww.y;
        pend[c] = isn[c];
        fresh[c] = false;
    }
    while (__ballot(pend[0] | pend[1] | pend[2])) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (pend[c]) {
                const uint32_t t0 = x[c] & 0xFFFFu, t1 = x[c] >> 16, t2 = y[c] & 0xFFFFu, t3 = y[c] >> 16;
                if ((t0 == tag[c]) | (t1 == tag[c]) | (t2 == tag[c]) | (t3 == tag[c])) {
                    pend[c] = false;  // visited before (fresh stays as it is: a lane never inserts a tag twice)
                } else if ((t0 != 0) & (t1 != 0) & (t2 != 0) & (t3 != 0)) {
                    reinterpret_cast<uint16_t *>(w[c])[tag[c] & 3u] = (uint16_t)tag[c];  // full set: overwrite a victim
                    fresh[c] = true;
                    pend[c] = false;
                } else {
                    const bool lowhalf = (t0 == 0) | (t1 == 0);
                    const uint32_t oldw = lowhalf ? x[c] : y[c];
                    const uint32_t neww = oldw | (((oldw & 0xFFFFu) == 0) ? tag[c] : (tag[c] << 16));
                    const uint32_t got = atomicCAS(w[c] + (lowhalf ? 0 : 1), oldw, neww);
                    if (got == oldw) {
                        fresh[c] = true;
                        pend[c] = false;
                    } else if (lowhalf) {
                        x[c] = got;
                    } else {
                        y[c] = got;
                    }
                }
            }
        }
    }
}

// One scoring pass: RR x 16 candidates starting at cid[p0], RR rows per 4-lane group, every load of the pass issued
// before the first FMA.  Loads and FMAs are unconditional (a group past the end re-reads the last candidate's row, a
// cache hit, and discards the result): with exec-masked loads the compiler's wait-count bookkeeping has to assume
// loads still in flight at the loop edge and drains the queue -- including the speculative adjacency fetch -- before
// it issues the next pass.
// Keys that a full pool would reject anyway (neighbor.h:151-153: not better than its last entry, `thr`; ~0 while the
// pool is not full) are dropped here; the others are packed into ck[*nout ...].  `thr` is read before the hop's
// merges, so it can only be looser than the reference's test at insert time: the merge repeats the exact test.
// SHARED: several waves append to the same list, the cursor lives in LDS.
struct PassProf {  // instrumented kernels only: cycles spent issuing a pass's loads, waiting + FMAs, reduction + packing
    long long issue = 0, fma = 0, tail = 0;
    uint32_t passes = 0;
};

template <bool IP, int NCH, bool TAIL, int RR, bool SHARED, bool PROF = false>
__device__ __forceinline__ void score_pass(const float *__restrict__ base, uint64_t ld, const float *lq, const float4 qt,
                                           const uint32_t *cid, u64 *ck, uint32_t p0, uint32_t m, u64 thr, uint32_t *nout,
                                           int t, int g, PassProf *pp = nullptr) {
    long long c0 = 0, c1 = 0;
    if (PROF) c0 = (long long)__builtin_readcyclecounter();
    const float4 *q4 = reinterpret_cast<const float4 *>(lq) + t;
    uint32_t id[RR];
    float4 x[RR][NCH];
    float4 tl[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) {
        const uint32_t c = p0 + 16 * r + g;
        id[r] = cid[c < m ? c : m - 1];
    }
#pragma unroll
    for (int r = 0; r < RR; ++r) {
        const float4 *row = reinterpret_cast<const float4 *>(base + (u64)id[r] * ld);
#pragma unroll
        for (int i = 0; i < NCH; ++i) x[r][i] = row[4 * i + t];
        if constexpr (TAIL) tl[r] = row[4 * NCH + (t & 1)];
    }
    // every row load of the pass is issued before the first query fragment is read or the first FMA is scheduled
    // (left alone, the scheduler trades this batch for registers and a pass becomes several dependent round trips)
    __builtin_amdgcn_sched_barrier(0);
    if (PROF) {
        c1 = (long long)__builtin_readcyclecounter();
        pp->issue += c1 - c0;
        c0 = c1;
    }
    float4 a[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) a[r] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const float4 y = q4[4 * i];
#pragma unroll
        for (int r = 0; r < RR; ++r) acc16<IP>(a[r], x[r][i], y);
    }
    if (PROF) {
        c1 = (long long)__builtin_readcyclecounter();
        pp->fma += c1 - c0;
        c0 = c1;
        pp->passes++;
    }
    u64 key[RR], bm[RR];
    uint32_t total = 0;
#pragma unroll
    for (int r = 0; r < RR; ++r) {
        // 16 -> 8 (distance.h:52-53 / :191-192), the 8-float step (:55-63 / :194-201), 8 -> 4 (:65-66 / :203-204), two hadd
        float4 s1;
        s1.x = quad_xor2(a[r].x) + a[r].x;
        s1.y = quad_xor2(a[r].y) + a[r].y;
        s1.z = quad_xor2(a[r].z) + a[r].z;
        s1.w = quad_xor2(a[r].w) + a[r].w;
        if constexpr (TAIL) acc16<IP>(s1, tl[r], qt);
        const float e0 = quad_xor1(s1.x) + s1.x;
        const float e1 = quad_xor1(s1.y) + s1.y;
        const float e2 = quad_xor1(s1.z) + s1.z;
        const float e3 = quad_xor1(s1.w) + s1.w;
        const float rr = (e0 + e1) + (e2 + e3);
        key[r] = make_key(IP ? -rr : rr, id[r]);
        bm[r] = __ballot(t == 0 && p0 + 16 * r + g < m && key[r] < thr);
        total += (uint32_t)__popcll(bm[r]);
    }
    if (total) {  // uniform
        uint32_t at;
        if constexpr (SHARED) {
            at = 0;
            if ((t | g) == 0) at = atomicAdd(nout, total);
            at = uni(at);
        } else {
            at = *nout;
            *nout = at + total;
        }
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            if ((bm[r] >> (4 * g + t)) & 1ull) ck[at + lanes_below(bm[r])] = key[r];
            at += (uint32_t)__popcll(bm[r]);
        }
    }
    if (PROF) pp->tail += (long long)__builtin_readcyclecounter() - c0;
}

// Score cid[0..m) against the query; keys that pass `thr` are appended to ck at *nout (see score_pass).  R x 16 rows
// per pass while that many are left, then narrower passes for the remainder.  NCH full 16-float steps (+ the 8-float
// step when TAIL) are known at compile time for the benchmark shapes (D = 128, 200, 512); NCH == 0 is the run-time
// dimension.
template <bool IP, int NCH, bool TAIL, int R, bool SHARED, bool PROF = false>
__device__ __forceinline__ void score_rows(const float *__restrict__ base, uint64_t ld, uint32_t dim, const float *lq,
                                           const uint32_t *cid, u64 *ck, uint32_t m, u64 thr, uint32_t *nout, int lane,
                                           PassProf *pp = nullptr) {
    const int t = lane & 3, g = lane >> 2;
    if constexpr (NCH > 0) {
        constexpr int RM = NCH > 16 ? 1 : R;  // a 512-d row is 32 float4 registers per lane
        float4 qt = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (TAIL) qt = *(reinterpret_cast<const float4 *>(lq) + 4 * NCH + (t & 1));
        uint32_t p0 = 0;
        if constexpr (RM >= 4)
            for (; p0 + 48 < m; p0 += 64) score_pass<IP, NCH, TAIL, 4, SHARED, PROF>(base, ld, lq, qt, cid, ck, p0, m, thr, nout, t, g, pp);
        if constexpr (RM >= 2)
            for (; p0 + 16 < m; p0 += 32) score_pass<IP, NCH, TAIL, 2, SHARED, PROF>(base, ld, lq, qt, cid, ck, p0, m, thr, nout, t, g, pp);
        for (; p0 < m; p0 += 16) score_pass<IP, NCH, TAIL, 1, SHARED, PROF>(base, ld, lq, qt, cid, ck, p0, m, thr, nout, t, g, pp);
    } else {
        for (uint32_t p0 = 0; p0 < m; p0 += 16) {
            const uint32_t c = p0 + g;
            float d = 0.f;
            uint32_t id = 0;
            if (c < m) {  // whole 4-lane groups are active or idle together
                id = cid[c];
                d = row_distance<IP, 0>(base + (u64)id * ld, lq, dim, t);
            }
            const u64 key = make_key(d, id);
            const u64 bm = __ballot(t == 0 && c < m && key < thr);
            const uint32_t total = (uint32_t)__popcll(bm);
            if (total == 0) continue;
            uint32_t at;
            if constexpr (SHARED) {
                at = 0;
                if (lane == 0) at = atomicAdd(nout, total);
                at = uni(at);
            } else {
                at = *nout;
                *nout = at + total;
            }
            if ((bm >> lane) & 1ull) ck[at + lanes_below(bm)] = key;
        }
    }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory queue, which would
// make wave 0 sit out the round trip of its speculative adjacency fetch before the other waves may start scoring;
// everything the waves of a query hand each other (candidate ids, keys, control words) lives in LDS.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int MODE, bool IP, int NCH, bool TAIL, int W, int R, bool PROF>
__device__ __forceinline__ void search_v2(const RoarSearchParams &p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = (W == 1) ? 0 : (int)(threadIdx.x >> 6);
    u64 *pool = reinterpret_cast<u64 *>(smem + p.off_pool);
    u64 *ck = reinterpret_cast<u64 *>(smem + p.off_ck);
    uint32_t *cid = reinterpret_cast<uint32_t *>(smem + p.off_cid);
    float *lq = reinterpret_cast<float *>(smem + p.off_q);
    uint16_t *vis = reinterpret_cast<uint16_t *>(smem + p.off_vis);
    volatile uint32_t *ctrl = reinterpret_cast<volatile uint32_t *>(smem + p.off_ctrl);
    const uint32_t L = p.L;
    const uint32_t sets_mask = (1u << p.vis_sets_log2) - 1u;
    const uint32_t NONE = 0xFFFFFFFFu;
    uint8_t *tags = (MODE == VIS_EXACT) ? p.tags + (u64)blockIdx.x * p.tag_stride : nullptr;
    uint32_t epoch = (MODE == VIS_EXACT) ? p.epochs[blockIdx.x] : 0u;
    auto block_sync = [&]() {
        if constexpr (W == 1) wave_sync();
        else lds_barrier();
    };

    for (;;) {
        uint32_t q = 0;
        if constexpr (W == 1) {
            if (lane == 0) q = atomicAdd(p.counter, 1u);
            q = uni(q);
        } else {
            if (threadIdx.x == 0) ctrl[0] = atomicAdd(p.counter, 1u);
            lds_barrier();
            q = ctrl[0];
        }
        if (q >= p.nq) break;
        long long tq0 = 0, t_adj = 0, t_filter = 0, t_score = 0, t_merge = 0, t_sel = 0;
        uint32_t pf_hit = 0, n_rt = 0, n_surv = 0;
        PassProf pprof;
        if (PROF) tq0 = (long long)__builtin_readcyclecounter();

        // query -> LDS (tests/test_search_roargraph.cpp:205 passes aligned_query_data + i * q_dim)
        const float *qp = p.queries + (u64)q * p.qstride;
        for (uint32_t i = threadIdx.x * 4; i < p.dim; i += 256 * W)
            *reinterpret_cast<float4 *>(lq + i) = *reinterpret_cast<const float4 *>(qp + i);
        if (MODE == VIS_LDS) {  // fresh visited table
            const uint32_t n16 = (4u << p.vis_sets_log2) >> 3;  // sets * 4 ways * 2 B / 16 B
            for (uint32_t i = threadIdx.x; i < n16; i += 64 * W) reinterpret_cast<uint4 *>(vis)[i] = make_uint4(0, 0, 0, 0);
        } else {  // VisitedList::reset (visited_list_pool.h:20-26) with a u8 epoch
            epoch++;
            if (epoch == 256u) {
                for (u64 i = (u64)threadIdx.x * 16; i < p.tag_stride; i += 1024 * W)
                    *reinterpret_cast<uint4 *>(tags + i) = make_uint4(0, 0, 0, 0);
                __threadfence();
                epoch = 1;
            }
        }
        block_sync();

        // state of the sequential part (wave 0)
        uint32_t psize = 1, cur = 0, cmps = 0, hops = 0, scanned = 0;
        uint32_t a0 = 0, a1 = 0, a2 = 0, pf = NONE, vcur = p.ep, deg = 0, wpos = 1;  // wpos > deg: the current row is exhausted
        const uint32_t *arow = p.adj;
        if (wid == 0) {
            // the entry point is always the first expansion: its adjacency row travels with its base row
            arow = p.adj + (u64)p.ep * p.adj_stride;
            a0 = arow[lane];
            a1 = arow[64 + lane];
            a2 = arow[128 + lane];
            pf = p.ep;
            // entry point: scored, inserted, NOT marked visited (index_bipartite.cpp:2337-2353)
            if (lane == 0) cid[0] = p.ep;
            wave_sync();
            uint32_t one = 0;
            score_rows<IP, NCH, TAIL, 1, false>(p.base, p.ld, p.dim, lq, cid, ck, 1, ~0ull, &one, lane);
            wave_sync();
            if (lane == 0) pool[0] = ck[0];
            wave_sync();
        }

        for (;;) {
            uint32_t m = 0;
            bool done = false;
            if (wid == 0) {
                long long c0 = 0, c1 = 0;
                if (PROF) c0 = (long long)__builtin_readcyclecounter();
                if (wpos > deg) {
                    if (cur >= psize) {  // !has_unexpanded_node, :2356
                        done = true;
                    } else {
                        // closest_unexpanded (neighbor.h:185-192): flag the entry, move the cursor past flagged ones
                        const u64 e = pool[cur];
                        const uint32_t v = (uint32_t)e >> 1;
                        if (lane == 0) pool[cur] = e | 1ull;
                        wave_sync();
                        uint32_t found = psize;
                        for (uint32_t c = cur + 1; c < psize; c += 64) {
                            const uint32_t kidx = c + (uint32_t)lane;
                            const bool unf = kidx < psize && !(pool[kidx] & 1ull);
                            const u64 b = __ballot(unf);
                            if (b) {
                                found = c + (uint32_t)__ffsll((long long)b) - 1u;
                                break;
                            }
                        }
                        cur = found;
                        ++hops;
                        if (PROF) {
                            c1 = (long long)__builtin_readcyclecounter();
                            t_sel += c1 - c0;
                            c0 = c1;
                        }
                        // adjacency row: word 0 = degree, word w (>= 1) = neighbour w - 1
                        arow = p.adj + (u64)v * p.adj_stride;
                        vcur = v;
                        if (v != pf) {  // the speculative fetch went to another row: pay the round trip now
                            a0 = arow[lane];
                            a1 = arow[64 + lane];
                            a2 = arow[128 + lane];
                            if (PROF) n_rt++;
                        } else if (PROF) {
                            pf_hit++;
                        }
                        deg = uni(a0);
                        scanned += deg;
                        wpos = 0;
                    }
                }  // else: a row longer than 191 entries; its next 192 words were fetched under the last scoring pass
                if (!done) {
                    // visited filter over the whole round trip (index_bipartite.cpp:2378-2385), survivors packed in cid
                    uint32_t nb[3];
                    bool isn[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const uint32_t w = wpos + 64u * (uint32_t)c + (uint32_t)lane;
                        isn[c] = w >= 1 && w <= deg;
                        nb[c] = isn[c] ? (c == 0 ? a0 : c == 1 ? a1 : a2) : 0u;
                    }
                    if (PROF) {
                        c1 = (long long)__builtin_readcyclecounter();
                        t_adj += c1 - c0;
                        c0 = c1;
                    }
                    if (MODE == VIS_EXACT) {
                        uint32_t tg[3] = {0, 0, 0};
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            if (isn[c]) tg[c] = __builtin_nontemporal_load(tags + nb[c]);
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const bool fresh = isn[c] && tg[c] != epoch;
                            if (fresh) __builtin_nontemporal_store((uint8_t)epoch, tags + nb[c]);
                            const u64 fm = __ballot(fresh);
                            if (fresh) cid[m + lanes_below(fm)] = nb[c];
                            m += (uint32_t)__popcll(fm);
                        }
                    } else {
                        bool fresh[3];
                        visited_lds3(isn, nb, vis, sets_mask, p.vis_sets_log2, fresh);
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const u64 fm = __ballot(fresh[c]);
                            if (fresh[c]) cid[m + lanes_below(fm)] = nb[c];
                            m += (uint32_t)__popcll(fm);
                        }
                    }
                    wpos += 192u;
                    cmps += m;  // ++cmps per scored neighbour, :2397
                    // Next adjacency words, in flight while this round trip's rows are scored: the rest of this row
                    // if it is longer, else the row of the closest unexpanded entry as the pool stands now -- if this
                    // hop's merges put something in front of it the fetch is simply dropped (pf is compared at select).
                    {
                        const uint32_t *nrow = arow + wpos;
                        if (wpos > deg) {
                            pf = (cur < psize) ? (uint32_t)pool[cur] >> 1 : vcur;  // (nothing left: any valid row)
                            nrow = p.adj + (u64)pf * p.adj_stride;
                        }
                        a0 = nrow[lane];
                        a1 = nrow[64 + lane];
                        a2 = nrow[128 + lane];
                    }
                    if (PROF) {
                        c1 = (long long)__builtin_readcyclecounter();
                        t_filter += c1 - c0;
                    }
                }
                if (W > 1 && lane == 0) {
                    ctrl[1] = done ? NONE : m;
                    ctrl[2] = 0;                     // survivors appended by the waves
                    ctrl[3] = (psize == L) ? 1u : 0u;  // pool full: keys are tested against its last entry
                }
            }
            long long s0 = 0;
            if (PROF) s0 = (long long)__builtin_readcyclecounter();
            uint32_t nsurv = 0;
            if constexpr (W > 1) {
                lds_barrier();
                m = ctrl[1];
                if (m == NONE) break;
                if (m) {  // even split in units of 16 rows
                    const u64 thr = ctrl[3] ? (pool[L - 1] & ~1ull) : ~0ull;
                    const uint32_t per = (((m + W - 1) / W) + 15u) & ~15u;
                    const uint32_t lo = (uint32_t)wid * per;
                    if (lo < m)
                        score_rows<IP, NCH, TAIL, R, true, PROF>(p.base, p.ld, p.dim, lq, cid + lo, ck, (m - lo < per) ? m - lo : per,
                                                                 thr, const_cast<uint32_t *>(ctrl + 2), lane, &pprof);
                }
                lds_barrier();
                nsurv = ctrl[2];
            } else {
                if (done) break;
                wave_sync();
                if (m) {
                    const u64 thr = (psize == L) ? (pool[L - 1] & ~1ull) : ~0ull;
                    score_rows<IP, NCH, TAIL, R, false, PROF>(p.base, p.ld, p.dim, lq, cid, ck, m, thr, &nsurv, lane, &pprof);
                }
                wave_sync();
            }
            if (PROF && wid == 0) {
                const long long s1 = (long long)__builtin_readcyclecounter();
                t_score += s1 - s0;
                s0 = s1;
            }
            if (wid == 0) {
                if (PROF) n_surv += nsurv;
                for (uint32_t b = 0; b < nsurv; b += 64)
                    merge_candidates(pool, ck + b, (nsurv - b < 64u) ? nsurv - b : 64u, L, psize, cur, lane);
                if (PROF) t_merge += (long long)__builtin_readcyclecounter() - s0;
            }
        }

        // results (:2408-2418); a short pool is reported, not thrown
        if (wid == 0) {
            for (uint32_t i = lane; i < p.k; i += 64) {
                uint32_t id = 0xFFFFFFFFu;
                float d = __builtin_inff();
                if (i < psize) {
                    const u64 e = pool[i];
                    id = (uint32_t)e >> 1;
                    d = ord2f((uint32_t)(e >> 32));
                }
                p.out_ids[(u64)q * p.k + i] = id;
                p.out_dists[(u64)q * p.k + i] = d;
            }
            if (lane == 0) {
                if (p.out_cmps) p.out_cmps[q] = cmps;
                if (p.out_hops) p.out_hops[q] = hops;
                if (p.out_scanned) p.out_scanned[q] = scanned;
                if (psize < p.k && p.short_count) atomicAdd(p.short_count, 1u);
                if (PROF && p.prof) {
                    unsigned long long *o = p.prof + (u64)q * 16;
                    o[8] = (unsigned long long)pprof.issue;
                    o[9] = (unsigned long long)pprof.fma;
                    o[10] = (unsigned long long)pprof.tail;
                    o[11] = pprof.passes;
                    o[12] = n_surv;
                    o[13] = o[14] = o[15] = 0;
                    o[0] = (unsigned long long)((long long)__builtin_readcyclecounter() - tq0);
                    o[1] = (unsigned long long)t_sel;
                    o[2] = (unsigned long long)t_adj;
                    o[3] = (unsigned long long)t_filter;
                    o[4] = (unsigned long long)t_score;
                    o[5] = (unsigned long long)t_merge;
                    o[6] = pf_hit;
                    o[7] = n_rt;
                }
            }
        }
        block_sync();
    }
    if (MODE == VIS_EXACT && threadIdx.x == 0) p.epochs[blockIdx.x] = epoch;
}

// register budget: three waves per SIMD with two rows per group and pass in flight, two waves with four rows (the
// explicit maximum keeps the scheduler from trading the batched loads of a pass for an occupancy LDS does not allow)
template <int MODE, bool IP, int NCH, bool TAIL, int W, int R, bool PROF>
__global__ void __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(R == 4 ? 2 : 3, R == 4 ? 2 : 3)))
roar_search_v2_kernel(const RoarSearchParams p) {
    search_v2<MODE, IP, NCH, TAIL, W, R, PROF>(p);
}


// SearchBipartiteGraph (src/index_bipartite.cpp:282-356), the two-hop walk over the bipartite graph
// (base node -> training-query nodes -> base nodes).  Same wave-per-query machinery as above; what
// differs from SearchRoarGraph is kept as the reference has it:
//   * n_init explicit entry points, every one scored, inserted and marked visited (:304-316; the
//     reference draws them from std::random_device, here the caller passes them so a run can be replayed);
//     they are not counted in cmps;
//   * a hop with an empty row still counts (:324-327);
//   * candidates of a hop = rows of the hop's query nodes, first occurrence wins (:330-334).
// The rows are CSR (a base node's row is as long as the number of training queries it is nearest to,
// which is unbounded).  64 query nodes of a hop are handled together: their row extents are prefix-
// summed, the flattened candidate range is walked 64 entries at a time, survivors of the visited filter
// are packed in LDS and scored 64 rows at a time.  Within a hop the order of the inserts cannot change
// the pool (top-L of a set, duplicates dropped), so packing across rows is exact.
template <int MODE, bool IP, int NCH>
__global__ void __launch_bounds__(64, 3) roar_bipartite_kernel(const RoarSearchParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    u64 *pool = reinterpret_cast<u64 *>(smem + p.off_pool);
    u64 *ck = reinterpret_cast<u64 *>(smem + p.off_ck);
    uint32_t *cid = reinterpret_cast<uint32_t *>(smem + p.off_cid);
    float *lq = reinterpret_cast<float *>(smem + p.off_q);
    uint16_t *vis = reinterpret_cast<uint16_t *>(smem + p.off_vis);
    u64 *rstart = reinterpret_cast<u64 *>(smem + p.off_bp);              // [64] first word of each query node's row
    uint32_t *rpre = reinterpret_cast<uint32_t *>(smem + p.off_bp + 512);  // [64] exclusive prefix of the row lengths
    const uint32_t L = p.L;
    const uint32_t sets_mask = (1u << p.vis_sets_log2) - 1u;
    uint8_t *tags = (MODE == VIS_EXACT) ? p.tags + (u64)blockIdx.x * p.tag_stride : nullptr;
    uint32_t epoch = (MODE == VIS_EXACT) ? p.epochs[blockIdx.x] : 0u;

    for (;;) {
        uint32_t q = 0;
        if (lane == 0) q = atomicAdd(p.counter, 1u);
        q = uni(q);
        if (q >= p.nq) break;

        const float *qp = p.queries + (u64)q * p.qstride;
        for (uint32_t i = lane * 4; i < p.dim; i += 256)
            *reinterpret_cast<float4 *>(lq + i) = *reinterpret_cast<const float4 *>(qp + i);
        if (MODE == VIS_LDS) {
            const uint32_t n16 = (4u << p.vis_sets_log2) >> 3;
            for (uint32_t i = lane; i < n16; i += 64) reinterpret_cast<uint4 *>(vis)[i] = make_uint4(0, 0, 0, 0);
        } else {
            epoch++;
            if (epoch == 256u) {
                for (u64 i = (u64)lane * 16; i < p.tag_stride; i += 1024)
                    *reinterpret_cast<uint4 *>(tags + i) = make_uint4(0, 0, 0, 0);
                __threadfence();
                epoch = 1;
            }
        }
        // entry points (:304-316): scored, inserted, marked
        const bool isinit = (uint32_t)lane < p.n_init;
        const uint32_t iid = isinit ? p.init_ids[(u64)q * p.n_init + lane] : 0u;
        if (isinit) cid[lane] = iid;
        wave_sync();
        score_candidates<IP, NCH>(p.base, p.ld, p.dim, lq, cid, ck, p.n_init, lane);
        wave_sync();
        if (lane == 0) pool[0] = ck[0];
        wave_sync();
        uint32_t psize = 1, cur = 0, cmps = 0, hops = 0, scanned = 0;
        merge_candidates(pool, ck, p.n_init, L, psize, cur, lane);  // ck[0] again: dropped as a duplicate
        (void)visited_filter<MODE>(isinit, iid, vis, sets_mask, p.vis_sets_log2, tags, epoch);
        wave_sync();

        while (cur < psize) {
            const u64 e = pool[cur];
            const uint32_t v = (uint32_t)e >> 1;
            if (lane == 0) pool[cur] = e | 1ull;
            wave_sync();
            {
                uint32_t found = psize;
                for (uint32_t c = cur + 1; c < psize; c += 64) {
                    const uint32_t kidx = c + (uint32_t)lane;
                    const bool unf = kidx < psize && !(pool[kidx] & 1ull);
                    const u64 b = __ballot(unf);
                    if (b) {
                        found = c + (uint32_t)__ffsll((long long)b) - 1u;
                        break;
                    }
                }
                cur = found;
            }
            ++hops;
            const u64 b0 = p.bp_off[v], b1 = p.bp_off[v + 1];
            scanned += (uint32_t)(b1 - b0);
            uint32_t cnt = 0;  // survivors waiting in cid[0, cnt)
            for (u64 r = b0; r < b1; r += 64) {
                // this lane's query node and the extent of its row
                const bool hasq = r + (u64)lane < b1;
                u64 s0 = 0;
                uint32_t len = 0;
                if (hasq) {
                    const uint32_t qn = p.bp_nbrs[r + lane];
                    s0 = p.bp_off[qn];
                    len = (uint32_t)(p.bp_off[qn + 1] - s0);
                }
                uint32_t incl = len;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t y = __shfl_up(incl, d);
                    if (lane >= d) incl += y;
                }
                const uint32_t total = lane_u32(incl, 63);
                rstart[lane] = s0;
                rpre[lane] = incl - len;
                scanned += total;
                wave_sync();
                for (uint32_t c0 = 0; c0 < total; c0 += 64) {
                    const uint32_t c = c0 + (uint32_t)lane;
                    const bool has = c < total;
                    uint32_t own = 0;  // last row whose prefix is <= c (empty rows share a prefix with their successor)
#pragma unroll
                    for (uint32_t step = 32; step; step >>= 1)
                        if (rpre[own + step] <= c) own += step;
                    uint32_t nb = 0;
                    if (has) nb = p.bp_nbrs[rstart[own] + (c - rpre[own])];
                    bool fresh = visited_filter<MODE>(has, nb, vis, sets_mask, p.vis_sets_log2, tags, epoch);
                    // the same node twice in this chunk: the later lane is the repeat (:331-334)
                    for (u64 it = __ballot(fresh); it;) {
                        const int j = __ffsll((long long)it) - 1;
                        it &= it - 1;
                        const uint32_t idj = lane_u32(nb, j);
                        if (lane > j && nb == idj) fresh = false;
                    }
                    const u64 fm = __ballot(fresh);
                    const uint32_t m = (uint32_t)__popcll(fm);
                    if (m == 0) continue;
                    if (fresh) cid[cnt + lanes_below(fm)] = nb;
                    cnt += m;
                    cmps += m;  // :342
                    wave_sync();
                    if (cnt >= 64) {
                        score_candidates<IP, NCH>(p.base, p.ld, p.dim, lq, cid, ck, 64, lane);
                        wave_sync();
                        merge_candidates(pool, ck, 64, L, psize, cur, lane);
                        const uint32_t rest = cnt - 64;
                        const uint32_t tmp = ((uint32_t)lane < rest) ? cid[64 + lane] : 0u;
                        wave_sync();
                        if ((uint32_t)lane < rest) cid[lane] = tmp;
                        cnt = rest;
                        wave_sync();
                    }
                }
                wave_sync();  // rstart / rpre are rewritten by the next 64 query nodes
            }
            if (cnt) {
                score_candidates<IP, NCH>(p.base, p.ld, p.dim, lq, cid, ck, cnt, lane);
                wave_sync();
                merge_candidates(pool, ck, cnt, L, psize, cur, lane);
            }
        }

        for (uint32_t i = lane; i < p.k; i += 64) {
            uint32_t id = 0xFFFFFFFFu;
            float d = __builtin_inff();
            if (i < psize) {
                const u64 e = pool[i];
                id = (uint32_t)e >> 1;
                d = ord2f((uint32_t)(e >> 32));
            }
            p.out_ids[(u64)q * p.k + i] = id;
            p.out_dists[(u64)q * p.k + i] = d;
        }
        if (lane == 0) {
            if (p.out_cmps) p.out_cmps[q] = cmps;
            if (p.out_hops) p.out_hops[q] = hops;
            if (p.out_scanned) p.out_scanned[q] = scanned;
            if (psize < p.k && p.short_count) atomicAdd(p.short_count, 1u);
        }
        wave_sync();
    }
    if (MODE == VIS_EXACT && lane == 0) p.epochs[blockIdx.x] = epoch;
}

// Workgroup form of the two-hop walk: 8 waves (512 threads) per query.  Work per query is extremely uneven
// on a bipartite graph (most random entry points are dead ends, a hub hop has tens of thousands of two-hop
// candidates), so a launch of the one-wave form lasts as long as its single heaviest query.  Here the eight
// waves split every block of 512 flattened candidates: each wave filters its 64 entries with an exact
// test-and-set on u32 epoch tags in HBM (atomicExch: no entry is taken twice, within or across waves, so cmps
// equals the reference's), scores its survivors, drops those a full pool would reject (neighbor.h:151-153) and
// only then takes its turn at the shared pool; wave 0 selects the next node and sets up the row extents.
// Same results as roar_bipartite_kernel by the same argument (within a hop the pool is order-free).
constexpr int BPW = 8;  // waves per query
template <bool IP, int NCH>
__global__ void __launch_bounds__(64 * BPW) roar_bipartite_kernel_wg(const RoarSearchParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    u64 *pool = reinterpret_cast<u64 *>(smem + p.off_pool);
    u64 *ck = reinterpret_cast<u64 *>(smem + p.off_ck) + 64 * wid;
    uint32_t *cid = reinterpret_cast<uint32_t *>(smem + p.off_cid) + 64 * wid;
    float *lq = reinterpret_cast<float *>(smem + p.off_q);
    u64 *rstart = reinterpret_cast<u64 *>(smem + p.off_bp);
    uint32_t *rpre = reinterpret_cast<uint32_t *>(smem + p.off_bp + 512);
    // ctrl: 0 query, 1 psize, 2 cur, 3 cmps, 4 merge mask, 5 node being expanded, 6 "has node", 7 row total
    volatile uint32_t *ctrl = reinterpret_cast<volatile uint32_t *>(smem + p.off_ctrl);
    uint32_t *ctrl_a = reinterpret_cast<uint32_t *>(smem + p.off_ctrl);
    const uint32_t L = p.L;
    uint32_t *tags = p.tags32 + (u64)blockIdx.x * p.tag_stride32;
    uint32_t epoch = p.epochs[blockIdx.x];

    for (;;) {
        if (threadIdx.x == 0) ctrl[0] = atomicAdd(p.counter, 1u);
        __syncthreads();
        const uint32_t q = ctrl[0];
        if (q >= p.nq) break;
        const float *qp = p.queries + (u64)q * p.qstride;
        for (uint32_t i = threadIdx.x * 4; i < p.dim; i += 256 * BPW)
            *reinterpret_cast<float4 *>(lq + i) = *reinterpret_cast<const float4 *>(qp + i);
        epoch++;
        if (epoch == 0u) {  // VisitedList::reset wrap (visited_list_pool.h:20-26)
            for (u64 i = threadIdx.x; i < p.tag_stride32; i += 64 * BPW) tags[i] = 0u;
            __threadfence();
            epoch = 1;
        }
        __syncthreads();
        if (wid == 0) {  // entry points (:304-316): scored, inserted, marked; not counted
            const bool isinit = (uint32_t)lane < p.n_init;
            const uint32_t iid = isinit ? p.init_ids[(u64)q * p.n_init + lane] : 0u;
            if (isinit) cid[lane] = iid;
            wave_sync();
            score_candidates<IP, NCH>(p.base, p.ld, p.dim, lq, cid, ck, p.n_init, lane);
            wave_sync();
            if (lane == 0) pool[0] = ck[0];
            wave_sync();
            uint32_t psize = 1, cur = 0;
            merge_candidates(pool, ck, p.n_init, L, psize, cur, lane);
            if (isinit) (void)atomicExch(&tags[iid], epoch);
            if (lane == 0) {
                ctrl[1] = psize;
                ctrl[2] = cur;
                ctrl[3] = 0;
                ctrl[4] = 0;
            }
        }
        __syncthreads();
        uint32_t hops = 0, scanned = 0;
        for (;;) {
            if (wid == 0) {  // closest_unexpanded (neighbor.h:185-192)
                const uint32_t psize = ctrl[1];
                uint32_t cur = ctrl[2];
                if (cur < psize) {
                    const u64 e = pool[cur];
                    if (lane == 0) pool[cur] = e | 1ull;
                    wave_sync();
                    uint32_t found = psize;
                    for (uint32_t c = cur + 1; c < psize; c += 64) {
                        const uint32_t kidx = c + (uint32_t)lane;
                        const bool unf = kidx < psize && !(pool[kidx] & 1ull);
                        const u64 b = __ballot(unf);
                        if (b) {
                            found = c + (uint32_t)__ffsll((long long)b) - 1u;
                            break;
                        }
                    }
                    if (lane == 0) {
                        ctrl[2] = found;
                        ctrl[5] = (uint32_t)e >> 1;
                        ctrl[6] = 1;
                    }
                } else if (lane == 0) {
                    ctrl[6] = 0;
                }
            }
            __syncthreads();
            if (!ctrl[6]) break;
            const uint32_t v = ctrl[5];
            ++hops;
            const u64 b0 = p.bp_off[v], b1 = p.bp_off[v + 1];
            scanned += (uint32_t)(b1 - b0);
            for (u64 r = b0; r < b1; r += 64) {
                if (wid == 0) {  // row extents of the next 64 query nodes
                    const bool hasq = r + (u64)lane < b1;
                    u64 s0 = 0;
                    uint32_t len = 0;
                    if (hasq) {
                        const uint32_t qn = p.bp_nbrs[r + lane];
                        s0 = p.bp_off[qn];
                        len = (uint32_t)(p.bp_off[qn + 1] - s0);
                    }
                    uint32_t incl = len;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const uint32_t y = __shfl_up(incl, d);
                        if (lane >= d) incl += y;
                    }
                    rstart[lane] = s0;
                    rpre[lane] = incl - len;
                    if (lane == 63) ctrl[7] = incl;
                }
                __syncthreads();
                const uint32_t total = ctrl[7];
                scanned += total;
                for (uint32_t c0 = 0; c0 < total; c0 += 64 * BPW) {
                    const uint32_t c = c0 + 64u * (uint32_t)wid + (uint32_t)lane;
                    const bool has = c < total;
                    uint32_t own = 0;
#pragma unroll
                    for (uint32_t step = 32; step; step >>= 1)
                        if (rpre[own + step] <= c) own += step;
                    uint32_t nb = 0;
                    bool fresh = false;
                    if (has) {
                        nb = p.bp_nbrs[rstart[own] + (c - rpre[own])];
                        fresh = atomicExch(&tags[nb], epoch) != epoch;  // :330-334, exact across lanes and waves
                    }
                    const u64 fm = __ballot(fresh);
                    const uint32_t m = (uint32_t)__popcll(fm);
                    bool want = false;
                    if (m) {
                        if (fresh) cid[lanes_below(fm)] = nb;
                        wave_sync();
                        score_candidates<IP, NCH>(p.base, p.ld, p.dim, lq, cid, ck, m, lane);
                        wave_sync();
                        // a full pool rejects what is not better than its last entry (neighbor.h:151-153); the
                        // pool cannot change between the two barriers around this block
                        const bool mine = (uint32_t)lane < m;
                        const u64 key = mine ? ck[lane] : ~0ull;
                        const bool ok = mine && (ctrl[1] < L || key < (pool[L - 1] & ~1ull));
                        want = __ballot(ok) != 0;
                        if (lane == 0) {
                            atomicAdd(&ctrl_a[3], m);  // ++cmps, :342
                            if (want) atomicOr(&ctrl_a[4], 1u << wid);
                        }
                    }
                    __syncthreads();
                    uint32_t mask = ctrl[4];
                    __syncthreads();
                    if (threadIdx.x == 0) ctrl[4] = 0;
                    while (mask) {  // the waves that have something to insert take turns at the pool
                        const int w = __ffs((int)mask) - 1;
                        mask &= mask - 1;
                        if (wid == w) {
                            uint32_t psize = ctrl[1], cur = ctrl[2];
                            merge_candidates(pool, ck, m, L, psize, cur, lane);
                            if (lane == 0) {
                                ctrl[1] = psize;
                                ctrl[2] = cur;
                            }
                        }
                        __syncthreads();
                    }
                    __syncthreads();  // the mask reset is in place before the next round's atomicOr
                }
                __syncthreads();  // rstart / rpre are rewritten by the next 64 query nodes
            }
            __syncthreads();  // ctrl[5], ctrl[6] are rewritten by the next select (a hop over an empty row has no other barrier)
        }
        if (wid == 0) {
            const uint32_t psize = ctrl[1];
            for (uint32_t i = lane; i < p.k; i += 64) {
                uint32_t id = 0xFFFFFFFFu;
                float d = __builtin_inff();
                if (i < psize) {
                    const u64 e = pool[i];
                    id = (uint32_t)e >> 1;
                    d = ord2f((uint32_t)(e >> 32));
                }
                p.out_ids[(u64)q * p.k + i] = id;
                p.out_dists[(u64)q * p.k + i] = d;
            }
            if (lane == 0) {
                if (p.out_cmps) p.out_cmps[q] = ctrl[3];
                if (p.out_hops) p.out_hops[q] = hops;
                if (p.out_scanned) p.out_scanned[q] = scanned;
                if (psize < p.k && p.short_count) atomicAdd(p.short_count, 1u);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) p.epochs[blockIdx.x] = epoch;
}

// distance-only kernel: one wave per query, m explicit ids each (parity surface for A2/A3)
template <bool IP>
__global__ void __launch_bounds__(64) roar_distance_kernel(const RoarDistParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    u64 *ck = reinterpret_cast<u64 *>(smem + p.off_ck);
    uint32_t *cid = reinterpret_cast<uint32_t *>(smem + p.off_cid);
    float *lq = reinterpret_cast<float *>(smem + p.off_q);
    for (uint32_t q = blockIdx.x; q < p.nq; q += gridDim.x) {
        const float *qp = p.queries + (u64)q * p.qstride;
        for (uint32_t i = lane * 4; i < p.dim; i += 256)
            *reinterpret_cast<float4 *>(lq + i) = *reinterpret_cast<const float4 *>(qp + i);
        for (uint32_t c0 = 0; c0 < p.m; c0 += 64) {
            const uint32_t mm = (p.m - c0 < 64u) ? p.m - c0 : 64u;
            if ((uint32_t)lane < mm) cid[lane] = p.ids[(u64)q * p.m + c0 + lane];
            wave_sync();
            score_candidates<IP, 0>(p.base, p.ld, p.dim, lq, cid, ck, mm, lane);
            wave_sync();
            if ((uint32_t)lane < mm) p.out[(u64)q * p.m + c0 + lane] = ord2f((uint32_t)(ck[lane] >> 32));
            wave_sync();
        }
    }
}

// k-way merge of per-shard sorted top-k lists by (distance, id); one thread per query.
__global__ void roar_merge_topk_kernel(uint32_t nshards, u64 nq, uint32_t k, const uint32_t *__restrict__ ids,
                                       const float *__restrict__ dists, uint32_t *__restrict__ out_ids,
                                       float *__restrict__ out_dists) {
    const u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    uint32_t head[32];
    for (uint32_t s = 0; s < nshards; ++s) head[s] = 0;
    for (uint32_t i = 0; i < k; ++i) {
        u64 best = ~0ull;
        uint32_t bs = 0xFFFFFFFFu;
        for (uint32_t s = 0; s < nshards; ++s) {
            if (head[s] >= k) continue;
            const u64 off = ((u64)s * nq + q) * k + head[s];
            const uint32_t id = ids[off];
            if (id == 0xFFFFFFFFu) {
                head[s] = k;
                continue;
            }
            const u64 key = ((u64)f2ord(dists[off]) << 32) | id;
            if (key < best) {
                best = key;
                bs = s;
            }
        }
        if (bs == 0xFFFFFFFFu) {
            out_ids[q * k + i] = 0xFFFFFFFFu;
            out_dists[q * k + i] = __builtin_inff();
        } else {
            const u64 off = ((u64)bs * nq + q) * k + head[bs];
            out_ids[q * k + i] = ids[off];
            out_dists[q * k + i] = dists[off];
            head[bs]++;
        }
    }
}

// The v2 kernel under its own symbol for the searches the graph builder issues (LDS visited mode, one wave per
// query), so that a kernel trace of a process that builds and then serves keeps the two apart.
template <bool IP, int NCH, bool TAIL>
__global__ void __launch_bounds__(64) roar_build_search_v2_kernel(const RoarSearchParams p) {
    search_v2<VIS_LDS, IP, NCH, TAIL, 1, 2, false>(p);
}

// NCH1: compile-time row length of the first-generation kernels (they take the 8-float step at run time)
template <int MODE, bool IP, int NCH, bool TAIL, int NCH1>
const void *pick_form(const RoarKernelSel &s) {
    switch (s.form) {
        case ROAR_FORM_BIPARTITE: return (const void *)roar_bipartite_kernel<MODE, IP, NCH1>;
        case ROAR_FORM_BIPARTITE_WG: return (const void *)roar_bipartite_kernel_wg<IP, NCH1>;
        case ROAR_FORM_BUILDER: return (const void *)roar_build_search_v2_kernel<IP, NCH, TAIL>;
        case ROAR_FORM_V1:
            if (s.waves == 4) return (const void *)roar_search_kernel_mw<MODE, IP, NCH1, 4>;
            if (s.waves == 2) return (const void *)roar_search_kernel_mw<MODE, IP, NCH1, 2>;
            return (const void *)roar_search_kernel<MODE, IP, NCH1>;
        default: break;
    }
    if (s.prof) {
        if (s.waves == 4) return (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 4, 2, true>;
        if (s.waves == 2) return (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 2, 2, true>;
        return (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, 2, true>;
    }
    if (s.waves == 4) return (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 4, 2, false>;
    if (s.waves == 2)
        return s.rows == 4 ? (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 2, 4, false>
                           : (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 2, 2, false>;
    return s.rows == 4 ? (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, 4, false>
                       : (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, 2, false>;
}

template <int MODE, bool IP>
const void *pick_nch(const RoarKernelSel &s, uint32_t dim) {
    switch (dim) {
        case 128: return pick_form<MODE, IP, 8, false, 8>(s);    // SIFT shape
        case 200: return pick_form<MODE, IP, 12, true, 12>(s);   // Text-to-Image shape
        case 512: return pick_form<MODE, IP, 32, false, 0>(s);   // WebVid shape
        default: break;
    }
    switch (dim >> 4) {  // other lengths: the v2 kernels walk the row at run time
        case 8: return pick_form<MODE, IP, 0, false, 8>(s);
        case 12: return pick_form<MODE, IP, 0, false, 12>(s);
        default: return pick_form<MODE, IP, 0, false, 0>(s);
    }
}

int block_threads(const RoarKernelSel &s) {
    if (s.form == ROAR_FORM_BIPARTITE_WG) return 64 * BPW;
    if (s.form == ROAR_FORM_BIPARTITE || s.form == ROAR_FORM_BUILDER) return 64;
    return 64 * (s.waves == 4 ? 4 : s.waves == 2 ? 2 : 1);
}

const void *pick_kernel(const RoarKernelSel &s, uint32_t dim) {
    if (s.visited_mode == VIS_EXACT) return s.metric_ip ? pick_nch<VIS_EXACT, true>(s, dim) : pick_nch<VIS_EXACT, false>(s, dim);
    return s.metric_ip ? pick_nch<VIS_LDS, true>(s, dim) : pick_nch<VIS_LDS, false>(s, dim);
}

}  // namespace

extern "C" int roar_launch_search(const RoarSearchParams *p, const RoarKernelSel *sel, uint32_t grid, void *stream) {
    const void *fn = pick_kernel(*sel, p->dim);
    if (p->lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    RoarSearchParams params = *p;
    void *args[] = {&params};
    return (int)hipLaunchKernel(fn, dim3(grid), dim3(block_threads(*sel)), args, p->lds_bytes, (hipStream_t)stream);
}

extern "C" int roar_search_occupancy(const RoarKernelSel *sel, uint32_t dim, uint32_t lds_bytes, int *blocks_per_cu) {
    const void *fn = pick_kernel(*sel, dim);
    if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, fn, block_threads(*sel), lds_bytes);
}

extern "C" int roar_launch_distances(const RoarDistParams *p, int metric_ip, void *stream) {
    uint32_t grid = p->nq < 4096u ? p->nq : 4096u;
    if (grid == 0) return 0;
    if (metric_ip)
        hipLaunchKernelGGL(roar_distance_kernel<true>, dim3(grid), dim3(64), p->lds_bytes, (hipStream_t)stream, *p);
    else
        hipLaunchKernelGGL(roar_distance_kernel<false>, dim3(grid), dim3(64), p->lds_bytes, (hipStream_t)stream, *p);
    return (int)hipGetLastError();
}

extern "C" int roar_launch_merge_topk(uint32_t nshards, uint64_t nq, uint32_t k, const uint32_t *ids,
                                      const float *dists, uint32_t *out_ids, float *out_dists, void *stream) {
    if (nq == 0) return 0;
    const uint32_t block = 256;
    const uint32_t grid = (uint32_t)((nq + block - 1) / block);
    hipLaunchKernelGGL(roar_merge_topk_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, nshards, (u64)nq, k, ids,
                       dists, out_ids, out_dists);
    return (int)hipGetLastError();
}
