// roar_search_v2.h -- the SearchRoarGraph kernel (src/index_bipartite.cpp:2311-2420) for gfx950, included by one
// translation unit per (visited mode, metric) pair: roar_search_{lds,exact}_{ip,l2}.hip define ROAR_TU_MODE / ROAR_TU_IP.
//
// One workgroup of W waves (1, 2 or 4) owns one query at a time (persistent, dynamic dequeue).  Wave 0 runs the
// sequential part of the reference's loop (:2356-2405) -- closest_unexpanded, the adjacency row, the visited test,
// the inserts -- and every wave scores its share of the hop's base rows.  What a hop costs is counted in dependent
// memory round trips and in instruction latency, so:
//   * the whole adjacency row (up to 3 x 64 words) is fetched in ONE round trip;
//   * while the rows of a hop are being scored, the adjacency row of the LIKELY next expansion (the closest
//     unexpanded pool entry as it stands before the hop's merges) is already in flight; it is used if the merges
//     leave that entry in front, dropped otherwise (results cannot depend on it);
//   * the visited filter runs over the whole row at once (LDS modes: three compare-and-swaps per lane in flight, a full
//     set either spills to the workgroup's epoch list in HBM -- exact -- or overwrites a victim -- forgetful, see
//     visited_ldsn; EXACT mode: the tag loads of all three chunks in flight; long pools: the highest-in-degree nodes have their
//     visited bits in an exact on-chip bitmap, named by slot in a hub-annotated copy of the adjacency, and only the
//     other neighbours go to the tag table / the HBM tags), the survivors of the whole hop are packed, scored
//     R x 16 rows per pass -- every load of a pass, the 8-float tail of a D = 200 row included, issued before the
//     first FMA -- and keys a full pool would reject are dropped before the merge;
//   * for L_pq <= 64 the candidate pool lives in ONE VGPR PAIR of wave 0 (lane i = i-th entry): closest_unexpanded is
//     a ballot, the reject threshold a readlane, and a merge is one pass over the surviving candidates in which a
//     ballot + popcount gives the lower bound (no binary search), the duplicate test and every other entry's shift.
// Within a hop the order of the inserts cannot change the pool (top-L of a set, duplicate keys dropped), so packing a
// hop's candidates is exact; cmps / hops / scanned are counted as the reference counts them.
//
// The kernel is HBM-gather bound (0.5 flop/byte): nothing here goes through MFMA on purpose.
#pragma once
#include <type_traits>

#include "roar_wave.h"

namespace {

// A kernel argument read WHERE IT IS USED -- once per query, or once per hop -- instead of living in scalar registers through every hop
// (the kernels of the benchmark shapes sit at their register budget).  Rounds 2-4 did that with a `volatile` read of the argument
// block: the compiler turns a volatile read into a system-scope VECTOR load (flat_load ... sc0 sc1: past every cache, from the memory
// the block lives in) followed by s_waitcnt vmcnt(0) -- a whole dependent round trip in front of every hop's screening pass, which
// also drained the speculative adjacency fetch issued just before it.  Now: a SCALAR load from the constant address space (the scalar
// cache holds the block after the first touch), which the compiler cannot hoist because the pointer passes through an empty asm.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) RoarSearchParams *RoarKArgs;
__device__ __forceinline__ RoarKArgs kargs() {
    RoarKArgs ka = (RoarKArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    return ka;
}
#define KARG(field) (kargs()->field)
#else
#define KARG(field) (decltype(RoarSearchParams::field){})  // (host pass of the compiler: never executed)
#endif

// LDS visited table, test-and-set of NC (1..3) ids per lane at once (one adjacency round trip; NC = the 64-word chunks
// of the row that hold neighbours -- a row of at most 63 neighbours, the common case on a sparse graph, is one chunk).  Same table as
// visited_filter<VIS_LDS> -- sets (2^k or 3 * 2^k) x 4 ways of u16 tags, (set, tag) <-> id a bijection, so a hit is never false -- but an
// empty way is claimed with a 32-bit compare-and-swap on the half of the set that holds it: no write / barrier /
// re-read round per insertion; a lane that loses a race sees the winner's word in the CAS result and tries the next
// way.  The three probes of a lane are independent, so a round is written branch-free with the three atomics in
// flight together.
// A FULL set (VisitedList semantics, visited_list_pool.h:8-29, must hold for any number of nodes):
//   * exact form (default): the table never gives an entry up.  An id that finds its set full and itself absent is
//     looked up -- and, when new, marked -- in the workgroup's spill list in HBM: u16 epoch tags, one per base point,
//     the reference's own `mass[id] == curV` test.  An id can only be in the spill list if its set was full when it came,
//     and sets only fill up, so a miss in a set with a free way needs no look-up: "visited" is exactly "in the table or
//     tagged", and cmps equals the reference's at every L_pq.  The list's pointer is read from the kernel-argument
//     segment and the epoch from LDS where they are needed, not held in scalar registers through the hop.
//   * forgetful form (option visited_mode = LDS_FORGETFUL; the builder's searches): a victim way is overwritten.  The
//     forgotten node is scored again if it is met again, its key is dropped by the merge as a duplicate or as worse than
//     the pool's last entry: ids / distances / hops are unchanged, cmps can only exceed the reference's, and nothing is
//     ever stored to HBM (a one-word mark costs as much as 2-3.5 lines of a row gather, scripts/probe_cost.hip).
template <int NC>
__device__ __forceinline__ void visited_ldsn(const bool (&isn)[NC], const uint32_t (&nb)[NC], uint16_t *vis, uint32_t sets_mask,
                                             uint32_t sets_log2, uint32_t mult, bool forget, bool protect, bool (&fresh)[NC]) {
    const uint32_t tmask2 = protect ? 0x7FFF7FFFu : 0xFFFFFFFFu;  // (protect: bit 15 of a tag = "its node is in the pool", roar_wave.h TagProtect)
    uint32_t *w[NC];
    uint32_t tag[NC], x[NC], y[NC];
    bool pend[NC];
    uint32_t spillm = 0;  // bit c: chunk c goes to the spill list (a per-lane word, not one more lane mask in scalar registers)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        uint32_t set = nb[c] & sets_mask;
        uint32_t hi = nb[c] >> sets_log2;
        if (mult == 3u) {  // 3 << sets_log2 sets: set = id mod sets, tag = id div sets (exact division by 3 of a 32-bit value)
            const uint32_t q3 = __umulhi(hi, 0xAAAAAAABu) >> 1;
            set |= (hi - 3u * q3) << sets_log2;
            hi = q3;
        } else if (mult == 5u) {  // 5 << sets_log2 sets (round 5: the size between 2^k and 3 * 2^(k-1) that leaves a long pool one more workgroup per CU)
            const uint32_t q5 = __umulhi(hi, 0xCCCCCCCDu) >> 2;
            set |= (hi - 5u * q5) << sets_log2;
            hi = q5;
        }
        tag[c] = hi + 1u;
        w[c] = reinterpret_cast<uint32_t *>(vis + 4 * set);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const uint2 ww = *reinterpret_cast<const uint2 *>(w[c]);
        x[c] = ww.x;
        y[c] = ww.y;
        pend[c] = isn[c];
        fresh[c] = false;
    }
    bool any = false;
#pragma unroll
    for (int c = 0; c < NC; ++c) any |= isn[c];
    while (__ballot(any)) {
        bool cas[NC], low[NC];
        uint32_t oldw[NC], got[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const uint32_t xm = x[c] & tmask2, ym = y[c] & tmask2;
            const uint32_t t0 = xm & 0xFFFFu, t1 = xm >> 16, t2 = ym & 0xFFFFu, t3 = ym >> 16;  // (a protected tag is not 0 either)
            const bool hit = (t0 == tag[c]) | (t1 == tag[c]) | (t2 == tag[c]) | (t3 == tag[c]);
            const bool full = (t0 != 0) & (t1 != 0) & (t2 != 0) & (t3 != 0);
            low[c] = (t0 == 0) | (t1 == 0);
            oldw[c] = low[c] ? x[c] : y[c];
            cas[c] = pend[c] & !hit & !full;
            const bool over = pend[c] & !hit & full;  // the set is full and does not hold this id
            if (over & forget) {
                uint32_t way = tag[c] & 3u;
                if (protect) {  // the first way from there on whose node is not in the pool (all four are: that one after all)
                    const uint32_t pm = ((x[c] >> 15) & 1u) | ((x[c] >> 31) << 1) | (((y[c] >> 15) & 1u) << 2) | ((y[c] >> 31) << 3);
                    const uint32_t rot = ((pm | (pm << 4)) >> way) & 15u;
                    const uint32_t fr = ~rot & 15u;
                    way = (way + (fr ? (uint32_t)__builtin_ctz(fr) : 0u)) & 3u;
                }
                reinterpret_cast<uint16_t *>(w[c])[way] = (uint16_t)tag[c];
            }
            fresh[c] |= over & forget;
            spillm |= (over & !forget) ? (1u << c) : 0u;
            pend[c] = cas[c];  // visited before, victim written, or handed to the spill list: done here
            got[c] = oldw[c];
        }
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (cas[c])
                got[c] = atomicCAS(w[c] + (low[c] ? 0 : 1), oldw[c], oldw[c] | (((oldw[c] & 0xFFFFu) == 0) ? tag[c] : (tag[c] << 16)));
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bool won = cas[c] & (got[c] == oldw[c]);
            fresh[c] |= won;
            pend[c] &= !won;
            if (cas[c] & !won) {  // somebody else took the way: look at the word as it is now
                if (low[c]) x[c] = got[c];
                else y[c] = got[c];
            }
        }
        any = false;
#pragma unroll
        for (int c = 0; c < NC; ++c) any |= pend[c];
    }
    if (__ballot(spillm != 0)) {  // (never in the forgetful form)
#if defined(__HIP_DEVICE_COMPILE__)
        extern __shared__ __align__(16) unsigned char smem_[];
        uint16_t *const sp = reinterpret_cast<uint16_t *>(KARG(tags)) +
                             (u64)blockIdx.x * KARG(tag_stride);
        const uint32_t ep = reinterpret_cast<const uint32_t *>(smem_ + KARG(off_ctrl))[6];
#else
        uint16_t *const sp = nullptr;  // (host pass of the compiler: never executed)
        const uint32_t ep = 0;
#endif
        uint32_t tg[NC];  // the look-ups of a lane's chunks are issued together (ids of one row are distinct: no two lanes share a tag)
#pragma unroll
        for (int c = 0; c < NC; ++c) tg[c] = ((spillm >> c) & 1u) ? (uint32_t)__builtin_nontemporal_load(sp + nb[c]) : ep;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bool fr = tg[c] != ep;  // (a lane that did not look up holds ep)
            if (fr) __builtin_nontemporal_store((uint16_t)ep, sp + nb[c]);
            fresh[c] |= fr;
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

// ---- 8-bit screening in front of the exact scoring ----------------------------------------------------------------------------
// What a hop costs the memory system is counted in 128-byte line requests, and of the ~49 rows a hop scores
// only ~6 have a key under the pool's reject threshold.  The index therefore keeps a second copy of the base rows in 8-bit
// codes -- at 200-d 256 bytes = TWO lines per row instead of seven, at 512-d 528 bytes = five lines instead of sixteen -- and a
// hop first scores its unvisited neighbours on that copy, dropping a neighbour there only if its exact key is CERTAINLY above
// the reject threshold -- which the merge would have rejected (neighbor.h:151-153).  Everything else goes on to the exact pass,
// whose distances, keys and merges are the reference's, bit for bit.  cmps counts the neighbours BEFORE the screen, as the
// reference counts its distance evaluations.  (The threshold is read before the hop's merges: it can only be looser than the
// reference's.)
//
// Inner product (screen8i_pass): BOTH sides in 8 bits, the sum in integers.
//   row:   bytes 0..D-1 = c_i = rint(x_i * 127 / max|x|) in [-127, 127] (two's complement), then s = max|x| / 127 (f32) and two
//          bounds rounded UP to bf16 in one word: B >= |x - s c|_2 (this row's quantisation error, evaluated when the copy is made)
//          and N >= |s c|_2.  With x~ = s c:  x.q = x~.q~ + x~.(q - q~) + (x - x~).q.
//   query: once per query, q~ = t k with k_i = rint(q_i * 127 / max|q|), t = max|q| / 127, E >= |q - q~|_2 + 2^-22 |q| (the second
//          term pays for the three roundings of s t I below), |q| rounded up; k packed behind the f32 query in LDS.
//   hop:   I = sum c_i k_i with v_dot4_i32_i8 -- exact, |I| <= D 127^2 < 2^24 for D <= 1024 -- and
//              exact distance of the row  >=  -(s t I) - (E N + |q| B) - 1e-30     (1e-30: products lost to underflow, also if subnormals were flushed)
//          sixteen codes per instruction quartet instead of sixteen conversions and sixteen FMAs: the launch is bound by instruction
//          issue and by dependent round trips, not by bytes (SQ counters), and 200 bytes of row per 4-lane group leave
//          room for FOUR rows per group and pass -- a whole hop's neighbours in one round trip.
//   A row of zeros has s = 0, B = N = 0: its screened value is exact.  A row with a component that is not finite, or with a scale
//   near the subnormals, has B = +inf and is never dropped; a query like that switches the screen off for itself (|q| = +inf).
//   CENTRED codes (round 4): the copy codes y = x - mu, mu = the mean of the base rows (f32, kept with the copy), not x itself --
//   embedding sets sit in a narrow cone, most of every score is the same mu.q, and a bound in units of |q||x| would then pass nearly
//   every row (measured: pass rate 0.26 at |mu| = 3 x the spread, ~1 at 30 x, against 0.10-0.14 centred).  x.q = mu.q + y.q, the bound
//   above holds for y with N, B measured on y, and the query adds once  qd = m + dm + dx:  m = its f32 value of mu.q, dm >= what that
//   value can differ from the real mu.q by (1.25 (D / 64 + 8) 2^-24 sum |mu_i q_i|: 64 partial sums, six shuffle adds), dx = the part of
//   the exact pass's own summation error that |mu| adds to the |y| the rows carry (1.25 (D / 16 + 6) 2^-24 |q| |mu|):
//              exact distance of the row  >=  -(s t I) - (E N + |q| B) - qd - 1e-30
//
// L2 (screen8_pass): the query stays in f32.
//   row:   u_i in 0..255 with x~_i = lo + s u_i, then s, lo, |x~|^2 and B >= |x - x~| (roar_make_screen8_l2_kernel);
//   hop:   D^ = |x~|^2 - 2 (lo sum q_i + s sum q_i u_i) + |q|^2 in f32, less kappa(D) (|x~|^2 + |q|^2) for the rounding of that
//          expression (the three terms may cancel); exact distance >= (sqrt(D^) - B)^2 when sqrt(D^) > B, else 0.
__device__ __forceinline__ float dot16_u8(const uint4 v, const float *lq16, float acc) {
    const float4 q0 = *reinterpret_cast<const float4 *>(lq16), q1 = *reinterpret_cast<const float4 *>(lq16 + 4),
                 q2 = *reinterpret_cast<const float4 *>(lq16 + 8), q3 = *reinterpret_cast<const float4 *>(lq16 + 12);
    acc = __builtin_fmaf((float)(v.x & 0xFFu), q0.x, acc);
    acc = __builtin_fmaf((float)((v.x >> 8) & 0xFFu), q0.y, acc);
    acc = __builtin_fmaf((float)((v.x >> 16) & 0xFFu), q0.z, acc);
    acc = __builtin_fmaf((float)(v.x >> 24), q0.w, acc);
    acc = __builtin_fmaf((float)(v.y & 0xFFu), q1.x, acc);
    acc = __builtin_fmaf((float)((v.y >> 8) & 0xFFu), q1.y, acc);
    acc = __builtin_fmaf((float)((v.y >> 16) & 0xFFu), q1.z, acc);
    acc = __builtin_fmaf((float)(v.y >> 24), q1.w, acc);
    acc = __builtin_fmaf((float)(v.z & 0xFFu), q2.x, acc);
    acc = __builtin_fmaf((float)((v.z >> 8) & 0xFFu), q2.y, acc);
    acc = __builtin_fmaf((float)((v.z >> 16) & 0xFFu), q2.z, acc);
    acc = __builtin_fmaf((float)(v.z >> 24), q2.w, acc);
    acc = __builtin_fmaf((float)(v.w & 0xFFu), q3.x, acc);
    acc = __builtin_fmaf((float)((v.w >> 8) & 0xFFu), q3.y, acc);
    acc = __builtin_fmaf((float)((v.w >> 16) & 0xFFu), q3.z, acc);
    acc = __builtin_fmaf((float)(v.w >> 24), q3.w, acc);
    return acc;
}

__device__ __forceinline__ int quad_sum_i32(int v) {
    v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);  // lane ^ 1
    v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);  // lane ^ 2
    return v;
}

// Inner product.  D = padded row length: D / 16 full chunks of 16 codes, lane t of a row's group takes chunks t, t + 4, ...; the
// chunk after them holds the last D % 16 (0 or 8) codes (lane 0's), s and the two bounds.  lk = the query's codes in LDS.
// qa = |q| rounded up, qb = t, qc = E.
// CHECK (roar_screen_check_kernel, tests): row c is tested against its OWN threshold chk[c] and the verdict goes to chk[c] (1.0 = kept)
// instead of the compaction.
// meta: roar_screen8_split(D) -- the rows' 16 bytes of floats, a packed array behind the code rows (else they are the row's last chunk)
template <int D, int RR, bool CHECK = false>
// cmeta: non-null -- the candidates' floats are in LDS, beside cid (they came with the adjacency row: RoarSearchParams::adj_meta)
__device__ __forceinline__ void screen8i_pass(const unsigned char *__restrict__ scr, const uint4 *__restrict__ meta, const uint32_t *lk, uint32_t *cid, uint32_t p0, uint32_t m,
                                              float thr_d, float qa, float qb, float qc, uint32_t &w, int t, int g, float *chk = nullptr, const uint2 *cmeta = nullptr) {
    constexpr int NDC = D / 16, NJ = NDC / 4, REM = D % 16;
    static_assert(NDC % 4 == 0 && (REM == 0 || REM == 8), "screening copy: row lengths of 64 k or 64 k + 8 floats");
    constexpr uint32_t STRIDE = roar_screen8_stride(D);
    uint32_t id[RR];
    uint4 x[RR][NJ + 1];
#pragma unroll
    for (int r = 0; r < RR; ++r) {
        const uint32_t c = p0 + 16 * r + g;
        id[r] = cid[c < m ? c : m - 1];
        if constexpr (roar_screen8_split(D)) {
            if (cmeta) {
                const uint2 f = cmeta[c < m ? c : m - 1];
                x[r][NJ] = make_uint4(0u, 0u, f.x, f.y);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RR; ++r) {
        const uint4 *row = reinterpret_cast<const uint4 *>(scr + (u64)id[r] * STRIDE);
#pragma unroll
        for (int j = 0; j < NJ; ++j) x[r][j] = row[t + 4 * j];
        if constexpr (roar_screen8_split(D)) {
            if (!cmeta) x[r][NJ] = meta[id[r]];
        } else {
            x[r][NJ] = row[NDC];
        }
    }
    wave_sync();  // (the ids of this pass are in registers: the compaction below may overwrite their slots)
    int acc[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) acc[r] = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {  // one chunk of the query's codes serves the RR rows
        const uint4 k = *reinterpret_cast<const uint4 *>(lk + 4 * (t + 4 * j));
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            acc[r] = __builtin_amdgcn_sdot4((int)x[r][j].x, (int)k.x, acc[r], false);
            acc[r] = __builtin_amdgcn_sdot4((int)x[r][j].y, (int)k.y, acc[r], false);
            acc[r] = __builtin_amdgcn_sdot4((int)x[r][j].z, (int)k.z, acc[r], false);
            acc[r] = __builtin_amdgcn_sdot4((int)x[r][j].w, (int)k.w, acc[r], false);
        }
    }
    if constexpr (REM == 8) {  // the last 8 codes: lane 0 of the group only
        const uint2 k = *reinterpret_cast<const uint2 *>(lk + 4 * NDC);
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            int a2 = __builtin_amdgcn_sdot4((int)x[r][NJ].x, (int)k.x, 0, false);
            a2 = __builtin_amdgcn_sdot4((int)x[r][NJ].y, (int)k.y, a2, false);
            acc[r] += (t == 0) ? a2 : 0;
        }
    }
    u64 bm[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) {
        const int I = quad_sum_i32(acc[r]);
        const uint32_t bn = x[r][NJ].w;
        const float sc = __uint_as_float(x[r][NJ].z), bnd = __uint_as_float(bn << 16), nrm = __uint_as_float(bn & 0xFFFF0000u);
        // the exact distance is >= lower - qd; qd, the same for every row of the query, sits in the threshold (screen_threshold)
#ifdef ROAR_SCREEN_SLACK_MUL  // measurement builds only: the row side of the bound as codes of fewer bits would have it (profiles/r06_coarser_screen_probe.txt)
        const float lower = -((sc * qb) * (float)I) - (qc * nrm + (float)(ROAR_SCREEN_SLACK_MUL) * qa * bnd) - 1.0e-30f;
#else
        const float lower = -((sc * qb) * (float)I) - (qc * nrm + qa * bnd) - 1.0e-30f;
#endif
        const uint32_t c = p0 + 16 * r + g;
        float th = thr_d;
        if constexpr (CHECK) th = chk[c < m ? c : m - 1];
        const bool keep = t == 0 && c < m && !(lower > th);
        bm[r] = __ballot(keep);
    }
    if constexpr (CHECK) {
        wave_sync();
#pragma unroll
        for (int r = 0; r < RR; ++r)
            if (t == 0 && p0 + 16 * r + g < m) chk[p0 + 16 * r + g] = ((bm[r] >> (4 * g)) & 1ull) ? 1.0f : 0.0f;
    } else {
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            if ((bm[r] >> (4 * g + t)) & 1ull) cid[w + lanes_below(bm[r])] = id[r];
            w += (uint32_t)__popcll(bm[r]);
        }
    }
    wave_sync();
}

// L2.  Row: D codes (D % 16 == 0), then s, lo, |x~|^2, B.  qa = |q|^2, qb = sum q_i.
template <int D, int RR, bool CHECK = false>
__device__ __forceinline__ void screen8_pass(const unsigned char *__restrict__ scr, const uint4 *__restrict__ meta, const float *lq, uint32_t *cid, uint32_t p0, uint32_t m,
                                             float thr_d, float qa, float qb, uint32_t &w, int t, int g, float *chk = nullptr) {
    constexpr int NDC = D / 16, NJ = NDC / 4;
    static_assert(NDC % 4 == 0 && D % 16 == 0, "screening copy, L2: row lengths of 64 k floats");
    constexpr uint32_t STRIDE = roar_screen8_stride(D);
    uint32_t id[RR];
    uint4 x[RR][NJ + 1];
#pragma unroll
    for (int r = 0; r < RR; ++r) {
        const uint32_t c = p0 + 16 * r + g;
        id[r] = cid[c < m ? c : m - 1];
    }
#pragma unroll
    for (int r = 0; r < RR; ++r) {
        const uint4 *row = reinterpret_cast<const uint4 *>(scr + (u64)id[r] * STRIDE);
#pragma unroll
        for (int j = 0; j < NJ; ++j) x[r][j] = row[t + 4 * j];
        if constexpr (roar_screen8_split(D)) x[r][NJ] = meta[id[r]];
        else x[r][NJ] = row[NDC];
    }
    __builtin_amdgcn_sched_barrier(0);
    wave_sync();  // (the ids of this pass are in registers: the compaction below may overwrite their slots)
    u64 bm[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            acc = dot16_u8(x[r][j], lq + 16 * (t + 4 * j), acc);
            __builtin_amdgcn_sched_barrier(0);  // (one chunk's query fragments at a time: hoisting them all costs registers)
        }
        acc += quad_xor1(acc);
        acc += quad_xor2(acc);
        const float sc = __uint_as_float(x[r][NJ].x), lo = __uint_as_float(x[r][NJ].y), n2 = __uint_as_float(x[r][NJ].z),
                    bnd = __uint_as_float(x[r][NJ].w);
        const float xq = __builtin_fmaf(lo, qb, sc * acc);
        const float dlo = (n2 - 2.f * xq + qa) - (float)roar_screen8_l2_rounding(D) * (n2 + qa) - 1.0e-30f;
        const float tt = __builtin_sqrtf(__builtin_fmaxf(dlo, 0.f)) * 0.999999f - bnd;
        const float lower = tt > 0.f ? tt * tt * 0.99999f : 0.f;  // the exact distance is >= lower
        const uint32_t c = p0 + 16 * r + g;
        float th = thr_d;
        if constexpr (CHECK) th = chk[c < m ? c : m - 1];
        const bool keep = t == 0 && c < m && !(lower > th);
        bm[r] = __ballot(keep);
    }
    if constexpr (CHECK) {
        wave_sync();
#pragma unroll
        for (int r = 0; r < RR; ++r)
            if (t == 0 && p0 + 16 * r + g < m) chk[p0 + 16 * r + g] = ((bm[r] >> (4 * g)) & 1ull) ? 1.0f : 0.0f;
    } else {
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            if ((bm[r] >> (4 * g + t)) & 1ull) cid[w + lanes_below(bm[r])] = id[r];
            w += (uint32_t)__popcll(bm[r]);
        }
    }
    wave_sync();
}

// A row is dropped iff (its lower bound without the centre's part) > threshold + qd: the sum is rounded UP (a threshold that is too high
// only keeps more rows).
__device__ __forceinline__ float screen_threshold(float thr_d, float qd) {
    const float v = thr_d + qd;
    return v + __builtin_fabsf(v) * 1.2e-7f;
}

// lq = the f32 query in LDS, its 8-bit codes (inner product) behind it
template <int D, bool IP, bool CHECK = false>
__device__ __forceinline__ uint32_t screen8_rows(const unsigned char *__restrict__ scr, const uint4 *__restrict__ meta, const float *lq, uint32_t *cid, uint32_t m, float thr_d,
                                                 float qa, float qb, float qc, int lane, float *chk = nullptr, const uint2 *cmeta = nullptr) {
    const int t = lane & 3, g = lane >> 2;
    uint32_t w = 0, p0 = 0;
    if constexpr (IP) {
        const uint32_t *lk = reinterpret_cast<const uint32_t *>(lq + ((D + 3) & ~3));
        // rows per group and pass: four up to 256 codes (64 bytes of row per lane), two at 512 (72 bytes)
        constexpr int RB = D <= 256 ? 4 : 2;
        for (; p0 + 16 * (RB / 2) < m; p0 += 16 * RB) screen8i_pass<D, RB, CHECK>(scr, meta, lk, cid, p0, m, thr_d, qa, qb, qc, w, t, g, chk, cmeta);
        if constexpr (RB == 4)
            for (; p0 + 16 < m; p0 += 32) screen8i_pass<D, 2, CHECK>(scr, meta, lk, cid, p0, m, thr_d, qa, qb, qc, w, t, g, chk, cmeta);
        for (; p0 < m; p0 += 16) screen8i_pass<D, 1, CHECK>(scr, meta, lk, cid, p0, m, thr_d, qa, qb, qc, w, t, g, chk, cmeta);
    } else {
        if constexpr (D <= 128)
            for (; p0 + 48 < m; p0 += 64) screen8_pass<D, 4, CHECK>(scr, meta, lq, cid, p0, m, thr_d, qa, qb, w, t, g, chk);
        if constexpr (D <= 256)
            for (; p0 + 16 < m; p0 += 32) screen8_pass<D, 2, CHECK>(scr, meta, lq, cid, p0, m, thr_d, qa, qb, w, t, g, chk);
        for (; p0 < m; p0 += 16) screen8_pass<D, 1, CHECK>(scr, meta, lq, cid, p0, m, thr_d, qa, qb, w, t, g, chk);
    }
    return w;
}

// The query's side of the screen, one wave: inner product: qa = |q| rounded up, qb = t, qc = E, qd = the centre's part (mu = the mean
// the copy's codes are centred on, dim floats in HBM; null: none) and the codes k behind the f32 query in LDS; L2: qa = |q|^2, qb = sum q_i.
template <int SD, bool IP>
__device__ __forceinline__ void screen8_query(float *lq, uint32_t dim, int lane, float &qa, float &qb, float &qc, float &qd,
                                              const float *__restrict__ mu = nullptr) {
    float s2 = 0.f, s1 = 0.f, mx = 0.f;
    for (uint32_t i = lane; i < dim; i += 64) {
        s2 = __builtin_fmaf(lq[i], lq[i], s2);
        s1 += lq[i];
        mx = __builtin_fmaxf(mx, __builtin_fabsf(lq[i]));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s2 += __shfl_xor(s2, o, 64);
        s1 += __shfl_xor(s1, o, 64);
        mx = __builtin_fmaxf(mx, __shfl_xor(mx, o, 64));
    }
    float mq = 0.f, ma = 0.f, m2 = 0.f;  // (a pass of its own: six accumulators through one reduction spill in the kernels that sit at their budget)
    if (IP && mu) {
        for (uint32_t i = lane; i < dim; i += 64) {
            const float u = mu[i];
            mq = __builtin_fmaf(u, lq[i], mq);
            ma = __builtin_fmaf(__builtin_fabsf(u), __builtin_fabsf(lq[i]), ma);
            m2 = __builtin_fmaf(u, u, m2);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            mq += __shfl_xor(mq, o, 64);
            ma += __shfl_xor(ma, o, 64);
            m2 += __shfl_xor(m2, o, 64);
        }
    }
    qc = 0.f;
    qd = 0.f;
    if constexpr (IP) {
        const float tq = mx / 127.0f, inv = mx > 0.f ? 127.0f / mx : 0.f;
        uint32_t *lk = reinterpret_cast<uint32_t *>(lq + ((SD + 3) & ~3));
        float e2 = 0.f;
        for (uint32_t i = lane; i < SD / 4; i += 64) {  // four codes per lane and step
            const float4 v = *reinterpret_cast<const float4 *>(lq + 4 * i);
            const float f[4] = {v.x, v.y, v.z, v.w};
            uint32_t word = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float k = __builtin_rintf(f[c] * inv);
                k = __builtin_fminf(127.f, __builtin_fmaxf(-127.f, k));
                const float d = (f[c] - tq * k) * inv;  // in units of t (the square of a tiny d would underflow)
                e2 = __builtin_fmaf(d, d, e2);
                word |= ((uint32_t)(int)k & 0xFFu) << (8 * c);
            }
            lk[i] = word;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) e2 += __shfl_xor(e2, o, 64);
        qa = __builtin_sqrtf(s2) * 1.0001f;
        qb = tq;
        // |q - t k| rounded up, + 4e-7 |q| for the roundings of t k_i inside d, of s t I and of the last sums, + what the
        // EXACT pass's own f32 sum (16 partial sums of D / 16 products, then the tree: distance.h:179-223) can differ from
        // the real-number inner product by -- the bound has to hold for the distance as the reference computes it
        qc = 1.001f * tq * __builtin_sqrtf(e2) + (4.0e-7f + 1.25f * (float)(SD / 16 + 6) * 5.9604645e-8f) * qa;
        // the centre's part: mu.q as computed here, + what that sum can be off by, + the exact pass's summation error on the mu part of x
        qd = mq + 1.25f * (float)(SD / 64 + 8) * 5.9604645e-8f * 1.001f * ma +
             1.25f * (float)(SD / 16 + 6) * 5.9604645e-8f * 1.001f * qa * (__builtin_sqrtf(m2) * 1.0001f);
        if (!(qd == qd) || !(__builtin_fabsf(qd) <= 3.0e38f)) qa = __builtin_inff();  // (a centre that is not finite cannot happen -- the copy drops it -- but no bound then)
        // no bound is claimed for a query that is not finite, whose squares overflow or whose scale is near the subnormals
        if (!(mx <= 1.0e18f) || !(s2 <= 3.0e38f) || (mx > 0.f && mx < 1.0e-30f)) qa = __builtin_inff();
    } else {
        qa = s2;
        qb = s1;
    }
}

// ---- batch-stream form (SRC == 2; host side: roar_stream.cpp) --------------------------------------------------------------------
// A workgroup does not belong to a launch's batch: it takes queries from the ring of SUBMITTED batches (RoarStreamSlot, pinned host
// memory), in batch order, crosses batch boundaries, and leaves when nothing published is left -- so a batch's last queries run beside
// the next batches' first ones without any launch boundary (the loop being split: tests/test_search_roargraph.cpp:202-212).
// Its block in LDS (32 words at p.off_stream): 0-15 copy of the slot it works on, 16 `want` = the batch number it takes queries from,
// 17 batches published as last seen, 18 batch number + 1 of the copy (0: none).
// Host memory is touched by ONE workgroup per batch (the slot) and by one workgroup at a time (the host's `tail`): what they read is
// shared through device memory (RoarStreamDev; roar_device.h says why).  Memory the host or another launch may rewrite between batches
// (query rows) is read with system-scope loads -- a workgroup that lives across batches never passes the L2 invalidate of a kernel
// start -- and results are stored write-through (system scope), so that "every query of the batch has finished" (one 64-bit add per
// query on the batch's counter cell) means "its results are in device memory" for whoever the host lets read them next.  No ACQUIRE /
// SEQ_CST atomics anywhere here: at system scope they are a `buffer_inv sc0 sc1` -- every workgroup of the XCD loses its L2 lines, hub
// rows and all; every value that something depends on is waited for explicitly.
__device__ __forceinline__ uint32_t sys_load32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void sys_store32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ uint32_t dev_load32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void dev_store32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 lds_ptr(const uint32_t *sl, int w) { return ((u64)sl[w + 1] << 32) | (u64)sl[w]; }

// What the stream's bookkeeping needs from the kernel arguments is read from the kernel-argument segment where it is used, once or twice
// per QUERY, instead of living in scalar registers through every hop (the kernels of the benchmark shapes sit at their register budget).
struct StreamArgs {
    RoarStreamCtl *ctl;
    const RoarStreamSlot *slots;
    RoarStreamDone *done;
    RoarStreamCell *cells;
    RoarStreamDev *dev;
    uint32_t ring_mask;
    uint32_t *sl;  // (plain LDS accesses: every hand-over is between lanes of ONE wave, behind wave_sync -- a volatile access is a flat load and a wait each)
};
__device__ __forceinline__ StreamArgs stream_args() {
    StreamArgs a;
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __align__(16) unsigned char smem_[];
    a.ctl = const_cast<RoarStreamCtl *>(KARG(s_ctl));
    a.slots = KARG(s_slots);
    a.done = KARG(s_done);
    a.cells = KARG(s_cells);
    a.dev = KARG(s_dev);
    a.ring_mask = KARG(s_ring) - 1u;
    a.sl = reinterpret_cast<uint32_t *>(smem_ + KARG(off_stream));
#else
    a = StreamArgs{};  // (host pass of the compiler: never executed)
#endif
    return a;
}

constexpr unsigned long long STREAM_SPIN_LIMIT_TICKS = 200000000ull;  // 2 s: a workgroup waiting for ANOTHER workgroup gives up (RoarStreamCtl::error)
constexpr uint32_t STREAM_TAIL_LOOK_TICKS = 500;  // 5 us of the 100 MHz device clock: how old the device's copy of the host's tail may be
// RoarStreamDev::tail_time = 40 bits of the device clock (never 0: 0 = "not a look at the host") << 24 | the tail (at most 2^16 batches an epoch)
__device__ __forceinline__ unsigned long long stream_clock40() {
    const unsigned long long c = (unsigned long long)wall_clock64() & 0xFFFFFFFFFFull;
    return c ? c : 1ull;
}

// The workgroup has run past the batches it knows to be published (`want`): -> the number of batches published now, or 0 when
// nothing is left and the workgroup has counted itself out (it must then leave).  Lane 0's view, made uniform by the caller.
__device__ __forceinline__ uint32_t stream_more(const StreamArgs &a, uint32_t want) {
    // The device's copy of the host's tail (+ when it was read): good if it shows more batches, or if it is fresh.  Else ONE workgroup
    // reads the host's word (a PCIe round trip) and the others wait for its answer -- they neither ask themselves nor leave on a
    // stale copy (a fifth of a freshly launched grid was gone, in the first form, before the batches the host published
    // microseconds later were seen).
    uint32_t t = 0;
    const unsigned long long spin0 = (unsigned long long)wall_clock64();
    for (;;) {
        const unsigned long long tt = __hip_atomic_load(&a.dev->tail_time, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = (uint32_t)tt & 0xFFFFFFu;
        if (t > want) return t;
        if ((unsigned long long)wall_clock64() - spin0 > STREAM_SPIN_LIMIT_TICKS) {
            sys_store32(&a.ctl->error, 3u);
            break;
        }
        if ((tt >> 24) != 0 && ((stream_clock40() - (tt >> 24)) & 0xFFFFFFFFFFull) <= STREAM_TAIL_LOOK_TICKS) break;
        if (dev_load32(&a.dev->checking) == 0u && atomicCAS(&a.dev->checking, 0u, 1u) == 0u) {
            const uint32_t h = sys_load32(&a.ctl->tail);  // (one workgroup at a time: the copy is simply replaced, the host's tail only grows)
            (void)atomicExch(&a.dev->tail_time, (stream_clock40() << 24) | (unsigned long long)h);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            (void)atomicExch(&a.dev->checking, 0u);
            t = h > t ? h : t;
            break;
        }
        __builtin_amdgcn_s_sleep(32);
    }
    if (t > want) return t;
    // Nothing left.  Leave -- unless this is the last workgroup alive: that one tells the host how many have left and then looks at the
    // host's tail itself (the host publishes a batch and THEN reads `exited` to decide whether workgroups must be started for it: one
    // of the two sees the other's write; roar_search_stream_wait also starts workgroups when it finds a batch pending and too few
    // alive: a lost race costs time, never a batch).  The host's count of who is alive is refreshed by every 64th workgroup to leave
    // (a posted write each), by the last one, and with every batch.
    const unsigned long long se = atomicAdd(&a.dev->started_exited, 1ull) + 1ull;
    const uint32_t ex = (uint32_t)se, started = (uint32_t)(se >> 32);
    if (ex >= started) {
        sys_store32(&a.ctl->exited, ex);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the count is on its way before the tail is sampled)
        const uint32_t again = sys_load32(&a.ctl->tail);
        if (again > want) {
            (void)atomicExch(&a.dev->tail_time, (stream_clock40() << 24) | (unsigned long long)again);
            (void)atomicAdd(&a.dev->started_exited, ~0ull);  // (back in: one fewer has left)
            return again;
        }
    } else if ((ex & 63u) == 0u) {
        sys_store32(&a.ctl->exited, ex);
    }
    return 0u;
}

// Batch b's slot: host memory -> the device's mirror, by the workgroup that wins the election on the batch's state word (two PCIe round
// trips: the 64 bytes, then seq1 once more -- the host's seqlock).  -> 0 somebody else has it in hand (or had), 2 fetched (w = the slot,
// lanes 0-15), 3 the host's slot already held a later batch.  Wave-uniform.
__device__ __forceinline__ uint32_t stream_fetch(const StreamArgs &a, int lane, uint32_t b, uint32_t &w) {
    uint32_t st = 0;
    if (lane == 0) st = atomicCAS(&a.dev->state[b], 0u, 1u);
    if (uni(st) != 0u) return 0u;
    const RoarStreamSlot *slot = a.slots + (b & a.ring_mask);
    w = 0;
    if (lane < 16) w = sys_load32(reinterpret_cast<const uint32_t *>(slot) + lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the second look at seq1 must be sampled after the fields)
    uint32_t seq_again = 0;
    if (lane == 0) seq_again = sys_load32(&slot->seq1);
    const bool ok = lane_u32(w, 15) == b + 1u && uni(seq_again) == b + 1u;
    if (ok && lane < 16) dev_store32(reinterpret_cast<uint32_t *>(a.dev->mirror + b) + lane, w);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) (void)atomicExch(&a.dev->state[b], ok ? 2u : 3u);
    return ok ? 2u : 3u;
}

// One call per query: count the query that has just finished (`finish`: its result stores have been ISSUED, write-through) and take
// the next one.  -> true: q = a query of batch sl[16], whose slot is in sl[0..15]; false: nothing published is left and the workgroup
// has counted itself out (RoarStreamDev::started_exited).  Wave-uniform.
//
// What this costs is dependent round trips to device memory at agent scope -- ~4 us each while the chip is saturated with row gathers,
// against ~450 us for a query -- so they are counted: the first form (take, slot, finish as separate steps: eight of them per
// 1 250-query batch boundary) handed queries out 10 % slower than the per-launch form.  Here: the next take travels WITH the result
// stores of the finished query (one wait for both); a take is ONE bundle of independent requests {add on the batch's `next`, the batch's
// state word and mirrored slot if this workgroup has not got them, the frontier}; a finished query is counted with an add that nobody
// waits for -- the workgroup that took the batch's LAST query polls the counter when its own query is done and reports the batch
// (three or four workgroups of 3 072 poll at any time with 1 250-query batches).  A workgroup that stays within a batch: one round
// trip per query, as the per-launch form; across a batch boundary: two.
__device__ __forceinline__ bool stream_next(int lane, bool finish, bool was_short, uint32_t &q) {
    const StreamArgs a = stream_args();
    uint32_t *const sl = a.sl;
    uint32_t c = sl[16];
    uint32_t idx = 0, st = 0, f = 0, w = 0;
    bool have = false;  // a bundle for batch c has been issued (idx, st, f, w)
    auto bundle = [&]() {
        const bool need = sl[18] != c + 1u;
        if (lane == 0) {
            idx = atomicAdd(&a.cells[c].next, 1u);
            f = dev_load32(&a.dev->frontier);
            if (need) st = dev_load32(&a.dev->state[c]);
        }
        if (need && lane < 16) w = dev_load32(reinterpret_cast<const uint32_t *>(a.dev->mirror + c) + lane);
        have = true;
    };
    if (finish) {
        const uint32_t b_n = c, nq_n = sl[12];
        const bool claimer = sl[20] != 0u;
        if (!claimer && c < sl[17]) bundle();  // (in flight together with the result stores)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every result store of the finished query has been acknowledged by memory
        if (lane == 0) {
            (void)atomicAdd(&a.cells[b_n].done_short, 1ull + (was_short ? (1ull << 32) : 0ull));
            if (claimer) {  // this workgroup took the batch's last query: it reports the batch once every query of it has been counted
                unsigned long long d;
                const unsigned long long spin0 = (unsigned long long)wall_clock64();
                for (;;) {
                    d = atomicAdd(&a.cells[b_n].done_short, 0ull);  // (an atomic like the adds it waits for, not a LOAD of their line: RoarStreamDev::state)
                    if ((uint32_t)d >= nq_n) break;
                    if ((unsigned long long)wall_clock64() - spin0 > STREAM_SPIN_LIMIT_TICKS) {
                        sys_store32(&a.ctl->error, 2u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(64);
                }
                RoarStreamDone *dn = a.done + (b_n & a.ring_mask);
                __hip_atomic_store(&dn->t_done, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                sys_store32(&dn->nshort, (uint32_t)(d >> 32));
                sys_store32(&a.ctl->exited, (uint32_t)atomicAdd(&a.dev->started_exited, 0ull));  // (who has left by now; an add of zero, not a load: RoarStreamDev::state)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the record's fields before its seq1_done; every result store of the batch
                if ((uint32_t)d >= nq_n) sys_store32(&dn->seq1_done, b_n + 1u);  // was acknowledged before its workgroup counted the query)
            }
        }
    }
    for (;;) {
        if (!have) {
            if (c >= sl[17]) {  // past what this workgroup knows to be published
                uint32_t t = 0;
                if (lane == 0) t = stream_more(a, c);
                t = uni(t);
                if (t == 0u) return false;
                if (lane == 0) sl[17] = t;
                wave_sync();
            }
            bundle();
        }
        have = false;
        idx = uni(idx);
        f = uni(f);
        bool exhausted = false;
        if (sl[18] != c + 1u) {  // the batch's slot: from the device's mirror (state 2), else fetched from the host by ONE workgroup
            st = uni(st);
            // (the bundle read the state word and the mirror entry side by side: "there" may have been sampled after the entry was written
            // and the entry before -- an entry is taken only if it names its batch in its last word, else it is read again, now behind the state)
            bool got = st == 2u && lane_u32(w, 15) == c + 1u;
            if (st == 0u) {  // nobody has fetched it: this workgroup does, if it wins the election
                uint32_t w2 = 0;
                st = stream_fetch(a, lane, c, w2);
                if (st == 2u) w = w2, got = true;
                else if (st == 0u) {
                    if (lane == 0) st = dev_load32(&a.dev->state[c]);
                    st = uni(st);
                }
            }
            if (st == 1u) {  // somebody else is fetching it: two PCIe round trips away from done
                const unsigned long long spin0 = (unsigned long long)wall_clock64();
                while (st == 1u) {
                    __builtin_amdgcn_s_sleep(32);
                    if (lane == 0) {
                        st = dev_load32(&a.dev->state[c]);
                        if (st == 1u && (unsigned long long)wall_clock64() - spin0 > STREAM_SPIN_LIMIT_TICKS) {
                            sys_store32(&a.ctl->error, 1u);
                            st = 3u;
                        }
                    }
                    st = uni(st);
                }
            }
            if (st == 2u && !got) {  // there for good: read it (again)
                if (lane < 16) w = dev_load32(reinterpret_cast<const uint32_t *>(a.dev->mirror + c) + lane);
                got = lane_u32(w, 15) == c + 1u;
            }
            if (got) {
                if (lane < 16) sl[lane] = w;
                if (lane == 0) sl[18] = c + 1u;
                wave_sync();
            } else {
                exhausted = true;  // (3: the host's slot already held a later batch -- this one was complete long ago)
            }
        }
        if (exhausted || idx >= sl[12]) {
            // nothing left in this batch: on to the first batch that may have something -- the frontier the others have reached, not
            // simply the next one (with 1 250-query batches the chip's workgroups span two or three batches)
            c = f > c + 1u ? f : c + 1u;
            if (lane == 0) sl[16] = c;
            wave_sync();
            continue;
        }
        // Whoever takes a batch's second / third query fetches the slot of the next / next but one batch, if it is published and nobody
        // has: every workgroup that finishes a query needs one from the FRONTIER batch, and while that batch's slot is two PCIe round
        // trips away nothing starts anywhere (measured: 1 250-query batches handed out 9 % slower than 10 000-query ones; this fetch
        // happens a batch or two ahead instead, off everybody's path but its own)
        if ((idx == 1u || idx == 2u) && c + idx < sl[17]) {
            uint32_t stn = 1u;
            if (lane == 0) stn = dev_load32(&a.dev->state[c + idx]);
            if (uni(stn) == 0u) {
                uint32_t w2 = 0;
                (void)stream_fetch(a, lane, c + idx, w2);
            }
        }
        if (lane == 0) {
            sl[20] = (idx + 1u == sl[12]) ? 1u : 0u;
            if (idx == 0)
                __hip_atomic_store(&a.done[c & a.ring_mask].t_first, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (idx + 1u == sl[12]) {  // the batch's last query: nothing left in it for anybody else
                (void)atomicMax(&a.dev->frontier, c + 1u);
                sys_store32(&a.done[c & a.ring_mask].seq1_claimed, c + 1u);
            }
        }
        q = idx;
        return true;
    }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory queue, which would
// make wave 0 sit out the round trip of its speculative adjacency fetch before the other waves may start scoring;
// everything the waves of a query hand each other (candidate ids, keys, control words) lives in LDS.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ctrl words (LDS) of the multi-wave forms: 0 query, 1 candidates of the round trip (NONE = query finished),
// 2 survivors appended by the waves, 3 pool full, 4-5 reject threshold (register-pool form)
// STATS: the instantiation that also counts what a query asks the memory side for (RoarSearchParams::out_traffic) -- the same
// hop, plus one LDS add of lane 0 where something is fetched.  Its own instantiation because the kernels of the benchmark
// shapes sit exactly at their register budget (the counting costs the one-wave 200-d kernel 12 bytes of scratch inside the hop);
// what a query fetches does not depend on timing, so the counts are those of the plain kernel on the same launch plan.
// PROT: the instantiation whose forgetful tag table keeps the tags of pool members (roar_wave.h TagProtect; LDS pool, one wave per query).
// Its own instantiation for the same reason: the policy's code costs a launch that does not use it 2 % (SIFT shape, L_pq = 170).
// SRC: where the queries come from -- 0 the launch's own batch (p.queries, p.nq); 1 the same while a host thread is still packing the
// rows (waits on p.ready; pageable host query blocks); 2 the batch-stream form: a ring of submitted batches (stream_next).
// ANN: the screened neighbours' floats (scale, bounds) come with the adjacency row (RoarSearchParams::adj_meta) instead of from a 16-byte
// read per neighbour beside the code rows.
template <int MODE, bool IP, int NCH, bool TAIL, int W, bool RP, bool PROF, bool HUB, int SRC = 0, bool STATS = false, bool PROT = false, bool ANN = false>
__device__ __forceinline__ void search_v2(const RoarSearchParams &p) {
    constexpr bool GATED = SRC == 1, STREAM = SRC == 2;
    static_assert(!STREAM || (W == 1 && !PROF && !HUB && !STATS && MODE == VIS_LDS), "batch-stream form: one wave per query, on-chip tag table");
    constexpr int R = 2;  // base rows per 4-lane group and scoring pass
    // padded row length with an 8-bit screening copy in front of the exact scoring (inner product, 200-d and 512-d), else 0
    constexpr int SD = (NCH == 8 && !TAIL) ? 128 : (NCH == 32 && !TAIL) ? 512 : (IP && NCH == 12 && TAIL) ? 200 : 0;
    static_assert(!ANN || (W == 1 && !HUB && !PROF && !STATS && SRC != 1 && IP && MODE == VIS_LDS && roar_screen8_split(SD > 0 ? SD : 1)),
                  "edge floats: inner product, code rows of whole lines, the one-wave form");
    // (a build with R = 1 and four waves per SIMD for launches with the screen on -- sixteen workgroups per CU: with the screen's f32
    // form it spilled inside the hop, 2.94 against 1.83 ms at L_pq = 50; with the integer form it fits but for 4-8 registers and buys
    // nothing: 2 048 tags, sixteen / twelve per CU, L_pq = 50 1.71 / 1.72 ms, 100 3.02 / 3.09, 500 14.0 / 14.8, 1 000 36.3 / 34.9; dropped)
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = (W == 1) ? 0 : (int)(threadIdx.x >> 6);
    u64 *pool = reinterpret_cast<u64 *>(smem + p.off_pool);
    u64 *ck = reinterpret_cast<u64 *>(smem + p.off_ck);
    uint32_t *cid = reinterpret_cast<uint32_t *>(smem + p.off_cid);
    float *lq = reinterpret_cast<float *>(smem + p.off_q);
    uint16_t *vis = reinterpret_cast<uint16_t *>(smem + p.off_vis);
    // (several waves per query hand words to each other through ctrl: volatile there; one wave: plain LDS accesses behind wave_sync -- a
    // volatile access is compiled as a flat load and a wait each, four of them in front of every hop's screening pass in rounds 3-4)
    using CtrlWord = std::conditional_t<W == 1, uint32_t, volatile uint32_t>;
    CtrlWord *ctrl = reinterpret_cast<CtrlWord *>(smem + p.off_ctrl);
    // HUB form (long pools): rows come from the hub-annotated adjacency -- word 0 = degree | hubs << 16, the row's hub
    // neighbours first, then the others, then one u16 bitmap slot per hub neighbour -- in one round trip of 4 x 64 words
    // (such a graph has no row longer than 170); hub neighbours are tested and marked in an exact LDS bitmap (hbits,
    // 65 536 bits), only the others go to the tag table / the HBM tags.
    uint32_t *rowbuf = reinterpret_cast<uint32_t *>(smem + p.off_rowbuf);
    uint32_t *hbits = reinterpret_cast<uint32_t *>(smem + p.off_hbits);
    const uint32_t *const ADJ = HUB ? p.adj_hub : p.adj;
    const uint32_t AS = HUB ? p.hub_stride : p.adj_stride;
    const uint32_t L = p.L;
    const uint32_t sets_log2 = p.vis_sets_log2 & 31u;  // bit 31: three times that many sets; bit 30: forgetful form; bit 29: see `protect`
    const uint32_t smult = (p.vis_sets_log2 >> 31) ? 3u : ((p.vis_sets_log2 >> 28) & 1u) ? 5u : 1u;  // (bit 28: five times that many sets)
    const bool forget = ((p.vis_sets_log2 >> 30) & 1u) != 0;
    const bool protect = PROT && !RP && forget && ((p.vis_sets_log2 >> 29) & 1u) != 0;  // bit 29: tags carry a "node is in the pool" bit
    // (exact LDS form: the workgroup's spill list -- u16 epoch tags in HBM -- has its epoch in LDS, ctrl[6])
    const uint32_t sets_mask = (1u << sets_log2) - 1u;
    const uint32_t NONE = 0xFFFFFFFFu;
    uint8_t *tags = (MODE == VIS_EXACT) ? p.tags + (u64)blockIdx.x * p.tag_stride : nullptr;
    uint32_t epoch = (MODE == VIS_EXACT) ? p.epochs[blockIdx.x] : 0u;
    auto block_sync = [&]() {
        if constexpr (W == 1) wave_sync();
        else lds_barrier();
    };
    uint32_t ready_seen = 0;
    if (MODE == VIS_LDS && !forget) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (threadIdx.x == 0) ctrl[6] = (KARG(epochs))[blockIdx.x];
#endif
        block_sync();
    }

    if constexpr (STREAM) {
        if (lane == 0) {
            const StreamArgs a = stream_args();
            a.sl[16] = p.s_start, a.sl[17] = p.s_tail, a.sl[18] = 0u, a.sl[20] = 0u;
            (void)atomicAdd(&a.dev->started_exited, 1ull << 32);
            // (what this launch was told at its start, sl[17], is not shared: the device's copy of the host's tail is written by looks at the host only)
        }
        wave_sync();
    }

    if constexpr (!STREAM) {
        // The launch cell must be as its last user left it: that launch's stamp in the cell's stamp word (the cell is then zero).  Anything else means
        // that launch did not finish -- a device fault, or the ring went round while it was still running -- and a search on its
        // leftovers would skip queries silently: refuse instead (every workgroup sees the same word and leaves; the host-buffer calls
        // turn the 0xFFFFFFFF into ROAR_ERR_DEVICE, roar_launch.cpp says how a caller of the device-buffer form recovers).
#if defined(__HIP_DEVICE_COMPILE__)
        if (dev_load32(KARG(stamp)) != KARG(cell_prev)) {
            uint32_t *const o_short = KARG(short_count);
            if (threadIdx.x == 0 && blockIdx.x == 0 && o_short) *o_short = 0xFFFFFFFFu;
            return;
        }
#endif
    }
    [[maybe_unused]] bool s_fin = false, s_short = false;  // (batch-stream form: a finished query waits to be counted, with the next take)
    for (;;) {
        uint32_t q = 0;
        if constexpr (STREAM) {
            const long long tk0 = wall_clock64();
            const bool more = stream_next(lane, s_fin, s_short, q);
            if (lane == 0) {  // (what the dequeues cost, all workgroups together: two adds nobody waits for)
                const StreamArgs a = stream_args();
                (void)atomicAdd(&a.dev->take_ticks, (unsigned long long)(wall_clock64() - tk0));
                (void)atomicAdd(&a.dev->take_calls, 1ull);
                a.sl[22] = (uint32_t)wall_clock64();
            }
            if (!more) return;  // (counted out in RoarStreamDev; the launch cell is not used)
        } else if constexpr (W == 1) {
            if (lane == 0) q = atomicAdd(p.counter, 1u);
            q = uni(q);
        } else {
            if (threadIdx.x == 0) ctrl[0] = atomicAdd(p.counter, 1u);
            lds_barrier();
            q = ctrl[0];
        }
        if (!STREAM && q >= p.nq) break;
        if (GATED && q >= ready_seen) {
            // the host is still packing the query block behind this launch: wait until row q is there (the counter only
            // grows, so a workgroup looks at it again only when it runs past what it has seen)
            for (;;) {
                ready_seen = __hip_atomic_load(p.ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                if (ready_seen > q) break;
                __builtin_amdgcn_s_sleep(100);
            }
        }
        long long tq0 = 0, t_adj = 0, t_filter = 0, t_score = 0, t_merge = 0, t_sel = 0;
        [[maybe_unused]] long long t_mphase[3] = {0, 0, 0};  // inside the merge: lower bounds, ranks, shift (roar_wave.h merge_candidates)
        uint32_t pf_hit = 0, n_rt = 0, n_surv = 0;
        PassProf pprof;
        if (PROF) tq0 = (long long)__builtin_readcyclecounter();

        // query -> LDS (tests/test_search_roargraph.cpp:205 passes aligned_query_data + i * q_dim)
        if constexpr (STREAM) {  // (rows another launch or a copy may have rewritten since this workgroup last saw them: system-scope loads)
            uint32_t *const sl = stream_args().sl;
            const uint32_t *qp = reinterpret_cast<const uint32_t *>(lds_ptr(sl, 0)) + (u64)q * sl[13];
            for (uint32_t i = threadIdx.x; i < p.dim; i += 64) reinterpret_cast<uint32_t *>(lq)[i] = sys_load32(qp + i);
        } else {
            const float *qp = p.queries + (u64)q * p.qstride;
            for (uint32_t i = threadIdx.x * 4; i < p.dim; i += 256 * W)
                *reinterpret_cast<float4 *>(lq + i) = *reinterpret_cast<const float4 *>(qp + i);
        }
        if (MODE == VIS_LDS) {  // fresh visited table
            const uint32_t n16 = ((4u * smult) << sets_log2) >> 3;  // sets * 4 ways * 2 B / 16 B
            for (uint32_t i = threadIdx.x; i < n16; i += 64 * W) reinterpret_cast<uint4 *>(vis)[i] = make_uint4(0, 0, 0, 0);
        }
        if (HUB)
            for (uint32_t i = threadIdx.x; i < 512u; i += 64 * W) reinterpret_cast<uint4 *>(hbits)[i] = make_uint4(0, 0, 0, 0);
        if (MODE == VIS_LDS && !forget) {  // VisitedList::reset (visited_list_pool.h:20-26) of the spill list, u16 epoch
            uint32_t e16 = ctrl[6] + 1u;
            if (e16 == 65536u) {  // (every 65 535 queries of this workgroup slot)
#if defined(__HIP_DEVICE_COMPILE__)
                const u64 bytes = KARG(tag_stride) * 2u;
                uint8_t *sp = KARG(tags) + (u64)blockIdx.x * bytes;
                for (u64 i = (u64)threadIdx.x * 16; i < bytes; i += 1024 * W) *reinterpret_cast<uint4 *>(sp + i) = make_uint4(0, 0, 0, 0);
                __threadfence();
#endif
                e16 = 1;
            }
            block_sync();
            if (threadIdx.x == 0) ctrl[6] = e16;
        }
        if (MODE == VIS_EXACT) {  // VisitedList::reset (visited_list_pool.h:20-26) with a u8 epoch
            epoch++;
            if (epoch == 256u) {
                for (u64 i = (u64)threadIdx.x * 16; i < p.tag_stride; i += 1024 * W)
                    *reinterpret_cast<uint4 *>(tags + i) = make_uint4(0, 0, 0, 0);
                __threadfence();
                epoch = 1;
            }
        }
        block_sync();

        // the query's side of the screen (screen8i_pass / screen8_pass), kept in LDS (ctrl[8..10]), not in registers through the hop:
        // inner product: |q| rounded up, t, E and the codes k behind the f32 query; L2: |q|^2 and sum q_i
        if constexpr (SD > 0) {
#if defined(__HIP_DEVICE_COMPILE__)
            const bool screen_on = KARG(screen) != nullptr;
#else
            const bool screen_on = false;
#endif
            if (wid == 0 && screen_on) {  // (the LDS behind the query is there for the codes only when the launch has a screening copy)
                float qa, qb, qc, qd;
#if defined(__HIP_DEVICE_COMPILE__)
                const float *mu = KARG(screen_mu);
#else
                const float *mu = nullptr;
#endif
                screen8_query<SD, IP>(lq, p.dim, lane, qa, qb, qc, qd, mu);
                if (lane == 0) {
                    ctrl[8] = __float_as_uint(qa);
                    ctrl[9] = __float_as_uint(qb);
                    ctrl[10] = __float_as_uint(qc);
                    ctrl[11] = __float_as_uint(qd);
                }
                wave_sync();
            }
        }
        // state of the sequential part (wave 0).  Pool: `pk` (register form, L_pq <= 64) or pool[] in LDS with the
        // cursor `cur` = index of the closest unexpanded entry.
        u64 pk = ~0ull;
        uint32_t psize = 1, cur = 0, cmps = 0, hops = 0, scanned = 0;
        // LDS pool: every entry below `front` is expanded except at most `ub` of them (the merge counts what it puts there): once those are
        // taken, the search for the next unexpanded entry starts AT the front instead of walking the expanded entries between a survivor
        // that landed low and the front, 64 per LDS round trip (L_pq = 1 000: four of them per hop, round 5)
        [[maybe_unused]] uint32_t front = 0, ub = 0;
        // what the query asks the memory side for, beside cmps (RoarSearchParams::out_traffic): one packed LDS word, ctrl[12..13] =
        // rows scored on the 8-bit copy | rows that passed it << 24 | 64-word adjacency chunks fetched << 44 -- one LDS add of
        // lane 0 where something is fetched, no register lives through the hop for it
        u64 *const traffic = reinterpret_cast<u64 *>(const_cast<uint32_t *>(ctrl + 12));
        auto count_traffic = [&](u64 v) {
            if constexpr (STATS)
                if (lane == 0) atomicAdd(traffic, v);
        };
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, pf = NONE, vcur = p.ep, deg = 0, wpos = 1;  // wpos > deg: the current row is exhausted
        // ANN: entry w of the row's "edge floats" travels with adjacency word w, in the same lane (RoarSearchParams::adj_meta; rows of at most 127 neighbours)
        [[maybe_unused]] u64 m0 = 0, m1 = 0;
        [[maybe_unused]] uint2 *const cmeta = reinterpret_cast<uint2 *>(smem + (ANN ? p.off_cmeta : 0u));
        uint32_t nhub = 0;  // HUB: hub neighbours of the current row (they come first)
        const uint32_t *arow = ADJ;
        if (wid == 0) {
            // the entry point is always the first expansion: its adjacency row travels with its base row
            arow = ADJ + (u64)p.ep * AS;
            // (plain form: the first 64 words only -- degree + up to 63 neighbours, the common case on a sparse graph; the other
            // chunks follow once the degree says they hold neighbours.  Hub form: ids and slots, the whole row.)
            a0 = arow[lane];
            if constexpr (ANN) m0 = (KARG(adj_meta) + (u64)p.ep * AS)[lane];
            if (HUB) {
                a1 = arow[64 + lane];
                if (AS > 128u) a2 = arow[128 + lane];
                a3 = arow[192 + lane];
            }
            pf = p.ep;
            // entry point: scored, inserted, NOT marked visited (index_bipartite.cpp:2337-2353)
            if (lane == 0) {
                cid[0] = p.ep;
                if constexpr (STATS) *traffic = (u64)(HUB ? (AS > 128u ? 4u : 3u) : 1u) << 44;
            }
            wave_sync();
            uint32_t one = 0;
            score_rows<IP, NCH, TAIL, 1, false>(p.base, p.ld, p.dim, lq, cid, ck, 1, ~0ull, &one, lane);
            wave_sync();
            if constexpr (RP) {
                if (lane == 0) pk = ck[0];
            } else {
                if (lane == 0) pool[0] = ck[0];
            }
            wave_sync();
        }

        for (;;) {
            uint32_t m = 0;
            bool done = false;
            if (wid == 0) {
                long long c0 = 0, c1 = 0;
                if (PROF) c0 = (long long)__builtin_readcyclecounter();
                uint32_t nextv = NONE;  // closest unexpanded entry once the current one is flagged (prefetch target)
                if (wpos > deg) {
                    // closest_unexpanded (neighbor.h:185-192): flag the entry, the cursor moves past flagged ones
                    uint32_t v = 0;
                    if constexpr (RP) {
                        const u64 um = __ballot((uint32_t)lane < psize && !(pk & 1ull));
                        if (um == 0) {  // !has_unexpanded_node, :2356
                            done = true;
                        } else {
                            const int c = __ffsll((long long)um) - 1;
                            v = (uint32_t)lane_u64(pk, c) >> 1;
                            if (lane == c) pk |= 1ull;
                            const u64 rest = um & (um - 1);
                            if (rest) nextv = (uint32_t)lane_u64(pk, __ffsll((long long)rest) - 1) >> 1;
                        }
                    } else {
                        if (cur >= psize) {
                            done = true;
                        } else {
                            // ONE read serves the entry itself (lane 0), the search for the next unexpanded entry when it starts right behind
                            // it (lanes 1-63: the common case) and that entry's id -- the target of the speculative adjacency fetch, which
                            // rounds 3-5 read again from pool[cur] after the filter: three or four dependent LDS round trips per hop less
                            const uint32_t w0 = cur + (uint32_t)lane;
                            const u64 ew = pool[w0 < psize ? w0 : cur];
                            const u64 e = lane_u64(ew, 0);
                            v = (uint32_t)e >> 1;
                            if (lane == 0) pool[cur] = e | 1ull;
                            wave_sync();
                            if (cur < front) ub = ub ? ub - 1u : 0u;
                            else front = cur + 1u, ub = 0;
                            uint32_t found = psize;
                            uint32_t c = (ub == 0 && front > cur + 1u) ? front : cur + 1u;
                            if (c == cur + 1u) {
                                const u64 b = __ballot(lane >= 1 && w0 < psize && !(ew & 1ull));
                                if (b) {
                                    const int l = __ffsll((long long)b) - 1;
                                    found = cur + (uint32_t)l;
                                    nextv = lane_u32((uint32_t)ew, l) >> 1;
                                }
                                c = cur + 64u;
                            }
                            if (found == psize) {
                                for (; c < psize; c += 64) {
                                    const uint32_t kidx = c + (uint32_t)lane;
                                    const u64 ek = pool[kidx < psize ? kidx : cur];
                                    const u64 b = __ballot(kidx < psize && !(ek & 1ull));
                                    if (b) {
                                        const int l = __ffsll((long long)b) - 1;
                                        found = c + (uint32_t)l;
                                        nextv = lane_u32((uint32_t)ek, l) >> 1;
                                        break;
                                    }
                                }
                            }
                            cur = found;
                            if (found >= front) front = found, ub = 0;  // (nothing unexpanded below it)
                        }
                    }
                    if (!done) {
                        ++hops;
                        if (PROF) {
                            c1 = (long long)__builtin_readcyclecounter();
                            t_sel += c1 - c0;
                            c0 = c1;
                        }
                        // adjacency row: word 0 = degree, word w (>= 1) = neighbour w - 1
                        arow = ADJ + (u64)v * AS;
                        vcur = v;
                        if (v != pf) {  // the speculative fetch went to another row: pay the round trip now
                            a0 = arow[lane];
                            if constexpr (ANN) m0 = (KARG(adj_meta) + (u64)v * AS)[lane];
                            if (HUB) {
                                a1 = arow[64 + lane];
                                if (AS > 128u) a2 = arow[128 + lane];
                                a3 = arow[192 + lane];
                            }
                            count_traffic((u64)(HUB ? (AS > 128u ? 4u : 3u) : 1u) << 44);
                            if (PROF) n_rt++;
                        } else if (PROF) {
                            pf_hit++;
                        }
                        deg = uni(a0);
                        if (HUB) {
                            nhub = deg >> 16;
                            deg &= 0xFFFFu;
                        } else if (deg > 63u) {  // a long row: its other chunks (a second round trip, rare on a sparse graph)
                            a1 = arow[64 + lane];
                            if constexpr (ANN) m1 = (KARG(adj_meta) + (u64)v * AS)[64 + lane];
                            if (deg > 127u) a2 = arow[128 + lane];
                            count_traffic((u64)(deg > 127u ? 2u : 1u) << 44);
                        }
                        scanned += deg;
                        wpos = 0;
                    }
                } else if constexpr (RP) {  // a row longer than 191 entries continues (its next words are already here)
                    const u64 um = __ballot((uint32_t)lane < psize && !(pk & 1ull));
                    if (um) nextv = (uint32_t)lane_u64(pk, __ffsll((long long)um) - 1) >> 1;
                } else {
                    if (cur < psize) nextv = (uint32_t)pool[cur] >> 1;
                }
                if (!done) {
                    // visited filter over the whole round trip (index_bipartite.cpp:2378-2385), survivors packed in cid
                    // (chunks of 64 words; only those that hold neighbours are looked at: a row of at most 63
                    // neighbours -- the common case on a sparse graph -- costs a third of the filter's instructions)
                    auto run_filter = [&](auto nc_tag) {
                        constexpr int NC = decltype(nc_tag)::value;
                        uint32_t nb[NC];
                        bool isn[NC];
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            const uint32_t w = wpos + 64u * (uint32_t)c + (uint32_t)lane;
                            isn[c] = w >= 1 && w <= deg;
                            nb[c] = isn[c] ? (c == 0 ? a0 : c == 1 ? a1 : a2) : 0u;
                        }
                        if (PROF) {
                            c1 = (long long)__builtin_readcyclecounter();
                            t_adj += c1 - c0;
                            c0 = c1;
                        }
                        bool hfresh[NC];
#pragma unroll
                        for (int c = 0; c < NC; ++c) hfresh[c] = false;
                        if (HUB) {
                            // the slots sit behind the ids, two per word: pass the four registers through LDS to index them
                            rowbuf[lane] = a0, rowbuf[64 + lane] = a1, rowbuf[128 + lane] = a2, rowbuf[192 + lane] = a3;
                            wave_sync();
#pragma unroll
                            for (int c = 0; c < NC; ++c) {
                                const uint32_t j = 64u * (uint32_t)c + (uint32_t)lane - 1u;  // neighbour index of word w
                                const bool hub = isn[c] && j < nhub;
                                if (hub) {
                                    const uint32_t slot = (rowbuf[1 + deg + (j >> 1)] >> (16 * (j & 1))) & 0xFFFFu;
                                    const uint32_t bit = 1u << (slot & 31);
                                    hfresh[c] = (atomicOr(&hbits[slot >> 5], bit) & bit) == 0;
                                    isn[c] = false;  // settled: the filters below see only the other neighbours
                                }
                            }
                            wave_sync();
                        }
                        bool fresh[NC];
                        if (MODE == VIS_EXACT) {  // visited_array[nbr] != tag ... = tag  (:2378,2385)
                            uint32_t tg[NC];  // the loads are issued together (a lane without a neighbour reads tag 0)
#pragma unroll
                            for (int c = 0; c < NC; ++c) tg[c] = __builtin_nontemporal_load(tags + (isn[c] ? nb[c] : 0u));
#pragma unroll
                            for (int c = 0; c < NC; ++c) {
                                fresh[c] = isn[c] && tg[c] != epoch;
                                if (fresh[c]) __builtin_nontemporal_store((uint8_t)epoch, tags + nb[c]);
                            }
                        } else {
                            visited_ldsn<NC>(isn, nb, vis, sets_mask, sets_log2, smult, forget, protect, fresh);
                        }
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            const bool f = fresh[c] | hfresh[c];
                            const u64 fm = __ballot(f);
                            if (f) cid[m + lanes_below(fm)] = nb[c];
                            if constexpr (ANN) {
                                if (f) {
                                    const u64 e = c == 0 ? m0 : m1;  // (rows of at most 127 neighbours: two chunks)
                                    cmeta[m + lanes_below(fm)] = make_uint2((uint32_t)e, (uint32_t)(e >> 32));
                                }
                            }
                            m += (uint32_t)__popcll(fm);
                        }
                    };
                    if (!HUB && deg < wpos + 64u) run_filter(std::integral_constant<int, 1>());
                    else if (!HUB && deg < wpos + 128u) run_filter(std::integral_constant<int, 2>());
                    else run_filter(std::integral_constant<int, 3>());
                    wpos += HUB ? 256u : 192u;
                    cmps += m;  // ++cmps per scored neighbour, :2397
                    // Next adjacency words, in flight while this round trip's rows are scored: the rest of this row
                    // if it is longer, else the row of the closest unexpanded entry as the pool stands now -- if this
                    // hop's merges put something in front of it the fetch is simply dropped (pf is compared at select).
                    {
                        const uint32_t *nrow = arow + wpos;
                        bool whole = true;  // the rest of a long row: every chunk; the next row, plain form: its first chunk
                        if (wpos > deg) {
                            pf = (nextv != NONE) ? nextv : vcur;  // (nothing left: any valid row)
                            nrow = ADJ + (u64)pf * AS;
                            whole = HUB;
                        }
                        a0 = nrow[lane];
                        if constexpr (ANN) m0 = (KARG(adj_meta) + (u64)(nrow - ADJ))[lane];  // (the same word of the parallel array)
                        if (whole) {
                            a1 = nrow[64 + lane];
                            if constexpr (ANN) m1 = (KARG(adj_meta) + (u64)(nrow - ADJ))[64 + lane];
                            if (AS > 128u) a2 = nrow[128 + lane];
                        }
                        if (HUB) a3 = nrow[192 + lane];
                        count_traffic((u64)(1u + (whole ? (AS > 128u ? 2u : 1u) : 0u) + (HUB ? 1u : 0u)) << 44);
                    }
                    if constexpr (SD > 0) {  // 8-bit screen (see screen8_pass), once the pool is full
                        if (psize == L && m > 4) {
#if defined(__HIP_DEVICE_COMPILE__)
                            const unsigned char *scr = KARG(screen);
#else
                            const unsigned char *scr = nullptr;
#endif
                            if (scr) {
                                wave_sync();
                                const u64 last = (RP ? lane_u64(pk, (int)L - 1) : pool[L - 1]) & ~1ull;
                                [[maybe_unused]] const uint32_t coded = m;
                                // (roar_screen8_split: the rows' floats sit behind the n code rows)
                                const uint4 *meta = nullptr;
                                if constexpr (roar_screen8_split(SD)) {
#if defined(__HIP_DEVICE_COMPILE__)
                                    meta = reinterpret_cast<const uint4 *>(scr + (u64)(KARG(n)) * SD);
#endif
                                }
                                m = screen8_rows<SD, IP>(scr, meta, lq, cid, m, screen_threshold(ord2f((uint32_t)(last >> 32)), __uint_as_float(ctrl[11])),
                                                         __uint_as_float(ctrl[8]), __uint_as_float(ctrl[9]), __uint_as_float(ctrl[10]), lane, nullptr,
                                                         ANN ? cmeta : nullptr);
                                count_traffic((u64)coded | ((u64)m << 24));
                            }
                        }
                    }
                    if (PROF) {
                        c1 = (long long)__builtin_readcyclecounter();
                        t_filter += c1 - c0;
                    }
                }
                if (W > 1 && lane == 0) {
                    ctrl[1] = done ? NONE : m;
                    ctrl[2] = 0;
                    ctrl[3] = (psize == L) ? 1u : 0u;
                    if constexpr (RP) {
                        const u64 last = lane_u64(pk, (int)L - 1) & ~1ull;  // (lane 0 only reads the broadcast)
                        ctrl[4] = (uint32_t)last;
                        ctrl[5] = (uint32_t)(last >> 32);
                    }
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
                    u64 thr = ~0ull;
                    if (ctrl[3]) thr = RP ? (((u64)ctrl[5] << 32) | ctrl[4]) : (pool[L - 1] & ~1ull);
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
                u64 thr = ~0ull;
                if (psize == L) thr = (RP ? lane_u64(pk, (int)L - 1) : pool[L - 1]) & ~1ull;
                if (m) score_rows<IP, NCH, TAIL, R, false, PROF>(p.base, p.ld, p.dim, lq, cid, ck, m, thr, &nsurv, lane, &pprof);
                wave_sync();
            }
            if (PROF && wid == 0) {
                const long long s1 = (long long)__builtin_readcyclecounter();
                t_score += s1 - s0;
                s0 = s1;
            }
            if (wid == 0) {
                if (PROF) n_surv += nsurv;
                const uint32_t ns = uni(nsurv);  // (it came through LDS: said to be uniform, the loop is a scalar one)
                for (uint32_t b = 0; b < ns; b += 64) {
                    const uint32_t cnt = (ns - b < 64u) ? ns - b : 64u;
                    if constexpr (RP) merge_regpool(pk, psize, ck + b, cnt, L, pool, lane);
                    else {
                        TagProtect tp;
                        if (MODE == VIS_LDS && protect) tp.vis = vis, tp.sets_mask = sets_mask, tp.sets_log2 = sets_log2, tp.mult = smult;
                        merge_candidates(pool, ck + b, cnt, L, psize, cur, lane, tp, &front, &ub, PROF ? t_mphase : nullptr);
                    }
                }
#ifdef ROAR_HOP_PAD
                // measurement builds only (scripts/build_variant.sh -DROAR_HOP_PAD=n): the wave sits out n x 64 more cycles per hop where
                // the merge is, issuing nothing -- the slope of the kernel time over the pad says how much of a hop's own time the
                // launch's rate hangs on (profiles/r06_hop_pad_probe.txt)
                __builtin_amdgcn_s_sleep(ROAR_HOP_PAD);
#endif
                if (PROF) t_merge += (long long)__builtin_readcyclecounter() - s0;
            }
        }

        // results (:2408-2418); a short pool is reported, not thrown
        if (wid == 0) {
            // The output pointers are used here only, once per query: they are read from the kernel-argument segment
            // where they are used instead of living in (spilled) scalar registers through every hop.
            // (batch-stream form: the batch's own arrays, from the workgroup's copy of its slot; stores are write-through)
            [[maybe_unused]] uint32_t *sl = nullptr;
            if constexpr (STREAM) sl = stream_args().sl;
            uint32_t *const o_ids = STREAM ? reinterpret_cast<uint32_t *>(lds_ptr(sl, 2)) : KARG(out_ids);
            float *const o_dists = STREAM ? reinterpret_cast<float *>(lds_ptr(sl, 4)) : KARG(out_dists);
            uint32_t *const o_cmps = STREAM ? reinterpret_cast<uint32_t *>(lds_ptr(sl, 6)) : KARG(out_cmps);
            uint32_t *const o_hops = STREAM ? reinterpret_cast<uint32_t *>(lds_ptr(sl, 8)) : KARG(out_hops);
            uint32_t *const o_scanned = STREAM ? reinterpret_cast<uint32_t *>(lds_ptr(sl, 10)) : KARG(out_scanned);
            auto put = [&](uint32_t *dst, uint32_t v) {
                if constexpr (STREAM) sys_store32(dst, v);
                else *dst = v;
            };
            for (uint32_t i = lane; i < p.k; i += 64) {
                uint32_t id = 0xFFFFFFFFu;
                float d = __builtin_inff();
                if (i < psize) {
                    const u64 e = RP ? pk : pool[i];
                    id = (uint32_t)e >> 1;
                    d = ord2f((uint32_t)(e >> 32));
                }
                put(o_ids + (u64)q * p.k + i, id);
                put(reinterpret_cast<uint32_t *>(o_dists) + (u64)q * p.k + i, __float_as_uint(d));
            }
            if (lane == 0) {
                if (o_cmps) put(o_cmps + q, cmps);
                if (o_hops) put(o_hops + q, hops);
                if (o_scanned) put(o_scanned + q, scanned);
                if constexpr (STATS) {
                    uint32_t *const o_traffic = KARG(out_traffic);
                    const u64 t = *traffic;
                    const uint32_t coded = (uint32_t)t & 0xFFFFFFu, passed = (uint32_t)(t >> 24) & 0xFFFFFu;
                    // every neighbour counted in cmps was scored on the f32 rows, but for those the screen dropped; + the entry point
                    *reinterpret_cast<uint4 *>(o_traffic + 4 * (u64)q) = make_uint4(coded, 1u + cmps - coded + passed, (uint32_t)(t >> 44) * 64u, 0u);
                }
                if (!STREAM && psize < p.k) atomicAdd(KARG(counter) + 2, 1u);  // short queries of this launch
                if (PROF && p.prof) {
                    unsigned long long *o = p.prof + (u64)q * 16;
                    o[0] = (unsigned long long)((long long)__builtin_readcyclecounter() - tq0);
                    o[1] = (unsigned long long)t_sel;
                    o[2] = (unsigned long long)t_adj;
                    o[3] = (unsigned long long)t_filter;
                    o[4] = (unsigned long long)t_score;
                    o[5] = (unsigned long long)t_merge;
                    o[6] = pf_hit;
                    o[7] = n_rt;
                    o[8] = (unsigned long long)pprof.issue;
                    o[9] = (unsigned long long)pprof.fma;
                    o[10] = (unsigned long long)pprof.tail;
                    o[11] = pprof.passes;
                    o[12] = n_surv;
                    o[13] = (unsigned long long)t_mphase[0], o[14] = (unsigned long long)t_mphase[1], o[15] = (unsigned long long)t_mphase[2];
                }
            }
            if constexpr (STREAM) {
                s_fin = true, s_short = psize < p.k;  // (counted by stream_next, while the stores above are in flight)
                if (lane == 0) {
                    const StreamArgs a = stream_args();
                    (void)atomicAdd(&a.dev->busy_ticks, (unsigned long long)((uint32_t)wall_clock64() - a.sl[22]));
                }
            }
        }
        block_sync();
    }
    if (MODE == VIS_EXACT && threadIdx.x == 0) p.epochs[blockIdx.x] = epoch;
    if (MODE == VIS_LDS && !forget && threadIdx.x == 0) {
#if defined(__HIP_DEVICE_COMPILE__)
        (KARG(epochs))[blockIdx.x] = ctrl[6];
#endif
    }
    // Launch slot {next query, workgroups done, short queries, -} (one 16-byte cell of the index's ring): the last
    // workgroup to leave hands the short count to the caller and leaves the cell zeroed for the launch that takes it
    // next, so no memset is queued in front of a launch.  (This workgroup's dequeue and short-count atomics were
    // issued before the one below, to the same L2 channel.)
    if (threadIdx.x == 0) {
        uint32_t *const cell = KARG(counter);
        uint32_t *const o_short = KARG(short_count);
        __threadfence();
        if (atomicAdd(cell + 1, 1u) == gridDim.x - 1u) {
            const uint32_t nshort = atomicExch(cell + 2, 0u);
            if (o_short) *o_short = nshort;
            atomicExch(cell + 1, 0u);
            atomicExch(cell, 0u);
            // the stamp LAST, behind a fence: the cell's next user (LAUNCH_RING launches later, maybe on another stream) starts dequeuing as
            // soon as it reads this launch's stamp, and must find every word of the cell zero by then
            __threadfence();
            atomicExch(KARG(stamp), KARG(cell_stamp));  // (this launch has finished with the cell)
        }
    }
}

// Register budget: three waves per SIMD (the explicit maximum keeps the scheduler from trading the batched loads of
// a scoring pass for an occupancy the LDS footprint does not allow anyway).
// (512-d rows: two waves per SIMD -- a row is 32 registers per lane in the exact pass, nine in the screen; at three per SIMD both
// spill inside the hop: WebVid-2.5M shape, L_pq = 95, 10 000 queries, screen on, 7.79 ms at three, 6.75 ms at two)
#ifndef ROAR_WPE_512
#define ROAR_WPE_512 2
#endif
template <int MODE, bool IP, int NCH, bool TAIL, int W, bool RP, bool PROF, bool HUB = false, int SRC = 0, bool STATS = false, bool PROT = false, bool ANN = false>
__global__ void __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(NCH > 16 ? ROAR_WPE_512 : 3, NCH > 16 ? ROAR_WPE_512 : 3)))
roar_search_v2_kernel(const RoarSearchParams p) {
    search_v2<MODE, IP, NCH, TAIL, W, RP, PROF, HUB, SRC, STATS, PROT, ANN>(p);
}

// The same kernel under its own symbol for the searches the graph builder issues (LDS visited mode, one wave per
// query), so that a kernel trace of a process that builds and then serves keeps the two apart.
template <bool IP, int NCH, bool TAIL, bool RP>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
roar_build_search_v2_kernel(const RoarSearchParams p) {
    search_v2<VIS_LDS, IP, NCH, TAIL, 1, RP, false, false>(p);
}

template <int MODE, bool IP, int NCH, bool TAIL>
const void *pick_form(const RoarKernelSel &s) {
    const bool rp = s.regpool != 0;
    if (s.form == ROAR_FORM_BUILDER) {
        if constexpr (MODE == VIS_LDS)
            return rp ? (const void *)roar_build_search_v2_kernel<IP, NCH, TAIL, true>
                      : (const void *)roar_build_search_v2_kernel<IP, NCH, TAIL, false>;
        else
            return nullptr;
    }
    if (s.ann) {  // the neighbours' screening floats come with the adjacency row: inner product, 128-d / 512-d, one wave per query, on-chip tag table
        if constexpr (MODE == VIS_LDS && IP && !TAIL && (NCH == 8 || NCH == 32)) {
            if (s.waves != 1 || s.hub || s.gated || s.prof || s.stats || (s.protect && rp)) return nullptr;
            if (s.stream) {
                if (s.protect) return (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, false, false, false, 2, false, true, true>;
                return rp ? (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, true, false, false, 2, false, false, true>
                          : (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, false, false, false, 2, false, false, true>;
            }
            if (s.protect) return (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, false, false, false, 0, false, true, true>;
            return rp ? (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, true, false, false, 0, false, false, true>
                      : (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, false, false, false, 0, false, false, true>;
        } else {
            return nullptr;
        }
    }
    if (s.stream) {  // batch-stream form: one wave per query, forgetful on-chip tag table, queries from a ring of submitted batches
        if constexpr (MODE == VIS_LDS) {
            if (s.waves != 1 || s.hub || s.gated || s.prof || s.stats || (s.protect && rp)) return nullptr;
            if (s.protect) return (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, false, false, false, 2, false, true>;
            return rp ? (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, true, false, false, 2>
                      : (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, false, false, false, 2>;
        } else {
            return nullptr;
        }
    }
    if constexpr (MODE == VIS_LDS) {
        if (s.protect) {  // the forgetful table keeps the tags of pool members: one wave per query, pool in LDS, no hub bitmap / gate
            if (rp || s.waves != 1 || s.hub || s.gated) return nullptr;
            if (s.stats) return s.prof ? nullptr : (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, false, false, false, 0, true, true>;
            if (s.prof) {
                if constexpr ((IP && NCH == 12) || (!IP && NCH == 8))
                    return (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, false, true, false, 0, false, true>;
                else
                    return nullptr;
            }
            return (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, false, false, false, 0, false, true>;
        }
    } else {
        if (s.protect) return nullptr;
    }
    if (s.stats) {  // one wave per query, counting what each query fetches (RoarSearchParams::out_traffic)
        if (s.prof || s.waves != 1 || s.gated || (s.hub && rp)) return nullptr;
        if (s.hub) return (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, false, false, true, 0, true>;
        return rp ? (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, true, false, false, 0, true>
                  : (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, false, false, false, 0, true>;
    }
    if (s.gated) {  // one wave per query, waiting for the host's rows-written counter (pageable host query blocks)
        if (s.hub || s.prof || s.waves != 1) return nullptr;
        return rp ? (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, true, false, false, 1>
                  : (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, false, false, false, 1>;
    }
    if (s.hub) {  // long pools: LDS pool, hub bitmap
        if (rp) return nullptr;
        if (s.prof) {
            if constexpr (IP && NCH == 12) return s.waves == 1 ? (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, false, true, true> : nullptr;
            else return nullptr;
        }
        if (s.waves == 4) return (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 4, false, false, true>;
        if (s.waves == 2) return (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 2, false, false, true>;
        return (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, false, false, true>;
    }
    if (s.prof) {  // instrumented instantiations exist for the Text-to-Image shape
        if constexpr (IP && NCH == 12) {
            if (s.waves == 4)
                return rp ? (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 4, true, true>
                          : (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 4, false, true>;
            if (s.waves == 2)
                return rp ? (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 2, true, true>
                          : (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 2, false, true>;
            return rp ? (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, true, true>
                      : (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, false, true>;
        } else if constexpr (!IP && NCH == 8) {  // ... and, one wave per query, for the SIFT shape
            if (s.waves != 1) return nullptr;
            return rp ? (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, true, true>
                      : (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, false, true>;
        } else {
            return nullptr;
        }
    }
    if (s.waves == 4)
        return rp ? (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 4, true, false>
                  : (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 4, false, false>;
    if (s.waves == 2)
        return rp ? (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 2, true, false>
                  : (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 2, false, false>;
    return rp ? (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, true, false>
              : (const void *)roar_search_v2_kernel<MODE, IP, NCH, TAIL, 1, false, false>;
}

}  // namespace

#if ROAR_TU_MODE == 1
// Known-answer surface for the screen (roar_debug_screen_keeps, tests): workgroup (q, b) runs the search's own screening passes over
// rows ids[q][64 b .. 64 b + 64) of query q, each row against its OWN threshold chk[q][c]; chk[q][c] becomes 1.0 (kept) or 0.0.
template <bool IP, int D>
__global__ void __launch_bounds__(64) roar_screen_check_kernel(const unsigned char *__restrict__ scr, const float *__restrict__ mu,
                                                               const float *__restrict__ queries, u64 qstride, const uint32_t *__restrict__ ids, float *chk,
                                                               uint32_t m, u64 n) {
    __shared__ __align__(16) float lq[D + D / 4 + 4];
    __shared__ uint32_t cid[64];
    __shared__ float th[64];
    const int lane = threadIdx.x;
    const u64 q = blockIdx.x;
    const uint32_t b0 = blockIdx.y * 64u, mm = (m - b0 < 64u) ? m - b0 : 64u;
    for (int i = lane; i < D; i += 64) lq[i] = queries[q * qstride + i];
    cid[lane] = (uint32_t)lane < mm ? ids[q * m + b0 + lane] : 0u;
    th[lane] = (uint32_t)lane < mm ? chk[q * m + b0 + lane] : 0.f;
    wave_sync();
    float qa, qb, qc, qd;
    screen8_query<D, IP>(lq, D, lane, qa, qb, qc, qd, mu);
    wave_sync();
    th[lane] = screen_threshold(th[lane], qd);
    wave_sync();
    const uint4 *meta = roar_screen8_split(D) ? reinterpret_cast<const uint4 *>(scr + n * (u64)roar_screen8_stride(D)) : nullptr;
    (void)screen8_rows<D, IP, true>(scr, meta, lq, cid, mm, 0.f, qa, qb, qc, lane, th);
    wave_sync();
    if ((uint32_t)lane < mm) chk[q * m + b0 + lane] = th[lane];
}

#define ROAR_CHECK_NAME2(i) roar_launch_screen_check_##i
#define ROAR_CHECK_NAME(i) ROAR_CHECK_NAME2(i)
extern "C" int ROAR_CHECK_NAME(ROAR_TU_IP)(const unsigned char *scr, const float *mu, uint32_t dim, const float *queries, uint64_t qstride, uint64_t nq,
                                           const uint32_t *ids, float *chk, uint32_t m, uint64_t n_rows, void *stream) {
    constexpr bool IP = ROAR_TU_IP != 0;
    if (nq == 0 || m == 0) return 0;
    const dim3 grid((uint32_t)nq, (m + 63u) / 64u);
    if (dim == 128) hipLaunchKernelGGL((roar_screen_check_kernel<IP, 128>), grid, dim3(64), 0, (hipStream_t)stream, scr, mu, queries, (u64)qstride, ids, chk, m, (u64)n_rows);
    else if (dim == 512) hipLaunchKernelGGL((roar_screen_check_kernel<IP, 512>), grid, dim3(64), 0, (hipStream_t)stream, scr, mu, queries, (u64)qstride, ids, chk, m, (u64)n_rows);
    else if (dim == 200 && IP) {
        if constexpr (IP) hipLaunchKernelGGL((roar_screen_check_kernel<IP, 200>), grid, dim3(64), 0, (hipStream_t)stream, scr, mu, queries, (u64)qstride, ids, chk, m, (u64)n_rows);
    } else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}
#endif

// One of these per translation unit (visited mode x metric): kernel address for a selection, or null.
#define ROAR_PICK_NAME2(m, i) roar_pick_search_##m##_##i
#define ROAR_PICK_NAME(m, i) ROAR_PICK_NAME2(m, i)
extern "C" const void *ROAR_PICK_NAME(ROAR_TU_MODE, ROAR_TU_IP)(const RoarKernelSel *s, uint32_t dim) {
    constexpr int MODE = ROAR_TU_MODE;
    constexpr bool IP = ROAR_TU_IP != 0;
    switch (dim) {
        case 128: return pick_form<MODE, IP, 8, false>(*s);   // SIFT shape
        case 200: return pick_form<MODE, IP, 12, true>(*s);   // Text-to-Image shape
        case 512: return pick_form<MODE, IP, 32, false>(*s);  // WebVid shape
        default: return pick_form<MODE, IP, 0, false>(*s);    // any other length: the row is walked at run time
    }
}
