// roar_kernels.hip -- the remaining gfx950 kernels of the RoarGraph engine and the launch dispatcher:
//   * SearchBipartiteGraph (two-hop walk, src/index_bipartite.cpp:282-356), one and eight waves per query;
//   * the distance-only kernel (parity surface for distance.h) and the top-k merge of the index-sharded configuration;
//   * roar_launch_search / roar_search_occupancy, which pick the kernel for a RoarKernelSel.
// The SearchRoarGraph kernels themselves live in roar_search_v2.h (four translation units).  Shared wave-level
// building blocks: roar_wave.h.
#include "roar_wave.h"

namespace {

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
    const uint32_t sets_log2 = p.vis_sets_log2 & 31u;  // (bits 30 / 31 belong to the SearchRoarGraph kernels: never set for this one)
    const uint32_t sets_mask = (1u << sets_log2) - 1u;
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
            const uint32_t n16 = (4u << sets_log2) >> 3;
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
        (void)visited_filter<MODE>(isinit, iid, vis, sets_mask, sets_log2, tags, epoch);
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
                    bool fresh = visited_filter<MODE>(has, nb, vis, sets_mask, sets_log2, tags, epoch);
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

// 8-bit screening copy of the base rows, inner-product form (roar_search_v2.h, screen8i_pass): one wave per row, lane l owns floats
// 4l .. 4l+3 of each 256-float piece.  Row out: dim codes c_i in [-127, 127] (two's complement), (dim % 16 == 0: 8 bytes of zeros,)
// s = max|x| / 127, then one word = B rounded up to bf16 (low half) | N rounded up to bf16 (high half), B >= |x - s c|_2, N >= |s c|_2
// -- roar_screen8_stride(dim) bytes.
__device__ __forceinline__ uint32_t bf16_up(float v) {  // v >= 0 or +inf: the smallest bf16 >= v, as the high half of an f32
    const uint32_t b = __float_as_uint(v);
    return (b & 0xFFFFu) ? (b >> 16) + 1u : (b >> 16);  // (+inf and values whose low half is zero stay; the largest finite rounds to +inf)
}

__global__ void __launch_bounds__(256) roar_make_screen8_kernel(const float *__restrict__ base, u64 ld, uint32_t dim, u64 n, unsigned char *__restrict__ out,
                                                                uint32_t stride, const float *__restrict__ mu, unsigned char *__restrict__ meta_out) {
    const u64 v = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (v >= n) return;
    const float4 *row = reinterpret_cast<const float4 *>(base + v * ld);
    const uint32_t nf4 = dim / 4;  // <= 256
    float4 a[4];
    float mx = 0.f;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (64 * j + lane < nf4) {
            a[j] = row[64 * j + lane];
            if (mu) {  // the codes are those of y = x - mu (f32 subtraction: its rounding, <= 2^-24 |y_i|, is paid for in B below)
                const float4 u = reinterpret_cast<const float4 *>(mu)[64 * j + lane];
                a[j].x -= u.x, a[j].y -= u.y, a[j].z -= u.z, a[j].w -= u.w;
            }
        }
        mx = __builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a[j].x), __builtin_fabsf(a[j].y)),
                                                  __builtin_fmaxf(__builtin_fabsf(a[j].z), __builtin_fabsf(a[j].w))));
        bad |= (a[j].x != a[j].x) | (a[j].y != a[j].y) | (a[j].z != a[j].z) | (a[j].w != a[j].w);  // (fmaxf drops NaNs: test the components)
    }
    bad |= !(mx <= 1.0e30f);  // inf, or so large that the bounds may overflow
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = __builtin_fmaxf(mx, __shfl_xor(mx, o, 64));
    const bool anybad = __ballot(bad) != 0 || (mx > 0.f && mx < 1.0e-30f);  // (a scale near the subnormals: no bound is claimed)
    const float sc = mx / 127.0f;
    const float inv = mx > 0.f ? 127.0f / mx : 0.f;
    uint32_t *o32 = reinterpret_cast<uint32_t *>(out + v * stride);
    float e2 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float f[4] = {a[j].x, a[j].y, a[j].z, a[j].w};
        uint32_t word = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float c = __builtin_rintf(f[i] * inv);
            c = __builtin_fminf(127.f, __builtin_fmaxf(-127.f, c));
            const float d = (f[i] - sc * c) * inv;  // in units of s: the square of a tiny d itself would underflow
            e2 = __builtin_fmaf(d, d, e2);          // (lanes past the row hold zeros: c = 0, d = 0)
            c2 = __builtin_fmaf(c, c, c2);          // (integers, the sum stays below 2^24: exact)
            word |= ((uint32_t)(int)c & 0xFFu) << (8 * i);
        }
        if (64 * j + lane < nf4) o32[64 * j + lane] = anybad ? 0u : word;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        e2 += __shfl_xor(e2, o, 64);
        c2 += __shfl_xor(c2, o, 64);
    }
    const uint32_t meta = (dim / 16u) * 4u + 2u;  // the last two words of the chunk after the full ones
    // N: |s c| rounded up.  B: |y - s c| rounded up (1.001: the f32 evaluation of e2), + 2^-23 N: each d_i was measured against the ROUNDED
    // product s c_i, the kernel means the real one; + 2^-23 N more when the codes are centred: y_i here is the f32 difference x_i - mu_i, the
    // bound is about the real one (|fl(y) - y| <= 2^-24 |y|, |y| <= N + B)
    const float nrm = sc * __builtin_sqrtf(c2) * 1.0001f;
    const float bnd = 1.001f * sc * __builtin_sqrtf(e2) * 1.0001f + 2.4e-7f * nrm;
    if (meta_out) {  // (roar_screen8_split: the rows' floats are a packed array behind the code rows)
        uint32_t *m32 = reinterpret_cast<uint32_t *>(meta_out + v * 16u);
        if (lane < 2) m32[lane] = 0u;
        if (lane == 2) m32[2] = __float_as_uint(anybad ? 0.f : sc);
        if (lane == 3) m32[3] = anybad ? 0x7F80u : (bf16_up(bnd) | (bf16_up(nrm) << 16));
        return;
    }
    if (lane == 0) o32[meta] = __float_as_uint(anybad ? 0.f : sc);
    if (lane == 1) o32[meta + 1] = anybad ? 0x7F80u : (bf16_up(bnd) | (bf16_up(nrm) << 16));
    for (uint32_t w = nf4 + lane; w < stride / 4; w += 64)  // zeros between the codes and the meta words, and after them
        if (w != meta && w != meta + 1) o32[w] = 0u;
}

// The L2 form (screen8_pass, !IP): x~_i = lo + s u_i with u_i in 0..255, lo = the row's minimum and s = (max - min) / 255 -- or the
// power of two at or above that if the row is then coded without error (rows of small integers: SIFT).  Row out: dim codes (dim % 16
// == 0), s, lo, |x~|^2, B with B >= |x - x~| + the distance between the x~ the kernel means (lo + s u in real arithmetic) and the
// rounded one measured here.
__global__ void __launch_bounds__(256) roar_make_screen8_l2_kernel(const float *__restrict__ base, u64 ld, uint32_t dim, u64 n, unsigned char *__restrict__ out,
                                                                   uint32_t stride, unsigned char *__restrict__ meta_out) {
    const u64 v = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (v >= n) return;
    const float4 *row = reinterpret_cast<const float4 *>(base + v * ld);
    const uint32_t nf4 = dim / 4;  // <= 256
    float f[16];
    float mx = -3.0e38f, mn = 3.0e38f;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool in = 64 * j + lane < nf4;
        const float4 a = in ? row[64 * j + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        f[4 * j] = a.x, f[4 * j + 1] = a.y, f[4 * j + 2] = a.z, f[4 * j + 3] = a.w;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x = f[4 * j + i];
            bad |= in && (x != x || !(__builtin_fabsf(x) <= 1.0e17f));  // NaN, inf, or so large that |x~|^2 may overflow
            if (in) mx = __builtin_fmaxf(mx, x), mn = __builtin_fminf(mn, x);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        mx = __builtin_fmaxf(mx, __shfl_xor(mx, o, 64));
        mn = __builtin_fminf(mn, __shfl_xor(mn, o, 64));
    }
    bool anybad = __ballot(bad) != 0;
    anybad |= !anybad && mx - mn > 0.f && mx - mn < 1.0e-30f;  // (a scale near the subnormals: no bound is claimed)
    const float range = anybad ? 0.f : mx - mn, lo = anybad ? 0.f : mn;
    const float tight = range / 255.0f;
    uint32_t words[4];
    float sc = 0.f, e2 = 0.f, n2 = 0.f;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (attempt == 0) {  // the power of two at or above the tight scale
            int ex;
            const float mant = __builtin_frexpf(tight, &ex);  // tight = mant 2^ex, mant in [0.5, 1)
            sc = tight > 0.f ? __builtin_ldexpf(1.0f, mant == 0.5f ? ex - 1 : ex) : 0.f;
        } else {
            sc = tight;
        }
        const float inv = sc > 0.f ? 1.0f / sc : 0.f;
        e2 = 0.f, n2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            words[j] = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool in = 64 * j + lane < nf4;
                float c = __builtin_rintf((f[4 * j + i] - lo) * inv);
                c = __builtin_fminf(255.f, __builtin_fmaxf(0.f, c));
                const float xh = in ? __builtin_fmaf(sc, c, lo) : 0.f;
                const float d = in ? (f[4 * j + i] - xh) * inv : 0.f;  // in units of s (sc = 0: a constant row, x~ = x)
                e2 = __builtin_fmaf(d, d, e2);
                n2 = __builtin_fmaf(xh, xh, n2);
                words[j] |= (uint32_t)(int)c << (8 * i);
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            e2 += __shfl_xor(e2, o, 64);
            n2 += __shfl_xor(n2, o, 64);
        }
        if (e2 == 0.f) break;  // (uniform: e2 is the same in every lane)
    }
    uint32_t *o32 = reinterpret_cast<uint32_t *>(out + v * stride);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (64 * j + lane < nf4) o32[64 * j + lane] = anybad ? 0u : words[j];
    const uint32_t meta = (dim / 16u) * 4u;
    uint32_t *m32 = meta_out ? reinterpret_cast<uint32_t *>(meta_out + v * 16u) : o32 + meta;  // (roar_screen8_split: a packed array behind the code rows)
    if (lane == 0) m32[0] = __float_as_uint(anybad ? 0.f : sc);
    if (lane == 1) m32[1] = __float_as_uint(lo);
    if (lane == 2) m32[2] = __float_as_uint(anybad ? 0.f : n2);
    if (lane == 3)
        m32[3] = __float_as_uint(anybad ? __builtin_inff() : 1.001f * sc * __builtin_sqrtf(e2) * 1.0001f + 2.5e-7f * __builtin_sqrtf(n2) * 1.0001f);
    if (!meta_out)
        for (uint32_t w = meta + 4 + lane; w < stride / 4; w += 64) o32[w] = 0u;
}

// RoarSearchParams::adj_meta: entry (v, w) = {scale, bounds} of the screening row of the neighbour in word w of v's adjacency row
// (words 1 .. degree; zeros elsewhere).  One thread per adjacency word.
__global__ void __launch_bounds__(256) roar_edge_floats_kernel(const uint32_t *__restrict__ adj, uint32_t S, u64 n, const uint4 *__restrict__ row_floats,
                                                               unsigned long long *__restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * S) return;
    const u64 v = i / S;
    const uint32_t w = (uint32_t)(i - v * S);
    const uint32_t deg = adj[v * S];
    unsigned long long e = 0ull;
    if (w >= 1u && w <= deg) {
        const uint4 f = row_floats[adj[i]];
        e = (unsigned long long)f.z | ((unsigned long long)f.w << 32);
    }
    out[i] = e;
}

// local ids of a shard -> global ids (index-sharded configuration); "no result" stays "no result"
__global__ void roar_rebase_ids_kernel(uint32_t *__restrict__ ids, u64 count, uint32_t base) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count && ids[i] != 0xFFFFFFFFu) ids[i] += base;
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


// Known-answer surface for the queue (A4): replays a scripted NeighborPriorityQueue sequence -- op 0 = insert(id, dist)
// (neighbor.h:150-183), op 1 = closest_unexpanded() if has_unexpanded_node() (:185-194) -- on one wave with the very
// routines the search kernels use, and writes the state after every op as the reference harness of
// tests/golden/pq_trace.npz does: [size, cur, (id, distance bits, flag) x capacity].  form 0: pool in LDS
// (merge_candidates + the cursor scan), form 1: pool in a register pair (merge_regpool + ballot select; capacity <= 64).
__global__ void __launch_bounds__(64) roar_queue_trace_kernel(uint32_t capacity, uint32_t n_ops, const uint8_t *__restrict__ ops,
                                                             const uint32_t *__restrict__ ids, const float *__restrict__ dists,
                                                             int form, uint32_t *__restrict__ snap) {
    extern __shared__ __align__(16) unsigned char smem[];
    u64 *pool = reinterpret_cast<u64 *>(smem);             // capacity + 1
    u64 *ck = pool + ((capacity + 2) & ~1u);               // one candidate
    const int lane = threadIdx.x;
    const uint32_t L = capacity, row = 2 + 3 * capacity;
    u64 pk = ~0ull;
    uint32_t psize = 0, cur = 0;
    for (uint32_t i = 0; i < n_ops; ++i) {
        if (ops[i] == 0) {
            if (lane == 0) ck[0] = make_key(dists[i], ids[i]);
            wave_sync();
            if (psize == 0) {  // the kernels always start from a scored entry point: first insert by hand
                if (form) pk = lane == 0 ? ck[0] : ~0ull;
                else if (lane == 0) pool[0] = ck[0];
                psize = 1;
                cur = 0;
                wave_sync();
            } else if (form) {
                merge_regpool(pk, psize, ck, 1, L, pool, lane);
            } else {
                merge_candidates(pool, ck, 1, L, psize, cur, lane);
            }
        } else if (form) {
            const u64 um = __ballot((uint32_t)lane < psize && !(pk & 1ull));
            if (um && lane == __ffsll((long long)um) - 1) pk |= 1ull;
        } else if (cur < psize) {
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
            cur = found;
        }
        if (form) {  // the register form has no cursor: it is the first entry that is not expanded
            const u64 um = __ballot((uint32_t)lane < psize && !(pk & 1ull));
            cur = um ? (uint32_t)__ffsll((long long)um) - 1u : psize;
        }
        uint32_t *s = snap + (u64)i * row;
        for (uint32_t j = lane; j < row; j += 64) s[j] = 0;
        wave_sync();
        if (lane == 0) {
            s[0] = psize;
            s[1] = cur;
        }
        if ((uint32_t)lane < psize) {
            const u64 e = form ? pk : pool[lane];
            s[2 + 3 * lane] = (uint32_t)e >> 1;
            s[3 + 3 * lane] = __float_as_uint(ord2f((uint32_t)(e >> 32)));
            s[4 + 3 * lane] = (uint32_t)(e & 1ull);
        }
        wave_sync();
    }
}

template <int MODE, bool IP>
const void *pick_bipartite(const RoarKernelSel &s, uint32_t dim) {
    const bool wg = s.form == ROAR_FORM_BIPARTITE_WG;
    switch (dim >> 4) {
        case 8: return wg ? (const void *)roar_bipartite_kernel_wg<IP, 8> : (const void *)roar_bipartite_kernel<MODE, IP, 8>;
        case 12: return wg ? (const void *)roar_bipartite_kernel_wg<IP, 12> : (const void *)roar_bipartite_kernel<MODE, IP, 12>;
        default: return wg ? (const void *)roar_bipartite_kernel_wg<IP, 0> : (const void *)roar_bipartite_kernel<MODE, IP, 0>;
    }
}

int block_threads(const RoarKernelSel &s) {
    if (s.form == ROAR_FORM_BIPARTITE_WG) return 64 * BPW;
    if (s.form == ROAR_FORM_BIPARTITE || s.form == ROAR_FORM_BUILDER) return 64;
    return 64 * (s.waves == 4 ? 4 : s.waves == 2 ? 2 : 1);
}

}  // namespace

extern "C" {  // roar_search_{lds,exact}_{ip,l2}.hip
const void *roar_pick_search_1_1(const RoarKernelSel *s, uint32_t dim);
const void *roar_pick_search_1_0(const RoarKernelSel *s, uint32_t dim);
const void *roar_pick_search_0_1(const RoarKernelSel *s, uint32_t dim);
const void *roar_pick_search_0_0(const RoarKernelSel *s, uint32_t dim);
}

namespace {
const void *pick_kernel(const RoarKernelSel &s, uint32_t dim) {
    if (s.form == ROAR_FORM_BIPARTITE || s.form == ROAR_FORM_BIPARTITE_WG) {
        if (s.visited_mode == VIS_EXACT) return s.metric_ip ? pick_bipartite<VIS_EXACT, true>(s, dim) : pick_bipartite<VIS_EXACT, false>(s, dim);
        return s.metric_ip ? pick_bipartite<VIS_LDS, true>(s, dim) : pick_bipartite<VIS_LDS, false>(s, dim);
    }
    if (s.visited_mode == VIS_EXACT) return s.metric_ip ? roar_pick_search_0_1(&s, dim) : roar_pick_search_0_0(&s, dim);
    return s.metric_ip ? roar_pick_search_1_1(&s, dim) : roar_pick_search_1_0(&s, dim);
}
}  // namespace

extern "C" int roar_launch_search(const RoarSearchParams *p, const RoarKernelSel *sel, uint32_t grid, void *stream) {
    const void *fn = pick_kernel(*sel, p->dim);
    if (!fn) return (int)hipErrorInvalidDeviceFunction;
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
    if (!fn) return (int)hipErrorInvalidDeviceFunction;
    if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, fn, block_threads(*sel), lds_bytes);
}

extern "C" int roar_launch_queue_trace(uint32_t capacity, uint32_t n_ops, const uint8_t *ops, const uint32_t *ids,
                                       const float *dists, int form, uint32_t *snap, void *stream) {
    hipLaunchKernelGGL(roar_queue_trace_kernel, dim3(1), dim3(64), (capacity + 4) * 8 + 64, (hipStream_t)stream, capacity, n_ops,
                       ops, ids, dists, form, snap);
    return (int)hipGetLastError();
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

extern "C" int roar_launch_edge_floats(const uint32_t *adj, uint32_t stride, uint64_t n, const void *row_floats, unsigned long long *out, void *stream) {
    if (n == 0) return 0;
    const u64 total = (u64)n * stride;
    hipLaunchKernelGGL(roar_edge_floats_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, adj, stride, (u64)n,
                       reinterpret_cast<const uint4 *>(row_floats), out);
    return (int)hipGetLastError();
}

extern "C" int roar_launch_make_screen8(const float *base, uint64_t ld, uint32_t dim, uint64_t n, unsigned char *out, int l2, const float *mu, void *stream) {
    if (n == 0) return 0;
    if (dim % 8 || dim > 1024 || (l2 && dim % 16)) return (int)hipErrorInvalidValue;
    unsigned char *const meta_out = roar_screen8_split(dim) ? out + (u64)n * roar_screen8_stride(dim) : nullptr;
    if (l2) {
        hipLaunchKernelGGL(roar_make_screen8_l2_kernel, dim3((uint32_t)((n * 64 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, base, (u64)ld, dim, (u64)n, out,
                           roar_screen8_stride(dim), meta_out);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(roar_make_screen8_kernel, dim3((uint32_t)((n * 64 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, base, (u64)ld, dim, (u64)n, out,
                       roar_screen8_stride(dim), mu, meta_out);
    return (int)hipGetLastError();
}

extern "C" int roar_launch_rebase_ids(uint32_t *ids, uint64_t count, uint32_t base, void *stream) {
    if (count == 0 || base == 0) return 0;
    hipLaunchKernelGGL(roar_rebase_ids_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ids, (u64)count, base);
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
