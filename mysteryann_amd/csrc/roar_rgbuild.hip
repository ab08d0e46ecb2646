// roar_rgbuild.hip -- the reference's RoarGraph builder (BuildRoarGraph -> LinkProjection, src/index_bipartite.cpp:143-233,
// 1043-1277 and its prune / reverse helpers :1352-1694, :1846-1940) re-designed for the GPU: kernels of the
// batch-synchronous form driven by roar_build_roargraph (roar_capi.cpp).
//
// The reference walks training queries and nodes one by one under OpenMP with per-node locks, so its edge order and
// the outcome of every "row is full, prune it again" step depend on thread timing.  Here every phase is a few wide
// kernels over all lists of a batch and a graph update that does not depend on scheduling:
//
//   projection (:1059-1129)   per training query: pivot = its nearest base point, the next M_sq - 1 candidates ordered
//                             by their distance to the pivot and pruned by PruneBiSearchBaseGetBase (:1612-1694) to
//                             <= M_pjbp (rg_list_prune_kernel); the pivot's row is the list of its LAST training query
//                             (the reference overwrites, :1075); every (kept -> pivot) edge is a reverse proposal;
//   reverse edges (:1391-1432, :1352-1389)  proposals are sorted by destination; a destination row that still fits its cap
//                             takes them, one that does not is pruned again over {row + proposals} ordered by distance
//                             to the destination (PruneProjectionReverseCandidates :1527-1610 with back-fill in list
//                             order / ...InternalReverse... :1434-1525 without) -- one pass over all proposals of a
//                             batch where the reference prunes after every single insertion;
//   connectivity (:1183-1262) nodes in growing batches: search the supply graph from the entry point with the node's
//                             own vector (the search kernel, k = L_pjpq: the expanded set of that search is its final
//                             pool), candidates = the whole pool, nearest first, leading members of the node's
//                             projection row skipped, PruneProjectionBaseSearchCandidates (:1846-1940, no back-fill)
//                             -> the node's supply row; reverse supply edges with cap 2 M_pjbp as above;
//   merge (:1262-1275)        projection row + up to 2 M_pjbp supply edges not already in it.
//
// (One wave per SIMD: the kept rows take 28 KB of LDS per wave at D = 200 and a candidate row 200 registers per lane.)
// The greedy occlusion rule -- keep p unless an already kept r has dist(p, r) < dist(p, target) -- is sequential in the
// candidates; a wave takes 64 candidates at a time (one per lane, the candidate's row in registers), tests them against
// the rows kept so far (in LDS, broadcast reads) and resolves the chunk in order, one kept row at a time.
// Known deviations from the restated reference (oracle/roar_builder_oracle.c), none visible in recall-vs-cmps at
// 200 k points (tests/test_gpu_parity.py): the expanded set is the final pool (nodes expanded and later evicted are
// worse than every pool entry); the `std::vector prune_queue(size)` quirk of :1438 (leading zero entries) is not
// reproduced; distances inside the prune are plain fused sums, not the AVX-512 lane order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <new>
#include <type_traits>

#include <rocprim/rocprim.hpp>

namespace {

typedef unsigned long long u64;
constexpr u64 NO_KEY = ~0ull;
constexpr uint32_t NO_ID = 0xFFFFFFFFu;

__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t f2o(float f) {
    if (f == 0.0f) f = 0.0f;
    uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float o2f(uint32_t o) { return __uint_as_float((o >> 31) ? (o ^ 0x80000000u) : ~o); }
__device__ __forceinline__ u64 mkkey(float d, uint32_t id) { return ((u64)f2o(d) << 32) | id; }
__device__ __forceinline__ uint32_t below(u64 mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// A candidate's row, held by its lane: NF4 > 0 keeps the NF4 float4 pieces in registers, NF4 == 0 re-reads global memory.
template <int NF4>
struct LaneRow {
    float4 x[NF4 > 0 ? NF4 : 1];
    const float *g = nullptr;
    __device__ __forceinline__ void load(const float *row) {
        g = row;
        if constexpr (NF4 > 0) {
#pragma unroll
            for (int i = 0; i < NF4; ++i) x[i] = reinterpret_cast<const float4 *>(row)[i];
        }
    }
    // distance to a row in LDS (every lane reads the same address: broadcast)
    template <bool IP>
    __device__ __forceinline__ float dist(const float *lrow, uint32_t dim) const {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if constexpr (NF4 > 0) {
#pragma unroll
            for (int i = 0; i < NF4; ++i) {
                const float4 y = reinterpret_cast<const float4 *>(lrow)[i];
                if (IP) {
                    s0 = __builtin_fmaf(x[i].x, y.x, s0), s1 = __builtin_fmaf(x[i].y, y.y, s1);
                    s2 = __builtin_fmaf(x[i].z, y.z, s2), s3 = __builtin_fmaf(x[i].w, y.w, s3);
                } else {
                    const float a = x[i].x - y.x, b = x[i].y - y.y, c = x[i].z - y.z, e = x[i].w - y.w;
                    s0 = __builtin_fmaf(a, a, s0), s1 = __builtin_fmaf(b, b, s1);
                    s2 = __builtin_fmaf(c, c, s2), s3 = __builtin_fmaf(e, e, s3);
                }
            }
        } else {
            for (uint32_t i = 0; i < dim; i += 4) {
                const float4 xx = *reinterpret_cast<const float4 *>(g + i);
                const float4 y = *reinterpret_cast<const float4 *>(lrow + i);
                if (IP) {
                    s0 = __builtin_fmaf(xx.x, y.x, s0), s1 = __builtin_fmaf(xx.y, y.y, s1);
                    s2 = __builtin_fmaf(xx.z, y.z, s2), s3 = __builtin_fmaf(xx.w, y.w, s3);
                } else {
                    const float a = xx.x - y.x, b = xx.y - y.y, c = xx.z - y.z, e = xx.w - y.w;
                    s0 = __builtin_fmaf(a, a, s0), s1 = __builtin_fmaf(b, b, s1);
                    s2 = __builtin_fmaf(c, c, s2), s3 = __builtin_fmaf(e, e, s3);
                }
            }
        }
        const float s = (s0 + s1) + (s2 + s3);
        return IP ? -s : s;
    }
    __device__ __forceinline__ void store(float *lrow, uint32_t dim) const {
        if constexpr (NF4 > 0) {
#pragma unroll
            for (int i = 0; i < NF4; ++i) reinterpret_cast<float4 *>(lrow)[i] = x[i];
        } else {
            for (uint32_t i = 0; i < dim; i += 4) *reinterpret_cast<float4 *>(lrow + i) = *reinterpret_cast<const float4 *>(g + i);
        }
    }
};

// The occlusion rule over a list sorted by (distance to the target, id): keys[i] = ord(dist) << 32 | id.  Starts at
// `start`, keeps at most M, the first live candidate unconditionally (there is nothing to occlude it); `self` and
// repeated ids never enter.  Kept rows / ids land in LDS (krows: M x dim floats, kids: M).  Returns the number kept.
template <bool IP, int NF4>
__device__ __forceinline__ uint32_t greedy_prune(const float *__restrict__ base, u64 ld, uint32_t dim, const u64 *keys,
                                                 uint32_t ncand, uint32_t start, uint32_t self, uint32_t M, float *krows,
                                                 uint32_t *kids, int lane) {
    uint32_t nk = 0;
    for (uint32_t c0 = start; c0 < ncand && nk < M; c0 += 64) {
        const uint32_t i = c0 + (uint32_t)lane;
        const bool has = i < ncand;
        const u64 key = has ? keys[i] : NO_KEY;
        const uint32_t id = (uint32_t)key;
        const float dpt = o2f((uint32_t)(key >> 32));
        bool alive = has && key != NO_KEY && id != self && !(i > start && (uint32_t)keys[i - 1] == id);
        LaneRow<NF4> row;
        if (alive) row.load(base + (u64)id * ld);
        for (uint32_t j = 0; j < nk; ++j)  // against what earlier chunks kept
            if (alive && row.template dist<IP>(krows + (u64)j * dim, dim) < dpt) alive = false;
        while (nk < M) {  // the chunk itself, in order
            const u64 m = __ballot(alive);
            if (m == 0) break;
            const int c = __ffsll((long long)m) - 1;
            if (lane == c) {
                row.store(krows + (u64)nk * dim, dim);
                kids[nk] = id;
                alive = false;
            }
            nk++;
            wsync();
            if (alive && row.template dist<IP>(krows + (u64)(nk - 1) * dim, dim) < dpt) alive = false;
        }
    }
    return nk;
}

// back-fill kids[nk..M) with the first ids of list[0..n) that are neither kept nor `self` (list: u64 keys or u32 ids)
template <typename T>
__device__ __forceinline__ uint32_t backfill(const T *list, uint32_t n, uint32_t self, uint32_t M, uint32_t *kids, uint32_t nk,
                                             int lane) {
    for (uint32_t c0 = 0; c0 < n && nk < M; c0 += 64) {
        const uint32_t i = c0 + (uint32_t)lane;
        const uint32_t id = i < n ? (uint32_t)list[i] : NO_ID;
        bool take = i < n && id != NO_ID && id != self;
        for (uint32_t j = 0; j < nk && take; ++j) take = kids[j] != id;
        // an id repeated inside this chunk: the first occurrence wins
        for (u64 it = __ballot(take); it;) {
            const int j = __ffsll((long long)it) - 1;
            it &= it - 1;
            const uint32_t idj = (uint32_t)__builtin_amdgcn_readlane((int)id, j);
            if (lane > j && take && id == idj) take = false;
        }
        const u64 m = __ballot(take);
        const uint32_t pos = nk + below(m);
        if (take && pos < M) kids[pos] = id;
        const uint32_t add = (uint32_t)__popcll(m);
        nk = nk + add < M ? nk + add : M;
        wsync();
    }
    return nk;
}

// distance of every candidate of a wave-owned list to a target row held in LDS; lane = candidate, its row streamed
template <bool IP>
__device__ __forceinline__ float lane_dist_to_lds(const float *__restrict__ crow, const float *ltarget, uint32_t dim) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (uint32_t i = 0; i < dim; i += 4) {
        const float4 x = *reinterpret_cast<const float4 *>(crow + i);
        const float4 y = *reinterpret_cast<const float4 *>(ltarget + i);
        if (IP) {
            s0 = __builtin_fmaf(x.x, y.x, s0), s1 = __builtin_fmaf(x.y, y.y, s1);
            s2 = __builtin_fmaf(x.z, y.z, s2), s3 = __builtin_fmaf(x.w, y.w, s3);
        } else {
            const float a = x.x - y.x, b = x.y - y.y, c = x.z - y.z, e = x.w - y.w;
            s0 = __builtin_fmaf(a, a, s0), s1 = __builtin_fmaf(b, b, s1);
            s2 = __builtin_fmaf(c, c, s2), s3 = __builtin_fmaf(e, e, s3);
        }
    }
    const float s = (s0 + s1) + (s2 + s3);
    return IP ? -s : s;
}

// in-wave sort of n <= 256 keys in LDS (rank by counting; equal keys keep one copy, the rest become NO_KEY at the end)
__device__ __forceinline__ void sort_small(u64 *keys, u64 *tmp, uint32_t n, int lane) {
    for (uint32_t i = lane; i < n; i += 64) tmp[i] = keys[i];
    wsync();
    for (uint32_t i = lane; i < n; i += 64) keys[i] = NO_KEY;
    wsync();
    for (uint32_t i = lane; i < n; i += 64) {
        const u64 k = tmp[i];
        uint32_t r = 0;
        bool dup = false;
        for (uint32_t q = 0; q < n; ++q) {
            const u64 kq = tmp[q];
            r += kq < k ? 1u : 0u;
            dup |= kq == k && q < i;
        }
        if (!dup && k != NO_KEY) keys[r] = k;  // ranks of distinct keys are distinct; a gap is left where copies were
    }
    wsync();
}

// LDS layout of the list kernels: [target row dim][kept rows M x dim][kids M][keys 256][tmp 256][ids 256]
struct ListLds {
    float *target, *krows;
    uint32_t *kids, *ids;
    u64 *keys, *tmp;
};
__device__ __forceinline__ ListLds carve(unsigned char *smem, uint32_t dim, uint32_t M) {
    ListLds l;
    l.target = reinterpret_cast<float *>(smem);
    l.krows = l.target + dim;
    l.keys = reinterpret_cast<u64 *>(l.krows + (u64)M * dim);
    l.tmp = l.keys + 256;
    l.kids = reinterpret_cast<uint32_t *>(l.tmp + 256);
    l.ids = l.kids + ((M + 3) & ~3u);
    return l;
}

// A short list per target (<= 255 candidates, unordered): order by distance to the target, prune.
//   mode 0  projection (PruneBiSearchBaseGetBase, :1612-1694): list t = lists[t * K .. + C), target = its first entry,
//           candidates = the others; back-fill from the sorted order;
//   mode 1  re-prune of an over-long supply row (:1230-1258): target = node t0 + t, candidates = its row in `rows`,
//           leading members of its projection row (grows) skipped, no back-fill.
// out[t][0..M) = kept ids (NO_ID padded), cnt[t] = number kept.
template <bool IP, int NF4>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) rg_list_prune_kernel(const float *__restrict__ base, u64 ld, uint32_t dim, int mode,
                                                          const uint32_t *__restrict__ lists, uint32_t K, uint32_t C,
                                                          uint32_t count, uint32_t t0, const uint32_t *__restrict__ rows,
                                                          uint32_t S, const uint32_t *__restrict__ grows, uint32_t Sg,
                                                          uint32_t M, uint32_t only_longer_than, uint32_t *__restrict__ out,
                                                          uint32_t *__restrict__ cnt) {
    extern __shared__ __align__(16) unsigned char smem[];
    const ListLds L = carve(smem, dim, M);
    const int lane = threadIdx.x;
    for (uint32_t t = blockIdx.x; t < count; t += gridDim.x) {
        uint32_t target, ncand;
        const uint32_t *src;
        if (mode == 0) {
            src = lists + (u64)t * K;
            target = src[0];
            ncand = C;
        } else {
            target = t0 + t;
            src = rows + (u64)target * S + 1;
            ncand = rows[(u64)target * S];
            if (ncand <= only_longer_than) {  // the row stands as it is
                if (lane == 0) cnt[t] = NO_ID;
                continue;
            }
        }
        if (ncand > 255) ncand = 255;
        for (uint32_t i = lane * 4; i < dim; i += 256)
            *reinterpret_cast<float4 *>(L.target + i) = *reinterpret_cast<const float4 *>(base + (u64)target * ld + i);
        wsync();
        for (uint32_t i = lane; i < 256; i += 64) {
            u64 key = NO_KEY;
            if (i < ncand) {
                const uint32_t id = src[i];
                if (id != NO_ID && id != target) key = mkkey(lane_dist_to_lds<IP>(base + (u64)id * ld, L.target, dim), id);
            }
            L.keys[i] = key;
        }
        wsync();
        sort_small(L.keys, L.tmp, 256, lane);
        uint32_t n = 0;  // live candidates sit in front
        for (uint32_t c0 = 0; c0 < 256; c0 += 64) n += (uint32_t)__popcll(__ballot(L.keys[c0 + lane] != NO_KEY));
        // the sort leaves NO_KEY gaps only behind repeated keys; compact
        {
            uint32_t w = 0;
            for (uint32_t c0 = 0; c0 < 256; c0 += 64) {
                const u64 k = L.keys[c0 + lane];
                const u64 m = __ballot(k != NO_KEY);
                if (k != NO_KEY) L.tmp[w + below(m)] = k;
                w += (uint32_t)__popcll(m);
            }
            wsync();
            for (uint32_t i = lane; i < 256; i += 64) L.keys[i] = i < n ? L.tmp[i] : NO_KEY;
            wsync();
        }
        uint32_t start = 0;
        if (mode == 1 && grows) {  // :1222-1224: skip the leading candidates the projection row already holds
            const uint32_t *gr = grows + (u64)target * Sg;
            const uint32_t gd = gr[0];
            while (start < n) {
                const uint32_t id = (uint32_t)L.keys[start];
                bool in = false;
                for (uint32_t j = lane; j < gd; j += 64) in |= gr[1 + j] == id;
                if (__ballot(in) == 0) break;
                ++start;
            }
        }
        uint32_t nk = greedy_prune<IP, NF4>(base, ld, dim, L.keys, n, start, target, M, L.krows, L.kids, lane);
        wsync();
        if (mode == 0) nk = backfill(L.keys, n, target, M, L.kids, nk, lane);  // :1685-1691
        wsync();
        for (uint32_t i = lane; i < M; i += 64) out[(u64)t * M + i] = i < nk ? L.kids[i] : NO_ID;
        if (lane == 0) cnt[t] = nk;
        wsync();
    }
}

// Connectivity enhancement, one wave per node of the batch: candidates = the node's search results, already ordered by
// distance to the node (res_ids / res_d: [count][Lres]); PruneProjectionBaseSearchCandidates (:1846-1940).
template <bool IP, int NF4>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) rg_supply_prune_kernel(const float *__restrict__ base, u64 ld, uint32_t dim,
                                                            const uint32_t *__restrict__ res_ids,
                                                            const float *__restrict__ res_d, uint32_t Lres, uint32_t count,
                                                            uint32_t t0, const uint32_t *__restrict__ grows, uint32_t Sg,
                                                            uint32_t M, u64 *__restrict__ keyscratch,
                                                            uint32_t *__restrict__ out, uint32_t *__restrict__ cnt) {
    extern __shared__ __align__(16) unsigned char smem[];
    const ListLds L = carve(smem, dim, M);
    const int lane = threadIdx.x;
    for (uint32_t t = blockIdx.x; t < count; t += gridDim.x) {
        const uint32_t v = t0 + t;
        u64 *keys = keyscratch + (u64)blockIdx.x * Lres;  // this wave's list, in global memory (L_pjpq entries)
        uint32_t n = 0;
        for (uint32_t c0 = 0; c0 < Lres; c0 += 64) {
            const uint32_t i = c0 + (uint32_t)lane;
            const uint32_t id = i < Lres ? res_ids[(u64)t * Lres + i] : NO_ID;
            const bool ok = id != NO_ID && id != v;  // :1208-1213: the node itself is dropped
            const u64 m = __ballot(ok);
            if (ok) keys[n + below(m)] = mkkey(res_d[(u64)t * Lres + i], id);
            n += (uint32_t)__popcll(m);
        }
        __threadfence_block();
        wsync();
        uint32_t start = 0;
        {
            const uint32_t *gr = grows + (u64)v * Sg;
            const uint32_t gd = gr[0];
            while (start < n) {
                const uint32_t id = (uint32_t)keys[start];
                bool in = false;
                for (uint32_t j = lane; j < gd; j += 64) in |= gr[1 + j] == id;
                if (__ballot(in) == 0) break;
                ++start;
            }
        }
        const uint32_t nk = greedy_prune<IP, NF4>(base, ld, dim, keys, n, start, v, M, L.krows, L.kids, lane);
        wsync();
        for (uint32_t i = lane; i < M; i += 64) out[(u64)t * M + i] = i < nk ? L.kids[i] : NO_ID;
        if (lane == 0) cnt[t] = nk;
        wsync();
    }
}

// ---- graph updates ---------------------------------------------------------------------------------------------------------

// the last training query of every pivot (the reference overwrites the pivot's list, :1075)
__global__ void rg_last_list_kernel(const uint32_t *__restrict__ lists, uint32_t K, uint32_t count, uint32_t sq0,
                                    uint32_t *__restrict__ last) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint32_t piv = lists[(u64)t * K];
    if (piv != NO_ID) atomicMax(&last[piv], sq0 + t + 1);  // 0 = no list
}

// rows[target(t)] = out[t] for the lists that own their target's row (projection: last list of the pivot; supply: every
// node of the batch), and the reverse proposals (kept -> target) of EVERY list
__global__ void rg_set_rows_kernel(uint32_t *rows, uint32_t S, const uint32_t *__restrict__ lists, uint32_t K, uint32_t count,
                                   uint32_t sq0, uint32_t t0, int by_pivot, const uint32_t *__restrict__ last,
                                   const uint32_t *__restrict__ out, const uint32_t *__restrict__ cnt, uint32_t M,
                                   u64 *__restrict__ props) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint32_t target = by_pivot ? lists[(u64)t * K] : t0 + t;
    const uint32_t n = cnt[t];
    const bool live = target != NO_ID && n != NO_ID;
    if (live && (!by_pivot || last[target] == sq0 + t + 1)) {
        uint32_t *row = rows + (u64)target * S;
        row[0] = n;
        for (uint32_t j = 0; j < n; ++j) row[1 + j] = out[(u64)t * M + j];
    }
    for (uint32_t j = 0; j < M; ++j) {
        const uint32_t u = live && j < n ? out[(u64)t * M + j] : NO_ID;
        props[(u64)t * M + j] = u != NO_ID ? (((u64)u << 32) | target) : NO_KEY;
    }
}

// second reverse pass of the projection (:1096-1099): every row proposes its owner to the rows it names
__global__ void rg_props_from_rows_kernel(const uint32_t *__restrict__ rows, uint32_t S, u64 v0, u64 count, uint32_t M,
                                          u64 *__restrict__ props) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint32_t *row = rows + (v0 + t) * S;
    const uint32_t deg = row[0];
    for (uint32_t j = 0; j < M; ++j) props[t * M + j] = j < deg ? (((u64)row[1 + j] << 32) | (uint32_t)(v0 + t)) : NO_KEY;
}

// Reverse edges of a batch.  One wave per run of equal destinations in the sorted proposals: new sources (not yet in the
// row, not the destination itself, each once) are appended while the row stays within `cap`; a row that would overflow
// is pruned again over {row entries, new sources} (PruneProjectionReverseCandidates / ...InternalReverse...): distances
// to the destination, sort, occlusion, back-fill in list order when asked.  Lists longer than 255 keep their 255
// nearest (a hub's row is decided by its nearest candidates: the scan stops once M are kept).
template <bool IP, int NF4>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) rg_reverse_kernel(const float *__restrict__ base, u64 ld, uint32_t dim, uint32_t *rows,
                                                       uint32_t S, uint32_t cap, uint32_t M, int list_backfill,
                                                       const u64 *__restrict__ props, u64 nprops,
                                                       const uint32_t *__restrict__ heads, uint32_t nheads) {
    extern __shared__ __align__(16) unsigned char smem[];
    const ListLds L = carve(smem, dim, M);
    const int lane = threadIdx.x;
    for (uint32_t h = blockIdx.x; h < nheads; h += gridDim.x) {
        const u64 p0 = heads[h];
        const uint32_t des = (uint32_t)(props[p0] >> 32);
        uint32_t *row = rows + (u64)des * S;
        const uint32_t deg = row[0] < cap ? row[0] : cap;
        // pass 1: count the new sources
        uint32_t nnew = 0;
        u64 pend = p0;
        for (u64 c0 = p0;; c0 += 64) {
            const u64 i = c0 + (u64)lane;
            const u64 key = i < nprops ? props[i] : NO_KEY;
            const bool mine = key != NO_KEY && (uint32_t)(key >> 32) == des;
            const uint32_t src = (uint32_t)key;
            bool isnew = mine && src != des && !(i > p0 && props[i - 1] == key);
            for (uint32_t j = 0; j < deg && isnew; ++j) isnew = row[1 + j] != src;
            nnew += (uint32_t)__popcll(__ballot(isnew));
            const u64 mm = __ballot(mine);
            pend = c0 + (u64)__popcll(mm);
            if (mm != ~0ull) break;
        }
        if (nnew == 0) continue;
        if (deg + nnew <= cap) {  // room for all: append in ascending source id
            uint32_t w = deg;
            for (u64 c0 = p0; c0 < pend; c0 += 64) {
                const u64 i = c0 + (u64)lane;
                const u64 key = i < pend ? props[i] : NO_KEY;
                const uint32_t src = (uint32_t)key;
                bool isnew = i < pend && src != des && !(i > p0 && props[i - 1] == key);
                for (uint32_t j = 0; j < deg && isnew; ++j) isnew = row[1 + j] != src;
                const u64 m = __ballot(isnew);
                if (isnew) row[1 + w + below(m)] = src;
                w += (uint32_t)__popcll(m);
            }
            if (lane == 0) row[0] = w;
            __threadfence_block();
            wsync();
            continue;
        }
        // overflow: prune {row, new sources} again.  Distances to the destination; keep the 255 nearest as the list.
        for (uint32_t i = lane * 4; i < dim; i += 256)
            *reinterpret_cast<float4 *>(L.target + i) = *reinterpret_cast<const float4 *>(base + (u64)des * ld + i);
        for (uint32_t i = lane; i < 256; i += 64) L.keys[i] = NO_KEY, L.ids[i] = NO_ID;
        wsync();
        uint32_t n = 0;       // entries in L.keys (unsorted), L.ids = the same ids in list order (for the back-fill)
        u64 worst = 0;        // largest key held once the list is full
        auto offer = [&](bool ok, uint32_t id) {
            // lanes with ok offer (distance, id); the list keeps the 255 smallest keys
            float d = 0.f;
            if (ok) d = lane_dist_to_lds<IP>(base + (u64)id * ld, L.target, dim);
            const u64 key = ok ? mkkey(d, id) : NO_KEY;
            for (u64 it = __ballot(ok && (n < 255 || key < worst)); it;) {  // (re-checked below: `worst` only shrinks)
                const int j = __ffsll((long long)it) - 1;
                it &= it - 1;
                const uint32_t klo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, j);
                const uint32_t khi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), j);
                const u64 kj = ((u64)khi << 32) | klo;
                if (n < 255) {
                    if (lane == 0) L.keys[n] = kj, L.ids[n] = klo;
                    n++;
                    worst = kj > worst ? kj : worst;
                } else if (kj < worst) {  // replace the worst, find the new worst
                    u64 loc = 0;
                    uint32_t at = 0;
                    for (uint32_t q = lane; q < 255; q += 64) {
                        const u64 kq = L.keys[q];
                        if (kq == worst) at = q + 1;
                    }
                    const u64 am = __ballot(at != 0);
                    const int al = __ffsll((long long)am) - 1;
                    const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)at, al) - 1;
                    if (lane == 0) L.keys[slot] = kj, L.ids[slot] = klo;
                    wsync();
                    for (uint32_t q = lane; q < 255; q += 64) loc = L.keys[q] > loc ? L.keys[q] : loc;
                    for (int o = 32; o > 0; o >>= 1) {
                        const u64 other = ((u64)(uint32_t)__shfl_xor((int)(uint32_t)(loc >> 32), o) << 32) |
                                          (uint32_t)__shfl_xor((int)(uint32_t)loc, o);
                        loc = other > loc ? other : loc;
                    }
                    worst = loc;
                }
                wsync();
            }
        };
        for (uint32_t c0 = 0; c0 < deg; c0 += 64) {
            const uint32_t i = c0 + (uint32_t)lane;
            const uint32_t id = i < deg ? row[1 + i] : NO_ID;
            offer(i < deg && id != des && id != NO_ID, id);
        }
        for (u64 c0 = p0; c0 < pend; c0 += 64) {
            const u64 i = c0 + (u64)lane;
            const u64 key = i < pend ? props[i] : NO_KEY;
            const uint32_t src = (uint32_t)key;
            bool isnew = i < pend && src != des && !(i > p0 && props[i - 1] == key);
            for (uint32_t j = 0; j < deg && isnew; ++j) isnew = row[1 + j] != src;
            offer(isnew, src);
        }
        wsync();
        sort_small(L.keys, L.tmp, 256, lane);
        uint32_t live = 0;
        {
            uint32_t w = 0;
            for (uint32_t c0 = 0; c0 < 256; c0 += 64) {
                const u64 k = L.keys[c0 + lane];
                const u64 m = __ballot(k != NO_KEY);
                if (k != NO_KEY) L.tmp[w + below(m)] = k;
                w += (uint32_t)__popcll(m);
            }
            live = w;
            wsync();
            for (uint32_t i = lane; i < 256; i += 64) L.keys[i] = i < live ? L.tmp[i] : NO_KEY;
            wsync();
        }
        uint32_t nk = greedy_prune<IP, NF4>(base, ld, dim, L.keys, live, 0, des, M, L.krows, L.kids, lane);
        wsync();
        if (list_backfill) nk = backfill(L.ids, n, des, M, L.kids, nk, lane);  // :1602-1606, list order
        wsync();
        for (uint32_t i = lane; i < nk; i += 64) row[1 + i] = L.kids[i];
        if (lane == 0) row[0] = nk;
        __threadfence_block();
        wsync();
    }
}

// heads of the runs of equal destinations in a sorted proposal array (compacted with an atomic cursor, then sorted by the
// caller so that the order of the waves is fixed; each run is independent of the others anyway)
__global__ void rg_heads_kernel(const u64 *__restrict__ props, u64 nprops, uint32_t *heads, uint32_t *nheads) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nprops) return;
    const u64 k = props[i];
    if (k == NO_KEY) return;
    if (i > 0 && (props[i - 1] >> 32) == (k >> 32)) return;
    heads[atomicAdd(nheads, 1u)] = (uint32_t)i;
}

// final rows (:1262-1275): the projection row, then up to 2 M supply edges it does not hold yet
__global__ void rg_merge_rows_kernel(const uint32_t *__restrict__ grows, uint32_t Sg, const uint32_t *__restrict__ srows,
                                     uint32_t Ss, uint32_t *__restrict__ adj, uint32_t Sf, u64 n, uint32_t M) {
    const u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint32_t *g = grows + v * Sg, *s = srows + v * Ss;
    uint32_t *o = adj + v * Sf;
    const uint32_t gd = g[0], sd = s[0];
    uint32_t w = 0;
    for (uint32_t j = 0; j < gd; ++j) {
        const uint32_t u = g[1 + j];
        bool dup = u == (uint32_t)v;
        for (uint32_t q = 0; q < w && !dup; ++q) dup = o[1 + q] == u;
        if (!dup) o[1 + w++] = u;
    }
    const uint32_t g0 = w;
    uint32_t added = 0;
    for (uint32_t j = 0; j < sd && added < 2 * M; ++j) {
        const uint32_t u = s[1 + j];
        bool in = u == (uint32_t)v;
        for (uint32_t q = 0; q < w && !in; ++q) in = o[1 + q] == u;
        (void)g0;
        if (!in) o[1 + w++] = u, added++;
    }
    o[0] = w;
}

__global__ void rg_copy_rows_kernel(const uint32_t *__restrict__ src, uint32_t Ssrc, uint32_t *__restrict__ dst, uint32_t Sdst,
                                    u64 n) {
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 v = w / Sdst;
    const uint32_t j = (uint32_t)(w % Sdst);
    if (v >= n) return;
    const uint32_t deg = src[v * Ssrc];
    dst[v * Sdst + j] = (j <= deg && j < Ssrc) ? src[v * Ssrc + j] : 0u;
}

template <typename F>
int with_shape(int ip, uint32_t dim, F &&f) {
    // compile-time row lengths for the benchmark shapes (candidate rows live in registers); anything else re-reads rows
    if (ip) {
        if (dim == 200) return f(std::integral_constant<bool, true>(), std::integral_constant<int, 50>());
        if (dim == 128) return f(std::integral_constant<bool, true>(), std::integral_constant<int, 32>());
        return f(std::integral_constant<bool, true>(), std::integral_constant<int, 0>());
    }
    if (dim == 200) return f(std::integral_constant<bool, false>(), std::integral_constant<int, 50>());
    if (dim == 128) return f(std::integral_constant<bool, false>(), std::integral_constant<int, 32>());
    return f(std::integral_constant<bool, false>(), std::integral_constant<int, 0>());
}

size_t list_lds_bytes(uint32_t dim, uint32_t M) {
    return (size_t)dim * 4 + (size_t)M * dim * 4 + 256 * 8 * 2 + ((M + 3) & ~3u) * 4 + 256 * 4 + 64;
}

template <typename K>
int set_lds(K kernel, size_t lds) {
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

}  // namespace

extern "C" {

int roar_rg_list_prune(const float *base, uint64_t ld, uint32_t dim, int ip, int mode, const uint32_t *lists, uint32_t K,
                       uint32_t C, uint32_t count, uint32_t t0, const uint32_t *rows, uint32_t S, const uint32_t *grows,
                       uint32_t Sg, uint32_t M, uint32_t only_longer_than, uint32_t *out, uint32_t *cnt, void *stream) {
    if (count == 0) return 0;
    const size_t lds = list_lds_bytes(dim, M);
    const uint32_t grid = count < 16384u ? count : 16384u;
    return with_shape(ip, dim, [&](auto IPc, auto NF) -> int {
        auto kern = rg_list_prune_kernel<decltype(IPc)::value, decltype(NF)::value>;
        int rc = set_lds(kern, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, (hipStream_t)stream, base, (u64)ld, dim, mode, lists, K, C, count, t0,
                           rows, S, grows, Sg, M, only_longer_than, out, cnt);
        return (int)hipGetLastError();
    });
}

int roar_rg_supply_prune(const float *base, uint64_t ld, uint32_t dim, int ip, const uint32_t *res_ids, const float *res_d,
                         uint32_t Lres, uint32_t count, uint32_t t0, const uint32_t *grows, uint32_t Sg, uint32_t M,
                         unsigned long long *keyscratch, uint32_t scratch_waves, uint32_t *out, uint32_t *cnt, void *stream) {
    if (count == 0) return 0;
    const size_t lds = list_lds_bytes(dim, M);
    uint32_t grid = count < scratch_waves ? count : scratch_waves;
    return with_shape(ip, dim, [&](auto IPc, auto NF) -> int {
        auto kern = rg_supply_prune_kernel<decltype(IPc)::value, decltype(NF)::value>;
        int rc = set_lds(kern, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, (hipStream_t)stream, base, (u64)ld, dim, res_ids, res_d, Lres, count,
                           t0, grows, Sg, M, (u64 *)keyscratch, out, cnt);
        return (int)hipGetLastError();
    });
}

// lists[count][K]: the first C ids of every list must name base points (id < n; 0xFFFFFFFF = no entry) -- the device form of the
// check roar_build_roargraph makes on the host ("learn base knn file error").  *bad is raised, never cleared.
__global__ void rg_check_lists_kernel(const uint32_t *__restrict__ lists, uint32_t K, uint32_t C, u64 cells, uint32_t n, uint32_t *bad) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells) return;
    const uint32_t id = lists[(i / C) * K + (i % C)];
    if (id >= n && id != 0xFFFFFFFFu) *bad = 1u;
}

extern "C" int roar_rg_check_lists(const uint32_t *lists, uint32_t K, uint32_t C, uint64_t count, uint32_t n, uint32_t *bad_dev, void *stream) {
    const u64 cells = (u64)count * C;
    if (cells == 0) return 0;
    hipLaunchKernelGGL(rg_check_lists_kernel, dim3((uint32_t)((cells + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lists, K, C,
                       cells, n, bad_dev);
    return (int)hipGetLastError();
}

int roar_rg_last_list(const uint32_t *lists, uint32_t K, uint32_t count, uint32_t sq0, uint32_t *last, void *stream) {
    if (count == 0) return 0;
    hipLaunchKernelGGL(rg_last_list_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, lists, K, count, sq0,
                       last);
    return (int)hipGetLastError();
}

int roar_rg_set_rows(uint32_t *rows, uint32_t S, const uint32_t *lists, uint32_t K, uint32_t count, uint32_t sq0, uint32_t t0,
                     int by_pivot, const uint32_t *last, const uint32_t *out, const uint32_t *cnt, uint32_t M,
                     unsigned long long *props, void *stream) {
    if (count == 0) return 0;
    hipLaunchKernelGGL(rg_set_rows_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, rows, S, lists, K, count,
                       sq0, t0, by_pivot, last, out, cnt, M, (u64 *)props);
    return (int)hipGetLastError();
}

// sort `nprops` proposals (keys_in -> keys_out), find the runs, merge them into the rows
int roar_rg_reverse(const float *base, uint64_t ld, uint32_t dim, int ip, uint32_t *rows, uint32_t S, uint32_t cap, uint32_t M,
                    int list_backfill, unsigned long long *keys_in, unsigned long long *keys_out, uint64_t nprops, void *sort_tmp,
                    size_t sort_tmp_bytes, uint32_t *heads, uint32_t *heads_sorted, uint32_t *nheads_dev, void *stream) {
    if (nprops == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    size_t tb = sort_tmp_bytes;
    hipError_t e = rocprim::radix_sort_keys(sort_tmp, tb, (u64 *)keys_in, (u64 *)keys_out, (size_t)nprops, 0, 64, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(nheads_dev, 0, 4, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(rg_heads_kernel, dim3((uint32_t)((nprops + 255) / 256)), dim3(256), 0, st, (const u64 *)keys_out,
                       (u64)nprops, heads, nheads_dev);
    uint32_t nheads = 0;
    e = hipMemcpyAsync(&nheads, nheads_dev, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return (int)e;
    if (nheads == 0) return 0;
    tb = sort_tmp_bytes;
    e = rocprim::radix_sort_keys(sort_tmp, tb, heads, heads_sorted, (size_t)nheads, 0, 32, st);
    if (e != hipSuccess) return (int)e;
    const size_t lds = list_lds_bytes(dim, M);
    const uint32_t grid = nheads < 16384u ? nheads : 16384u;
    return with_shape(ip, dim, [&](auto IPc, auto NF) -> int {
        auto kern = rg_reverse_kernel<decltype(IPc)::value, decltype(NF)::value>;
        int rc = set_lds(kern, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, st, base, (u64)ld, dim, rows, S, cap, M, list_backfill,
                           (const u64 *)keys_out, (u64)nprops, (const uint32_t *)heads_sorted, nheads);
        return (int)hipGetLastError();
    });
}

int roar_rg_props_from_rows(const uint32_t *rows, uint32_t S, uint64_t v0, uint64_t count, uint32_t M, unsigned long long *props,
                            void *stream) {
    if (count == 0) return 0;
    hipLaunchKernelGGL(rg_props_from_rows_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows,
                       S, (u64)v0, (u64)count, M, (u64 *)props);
    return (int)hipGetLastError();
}

int roar_rg_sort_tmp_bytes(uint64_t max_keys, size_t *bytes) {
    size_t a = 0, b = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, a, (u64 *)nullptr, (u64 *)nullptr, (size_t)max_keys, 0, 64, nullptr);
    if (e != hipSuccess) return (int)e;
    e = rocprim::radix_sort_keys(nullptr, b, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)max_keys, 0, 32, nullptr);
    if (e != hipSuccess) return (int)e;
    *bytes = a > b ? a : b;
    return 0;
}

int roar_rg_merge_rows(const uint32_t *grows, uint32_t Sg, const uint32_t *srows, uint32_t Ss, uint32_t *adj, uint32_t Sf,
                       uint64_t n, uint32_t M, void *stream) {
    hipLaunchKernelGGL(rg_merge_rows_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grows, Sg,
                       srows, Ss, adj, Sf, (u64)n, M);
    return (int)hipGetLastError();
}

int roar_rg_copy_rows(const uint32_t *src, uint32_t Ssrc, uint32_t *dst, uint32_t Sdst, uint64_t n, void *stream) {
    const uint64_t words = n * Sdst;
    hipLaunchKernelGGL(rg_copy_rows_kernel, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, Ssrc,
                       dst, Sdst, (u64)n);
    return (int)hipGetLastError();
}

}  // extern "C"
