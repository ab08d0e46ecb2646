// roar_build.hip -- in-HBM bootstrap graph builder (gfx950).
//
// Why it exists: no dataset or index can be downloaded to the GPU box, so every measurement at
// 1 M - 100 M points needs an index built there, in minutes, from synthetic vectors.  This is NOT
// the reference's LinkProjection (src/index_bipartite.cpp:1043-1277, SURVEY 8(f) "next" #1); it is an
// incremental batch insertion that reuses the search kernel as its work-horse and emits the
// same fixed-stride adjacency the search reads, from which the reference's index file
// (u32 ep; u32 npts; npts x {u32 deg; u32 nbr[deg]}) is written unchanged.
//
//   entry point   node nearest the centroid, as CalculateProjectionep (index_bipartite.cpp:2004-2041)
//   seeds         the first `seed` nodes (+ the entry point) start with pseudo-random links
//   batches       nodes [a, b): search the graph built so far with the node's own vector
//                 (L_build), keep the M nearest as out-edges, append the node to each of them
//                 (reverse edge) while that row has room (max_degree)
//   projection    optional: training queries (drawn from the QUERY distribution) are searched and
//                 the nearest result (the pivot) is linked to the other results, which is the idea
//                 of the projected bipartite graph in its simplest form
//   rescue        nodes nobody points to are forced into a row of one of their own neighbours
//
// Determinism.  No kernel here appends to an adjacency row through an atomic cursor.  Every step that links nodes
// writes its edges as PROPOSALS -- u64 keys (destination row << 32 | new neighbour) at slots fixed by the list that
// makes them -- the proposals of a batch are radix-sorted (rocPRIM) and applied by one thread per destination row in
// ascending neighbour id: skip what the row already holds, append while the row has room, drop (or, for the orphan
// rescue, overwrite the last slot) beyond.  Searches return bit-identical results whatever the scheduling, so two builds
// of the same data make the same graph (roar_get_info "graph_hash"; tests/test_gpu_parity.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <new>

#include <rocprim/rocprim.hpp>

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ uint32_t f2ord_b(float f) {
    uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// column sums of the base matrix: sums[d] += sum over a slab of rows (atomics, one per column per block)
__global__ void colsum_kernel(const float *__restrict__ base, u64 ld, uint32_t dim, u64 n, u64 rows_per_block,
                              float *sums) {
    const u64 r0 = (u64)blockIdx.x * rows_per_block;
    const u64 r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    for (uint32_t c = threadIdx.x; c < dim; c += blockDim.x) {
        float s = 0.f;
        for (u64 r = r0; r < r1; ++r) s += base[r * ld + c];
        atomicAdd(&sums[c], s);
    }
}

// squared L2 distance of every row to centre (= sums / n); global argmin through a u64 atomicMin
__global__ void nearest_to_centre_kernel(const float *__restrict__ base, u64 ld, uint32_t dim, u64 n,
                                         const float *__restrict__ sums, u64 *best) {
    const int lane = threadIdx.x & 63;
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    const float inv = 1.0f / (float)n;
    u64 local = ~0ull;
    for (u64 r = wave; r < n; r += nwaves) {
        float s = 0.f;
        for (uint32_t c = lane; c < dim; c += 64) {
            const float d = sums[c] * inv - base[r * ld + c];
            s += d * d;
        }
        s = wave_sum(s);
        const u64 key = ((u64)f2ord_b(s) << 32) | (uint32_t)r;
        local = key < local ? key : local;
    }
    if (lane == 0 && local != ~0ull) atomicMin(best, local);
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// seed rows: M pseudo-random links among the seed set {0..seed-1} (+ ep mapped to index `seed`)
__global__ void seed_rows_kernel(uint32_t *adj, uint32_t S, uint32_t seed, uint32_t ep, uint32_t M) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = seed + (ep >= seed ? 1u : 0u);
    if (i >= total) return;
    const uint32_t v = i < seed ? i : ep;
    uint32_t *row = adj + (u64)v * S;
    uint32_t deg = 0;
    for (uint32_t j = 0; j < M && j + 1 < total; ++j) {
        uint32_t t = mix32(i * 0x9E3779B9u + j * 0x85EBCA6Bu + 12345u) % total;
        if (t == i) t = (t + 1) % total;
        const uint32_t u = t < seed ? t : ep;
        bool dup = false;
        for (uint32_t q = 0; q < deg; ++q) dup |= row[1 + q] == u;
        if (!dup) row[1 + deg++] = u;
    }
    row[0] = deg;
}

constexpr u64 NO_PROP = ~0ull;
__device__ __forceinline__ u64 make_prop(uint32_t dst, uint32_t src) { return ((u64)dst << 32) | src; }

// link nodes a..a+count-1 (thread per node): out-edges = its M search results, and the reverse edges; 2 M proposal slots
// per node.  (relink: the node is a seed whose random links are being replaced -- its new out-edges are proposed like
// any other, the random ones stay until the row is cut.)
__global__ void link_kernel(uint32_t a, uint32_t count, const uint32_t *__restrict__ res, uint32_t M, uint32_t ep, int relink,
                            u64 *__restrict__ props) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t v = a + i;
    u64 *out = props + (u64)i * 2 * M;
    for (uint32_t j = 0; j < M; ++j) {
        const uint32_t u = res[(u64)i * M + j];
        const bool ok = u != 0xFFFFFFFFu && u != v && (relink || v != ep);  // the entry point was linked with the seeds
        out[2 * j] = ok ? make_prop(v, u) : NO_PROP;
        out[2 * j + 1] = ok ? make_prop(u, v) : NO_PROP;
    }
}

// One thread per run of equal destinations in the sorted proposal array: merge the run into row `dst`.
// overflow_last != 0: a full row takes the proposal in its last slot (orphan rescue); else it is dropped.
__global__ void apply_props_kernel(uint32_t *adj, uint32_t S, uint32_t cap, const u64 *__restrict__ props, u64 nprops,
                                   uint32_t *indeg, int overflow_last) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nprops) return;
    const u64 key = props[i];
    if (key == NO_PROP) return;
    const uint32_t dst = (uint32_t)(key >> 32);
    if (i > 0 && (uint32_t)(props[i - 1] >> 32) == dst) return;  // not the head of its run
    uint32_t *row = adj + (u64)dst * S;
    uint32_t deg = row[0] < cap ? row[0] : cap;
    u64 prev = NO_PROP;
    for (u64 j = i; j < nprops; ++j) {
        const u64 kj = props[j];
        if (kj == NO_PROP || (uint32_t)(kj >> 32) != dst) break;
        if (kj == prev) continue;  // the same edge proposed twice
        prev = kj;
        const uint32_t src = (uint32_t)kj;
        if (src == dst) continue;
        bool have = false;
        for (uint32_t q = 0; q < deg && !have; ++q) have = row[1 + q] == src;
        if (have) continue;
        if (deg < cap) {
            row[1 + deg++] = src;
            if (indeg) atomicAdd(&indeg[src], 1u);
        } else if (overflow_last) {
            const uint32_t victim = row[cap];
            row[cap] = src;
            if (indeg) {
                atomicAdd(&indeg[src], 1u);
                atomicSub(&indeg[victim], 1u);
            }
        } else {
            break;  // the row is full and stays as it is: nothing further in this run can change it
        }
    }
    row[0] = deg;
}

// 64-bit order-independent-across-rows hash of the graph: sum over rows of mix(row id, degree, neighbours in order)
__device__ __forceinline__ u64 mix64(u64 x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    return x;
}
__global__ void graph_hash_kernel(const uint32_t *__restrict__ adj, uint32_t S, u64 n, uint32_t ep, u64 *out) {
    const u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 h = 0;
    if (v < n) {
        const uint32_t *row = adj + v * S;
        const uint32_t deg = row[0];
        h = mix64(v * 0x9E3779B97F4A7C15ull + deg);
        for (uint32_t j = 0; j < deg; ++j) h = mix64(h ^ ((u64)row[1 + j] + 0x632BE59BD9B4E019ull * (j + 1)));
        if (v == 0) h += mix64(0xE9ull + ep);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, h);
}

// rows of stride S_old -> rows of stride S_new (<= S_old) in place, ascending row order: row v's new position never
// lies beyond its old one, and a block only moves rows whose old words nobody before it still has to read ... so the
// copy goes through a second buffer instead (a 10 M-row graph is 1.6 GB at stride 40 words).
__global__ void restride_rows_kernel(const uint32_t *__restrict__ src, uint32_t S_old, uint32_t *__restrict__ dst,
                                     uint32_t S_new, u64 n) {
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 v = w / S_new;
    const uint32_t j = (uint32_t)(w % S_new);
    if (v >= n) return;
    const uint32_t deg = src[v * S_old];
    dst[v * S_new + j] = (j <= deg && j < S_old) ? src[v * S_old + j] : 0u;
}

// Occlusion-prune-and-link, one wave per list.  Two ways to name the node that receives the edges:
//   pivot mode  (target_base < 0): list[0] is the target, list[1..C) are the candidates -- the projection
//               step: list = a training query's nearest base points (LinkProjection,
//               src/index_bipartite.cpp:1059-1093);
//   target mode (target_base >= 0): list t belongs to node target_base + t and all C entries are
//               candidates -- insertion of a new node from its own search results, and the connectivity
//               supplement (:1188-1262) where a node is searched in the finished graph.
// Rule (PruneBiSearchBaseGetBase, :1612-1694 / PruneProjectionBaseSearchCandidates, :1846-1940):
// candidates are ordered by their base-to-base distance to the target; one is kept unless an already
// kept one is closer to it than the target is; optionally back-filled with the nearest leftovers up to
// Mp.  The edges target -> kept and kept -> target leave as proposals (2 x 63 slots per list), see apply_props_kernel.
template <bool IP>
__global__ void __launch_bounds__(64) prune_link_kernel(const uint32_t *adj, uint32_t S, uint32_t cap_t,
                                                       const float *__restrict__ base, u64 ld, uint32_t dim,
                                                       uint32_t count, const uint32_t *__restrict__ res,
                                                       uint32_t M, uint32_t C, uint32_t Mp, long long target_base,
                                                       int skip_existing, int backfill, uint32_t skip_node, int flags,
                                                       u64 *__restrict__ props) {
    extern __shared__ __align__(16) float rows[];  // (ncand + 1) x (dim + 1) floats, row 0 = target
    const int lane = threadIdx.x;
    const uint32_t rs = dim + 1;
    const bool tmode = target_base >= 0;
    const uint32_t ncand = tmode ? C : C - 1;  // <= 63
    for (uint32_t t = blockIdx.x; t < count; t += gridDim.x) {
        const uint32_t *r = res + (u64)t * M;
        const uint32_t target = tmode ? (uint32_t)(target_base + t) : r[0];
        if (target == 0xFFFFFFFFu || target == skip_node) {
            if (lane > 0 && (uint32_t)lane <= ncand) {
                u64 *out = props + ((u64)t * ncand + (uint32_t)(lane - 1)) * 2;
                out[0] = out[1] = NO_PROP;
            }
            continue;
        }
        uint32_t myid = 0xFFFFFFFFu;
        if (lane == 0) myid = target;
        else if ((uint32_t)lane <= ncand) myid = r[tmode ? lane - 1 : lane];
        if (lane > 0 && myid == target) myid = 0xFFFFFFFFu;
        const uint32_t *trow = adj + (u64)target * S;
        if (skip_existing && lane > 0 && myid != 0xFFFFFFFFu) {
            const uint32_t td = trow[0] < cap_t ? trow[0] : cap_t;
            bool have = false;
            for (uint32_t q = 0; q < td; ++q) have |= trow[1 + q] == myid;
            if (have) myid = 0xFFFFFFFFu;
        }
        // gather the rows (coalesced: the wave walks one row at a time)
        for (uint32_t c = 0; c <= ncand; ++c) {
            const uint32_t id = (uint32_t)__builtin_amdgcn_readlane((int)myid, (int)c);
            if (id == 0xFFFFFFFFu) continue;
            const float *src = base + (u64)id * ld;
            for (uint32_t i = lane; i < dim; i += 64) rows[c * rs + i] = src[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        auto dist_rows = [&](uint32_t a, uint32_t b) -> float {
            float s = 0.f;
            const float *x = rows + a * rs, *y = rows + b * rs;
            for (uint32_t i = 0; i < dim; ++i) {
                if (IP) s = __builtin_fmaf(x[i], y[i], s);
                else { const float d = x[i] - y[i]; s = __builtin_fmaf(d, d, s); }
            }
            return IP ? -s : s;
        };
        const bool cand = lane > 0 && (uint32_t)lane <= ncand && myid != 0xFFFFFFFFu;
        float dp = cand ? dist_rows((uint32_t)lane, 0) : __builtin_inff();
        // flags bit 0: keep the list's own order (nearest to the training query first) instead of ordering
        // by distance to the target; bit 1: no occlusion test (the first Mp live candidates are kept)
        const u64 key = cand ? (((u64)((flags & 1) ? (uint32_t)lane : f2ord_b(dp)) << 32) | myid) : ~0ull;
        // order candidates by (distance to target, id); duplicates of an earlier lane drop out
        uint32_t rank = 0;
        bool dupl = false;
        for (uint32_t j = 1; j <= ncand; ++j) {
            const uint32_t klo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, (int)j);
            const uint32_t khi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), (int)j);
            const u64 kj = ((u64)khi << 32) | klo;
            if (kj == ~0ull || (int)j == lane) continue;
            if (kj < key) rank++;
            if ((uint32_t)kj == (uint32_t)key && (int)j < lane) dupl = true;  // same id earlier in the list
        }
        const bool live = cand && !dupl;
        // greedy occlusion in sorted order (a duplicate has an equal key, so it never raised a rank;
        // ranks of live lanes may have gaps, hence the loop over every possible rank)
        bool kept = false;
        uint32_t nkept = 0;
        for (uint32_t o = 0; o <= ncand && nkept < Mp; ++o) {
            const u64 who = __ballot(live && rank == o);
            if (!who) continue;
            const int c = __ffsll((long long)who) - 1;
            const float dc = __shfl(dp, c);
            bool occl = false;
            if (kept && !(flags & 2)) occl = dist_rows((uint32_t)lane, (uint32_t)c) < dc;
            if (__ballot(occl) == 0) {
                if (lane == c) kept = true;
                nkept++;
            }
        }
        if (backfill) {  // :1685-1691
            for (uint32_t o = 0; o <= ncand && nkept < Mp; ++o) {
                const u64 who = __ballot(live && !kept && rank == o);
                if (!who) continue;
                if (lane == __ffsll((long long)who) - 1) kept = true;
                nkept++;
            }
        }
        // target -> kept and kept -> target, as proposals: slot pair of lane l of list t (lanes 1..63)
        if (lane > 0 && (uint32_t)lane <= ncand) {
            u64 *out = props + ((u64)t * ncand + (uint32_t)(lane - 1)) * 2;
            out[0] = kept ? make_prop(target, myid) : NO_PROP;
            out[1] = kept ? make_prop(myid, target) : NO_PROP;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// connectivity supplement without pruning (cf. the reference's "connectivity enhancement", src/index_bipartite.cpp:
// 1188-1262): node v was searched in the finished graph with its own vector under the index metric; v <-> each of
// its first Kc results, as proposals (2 Kc slots per node).
__global__ void supply_kernel(uint32_t a, uint32_t count, const uint32_t *__restrict__ res, uint32_t M, uint32_t Kc,
                              u64 *__restrict__ props) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t v = a + i;
    u64 *out = props + (u64)i * 2 * Kc;
    uint32_t done = 0;
    for (uint32_t j = 0; j < M && done < Kc; ++j) {
        const uint32_t u = res[(u64)i * M + j];
        if (u == 0xFFFFFFFFu || u == v) continue;
        out[2 * done] = make_prop(u, v);
        out[2 * done + 1] = make_prop(v, u);
        done++;
    }
    for (; done < Kc; ++done) out[2 * done] = out[2 * done + 1] = NO_PROP;
}

__global__ void snapshot_degree_kernel(const uint32_t *adj, uint32_t S, uint32_t max_deg, u64 n, uint32_t *base_deg) {
    const u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint32_t d = adj[v * S];
    base_deg[v] = d < max_deg ? d : max_deg;
}

__global__ void clamp_degree_kernel(uint32_t *adj, uint32_t S, uint32_t max_deg, u64 n) {
    const u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    if (adj[v * S] > max_deg) adj[v * S] = max_deg;
}

// nodes without an in-edge propose themselves to the row of their r-th own neighbour (one slot per node; applied with
// overflow_last: a full row gives up its last slot)
__global__ void rescue_kernel(const uint32_t *adj, uint32_t S, u64 n, uint32_t ep, uint32_t r, const uint32_t *indeg,
                              uint32_t *orphans, u64 *__restrict__ props) {
    const u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    u64 p = NO_PROP;
    if (v != ep && indeg[v] == 0) {
        atomicAdd(orphans, 1u);
        const uint32_t *row = adj + v * S;
        const uint32_t deg = row[0];
        if (deg) p = make_prop(row[1 + (r % deg)], (uint32_t)v);
    }
    props[v] = p;
}

// final clean-up of a built graph: drop self loops and repeated neighbours (first occurrence kept,
// order preserved), so rows hold distinct ids like the rows made from a file (roar_set_graph_csr).
__global__ void dedupe_rows_kernel(uint32_t *adj, uint32_t S, uint32_t max_deg, u64 n) {
    const u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    uint32_t *row = adj + v * S;
    const uint32_t deg = row[0] < max_deg ? row[0] : max_deg;
    uint32_t out = 0;
    for (uint32_t j = 0; j < deg; ++j) {
        const uint32_t u = row[1 + j];
        bool dup = u == (uint32_t)v;
        for (uint32_t q = 0; q < out && !dup; ++q) dup = row[1 + q] == u;
        if (!dup) row[1 + out++] = u;
    }
    row[0] = out;
}

// Rows longer than `keep` are cut back to the `keep` neighbours nearest to the node itself (index metric),
// nearest first.  Hub rows collect far more proposals than they may hold; which of them stay decides how
// much a hop through the hub is worth (first come first kept: recall 0.9 at L_pq = 30 on T2I-10M; nearest kept:
// at L_pq = 20).  One wave per long row: distances to the node's own vector (row gathers, four neighbours in
// flight), rank by counting in LDS, rewrite.
template <bool IP>
__global__ void __launch_bounds__(64) cut_rows_kernel(uint32_t *adj, uint32_t S, uint32_t cap, uint32_t keep,
                                                      const float *__restrict__ base, u64 ld, uint32_t dim, u64 n,
                                                      uint32_t *base_deg) {
    extern __shared__ __align__(16) unsigned char cut_smem[];
    u64 *keys = reinterpret_cast<u64 *>(cut_smem);                  // [1024]
    uint32_t *ids_l = reinterpret_cast<uint32_t *>(cut_smem + 8192);  // [1024]
    float *hv = reinterpret_cast<float *>(cut_smem + 12288);          // [dim]
    const int lane = threadIdx.x;
    for (u64 v = blockIdx.x; v < n; v += gridDim.x) {
        uint32_t *row = adj + v * S;
        uint32_t deg = row[0] < cap ? row[0] : cap;
        if (deg > 1024u) deg = 1024u;
        if (deg <= keep) continue;
        const float *hrow = base + v * ld;
        for (uint32_t i = lane * 4; i < dim; i += 256)
            *reinterpret_cast<float4 *>(hv + i) = *reinterpret_cast<const float4 *>(hrow + i);
        for (uint32_t j = lane; j < deg; j += 64) ids_l[j] = row[1 + j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t j0 = 0; j0 < deg; j0 += 4) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t j = j0 + u < deg ? j0 + u : deg - 1;
                const float *r = base + (u64)ids_l[j] * ld;
                for (uint32_t i = lane * 4; i < dim; i += 256) {
                    const float4 x = *reinterpret_cast<const float4 *>(r + i);
                    const float4 y = *reinterpret_cast<const float4 *>(hv + i);
                    if (IP) {
                        acc[u] += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
                    } else {
                        const float a = x.x - y.x, b = x.y - y.y, c = x.z - y.z, e = x.w - y.w;
                        acc[u] += a * a + b * b + c * c + e * e;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float t = wave_sum(acc[u]);
                if (lane == 0 && j0 + u < deg) keys[j0 + u] = ((u64)f2ord_b(IP ? -t : t) << 32) | ids_l[j0 + u];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t i = lane; i < deg; i += 64) {
            const u64 key = keys[i];
            uint32_t rank = 0;
            for (uint32_t q = 0; q < deg; ++q) rank += keys[q] < key ? 1u : 0u;
            if (rank < keep) row[1 + rank] = (uint32_t)key;  // ties between equal distances fall to the smaller id
        }
        if (lane == 0) {
            row[0] = keep;
            if (base_deg) base_deg[v] = 0;  // the row is no longer [phase-1 edges | projection edges]
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// optional trim: a row that received at least `min_proj` projection edges keeps only its first
// `keep_base` phase-1 (base-to-base) edges; rows are [phase-1 edges | projection edges].
__global__ void trim_base_edges_kernel(uint32_t *adj, uint32_t S, uint32_t max_deg, const uint32_t *base_deg, u64 n,
                                       uint32_t min_proj, uint32_t keep_base) {
    const u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    uint32_t *row = adj + v * S;
    const uint32_t deg = row[0] < max_deg ? row[0] : max_deg;
    const uint32_t b = base_deg[v] < deg ? base_deg[v] : deg;
    if (deg - b < min_proj || b <= keep_base) return;
    uint32_t out = keep_base;
    for (uint32_t j = b; j < deg; ++j) row[1 + out++] = row[1 + j];
    row[0] = out;
}

__global__ void degree_stats_kernel(const uint32_t *adj, uint32_t S, u64 n, u64 *sum, uint32_t *maxd) {
    const u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t d = v < n ? adj[v * S] : 0u;
    uint32_t m = d;
    u64 s = d;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        const uint32_t t = __shfl_xor(m, o);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(sum, s);
        atomicMax(maxd, m);
    }
}

}  // namespace

extern "C" {

int roar_bk_dedupe(uint32_t *adj, uint32_t S, uint32_t max_deg, uint64_t n, void *stream) {
    hipLaunchKernelGGL(dedupe_rows_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, adj, S,
                       max_deg, (u64)n);
    return (int)hipGetLastError();
}

int roar_bk_cut_rows(uint32_t *adj, uint32_t S, uint32_t cap, uint32_t keep, const float *base, uint64_t ld, uint32_t dim,
                     int metric_ip, uint64_t n, uint32_t *base_deg, void *stream) {
    const size_t lds = 12288 + (size_t)dim * sizeof(float);
    const uint32_t grid = (uint32_t)(n < 8192 ? n : 8192);
    if (metric_ip)
        hipLaunchKernelGGL(cut_rows_kernel<true>, dim3(grid), dim3(64), lds, (hipStream_t)stream, adj, S, cap, keep, base,
                           (u64)ld, dim, (u64)n, base_deg);
    else
        hipLaunchKernelGGL(cut_rows_kernel<false>, dim3(grid), dim3(64), lds, (hipStream_t)stream, adj, S, cap, keep, base,
                           (u64)ld, dim, (u64)n, base_deg);
    return (int)hipGetLastError();
}

int roar_bk_trim(uint32_t *adj, uint32_t S, uint32_t max_deg, const uint32_t *base_deg, uint64_t n, uint32_t min_proj,
                 uint32_t keep_base, void *stream) {
    hipLaunchKernelGGL(trim_base_edges_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, adj,
                       S, max_deg, base_deg, (u64)n, min_proj, keep_base);
    return (int)hipGetLastError();
}

int roar_bk_degree_stats(const uint32_t *adj, uint32_t S, uint64_t n, unsigned long long *sum_dev, uint32_t *max_dev,
                         void *stream) {
    hipError_t e = hipMemsetAsync(sum_dev, 0, 8, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(max_dev, 0, 4, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(degree_stats_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, adj, S,
                       (u64)n, (u64 *)sum_dev, max_dev);
    return (int)hipGetLastError();
}

int roar_bk_entry_point(const float *base, uint64_t ld, uint32_t dim, uint64_t n, float *sums_dev /*dim*/,
                        unsigned long long *best_dev, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(sums_dev, 0, dim * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(best_dev, 0xFF, sizeof(unsigned long long), st);
    if (e != hipSuccess) return (int)e;
    const uint64_t rows_per_block = 4096;
    const uint32_t grid = (uint32_t)((n + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(colsum_kernel, dim3(grid), dim3(256), 0, st, base, (u64)ld, dim, (u64)n, (u64)rows_per_block,
                       sums_dev);
    hipLaunchKernelGGL(nearest_to_centre_kernel, dim3(2048), dim3(256), 0, st, base, (u64)ld, dim, (u64)n, sums_dev,
                       (u64 *)best_dev);
    return (int)hipGetLastError();
}

int roar_bk_seed(uint32_t *adj, uint32_t S, uint32_t seed, uint32_t ep, uint32_t M, void *stream) {
    const uint32_t total = seed + 1;
    hipLaunchKernelGGL(seed_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, adj, S, seed, ep,
                       M);
    return (int)hipGetLastError();
}

// ---- proposal workspace: two key buffers + the radix sort's scratch ------------------------------------------------
struct RoarBuildWs {
    u64 *a = nullptr, *b = nullptr;
    uint64_t cap = 0;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
};

int roar_bk_ws_create(uint64_t max_props, void **out) {
    RoarBuildWs *w = new (std::nothrow) RoarBuildWs();
    if (!w) return (int)hipErrorOutOfMemory;
    w->cap = max_props;
    hipError_t e = hipMalloc((void **)&w->a, (size_t)max_props * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&w->b, (size_t)max_props * 8);
    if (e == hipSuccess) e = rocprim::radix_sort_keys(nullptr, w->tmp_bytes, w->a, w->b, (size_t)max_props, 0, 64, nullptr);
    if (e == hipSuccess) e = hipMalloc(&w->tmp, w->tmp_bytes ? w->tmp_bytes : 16);
    if (e != hipSuccess) {
        (void)hipFree(w->a); (void)hipFree(w->b); (void)hipFree(w->tmp);
        delete w;
        return (int)e;
    }
    *out = w;
    return 0;
}

void roar_bk_ws_destroy(void *ws) {
    RoarBuildWs *w = (RoarBuildWs *)ws;
    if (!w) return;
    (void)hipFree(w->a); (void)hipFree(w->b); (void)hipFree(w->tmp);
    delete w;
}

uint64_t roar_bk_ws_capacity(void *ws) { return ws ? ((RoarBuildWs *)ws)->cap : 0; }
unsigned long long *roar_bk_ws_keys(void *ws) { return ws ? (unsigned long long *)((RoarBuildWs *)ws)->a : nullptr; }

// sort the proposals in w->a[0, nprops) and merge them into the rows (ascending destination, ascending neighbour)
int roar_bk_apply(void *ws, uint64_t nprops, uint32_t *adj, uint32_t S, uint32_t cap, uint32_t *indeg, int overflow_last,
                  void *stream) {
    RoarBuildWs *w = (RoarBuildWs *)ws;
    if (!w || nprops > w->cap) return (int)hipErrorInvalidValue;
    if (nprops == 0) return 0;
    size_t tb = w->tmp_bytes;
    hipError_t e = rocprim::radix_sort_keys(w->tmp, tb, w->a, w->b, (size_t)nprops, 0, 64, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(apply_props_kernel, dim3((uint32_t)((nprops + 255) / 256)), dim3(256), 0, (hipStream_t)stream, adj, S,
                       cap, w->b, (u64)nprops, indeg, overflow_last);
    return (int)hipGetLastError();
}

int roar_bk_link(void *ws, uint32_t *adj, uint32_t S, uint32_t max_deg, uint32_t a, uint32_t count, const uint32_t *res,
                 uint32_t M, uint32_t ep, uint32_t *indeg, int relink, void *stream) {
    if (count == 0) return 0;
    RoarBuildWs *w = (RoarBuildWs *)ws;
    const uint64_t np = (uint64_t)count * 2 * M;
    if (!w || np > w->cap) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(link_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, count, res, M, ep,
                       relink, w->a);
    return roar_bk_apply(ws, np, adj, S, max_deg, indeg, 0, stream);
}

int roar_bk_prune_link(void *ws, uint32_t *adj, uint32_t S, uint32_t cap, const float *base, uint64_t ld, uint32_t dim,
                       int metric_ip, uint32_t count, const uint32_t *res, uint32_t M, uint32_t C, uint32_t Mp,
                       uint32_t *indeg, long long target_base, int skip_existing, int backfill, uint32_t skip_node, int flags,
                       void *stream) {
    if (count == 0) return 0;
    RoarBuildWs *w = (RoarBuildWs *)ws;
    const uint32_t ncand = target_base >= 0 ? C : C - 1;
    if (ncand > 63 || ncand == 0) return (int)hipErrorInvalidValue;
    const uint64_t np = (uint64_t)count * ncand * 2;
    if (!w || np > w->cap) return (int)hipErrorInvalidValue;
    const uint32_t lds = (ncand + 1) * (dim + 1) * sizeof(float);
    const uint32_t grid = count < 8192u ? count : 8192u;
    const void *fn = metric_ip ? (const void *)prune_link_kernel<true> : (const void *)prune_link_kernel<false>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    if (metric_ip)
        hipLaunchKernelGGL(prune_link_kernel<true>, dim3(grid), dim3(64), lds, (hipStream_t)stream, adj, S, cap, base, (u64)ld,
                           dim, count, res, M, C, Mp, target_base, skip_existing, backfill, skip_node, flags, w->a);
    else
        hipLaunchKernelGGL(prune_link_kernel<false>, dim3(grid), dim3(64), lds, (hipStream_t)stream, adj, S, cap, base, (u64)ld,
                           dim, count, res, M, C, Mp, target_base, skip_existing, backfill, skip_node, flags, w->a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    return roar_bk_apply(ws, np, adj, S, cap, indeg, 0, stream);
}

int roar_bk_supply(void *ws, uint32_t *adj, uint32_t S, uint32_t max_deg, uint32_t a, uint32_t count, const uint32_t *res,
                   uint32_t M, uint32_t Kc, uint32_t *indeg, void *stream) {
    if (count == 0 || Kc == 0) return 0;
    RoarBuildWs *w = (RoarBuildWs *)ws;
    const uint64_t np = (uint64_t)count * 2 * Kc;
    if (!w || np > w->cap) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(supply_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, count, res, M, Kc,
                       w->a);
    return roar_bk_apply(ws, np, adj, S, max_deg, indeg, 0, stream);
}

int roar_bk_snapshot_degree(const uint32_t *adj, uint32_t S, uint32_t max_deg, uint64_t n, uint32_t *base_deg,
                            void *stream) {
    hipLaunchKernelGGL(snapshot_degree_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, adj,
                       S, max_deg, (u64)n, base_deg);
    return (int)hipGetLastError();
}

int roar_bk_clamp(uint32_t *adj, uint32_t S, uint32_t max_deg, uint64_t n, void *stream) {
    hipLaunchKernelGGL(clamp_degree_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, adj, S,
                       max_deg, (u64)n);
    return (int)hipGetLastError();
}

int roar_bk_rescue(void *ws, uint32_t *adj, uint32_t S, uint32_t max_deg, uint64_t n, uint32_t ep, uint32_t r, uint32_t *indeg,
                   uint32_t *orphans, void *stream) {
    RoarBuildWs *w = (RoarBuildWs *)ws;
    if (!w || n > w->cap) return (int)hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(orphans, 0, 4, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(rescue_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, adj, S, (u64)n, ep,
                       r, indeg, orphans, w->a);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    return roar_bk_apply(ws, n, adj, S, max_deg, indeg, 1, stream);  // a full row gives up its last slot
}

int roar_bk_graph_hash(const uint32_t *adj, uint32_t S, uint64_t n, uint32_t ep, unsigned long long *out_dev, void *stream) {
    hipError_t e = hipMemsetAsync(out_dev, 0, 8, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(graph_hash_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, adj, S, (u64)n,
                       ep, (u64 *)out_dev);
    return (int)hipGetLastError();
}

int roar_bk_restride(const uint32_t *src, uint32_t S_old, uint32_t *dst, uint32_t S_new, uint64_t n, void *stream) {
    const uint64_t words = n * S_new;
    hipLaunchKernelGGL(restride_rows_kernel, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       S_old, dst, S_new, (u64)n);
    return (int)hipGetLastError();
}
}
