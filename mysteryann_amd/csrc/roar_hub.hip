// roar_hub.hip -- the hub-annotated copy of the adjacency (RoarSearchParams::adj_hub; long pools on dense graphs), made on
// the device.  The H = min(65 535, n) nodes with the most in-edges (ties to the smaller id) get slots 0 .. H-1 of an exact
// on-chip visited bitmap; every row is rewritten as
//     [degree | hubs << 16] [hub neighbours, in row order] [the others, in row order] [u16 slot per hub neighbour]
// (within a hop the scan order cannot change the pool, so listing hubs first changes no result).  Three steps: in-degree
// histogram (one atomic per edge), ranking by a radix sort of (inverted in-degree << 32 | id) keys (rocPRIM), row rewrite
// (one wave per row).  Round 2 did this on the host -- 5 GB of rows through PCIe and back inside the first long-pool
// search of a 10 M-point index.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <rocprim/rocprim.hpp>

namespace {
typedef unsigned long long u64;

__global__ void hub_indegree_kernel(const uint32_t *__restrict__ adj, uint32_t S, u64 n, uint32_t *__restrict__ indeg) {
    // one wave per row: lane j adds for neighbours j, j + 64, ...
    const u64 v = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (v >= n) return;
    const uint32_t *row = adj + v * S;
    const uint32_t deg = row[0];
    for (uint32_t j = lane; j < deg; j += 64) atomicAdd(indeg + row[1 + j], 1u);
}

__global__ void hub_keys_kernel(const uint32_t *__restrict__ indeg, u64 n, u64 *__restrict__ keys) {
    const u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) keys[v] = ((u64)(0xFFFFFFFFu - indeg[v]) << 32) | v;  // ascending keys = descending in-degree, then ascending id
}

__global__ void hub_slots_kernel(const u64 *__restrict__ sorted, uint32_t H, uint16_t *__restrict__ slot_of) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < H) slot_of[(uint32_t)sorted[r]] = (uint16_t)r;
}

// one wave per row; rows of at most 170 neighbours (three 64-word chunks), stable partition by "has a slot"
__global__ void __launch_bounds__(256) hub_rows_kernel(const uint32_t *__restrict__ adj, uint32_t S, u64 n,
                                                       const uint16_t *__restrict__ slot_of, uint32_t *__restrict__ out, uint32_t Sh) {
    const u64 v = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (v >= n) return;
    const uint32_t *row = adj + v * S;
    uint32_t *o = out + v * Sh;
    const uint32_t deg = row[0];
    uint32_t nb[3], sl[3];
    bool is[3], hub[3];
    uint32_t nh = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint32_t j = 64u * c + lane;
        is[c] = j < deg;
        nb[c] = is[c] ? row[1 + j] : 0u;
        sl[c] = is[c] ? (uint32_t)slot_of[nb[c]] : 0xFFFFu;
        hub[c] = is[c] && sl[c] != 0xFFFFu;
        nh += (uint32_t)__popcll(__ballot(hub[c]));
    }
    uint32_t hpos = 0, opos = nh;
    uint16_t *slots = reinterpret_cast<uint16_t *>(o + 1 + deg);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const u64 hm = __ballot(hub[c]), om = __ballot(is[c] && !hub[c]);
        const u64 below = (1ull << lane) - 1ull;
        if (hub[c]) {
            const uint32_t at = hpos + (uint32_t)__popcll(hm & below);
            o[1 + at] = nb[c];
            slots[at] = (uint16_t)sl[c];
        } else if (is[c]) {
            o[1 + opos + (uint32_t)__popcll(om & below)] = nb[c];
        }
        hpos += (uint32_t)__popcll(hm);
        opos += (uint32_t)__popcll(om);
    }
    if (lane == 0) o[0] = deg | (nh << 16);
}
}  // namespace

// adj: n rows of S words (word 0 = degree), max degree <= 170.  out: n rows of Sh words, zeroed by the caller.  All device
// pointers; keys_a / keys_b: u64[n] each, indeg: u32[n], slot_of: u16[n]; sort_tmp of *sort_tmp_bytes bytes (call once with
// sort_tmp == nullptr to size it).  Returns a hipError_t value.
extern "C" int roar_hub_build_rows(const uint32_t *adj, uint32_t S, uint64_t n, uint32_t *out, uint32_t Sh, uint32_t *indeg,
                                   uint16_t *slot_of, unsigned long long *keys_a, unsigned long long *keys_b, void *sort_tmp,
                                   size_t *sort_tmp_bytes, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!sort_tmp) {
        size_t b = 0;
        hipError_t e = rocprim::radix_sort_keys(nullptr, b, (u64 *)nullptr, (u64 *)nullptr, (size_t)n, 0, 64, st);
        *sort_tmp_bytes = b ? b : 16;
        return (int)e;
    }
    const uint32_t H = (uint32_t)(n < 65535 ? n : 65535);  // slot 0xFFFF means "no slot"
    hipError_t e = hipMemsetAsync(indeg, 0, (size_t)n * 4, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(slot_of, 0xFF, (size_t)n * 2, st);
    if (e != hipSuccess) return (int)e;
    const uint32_t wg = (uint32_t)((n * 64 + 255) / 256);
    hipLaunchKernelGGL(hub_indegree_kernel, dim3(wg), dim3(256), 0, st, adj, S, (u64)n, indeg);
    hipLaunchKernelGGL(hub_keys_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, indeg, (u64)n, (u64 *)keys_a);
    size_t tb = *sort_tmp_bytes;
    e = rocprim::radix_sort_keys(sort_tmp, tb, (u64 *)keys_a, (u64 *)keys_b, (size_t)n, 0, 64, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(hub_slots_kernel, dim3((H + 255) / 256), dim3(256), 0, st, (const u64 *)keys_b, H, slot_of);
    hipLaunchKernelGGL(hub_rows_kernel, dim3(wg), dim3(256), 0, st, adj, S, (u64)n, slot_of, out, Sh);
    return (int)hipGetLastError();
}
