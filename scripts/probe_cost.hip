// probe_cost.hip -- what does a one-word visited probe cost beside the row gathers of the beam search, as a function of
// the footprint the probes of one workgroup fall into?  (Round 3: is a compact per-slot spill table -- L2 / Infinity
// Cache resident -- cheaper to probe than the n-byte tag list of the EXACT mode?)
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_cost scripts/probe_cost.hip && /tmp/probe_cost
//
// Model of a hop: P probes (one u32 load per lane, ~60 % followed by a store) into the workgroup's own region of F bytes,
// then R rows of 800 B gathered at random from an 8 GB array (4 lanes per row, 13 dwordx4 loads per lane, as the search
// kernel's scoring pass).  Grid = 6 one-wave workgroups per CU, as the search kernel.  Reported: time per hop without
// probes, with probes, and the marginal cost of a probe in units of one row line (a row = 7 lines of 128 B).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t rng(uint32_t &s) {
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    return s;
}

__global__ void __launch_bounds__(64) hop_kernel(const float4 *base, uint32_t nrows, uint32_t *tab, uint32_t fwords, int hops,
                                                 int rows_per_hop, int probes_per_hop, uint32_t epoch, float *out, int mode) {
    const int lane = threadIdx.x, t = lane & 3, g = lane >> 2;
    uint32_t s = 0x9E3779B9u * (blockIdx.x * 64 + lane + 1) + epoch;
    uint32_t *mine = tab + (size_t)blockIdx.x * fwords;
    float acc = 0.f;
    uint32_t prev_w = 0;
    for (int h = 0; h < hops; ++h) {
        uint32_t dep = 0;
        for (int p0 = 0; p0 < probes_per_hop; p0 += 64) {
            if (p0 + lane < probes_per_hop) {
                const uint32_t w = __umulhi(rng(s), fwords);
                // mode 0: load, 5 of 8 followed by a store (the EXACT mode's mix); 1: loads only; 2: load + store always;
                // 3: stores only; 4: byte loads + byte stores (u8 tags) with the mode-0 mix
                uint32_t v = 0;
                if (mode >= 5) {
                    // 5 / 6 / 7 / 8: read a whole aligned 16 / 32 / 64 / 128-byte piece, change one word, write the whole piece
                    // back (5 of 8 times): is a full-sector store cheaper than a one-word store?   9: 32-byte piece, loads only
                    const int nq4 = mode == 5 ? 1 : mode == 6 || mode == 9 ? 2 : mode == 7 ? 4 : 8;
                    uint4 *pp = reinterpret_cast<uint4 *>(mine) + (size_t)(w / (4u * nq4)) * nq4;
                    uint4 buf[8];
                    for (int i = 0; i < nq4; ++i) buf[i] = pp[i];
                    v = buf[0].x ^ buf[nq4 - 1].w;
                    if (mode != 9 && (s & 7u) < 5u) {
                        buf[(s >> 8) % nq4].y = epoch;
                        for (int i = 0; i < nq4; ++i) pp[i] = buf[i];
                    }
                } else if (mode >= 10) {
                    // 10: plain load + plain store always; 11: plain stores only; 12: nontemporal load + plain store 5/8;
                    // 13: load, then the store of the PREVIOUS hop's word (the store is off the hop's dependent chain)
                    if (mode != 11) v = mode == 12 ? __builtin_nontemporal_load(mine + w) : mine[w];
                    if (mode == 13) {
                        if (h > 0) mine[prev_w] = epoch;
                        prev_w = w;
                    } else if (mode != 12 || (s & 7u) < 5u) {
                        mine[w] = epoch;
                    }
                } else if (mode == 4) {
                    unsigned char *b = reinterpret_cast<unsigned char *>(mine) + (w * 4u + (s >> 30));
                    v = __builtin_nontemporal_load(b);
                    if ((s & 7u) < 5u) __builtin_nontemporal_store((unsigned char)epoch, b);
                } else {
                    if (mode != 3) v = __builtin_nontemporal_load(mine + w);
                    const bool st = mode == 0 ? (v != epoch && (s & 7u) < 5u) : mode == 1 ? false : true;
                    if (st) __builtin_nontemporal_store(epoch, mine + w);
                }
                dep |= v & 0x80000000u;  // (never set: the rows wait for the probes like the search's do)
            }
        }
        dep = __shfl(dep, 0) >> 31;
        for (int r0 = 0; r0 < rows_per_hop; r0 += 32) {
            // (the four lanes of a group read the same two rows: the seed depends on the group, not on the lane)
            uint32_t s2 = (0x9E3779B9u * (blockIdx.x * 1024u + (uint32_t)h) + 0x85EBCA6Bu * (uint32_t)(r0 + g + 1) + dep) | 1u;
            rng(s2);
            const uint32_t ra = __umulhi(rng(s2), nrows), rb = __umulhi(rng(s2), nrows);
            const float4 *pa = base + (size_t)ra * 50 + t, *pb = base + (size_t)rb * 50 + t;
            float4 xa[13], xb[13];
#pragma unroll
            for (int i = 0; i < 12; ++i) xa[i] = pa[4 * i], xb[i] = pb[4 * i];
            xa[12] = pa[48 - t + (t & 1)], xb[12] = pb[48 - t + (t & 1)];
#pragma unroll
            for (int i = 0; i < 13; ++i) acc += xa[i].x * xb[i].y + xa[i].z * xb[i].w;
        }
        rng(s);
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    const uint32_t nrows = 10000000;
    float4 *base;
    CHECK(hipMalloc(&base, (size_t)nrows * 800));
    CHECK(hipMemset(base, 0, (size_t)nrows * 800));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * 6;
    float *out;
    CHECK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int hops = 300, R = 64;
    const size_t fmax = 16u << 20;
    uint32_t *tab;
    CHECK(hipMalloc(&tab, fmax * grid));
    CHECK(hipMemset(tab, 0, fmax * grid));
    int mode = 0;
    auto run = [&](size_t fbytes, int probes, uint32_t epoch) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(hop_kernel, dim3(grid), dim3(64), 0, 0, base, nrows, tab, (uint32_t)(fbytes / 4), hops, R, probes, epoch + rep, out, mode);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        return best;
    };
    const float t0 = run(4096, 0, 1);
    const double lines = (double)grid * hops * R * 7;
    printf("grid %d, %d hops x %d rows of 800 B: %.3f ms without probes = %.2f TB/s, %.1f G lines/s\n", grid, hops, R, t0,
           (double)grid * hops * R * 800 / t0 / 1e9, lines / t0 / 1e6);
    uint32_t epoch = 10;
    const char *names[] = {"load + 5/8 store", "loads only", "load + store", "stores only", "u8 load + 5/8 u8 store",
                           "16-B piece read + 5/8 written back whole", "32-B piece read + 5/8 written back whole",
                           "64-B piece read + 5/8 written back whole", "128-B piece read + 5/8 written back whole", "32-B piece, loads only",
                           "plain load + plain store", "plain stores only", "nt load + 5/8 plain store", "plain load + store of the previous hop's word"};
    for (mode = 1; mode <= 13; ++mode) {
        if (mode >= 5 && mode <= 9) continue;
        for (size_t f : {(size_t)4096, (size_t)(1u << 20), (size_t)(10u << 20)}) {
            const float t1 = run(f, 49, epoch);
            epoch += 10;
            printf("[%s] probes/hop 49 footprint/workgroup %8zu B: %.3f ms (+%.1f %%), one probe costs %.2f row lines\n", names[mode], f, t1,
                   (t1 / t0 - 1) * 100, (t1 - t0) / ((double)grid * hops * 49) / (t0 / lines));
        }
    }
    mode = 0;
    for (int probes : {49, 98}) {
        for (size_t f : {(size_t)4096, (size_t)16384, (size_t)65536, (size_t)131072, (size_t)262144, (size_t)524288, (size_t)(1u << 20), (size_t)(4u << 20),
                         (size_t)(10u << 20), (size_t)(16u << 20)}) {
            const float t1 = run(f, probes, epoch);
            epoch += 10;
            const double per_probe_ns = (t1 - t0) * 1e6 / ((double)grid * hops * probes) * grid;  // wall ns per probe per workgroup
            const double line_cost = t0 / lines;
            printf("probes/hop %3d footprint/workgroup %8zu B (total %7.1f MB): %.3f ms (+%.1f %%), one probe costs %.2f row lines (%.1f ns of a workgroup's time)\n",
                   probes, f, (double)f * grid / 1e6, t1, (t1 / t0 - 1) * 100, (t1 - t0) / ((double)grid * hops * probes) / line_cost, per_probe_ns);
        }
    }
    return 0;
}
