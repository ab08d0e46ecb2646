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
                                                 int rows_per_hop, int probes_per_hop, uint32_t epoch, float *out) {
    const int lane = threadIdx.x, t = lane & 3, g = lane >> 2;
    uint32_t s = 0x9E3779B9u * (blockIdx.x * 64 + lane + 1) + epoch;
    uint32_t *mine = tab + (size_t)blockIdx.x * fwords;
    float acc = 0.f;
    for (int h = 0; h < hops; ++h) {
        uint32_t dep = 0;
        for (int p0 = 0; p0 < probes_per_hop; p0 += 64) {
            if (p0 + lane < probes_per_hop) {
                const uint32_t w = rng(s) % fwords;
                const uint32_t v = __builtin_nontemporal_load(mine + w);
                if (v != epoch && (s & 7u) < 5u) __builtin_nontemporal_store(epoch, mine + w);
                dep |= v & 0x80000000u;  // (never set: the rows wait for the probes like the search's do)
            }
        }
        dep = __shfl(dep, 0) >> 31;
        for (int r0 = 0; r0 < rows_per_hop; r0 += 32) {
            uint32_t s2 = s + 977u * (uint32_t)(r0 + g) + dep;
            const uint32_t ra = rng(s2) % nrows, rb = rng(s2) % nrows;
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
    auto run = [&](size_t fbytes, int probes, uint32_t epoch) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(hop_kernel, dim3(grid), dim3(64), 0, 0, base, nrows, tab, (uint32_t)(fbytes / 4), hops, R, probes, epoch + rep, out);
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
