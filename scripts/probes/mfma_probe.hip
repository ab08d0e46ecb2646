// What the matrix cores give a loop shaped like the exact-kNN pass kernel's: 8 waves per CU (2 per SIMD), A fragments in registers,
// 13 v_mfma_f32_32x32x16_bf16 per 32 x 32 block.  Variants: B from registers or from LDS (ds_read_b128, three buffers), one or two
// accumulator chains, with or without 32 verdict-like VALU instructions per block between the MFMAs (singly, or two per v_pk_fma_f32).
//   hipcc -O3 --offload-arch=gfx950 -o mfma_probe mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int KS = 13;

template <int OFF>
__device__ __forceinline__ void lds_read_b128(uint32_t a, u32x4 &f) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(f) : "v"(a), "n"(OFF) : "memory");
}
template <int LEFT>
__device__ __forceinline__ void lds_wait1(u32x4 &f) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(LEFT) : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>());
        static_for<I + 1, N>(f);
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool LDS, int CHAINS, bool VALU, bool PK = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) probe(const uint4 *src, float *out, int iters) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    uint4 *buf = reinterpret_cast<uint4 *>(smem);
    constexpr int KQ = 2 * KS + 1;
    for (int i = tid; i < 64 * KQ; i += 512) buf[i] = src[i];
    __syncthreads();
    bf16x8 af[KS];
    for (int ks = 0; ks < KS; ++ks) af[ks] = __builtin_bit_cast(bf16x8, src[(tid * KS + ks) % (64 * KQ)]);
    float thr[16], eq[16];
    for (int g = 0; g < 16; ++g) thr[g] = 1e30f + g + lane, eq[g] = 0.001f * (g + 1);
    f32x2 eq2[8];
    for (int g = 0; g < 8; ++g) eq2[g] = f32x2{eq[2 * g], eq[2 * g + 1]};
    f32x16 acc[2];
    for (int g = 0; g < 16; ++g) acc[0][g] = 0.f, acc[1][g] = 0.f;
    const uint32_t a0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)(buf + r * KQ + h);
    const uint32_t a1 = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)(buf + (32 + r) * KQ + h);
    unsigned long long anyacc = 0;
    float xn = 1.0f + lane * 1e-6f;
    for (int it = 0; it < iters; ++it) {
        for (int half = 0; half < 2; ++half) {
            f32x16 &cur = acc[CHAINS == 2 ? half : 0];
            const f32x16 &old = acc[CHAINS == 2 ? 1 - half : 0];
            const uint32_t a = half ? a1 : a0;
            u32x4 f[3];
            if constexpr (LDS) {
                lds_read_b128<0>(a, f[0]);
                lds_read_b128<32>(a, f[1]);
            }
            static_for<0, KS>([&](auto c) {
                constexpr int ks = decltype(c)::value;
                if constexpr (LDS) {
                    constexpr int nx = ks + 2;
                    if constexpr (nx < KS) lds_read_b128<(nx < KS ? nx : 0) * 32>(a, f[nx % 3]);
                    constexpr int later = (KS - 1 - ks) < 2 ? (KS - 1 - ks) : 2;
                    lds_wait1<later>(f[ks % 3]);
                    cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], __builtin_bit_cast(bf16x8, f[ks % 3]), cur, 0, 0, 0);
                } else {
                    cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], af[(ks + 1) % KS], cur, 0, 0, 0);
                }
                if constexpr (VALU && CHAINS == 2 && !PK) {
                    static_for<0, 16>([&](auto cg) {
                        constexpr int g = decltype(cg)::value;
                        if constexpr ((g * KS) / 16 == ks) anyacc |= __ballot(-__builtin_fmaf(eq[g], xn, old[g]) <= -thr[g]);
                    });
                }
                if constexpr (VALU && CHAINS == 2 && PK) {  // two verdicts per v_pk_fma_f32
                    static_for<0, 8>([&](auto cg) {
                        constexpr int g = decltype(cg)::value;
                        if constexpr ((g * KS) / 8 == ks) {
                            const f32x2 v = __builtin_elementwise_fma(eq2[g], f32x2{xn, xn}, f32x2{old[2 * g], old[2 * g + 1]});
                            anyacc |= __ballot(-v.x <= -thr[2 * g]);
                            anyacc |= __ballot(-v.y <= -thr[2 * g + 1]);
                        }
                    });
                }
            });
            if constexpr (VALU && CHAINS == 1) {
#pragma unroll
                for (int g = 0; g < 16; ++g) anyacc |= __ballot(-__builtin_fmaf(eq[g], xn, cur[g]) <= -thr[g]);
            }
        }
    }
    float s = (float)(anyacc & 1);
    for (int g = 0; g < 16; ++g) s += acc[0][g] + acc[1][g];
    out[blockIdx.x * 512 + tid] = s;
}

template <bool LDS, int CHAINS, bool VALU, bool PK = false>
void run(const char *name, const uint4 *src, float *out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const size_t lds = 64 * (2 * KS + 1) * 16;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<LDS, CHAINS, VALU, PK>), dim3(256), dim3(512), lds, 0, src, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 256.0 * 8 * iters * 2 * KS * 32768.0;
        if (rep) printf("%-58s %8.2f ms  %7.1f TFLOP/s  (%.2f of 2 500)\n", name, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500);
    }
}

int main() {
    uint4 *src;
    float *out;
    hipMalloc(&src, 64 * 27 * 16);
    hipMemset(src, 0x11, 64 * 27 * 16);
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 40000;
    run<false, 2, false>("B in registers, two chains", src, out, iters);
    run<false, 1, false>("B in registers, one chain", src, out, iters);
    run<true, 2, false>("B from LDS, two chains (alternating blocks)", src, out, iters);
    run<true, 1, false>("B from LDS, one chain", src, out, iters);
    run<true, 2, true>("B from LDS, two chains, verdicts under the MFMAs", src, out, iters);
    run<true, 1, true>("B from LDS, one chain, verdicts after the MFMAs", src, out, iters);
    run<true, 2, true, true>("B from LDS, two chains, verdicts in pairs (v_pk_fma_f32)", src, out, iters);
    return 0;
}
