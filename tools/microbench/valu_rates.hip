// valu_rates.hip -- issue rates on gfx950 that the additive-joint kernels depend on:
//   v_exp_f32, v_fma_f32, v_mfma_f32_32x32x2_f32 alone, and the Z kernel's mix (2 exp + 2 fma per MFMA),
// with the exps/fmas independent of the MFMA chain.  Reports shader cycles per instruction per SIMD
// (s_memtime deltas of wave 0 of each block; grid = 256 CUs x WPS blocks of 256 threads -> WPS waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kIters = 512;

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, unsigned long long* cyc, float seed) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-6f + i;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const float wa = seed * 0.5f, wb = seed * 0.25f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
        if constexpr (MODE == 0) {            // 8 independent v_exp_f32
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_exp2f(a[i]);
        } else if constexpr (MODE == 1) {     // 8 independent v_fma_f32
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
        } else if constexpr (MODE == 2) {     // 4 dependent MFMAs
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, wb, acc, 0, 0, 0);
        } else if constexpr (MODE == 3) {     // Z mix: per MFMA 2 exp + 2 fma (independent of the MFMA chain)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[2 * i] = __builtin_amdgcn_exp2f(__builtin_fmaf(a[2 * i], 1.0001f, -0.5f));
                a[2 * i + 1] = __builtin_amdgcn_exp2f(__builtin_fmaf(a[2 * i + 1], 1.0001f, -0.5f));
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, wb, acc, 0, 0, 0);
            }
        } else {                              // 4: exp results FEED the MFMA (operand dependency, as in DF/DG)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(a[2 * i], 1.0001f, -0.5f));
                a[2 * i] = e0 * 0.999f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, e0, acc, 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, int wps, double per_iter, float* out, unsigned long long* cyc) {
    const int blocks = 256 * wps;
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, 0.001f);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, 0.001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    // s_memtime ticks at 100 MHz on this part? report both raw ticks and wall time per instruction-group
    printf("%-34s waves/SIMD=%d: %8.1f memtime ticks/iter, kernel %.3f ms -> %.2f ns per iter per wave-slot (%.1f instr/iter)\n",
           name, wps, avg / kIters, ms, ms * 1e6 / kIters, per_iter);
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    for (int wps : {1, 2, 4}) {
        run<0>("8 x v_exp_f32", wps, 8, out, cyc);
        run<1>("8 x v_fma_f32", wps, 8, out, cyc);
        run<2>("4 x mfma_f32_32x32x2 (dependent)", wps, 4, out, cyc);
        run<3>("4 x (2 exp + 2 fma + mfma) indep.", wps, 20, out, cyc);
        run<4>("4 x (fma+exp -> mfma operand)", wps, 16, out, cyc);
    }
    return 0;
}
