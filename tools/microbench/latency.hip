// Development microbenchmark: dependent-chain latency of single instructions for ONE wavefront.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)
constexpr int ITER = 4096;
template <int OP> __global__ void chain(float* out, long long* cyc, float seed) {
    float v = seed + threadIdx.x * 1e-3f, w = seed * 0.5f;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < ITER; ++i) {
        if (OP == 0) v = v + w;
        if (OP == 1) v = __builtin_amdgcn_exp2f(v) - 1.0f;                 // exp + add
        if (OP == 2) v = __builtin_amdgcn_logf(v + 2.0f);                  // add + log
        if (OP == 3) v = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false)) + w;  // dpp + add
        if (OP == 4) v = fmaxf(v, w) - fminf(v, w);                        // max|min + sub
        if (OP == 5) v = v + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));   // readlane + add
        if (OP == 6) { float hi = fmaxf(v, w), lo = fminf(v, w); v = hi + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(lo - hi)); }  // log2_add
        if (OP == 7) v = fmaf(v, w, 1.0f);
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int OP> int run(const char* name, int ops) {
    float* out; long long* cyc; CK(hipMalloc(&out, 256)); CK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((chain<OP>), dim3(1), dim3(64), 0, 0, out, cyc, 0.7f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((chain<OP>), dim3(1), dim3(64), 0, 0, out, cyc, 0.7f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-22s %7.1f ns/iter  %6.1f counter-ticks/iter  (%d dependent ops) -> %.1f ns/op\n", name, ms * 1e6 / ITER, (double)c / ITER, ops, ms * 1e6 / ITER / ops);
    return 0;
}
int main() {
    run<0>("v_add", 1); run<7>("v_fma", 1); run<1>("exp2+add", 2); run<2>("add+log2", 2); run<3>("dpp_mov+add", 2);
    run<4>("max|min+sub", 2); run<5>("readlane+add", 2); run<6>("log2_add", 6);
    return 0;
}
