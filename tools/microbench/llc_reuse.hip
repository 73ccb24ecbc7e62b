// Development microbenchmark: does a second pass over a buffer that was just streamed come out of the 256 MB Infinity Cache
// faster than HBM delivers it?  (The question behind a sample-blocked schedule: statistics of a group of samples, their
// lattice, then the gradient pass re-reading the group while it is still on the die.)
//   pass 1: read X (S MB), temporal or non-temporal loads;  pass 2: Y = f(X), 16-byte packets, load / store policy selectable.
// Reported: pass 2 alone after a 2 GB flush read ("cold") and right behind pass 1 ("warm"), as GB/s of read + written bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __global__ __launch_bounds__(256) void read_pass(const u32x4* __restrict__ x, size_t n, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256) {
        const u32x4 v = NT ? __builtin_nontemporal_load(x + i) : x[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
template <bool NTL, bool NTS> __global__ __launch_bounds__(256) void copy_pass(const u32x4* __restrict__ x, u32x4* __restrict__ y, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256) {
        u32x4 v = NTL ? __builtin_nontemporal_load(x + i) : x[i];
        v.x += 1u;
        if (NTS) __builtin_nontemporal_store(v, y + i); else y[i] = v;
    }
}
int main(int argc, char** argv) {
    const size_t flush_n = (2ull << 30) / 16;
    u32x4 *x, *y, *fl; unsigned* sink;
    CK(hipMalloc(&x, 1ull << 30)); CK(hipMalloc(&y, 1ull << 30)); CK(hipMalloc(&fl, flush_n * 16)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(x, 1, 1ull << 30)); CK(hipMemset(y, 0, 1ull << 30)); CK(hipMemset(fl, 2, flush_n * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 16;
    printf("# MB  policy(pass1 load / pass2 load / pass2 store)  cold us (GB/s)   warm us (GB/s)   pass 1 warm-after-itself us (GB/s)\n");
    for (int mb : {16, 32, 63, 96, 126, 192, 256, 512}) {
        const size_t n = static_cast<size_t>(mb) * 1000 * 1000 / 16;
        for (int pol = 0; pol < 4; ++pol) {
            auto pass1 = [&] { if (pol & 1) hipLaunchKernelGGL(read_pass<true>, dim3(grid), dim3(256), 0, 0, x, n, sink); else hipLaunchKernelGGL(read_pass<false>, dim3(grid), dim3(256), 0, 0, x, n, sink); };
            auto pass2 = [&] { if (pol & 2) hipLaunchKernelGGL((copy_pass<true, true>), dim3(grid), dim3(256), 0, 0, x, y, n); else hipLaunchKernelGGL((copy_pass<false, true>), dim3(grid), dim3(256), 0, 0, x, y, n); };
            auto flush = [&] { hipLaunchKernelGGL(read_pass<false>, dim3(grid), dim3(256), 0, 0, fl, flush_n, sink); };
            float cold = 0, warm = 0, again = 0; const int reps = 10;
            for (int r = 0; r < reps + 2; ++r) {
                float ms;
                flush(); CK(hipEventRecord(e0)); pass2(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) cold += ms;
                flush(); pass1(); CK(hipEventRecord(e0)); pass2(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) warm += ms;
                flush(); pass1(); CK(hipEventRecord(e0)); pass1(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) again += ms;
            }
            cold /= reps; warm /= reps; again /= reps;
            const double bytes2 = 2.0 * n * 16, bytes1 = 1.0 * n * 16;
            printf("%4d  %s / %s / nt   %7.1f (%5.0f)   %7.1f (%5.0f)   %7.1f (%5.0f)\n", mb, (pol & 1) ? "nt" : "--", (pol & 2) ? "nt" : "--",
                   cold * 1e3, bytes2 / cold * 1e-6, warm * 1e3, bytes2 / warm * 1e-6, again * 1e3, bytes1 / again * 1e-6);
        }
    }
    return 0;
}
