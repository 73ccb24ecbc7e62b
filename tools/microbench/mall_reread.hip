// mall_reread.hip -- does a second read of a recently read buffer run faster than HBM speed?
// For each size S: kernel A reads S bytes (flat, 16-byte non-temporal or plain loads), kernel B reads
// the same S bytes again; B's bandwidth vs S shows what the L2 / Infinity Cache retain between kernels.
// Also: A reads, B reads+writes a second buffer (the stats -> gradient pattern).
// Build: hipcc --offload-arch=gfx950 -O3 mall_reread.hip -o mall_reread ; run: ./mall_reread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ in, size_t n, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
        const u32x4 v = NT ? __builtin_nontemporal_load(in + i) : in[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
        u32x4 v = NT ? __builtin_nontemporal_load(in + i) : in[i];
        v.x += 1;
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}

int main() {
    const size_t maxb = 4ull << 30;
    u32x4 *a, *b; unsigned* sink;
    hipMalloc(&a, maxb); hipMalloc(&b, maxb); hipMalloc(&sink, 4);
    hipMemset(a, 1, maxb); hipMemset(b, 0, maxb);
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    const size_t sizes[] = {32ull << 20, 64ull << 20, 128ull << 20, 192ull << 20, 256ull << 20, 384ull << 20, 1ull << 30, 4ull << 30};
    for (int nt = 0; nt < 2; ++nt)
        for (size_t S : sizes) {
            const size_t n = S / 16;
            const unsigned grid = 256 * 8;
            float ta = 0, tb = 0, tc = 0;
            const int reps = 5;
            for (int r = 0; r < reps + 1; ++r) {
                // flush: touch a different 1 GB region so the caches do not hold `a[0..S)` from the last rep
                hipLaunchKernelGGL(read_kernel<false>, dim3(grid), dim3(256), 0, 0, a + (3ull << 30) / 16, (1ull << 30) / 16, sink);
                hipEventRecord(e0);
                if (nt) hipLaunchKernelGGL(read_kernel<true>, dim3(grid), dim3(256), 0, 0, a, n, sink);
                else hipLaunchKernelGGL(read_kernel<false>, dim3(grid), dim3(256), 0, 0, a, n, sink);
                hipEventRecord(e1);
                if (nt) hipLaunchKernelGGL(read_kernel<true>, dim3(grid), dim3(256), 0, 0, a, n, sink);
                else hipLaunchKernelGGL(read_kernel<false>, dim3(grid), dim3(256), 0, 0, a, n, sink);
                hipEventRecord(e2);
                hipEventSynchronize(e2);
                float x, y; hipEventElapsedTime(&x, e0, e1); hipEventElapsedTime(&y, e1, e2);
                // read then copy (second pass reads a again and writes b)
                hipLaunchKernelGGL(read_kernel<false>, dim3(grid), dim3(256), 0, 0, a + (3ull << 30) / 16, (1ull << 30) / 16, sink);
                if (nt) hipLaunchKernelGGL(read_kernel<true>, dim3(grid), dim3(256), 0, 0, a, n, sink);
                else hipLaunchKernelGGL(read_kernel<false>, dim3(grid), dim3(256), 0, 0, a, n, sink);
                hipEventRecord(e1);
                if (nt) hipLaunchKernelGGL(copy_kernel<true>, dim3(grid), dim3(256), 0, 0, a, b, n);
                else hipLaunchKernelGGL(copy_kernel<false>, dim3(grid), dim3(256), 0, 0, a, b, n);
                hipEventRecord(e2);
                hipEventSynchronize(e2);
                float z; hipEventElapsedTime(&z, e1, e2);
                if (r) { ta += x; tb += y; tc += z; }
            }
            printf("%s S=%5zu MB: first read %6.2f TB/s | re-read %6.2f TB/s | re-read + write (copy) %6.2f TB/s of 2S\n",
                   nt ? "NT   " : "plain", S >> 20, S / (ta / reps) * 1e-9, S / (tb / reps) * 1e-9, 2.0 * S / (tc / reps) * 1e-9);
        }
    return 0;
}
