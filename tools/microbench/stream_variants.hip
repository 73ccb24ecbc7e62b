// Development microbenchmark: which streaming structure reaches the HBM ceiling for the
// gradient pass (read E*4 bytes, exp, write E*4 bytes) at the c3 footprint (8 GB + 8 GB)?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 stream_variants.hip -o stream_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../warp-transducer_amd/csrc/rnnt_kernels.h"
using namespace rnnt;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// A: flat copy, grid-stride, U packets in flight per thread
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_flat(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        uint4 r[U];
#pragma unroll
        for (int k = 0; k < U; ++k) r[k] = load_packet<NT>(in + i + k * stride);
#pragma unroll
        for (int k = 0; k < U; ++k) store_packet<NT>(out + i + k * stride, r[k]);
    }
    for (; i < n; i += stride) store_packet<NT>(out + i, load_packet<NT>(in + i));
}

// B: flat exp, block owns a contiguous chunk (CH packets per thread, consecutive 4 KB per wave-iteration)
template <int U, bool NT, bool EXP>
__global__ __launch_bounds__(256) void exp_chunk(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n, float c) {
    // each block handles U*256 consecutive packets per iteration, blocks grid-stride over chunks
    size_t chunk = (size_t)U * 256;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
        uint4 r[U];
#pragma unroll
        for (int k = 0; k < U; ++k) { size_t i = base + k * 256 + threadIdx.x; if (i < n) r[k] = load_packet<NT>(in + i); }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            size_t i = base + k * 256 + threadIdx.x;
            if (i < n) {
                if (EXP) {
                    float v[4]; unpack<F32>(r[k], v);
                    for (int j = 0; j < 4; ++j) v[j] = __expf(v[j] + c);
                    r[k] = pack<F32>(v);
                }
                store_packet<NT>(out + i, r[k]);
            }
        }
    }
}

int main(int argc, char** argv) {
    const int N = 128, T = 150, Uu = 21, A = argc > 1 ? atoi(argv[1]) : 5000;
    const size_t R = (size_t)N * T * Uu, E = R * A;
    float *acts, *grads; int *xlen, *ylen; Cell<float>* cells;
    CK(hipMalloc(&acts, E * 4)); CK(hipMalloc(&grads, E * 4));
    CK(hipMalloc(&xlen, N * 4)); CK(hipMalloc(&ylen, N * 4));
    const int D = T + Uu - 1;
    CK(hipMalloc(&cells, (size_t)N * D * Uu * 16));
    CK(hipMemset(acts, 0, E * 4)); CK(hipMemset(cells, 0, (size_t)N * D * Uu * 16));
    std::vector<int> hx(N, T), hy(N, Uu - 1);
    CK(hipMemcpy(xlen, hx.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ylen, hy.data(), N * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t npk = E / 4;
    auto timeit = [&](const char* name, auto&& launch, double bytes) {
        for (int i = 0; i < 2; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 8;
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-34s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
    };
    const double rw = 2.0 * E * 4, ro = 1.0 * E * 4;
    auto in = (const u32x4*)acts; auto out = (u32x4*)grads;
    for (int g : {2048, 4096, 8192}) {
        char nm[64];
        snprintf(nm, 64, "copy_flat U=4 grid=%d", g);
        timeit(nm, [&] { hipLaunchKernelGGL((copy_flat<4, false>), dim3(g), dim3(256), 0, 0, in, out, npk); }, rw);
        snprintf(nm, 64, "copy_flat U=8 NT grid=%d", g);
        timeit(nm, [&] { hipLaunchKernelGGL((copy_flat<8, true>), dim3(g), dim3(256), 0, 0, in, out, npk); }, rw);
    }
    for (int g : {2048, 8192, 65536}) {
        char nm[64];
        snprintf(nm, 64, "exp_chunk U=4 grid=%d", g);
        timeit(nm, [&] { hipLaunchKernelGGL((exp_chunk<4, false, true>), dim3(g), dim3(256), 0, 0, in, out, npk, 0.5f); }, rw);
        snprintf(nm, 64, "exp_chunk U=4 NT grid=%d", g);
        timeit(nm, [&] { hipLaunchKernelGGL((exp_chunk<4, true, true>), dim3(g), dim3(256), 0, 0, in, out, npk, 0.5f); }, rw);
        snprintf(nm, 64, "copy_chunk U=4 NT grid=%d", g);
        timeit(nm, [&] { hipLaunchKernelGGL((exp_chunk<4, true, false>), dim3(g), dim3(256), 0, 0, in, out, npk, 0.5f); }, rw);
        snprintf(nm, 64, "exp_chunk U=8 NT grid=%d", g);
        timeit(nm, [&] { hipLaunchKernelGGL((exp_chunk<8, true, true>), dim3(g), dim3(256), 0, 0, in, out, npk, 0.5f); }, rw);
    }
    {
        size_t chunks = (npk + 1023) / 1024;
        timeit("exp_chunk U=4 NT grid=all", [&] { hipLaunchKernelGGL((exp_chunk<4, true, true>), dim3(chunks), dim3(256), 0, 0, in, out, npk, 0.5f); }, rw);
        timeit("exp_chunk U=4 grid=all", [&] { hipLaunchKernelGGL((exp_chunk<4, false, true>), dim3(chunks), dim3(256), 0, 0, in, out, npk, 0.5f); }, rw);
    }
    // (the wavefront-per-row gradient kernel this file once raced against the flat form is gone from the tree:
    //  5.1 vs 6.4 TB/s, EXPERIMENTS.md section 3)
    timeit("hipMemcpyDtoD", [&] { CK(hipMemcpyAsync(grads, acts, E * 4, hipMemcpyDeviceToDevice, 0)); }, rw);
    timeit("hipMemset (write only)", [&] { CK(hipMemsetAsync(grads, 0, E * 4, 0)); }, ro);
    return 0;
}
