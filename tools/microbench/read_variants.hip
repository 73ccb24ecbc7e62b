// Development microbenchmark: read-only streaming ceiling (the row-stats pass reads E*s bytes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../warp-transducer_amd/csrc/rnnt_kernels.h"
using namespace rnnt;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int PPT, bool NT, bool EXP>
__global__ __launch_bounds__(256) void read_chunk(const u32x4* __restrict__ in, float* __restrict__ out, size_t n) {
    const size_t base = (size_t)blockIdx.x * PPT * 256;
    float acc = 0;
    uint4 r[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) { size_t i = base + k * 256 + threadIdx.x; r[k] = i < n ? load_packet<NT>(in + i) : make_uint4(0,0,0,0); }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        float v[4]; unpack<F32>(r[k], v);
        for (int j = 0; j < 4; ++j) acc += EXP ? __expf(v[j] - 1.0f) : v[j];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc;
}

int main(int argc, char** argv) {
    const int N = 128, T = 150, Uu = 21, A = argc > 1 ? atoi(argv[1]) : 5000;
    const size_t R = (size_t)N * T * Uu, E = R * A;
    const int Up = 64; const size_t Dp = lat_rows(T, Uu);
    float *acts, *out; int *xlen, *ylen, *labels; LogPair<float>* lp2; float* logz;
    CK(hipMalloc(&acts, E * 4)); CK(hipMalloc(&out, 64 << 20));
    CK(hipMalloc(&xlen, N * 4)); CK(hipMalloc(&ylen, N * 4)); CK(hipMalloc(&labels, N * Uu * 4));
    CK(hipMalloc(&lp2, N * Dp * Up * 8)); CK(hipMalloc(&logz, N * Dp * Up * 4));
    CK(hipMemset(acts, 0, E * 4)); CK(hipMemset(labels, 0, N * Uu * 4));
    std::vector<int> hx(N, T), hy(N, Uu - 1);
    CK(hipMemcpy(xlen, hx.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ylen, hy.data(), N * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t npk = E / 4;
    auto timeit = [&](const char* name, auto&& launch) {
        for (int i = 0; i < 2; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 8;
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-34s %8.3f ms  %7.1f GB/s\n", name, ms, E * 4.0 / ms / 1e6);
    };
    auto in = (const u32x4*)acts;
#define RUN(P, NTF, EX, name) timeit(name, [&] { hipLaunchKernelGGL((read_chunk<P, NTF, EX>), dim3((npk + P * 256 - 1) / (P * 256)), dim3(256), 0, 0, in, out, npk); })
    RUN(1, true, true, "read_chunk ppt=1 NT exp");
    RUN(2, true, true, "read_chunk ppt=2 NT exp");
    RUN(4, true, true, "read_chunk ppt=4 NT exp");
    RUN(8, true, true, "read_chunk ppt=8 NT exp");
    RUN(2, false, true, "read_chunk ppt=2 exp");
    RUN(4, false, true, "read_chunk ppt=4 exp");
    RUN(4, true, false, "read_chunk ppt=4 NT sum");
    for (int w : {2, 4, 8}) {
        char nm[64]; snprintf(nm, 64, "row_stats_kernel W=%d NT", w);
        dim3 rg((T * Uu + w - 1) / w, N);
        if (w == 2) timeit(nm, [&] { hipLaunchKernelGGL((row_stats_kernel<F32, 2, true>), rg, dim3(128), 0, 0, acts, labels, xlen, ylen, lp2, logz, T, Uu, Up, A, 0, 1, nullptr, 0ull); });
        if (w == 4) timeit(nm, [&] { hipLaunchKernelGGL((row_stats_kernel<F32, 4, true>), rg, dim3(256), 0, 0, acts, labels, xlen, ylen, lp2, logz, T, Uu, Up, A, 0, 1, nullptr, 0ull); });
        if (w == 8) timeit(nm, [&] { hipLaunchKernelGGL((row_stats_kernel<F32, 8, true>), rg, dim3(512), 0, 0, acts, labels, xlen, ylen, lp2, logz, T, Uu, Up, A, 0, 1, nullptr, 0ull); });
    }
    { dim3 rg(T * Uu, N);
#define RB(KK) timeit("row_stats_block_kernel NT K=" #KK, [&] { hipLaunchKernelGGL((row_stats_block_kernel<F32, true, KK>), rg, dim3(256), 0, 0, acts, labels, xlen, ylen, lp2, logz, T, Uu, Up, A, 0, 1, nullptr, 0ull); })
      RB(2); RB(3); RB(4); RB(5); RB(8); }
    return 0;
}
