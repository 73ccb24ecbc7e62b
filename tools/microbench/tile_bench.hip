// Development microbenchmark for the LDS-tile row-stats kernel (c4 shape).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../warp-transducer_amd/csrc/rnnt_kernels.h"
using namespace rnnt;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
static int xcd = 1;
template <int G> void run(const float* acts, int* labels, int* xlen, int* ylen, LogPair<float>* lp2, float* logz, unsigned long long R, int T, int U, int Up, int A, double bytes) {
    const int RT = 256 / G; const size_t lds = (size_t)RT * A * 4 + 32;
    if (lds > 64 * 1024) { printf("G=%d: tile too large\n", G); return; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&] { hipLaunchKernelGGL((row_stats_tile_kernel<F32, G>), dim3(((R + RT - 1) / RT + 7) / 8 * 8), dim3(256), lds, 0, acts, labels, xlen, ylen, lp2, logz, R, T, U, Up, A, 0, xcd, nullptr, 64); };
    for (int i = 0; i < 2; ++i) launch();
    CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("G=%-2d tile=%5.1f KB: %7.3f ms %7.1f GB/s\n", G, lds / 1024.0, ms, bytes / ms / 1e6);
}
int main(int argc, char** argv) {
    const int N = 64, T = 1500, U = 301, A = argc > 1 ? atoi(argv[1]) : 50;
    const unsigned long long R = (unsigned long long)N * T * U; const size_t E = R * A;
    const int Up = 320; const size_t Dp = lat_rows(T, U);
    float* acts; int *xlen, *ylen, *labels; LogPair<float>* lp2; float* logz;
    CK(hipMalloc(&acts, E * 4)); CK(hipMalloc(&xlen, N * 4)); CK(hipMalloc(&ylen, N * 4)); CK(hipMalloc(&labels, N * U * 4));
    CK(hipMalloc(&lp2, N * Dp * Up * 8)); CK(hipMalloc(&logz, N * Dp * Up * 4));
    CK(hipMemset(acts, 0, E * 4)); CK(hipMemset(labels, 0, N * U * 4));
    std::vector<int> hx(N, T), hy(N, U - 1);
    CK(hipMemcpy(xlen, hx.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ylen, hy.data(), N * 4, hipMemcpyHostToDevice));
    if (argc > 2) xcd = atoi(argv[2]);
    printf("XCD-aware tile order: %d\n", xcd);
    run<1>(acts, labels, xlen, ylen, lp2, logz, R, T, U, Up, A, E * 4.0);
    run<2>(acts, labels, xlen, ylen, lp2, logz, R, T, U, Up, A, E * 4.0);
    run<4>(acts, labels, xlen, ylen, lp2, logz, R, T, U, Up, A, E * 4.0);
    run<8>(acts, labels, xlen, ylen, lp2, logz, R, T, U, Up, A, E * 4.0);
    return 0;
}
