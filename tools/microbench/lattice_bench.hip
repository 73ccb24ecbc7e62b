// Development microbenchmark for the lattice kernel: time per diagonal.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../warp-transducer_amd/csrc/rnnt_kernels.h"
using namespace rnnt;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 16, T = argc > 2 ? atoi(argv[2]) : 1500, U = argc > 3 ? atoi(argv[3]) : 41;
    const int Up = (U + 63) / 64 * 64, W = Up / 64;
    const size_t Dp = lat_rows(T, U);
    LogPair<float>* cells; float* alpha; float* beta; double *offa, *offb, *llf, *llb; float* costs; int *xlen, *ylen;
    CK(hipMalloc(&cells, N * Dp * Up * 8)); CK(hipMalloc(&beta, (N * Dp * Up + Up + 64) * 4)); CK(hipMalloc(&alpha, (N * Dp * Up + Up + 64) * 4));
    CK(hipMalloc(&offa, N * W * Dp * 8)); CK(hipMalloc(&offb, (N * W * Dp + Dp) * 8));
    CK(hipMalloc(&llf, N * 8)); CK(hipMalloc(&llb, N * 8)); CK(hipMalloc(&costs, N * 4));
    CK(hipMalloc(&xlen, N * 4)); CK(hipMalloc(&ylen, N * 4));
    std::vector<LogPair<float>> h(N * Dp * Up);
    for (auto& c : h) { c.x = -1.0f - (rand() % 100) * 0.01f; c.y = -2.0f - (rand() % 100) * 0.01f; }
    CK(hipMemcpy(cells, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    std::vector<int> hx(N, T), hy(N, U - 1);
    CK(hipMemcpy(xlen, hx.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ylen, hy.data(), N * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&] {
        const int cols = argc > 4 ? atoi(argv[4]) : 2;          // columns per lane for U > 64
        if (W == 1) hipLaunchKernelGGL((lattice_kernel<float, 1, 1>), dim3(N * 2), dim3(64), 0, 0, cells, alpha, beta, offa, offb, llf, llb, costs, xlen, ylen, T, U, Up, 2);
        else if (cols == 1) hipLaunchKernelGGL((lattice_kernel<float, 8, 1>), dim3(N * 2), dim3(Up), 0, 0, cells, alpha, beta, offa, offb, llf, llb, costs, xlen, ylen, T, U, Up, 2);
        else if (lat_waves(Up, 2) <= 4) hipLaunchKernelGGL((lattice_kernel<float, 4, 2>), dim3(N * 2), dim3(lat_waves(Up, 2) * 64), 0, 0, cells, alpha, beta, offa, offb, llf, llb, costs, xlen, ylen, T, U, Up, 2);
        else hipLaunchKernelGGL((lattice_kernel<float, 8, 2>), dim3(N * 2), dim3(lat_waves(Up, 2) * 64), 0, 0, cells, alpha, beta, offa, offb, llf, llb, costs, xlen, ylen, T, U, Up, 2);
    };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    float c0; CK(hipMemcpy(&c0, costs, 4, hipMemcpyDeviceToHost));
    printf("cols=%d N=%d T=%d U=%d: %.1f us, %.1f ns/diagonal (cost[0]=%.3f)\n", (U > 64 ? (argc > 4 ? atoi(argv[4]) : 2) : 1), N, T, U, ms * 1e3, ms * 1e6 / (T + U - 2), c0);
    return 0;
}
