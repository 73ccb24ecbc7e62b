// Development microbenchmark for the lattice kernel: time per diagonal.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include "../../warp-transducer_amd/csrc/rnnt_kernels.h"
using namespace rnnt;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
// argv[5] = 1: the same measurement while a second stream keeps the memory system saturated with a streaming copy
// (what the lattice kernel would see beside the statistics / gradient kernel of another half of the batch)
__global__ __launch_bounds__(256) void hog_copy(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 16, T = argc > 2 ? atoi(argv[2]) : 1500, U = argc > 3 ? atoi(argv[3]) : 41;
    const int Up = lat_stride(U), W = (Up + 63) / 64;
    const size_t Dp = lat_rows(T, U);
    LogPair<float>* cells; float* alpha; float* beta; double *offa, *offb, *llf, *llb; float* costs; int *xlen, *ylen;
    CK(hipMalloc(&cells, N * Dp * Up * 8)); CK(hipMalloc(&beta, (N * Dp * Up + Up + 64) * 4)); CK(hipMalloc(&alpha, (N * Dp * Up + Up + 64) * 4));
    CK(hipMalloc(&offa, N * W * Dp * 8)); CK(hipMalloc(&offb, (N * W * Dp + Dp) * 8));
    CK(hipMalloc(&llf, N * 8)); CK(hipMalloc(&llb, N * 8)); CK(hipMalloc(&costs, N * 4));
    CK(hipMalloc(&xlen, N * 4)); CK(hipMalloc(&ylen, N * 4));
    int* padflag; CK(hipMalloc(&padflag, 4));
    std::vector<LogPair<float>> h(N * Dp * Up);
    for (auto& c : h) { c.x = -1.0f - (rand() % 100) * 0.01f; c.y = -2.0f - (rand() % 100) * 0.01f; }
    CK(hipMemcpy(cells, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    std::vector<int> hx(N, T), hy(N, U - 1);
    if (argc > 8 && atoi(argv[8]) != 0)                          // argv[8] = 1: ragged lengths
        for (int i = 1; i < N; ++i) { hx[i] = T / 2 + rand() % (T - T / 2 + 1); hy[i] = (U - 1) / 2 + rand() % (U - (U - 1) / 2); }
    CK(hipMemcpy(xlen, hx.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ylen, hy.data(), N * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t ls; CK(hipStreamCreateWithFlags(&ls, hipStreamNonBlocking));
    auto launch = [&] {
        const int cols = argc > 4 ? atoi(argv[4]) : 2;          // columns per lane for U > 64
        if (W == 1 && argc > 6 && atoi(argv[6]) != 0) hipLaunchKernelGGL((lattice_lin_kernel<0>), dim3(N * 2), dim3(kLinThreads), 0, ls, cells, alpha, beta, offa, offb, llf, llb, costs, xlen, ylen, T, U, Up, 2, 0, padflag);   // argv[6] = 1: the linear-domain kernel
        else if (W == 1) hipLaunchKernelGGL((lattice_kernel<float, 1, 1>), dim3(N * 2), dim3(64), 0, ls, cells, alpha, beta, offa, offb, llf, llb, costs, xlen, ylen, T, U, Up, 2, padflag);
        else if (cols == 1) hipLaunchKernelGGL((lattice_kernel<float, 8, 1>), dim3(N * 2), dim3(W * 64), 0, ls, cells, alpha, beta, offa, offb, llf, llb, costs, xlen, ylen, T, U, Up, 2, padflag);
        else if (lat_waves(Up, 2) <= 4) hipLaunchKernelGGL((lattice_kernel<float, 4, 2>), dim3(N * 2), dim3(lat_waves(Up, 2) * 64), 0, ls, cells, alpha, beta, offa, offb, llf, llb, costs, xlen, ylen, T, U, Up, 2, padflag);
        else hipLaunchKernelGGL((lattice_kernel<float, 8, 2>), dim3(N * 2), dim3(lat_waves(Up, 2) * 64), 0, ls, cells, alpha, beta, offa, offb, llf, llb, costs, xlen, ylen, T, U, Up, 2, padflag);
    };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    const bool contend = argc > 5 && atoi(argv[5]) != 0;
    hipStream_t hog; CK(hipStreamCreateWithFlags(&hog, hipStreamNonBlocking));
    u32x4 *hin = nullptr, *hout = nullptr;
    const size_t hn = (1ull << 30) / 16;                         // 1 GiB read + 1 GiB written per launch (~0.33 ms)
    const int reps = 10;
    if (contend) {
        CK(hipMalloc(&hin, hn * 16)); CK(hipMalloc(&hout, hn * 16));
        CK(hipMemset(hin, 1, hn * 16));
        CK(hipDeviceSynchronize());
        for (int i = 0; i < 40 * reps; ++i) hipLaunchKernelGGL(hog_copy, dim3(16384), dim3(256), 0, hog, hin, hout, hn);
    }
    CK(hipEventRecord(e0, ls));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, ls)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const bool still = contend && hipStreamQuery(hog) == hipErrorNotReady;   // the copy outlasted the measurement
    CK(hipDeviceSynchronize());
    if (contend) printf("[beside a streaming copy%s] ", still ? "" : " -- WARNING: the copy ended first");
    float c0; CK(hipMemcpy(&c0, costs, 4, hipMemcpyDeviceToHost));
    { double lf, lb; CK(hipMemcpy(&lf, llf, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&lb, llb, 8, hipMemcpyDeviceToHost)); printf("[llf %.6f llb %.6f] ", lf * 0.6931471805599453, lb); }
    if (argc > 7) {                                              // argv[7] = R: R more launches, every output compared bit by bit with the first
        const int R = atoi(argv[7]);
        const size_t na = static_cast<size_t>(N) * Dp * Up;
        std::vector<float> a0(na), b0(na), a1(na), b1(na);
        std::vector<double> f0(N), g0(N), f1(N), g1(N);
        size_t dl = 0, da = 0;
        for (int r = 0; r <= R; ++r) {
            CK(hipMemsetAsync(alpha, 0xff, na * 4, ls)); CK(hipMemsetAsync(beta, 0xff, na * 4, ls));
            launch(); CK(hipStreamSynchronize(ls));
            CK(hipMemcpy(r ? a1.data() : a0.data(), alpha, na * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(r ? b1.data() : b0.data(), beta, na * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(r ? f1.data() : f0.data(), llf, N * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(r ? g1.data() : g0.data(), llb, N * 8, hipMemcpyDeviceToHost));
            if (r) {
                for (int i = 0; i < N; ++i) dl += (memcmp(&f0[i], &f1[i], 8) != 0) + (memcmp(&g0[i], &g1[i], 8) != 0);
                for (size_t i = 0; i < na; ++i) da += (memcmp(&a0[i], &a1[i], 4) != 0) + (memcmp(&b0[i], &b1[i], 4) != 0);
            }
        }
        printf("[%d repeats: %zu likelihoods, %zu lattice values differ] ", R, dl, da);
    }
    printf("cols=%d N=%d T=%d U=%d: %.1f us, %.1f ns/diagonal (cost[0]=%.3f)\n", (U > 64 ? (argc > 4 ? atoi(argv[4]) : 2) : 1), N, T, U, ms * 1e3, ms * 1e6 / (T + U - 2), c0);
    return 0;
}
