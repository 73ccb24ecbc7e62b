// d2h_small.hip -- how the N-float costs get from device memory to a PAGEABLE host array behind the last kernel:
//   A  hipMemcpyAsync(pageable) + hipStreamSynchronize       (the reference's route, gpu_rnnt.h:208-213)
//   B  hipStreamSynchronize + hipMemcpy (synchronous copy)
//   C  kernel writes a pinned mapped buffer, hipStreamSynchronize, memcpy on the host   (the opt-in staging buffer)
//   D  hipStreamSynchronize alone (no copy: floor)
// wall clock per iteration of {tiny kernel, route}, median of 2000.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void fill(float* p, int n, float v) { int i = blockIdx.x * 64 + threadIdx.x; if (i < n) p[i] = v + i; }
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 16;
    float *dev, *pin, *pin_dev;
    CK(hipMalloc(&dev, N * 4));
    CK(hipHostMalloc(&pin, N * 4, hipHostMallocPortable | hipHostMallocMapped));
    CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&pin_dev), pin, 0));
    std::vector<float> host(N);
    hipStream_t s; CK(hipStreamCreate(&s));
    auto run = [&](int mode) -> double {
        std::vector<double> t;
        for (int it = 0; it < 2200; ++it) {
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(fill, dim3((N + 63) / 64), dim3(64), 0, s, mode == 2 ? pin_dev : dev, N, float(it));
            if (mode == 0) { (void)hipMemcpyAsync(host.data(), dev, N * 4, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); }
            else if (mode == 1) { (void)hipStreamSynchronize(s); (void)hipMemcpy(host.data(), dev, N * 4, hipMemcpyDeviceToHost); }
            else if (mode == 2) { (void)hipStreamSynchronize(s); memcpy(host.data(), pin, N * 4); }
            else { (void)hipStreamSynchronize(s); }
            const auto t1 = std::chrono::steady_clock::now();
            if (it >= 200) t.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
            if (mode < 3 && host[N - 1] != float(it) + N - 1) { printf("wrong value\n"); return -1; }
        }
        std::sort(t.begin(), t.end());
        return t[t.size() / 2];
    };
    const char* names[4] = {"A memcpyAsync(pageable) + sync", "B sync + hipMemcpy", "C pinned mapped + sync + host memcpy", "D sync only"};
    for (int m = 0; m < 4; ++m) printf("N=%d  %-40s %.2f us\n", N, names[m], run(m));
    return 0;
}
