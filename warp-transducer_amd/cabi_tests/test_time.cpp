// test_time <Batch size> <Time step> <Label length> <Alphabet size>
// The reference's GPU timing harness protocol (tests/test_time.cu:27-128): lattice maxU = L+1, all
// samples full length, acts uniform(0,1), 10 timed iterations of compute_rnnt_loss WITH gradients,
// wall clock around the call (it includes the costs D2H copy and the stream sync), workspace
// allocated outside the timed region, NO warm-up; prints every iteration, then mean and variance.
// (Inputs are generated on the device: the host mt19937 stream takes ~1 min at the c3 size.)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include <rnnt.h>

__global__ void fill_uniform(float* p, size_t n, unsigned seed) {
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (; i < n; i += stride) {
        unsigned long long x = (i + 1) * 0x9E3779B97F4A7C15ull + seed;      // splitmix64 hash of the index
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
        p[i] = static_cast<float>(x >> 40) * (1.0f / 16777216.0f);
    }
}

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "Arguments: <Batch size> <Time step> <Label length> <Alphabet size>\n"); return 1; }
    const int B = std::atoi(argv[1]), T = std::atoi(argv[2]), L = std::atoi(argv[3]), A = std::atoi(argv[4]);
    const int U = L + 1;
    std::printf("Arguments:\nBatch size: %d\nTime step: %d\nLabel length: %d\nAlphabet size: %d\n", B, T, L, A);
    const size_t len = static_cast<size_t>(B) * T * U * A;
    float *acts, *grads;
    if (hipMalloc(&acts, len * sizeof(float)) != hipSuccess || hipMalloc(&grads, len * sizeof(float)) != hipSuccess) {
        std::fprintf(stderr, "hipMalloc of %.2f GB failed\n", 2.0 * len * 4 / 1e9);   // the reference never checks (README N=128 row)
        return 1;
    }
    hipLaunchKernelGGL(fill_uniform, dim3(4096), dim3(256), 0, 0, acts, len, 0u);
    std::vector<int> labels(static_cast<size_t>(B) * L), ll(B, L), tl(B, T);
    std::mt19937 gen(1);
    std::uniform_int_distribution<> dis(1, A - 1);
    for (int i = 0; i < L; ++i) labels[i] = dis(gen);
    if (L >= 3) { labels[L / 2] = labels[L / 2 + 1]; labels[L / 2 - 1] = labels[L / 2]; }
    for (int b = 1; b < B; ++b) for (int i = 0; i < L; ++i) labels[static_cast<size_t>(b) * L + i] = labels[i];
    int *d_labels, *d_ll, *d_tl;
#define HIP_OK(x) do { if ((x) != hipSuccess) { std::fprintf(stderr, "HIP call failed: %s\n", #x); return 1; } } while (0)
    HIP_OK(hipMalloc(&d_labels, labels.size() * sizeof(int)));
    HIP_OK(hipMalloc(&d_ll, B * sizeof(int)));
    HIP_OK(hipMalloc(&d_tl, B * sizeof(int)));
    HIP_OK(hipMemcpy(d_labels, labels.data(), labels.size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_ll, ll.data(), B * sizeof(int), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_tl, tl.data(), B * sizeof(int), hipMemcpyHostToDevice));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    rnntOptions options{};
    options.maxT = T; options.maxU = U; options.blank_label = 0; options.loc = RNNT_GPU;
    options.stream = reinterpret_cast<CUstream>(stream);
    size_t bytes = 0;
    if (get_workspace_size(T, U, B, true, &bytes) != RNNT_STATUS_SUCCESS) return 1;
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> costs(B), time;
    for (int i = 0; i < 10; ++i) {
        void* ws;
        HIP_OK(hipMalloc(&ws, bytes));
        const auto start = std::chrono::high_resolution_clock::now();
        const rnntStatus_t st = compute_rnnt_loss(acts, grads, d_labels, d_ll, d_tl, A, B, costs.data(), ws, options);
        const auto end = std::chrono::high_resolution_clock::now();
        (void)hipFree(ws);
        if (st != RNNT_STATUS_SUCCESS) { std::fprintf(stderr, "compute_rnnt_loss: %s\n", rnntGetStatusString(st)); return 1; }
        const float ms = std::chrono::duration<float, std::milli>(end - start).count();
        time.push_back(ms);
        std::printf("compute_rnnt_loss elapsed time: %f ms\n", ms);
    }
    float sum = 0, var = 0;
    for (float t : time) sum += t;
    sum /= time.size();
    for (float t : time) var += (t - sum) * (t - sum);
    var /= time.size();
    double cost = 0;
    for (float c : costs) cost += c;
    std::printf("average 10 time cost: %f ms variance: %f (sum of costs %.3f)\n", sum, var, cost);
    return 0;
}
