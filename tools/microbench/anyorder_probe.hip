// anyorder_probe.hip -- can two kernels of ONE stream overlap, the later one feeding the earlier one through flags?
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/anyorder_probe.hip -o tools/microbench/anyorder_probe
// hipExtLaunchKernelGGL(..., flags = hipExtAnyOrderLaunch) clears the barrier bit of the dispatch packet: the command
// processor may start it before the previous packet of the queue has finished.  Questions (every wait below has a time
// limit, so a "no" is a number, not a hung GPU):
//   T1  does a consumer launched FIRST see a flag set by a producer launched SECOND with the any-order flag?
//   T2  lattice-like role: 128 long-lived polling blocks launched first, then a streaming kernel of ~100k blocks that
//       counts "tiles done" per sample: when does each poller get released, and what does their presence cost the
//       streaming kernel?
//   T3  a third kernel (any-order, launched last) whose blocks poll what the pollers of T2 publish: are its blocks
//       dispatched only after ALL blocks of the streaming kernel (in-order dispatch across packets of one queue)?
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned long long now() { return wall_clock64(); }        // 100 MHz
constexpr unsigned long long kLimit = 5000000ull;                                      // 50 ms

__global__ void consumer(int* flag, unsigned long long* out) {
    const unsigned long long t0 = now();
    int seen = 0;
    while (!(seen = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) && now() - t0 < kLimit) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) { out[0] = seen; out[1] = now() - t0; }
}
__global__ void producer(int* flag) { if (threadIdx.x == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

// T2 / T3
__global__ __launch_bounds__(256) void pollers(const int* done, int need, int* published, unsigned long long* rel, unsigned long long* t_start,
                                               int hold_us) {
    const int b = blockIdx.x >> 1;
    __shared__ int ok;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = now();
        if (blockIdx.x == 0) *t_start = t0;
        int v = 0;
        while ((v = __hip_atomic_load(done + b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < need && now() - t0 < kLimit) __builtin_amdgcn_s_sleep(64);
        ok = v >= need;
        rel[blockIdx.x] = now();
    }
    __syncthreads();
    // the dependent chain of the lattice: hold_us of sleeping
    const unsigned long long t1 = now();
    while (now() - t1 < static_cast<unsigned long long>(hold_us) * 100ull) __builtin_amdgcn_s_sleep(64);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(published + b, ok ? 1 : 1000, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// streaming kernel: block = one "tile" of 32 KB read; tiles of a sample are consecutive; the last thread counts it done
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void streamer(const u32x4_t* __restrict__ src, float* sink, int* done, int tiles_per_sample, unsigned long long* first_last) {
    const size_t base = static_cast<size_t>(blockIdx.x) * 2048;
    u32x4_t acc = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        acc ^= __builtin_nontemporal_load(src + base + k * 256 + threadIdx.x);
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1.0f;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(done + blockIdx.x / tiles_per_sample, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (blockIdx.x == 0) first_last[0] = now();
        if (blockIdx.x == gridDim.x - 1) first_last[1] = now();
    }
}

// third kernel: block i belongs to sample i / per; waits for published[sample] == 2
__global__ __launch_bounds__(256) void third(const int* published, int per, unsigned long long* stamp, int* bad) {
    const int b = blockIdx.x / per;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = now();
        int v = 0;
        while ((v = __hip_atomic_load(published + b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < 2 && now() - t0 < kLimit) __builtin_amdgcn_s_sleep(64);
        if (v != 2) atomicAdd(bad, 1);
        if (blockIdx.x == 0) stamp[0] = t0;
        if (blockIdx.x == gridDim.x - 1) stamp[1] = now();
    }
}

int main() {
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    int* flag; unsigned long long* out;
    CHECK(hipMalloc(&flag, 4)); CHECK(hipHostMalloc(&out, 64));
    for (int mode = 0; mode < 2; ++mode) {
        CHECK(hipMemsetAsync(flag, 0, 4, s));
        CHECK(hipStreamSynchronize(s));
        hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, s, flag, out);
        if (mode == 0) hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, s, flag);
        else hipExtLaunchKernelGGL(producer, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, flag);
        CHECK(hipStreamSynchronize(s));
        printf("T1 %-22s consumer saw the flag: %llu after %.1f us\n", mode ? "any-order producer:" : "ordinary producer:", out[0], out[1] / 100.0);
    }

    const int N = 64, TPS = 1600;                                  // 64 samples x 1600 tiles x 32 KB = 3.3 GB
    const size_t bytes = static_cast<size_t>(N) * TPS * 32768;
    u32x4_t* src; float* sink; int *done, *published, *bad;
    unsigned long long *rel, *tstart, *fl, *stamp;
    CHECK(hipMalloc(&src, bytes)); CHECK(hipMemset(src, 1, bytes)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMalloc(&done, N * 4)); CHECK(hipMalloc(&published, N * 4)); CHECK(hipMalloc(&bad, 4));
    CHECK(hipHostMalloc(&rel, 2 * N * 8)); CHECK(hipHostMalloc(&tstart, 8)); CHECK(hipHostMalloc(&fl, 16)); CHECK(hipHostMalloc(&stamp, 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto reset = [&]() { (void)hipMemsetAsync(done, 0, N * 4, s); (void)hipMemsetAsync(published, 0, N * 4, s); (void)hipMemsetAsync(bad, 0, 4, s); };
    // streaming kernel alone
    for (int rep = 0; rep < 3; ++rep) {
        reset();
        CHECK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(streamer, dim3(N * TPS), dim3(256), 0, s, src, sink, done, TPS, fl);
        CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("T2 streaming kernel alone: %.3f ms (%.2f TB/s)\n", ms, bytes / ms / 1e9);
    }
    for (int hold : {0, 270}) {
        for (int rep = 0; rep < 3; ++rep) {
            reset();
            CHECK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(pollers, dim3(2 * N), dim3(256), 0, s, done, TPS, published, rel, tstart, hold);
            hipExtLaunchKernelGGL(streamer, dim3(N * TPS), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, src, sink, done, TPS, fl);
            hipExtLaunchKernelGGL(third, dim3(N * 100), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, published, 100, stamp, bad);
            CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            int hbad = 0, hpub[N];
            CHECK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hpub, published, N * 4, hipMemcpyDeviceToHost));
            int timeouts = 0;
            for (int b = 0; b < N; ++b) timeouts += hpub[b] != 2;
            const double t0 = static_cast<double>(*tstart);
            printf("T2/T3 hold %3d us: pollers + any-order streamer + any-order third: %.3f ms total; pollers timed out: %d, third-kernel blocks that "
                   "gave up: %d\n   release of sample 0 / 21 / 42 / 63 at %.0f / %.0f / %.0f / %.0f us; streamer first block done %.0f us, last "
                   "block done %.0f us; third kernel: first block started %.0f us, last block done %.0f us (after the pollers' start)\n",
                   hold, ms, timeouts, hbad, (rel[0] - t0) / 100, (rel[42] - t0) / 100, (rel[84] - t0) / 100, (rel[126] - t0) / 100,
                   (fl[0] - t0) / 100, (fl[1] - t0) / 100, (stamp[0] - t0) / 100, (stamp[1] - t0) / 100);
        }
    }
    // the same three kernels in ORDINARY stream order must time out in the pollers (nothing feeds them): skipped -- T1 mode 0 shows it
    return 0;
}
