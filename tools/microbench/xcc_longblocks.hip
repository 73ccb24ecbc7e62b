// xcc_longblocks.hip -- does "workgroup i runs on XCD i % 8" survive LONG-LIVED blocks that fill the chip?
// Replicates the launch geometry of row_stats_tile_kernel on c4 (112 880 blocks of 256 threads, 52 KB of LDS each: three
// resident blocks per CU, each streaming 51 KB) and records HW_REG_XCC_ID per block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void probe(const f32x4* __restrict__ src, unsigned* __restrict__ out, float* __restrict__ sink, int per) {
    extern __shared__ f32x4 lds[];
    f32x4 acc = {0, 0, 0, 0};
    const f32x4* p = src + static_cast<size_t>(blockIdx.x) * per;
    for (int i = threadIdx.x; i < per; i += 256) { const f32x4 v = __builtin_nontemporal_load(p + i); lds[i] = v; }
    __syncthreads();
    for (int i = threadIdx.x; i < per; i += 256) acc += lds[(i * 7) % per];
    if (acc.x + acc.y + acc.z + acc.w == 12345.0f) sink[0] = 1.0f;
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((32 - 1) << 11)) & 0xf;
}
int main() {
    const int nblk = 112880, per = 3200;                 // 3200 float4 = 51 200 bytes per block
    f32x4* src; unsigned* d; float* sink;
    hipMalloc(&src, static_cast<size_t>(nblk) * per * 16); hipMalloc(&d, nblk * 4); hipMalloc(&sink, 4);
    hipMemset(src, 0, static_cast<size_t>(nblk) * per * 16);
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe, dim3(nblk), dim3(256), per * 16 + 32, 0, src, d, sink, per);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned> h(nblk);
        hipMemcpy(h.data(), d, nblk * 4, hipMemcpyDeviceToHost);
        long match = 0; long cnt[16] = {0};
        for (int i = 0; i < nblk; ++i) { cnt[h[i] & 15]++; if ((h[i] & 15) == static_cast<unsigned>(i % 8)) ++match; }
        // how far from round-robin: for windows of 8 consecutive blocks, how many contain all 8 XCDs
        long full = 0;
        for (int i = 0; i + 8 <= nblk; i += 8) { unsigned m = 0; for (int j = 0; j < 8; ++j) m |= 1u << (h[i + j] & 15); if (m == 0xff) ++full; }
        printf("rep %d: %.3f ms (%.0f GB/s); blocks with xcc == i %% 8: %ld of %d (%.1f %%); aligned windows of 8 holding all 8 XCDs: %ld of %d; per-XCD counts:",
               rep, ms, static_cast<double>(nblk) * per * 16 / ms / 1e6, match, nblk, 100.0 * match / nblk, full, nblk / 8);
        for (int x = 0; x < 8; ++x) printf(" %ld", cnt[x]);
        printf("\n");
    }
    return 0;
}
