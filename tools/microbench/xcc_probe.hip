// xcc_probe.hip -- which XCD does workgroup i run on?  Reads HW_REG_XCC_ID (hwreg 20 on gfx942/gfx950)
// in every block of a 1-D and a 2-D launch and prints the mapping statistics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out) {
    if (threadIdx.x == 0) {
        const unsigned id = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((32 - 1) << 11));   // XCC_ID, all 32 bits
        out[blockIdx.y * gridDim.x + blockIdx.x] = id;
    }
}
int main() {
    const int gx = 4096, gy = 3;
    unsigned* d; hipMalloc(&d, gx * gy * 4);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(gx, gy), dim3(256), rep ? 50 * 1024 : 0, 0, d);
        hipDeviceSynchronize();
        std::vector<unsigned> h(gx * gy);
        hipMemcpy(h.data(), d, gx * gy * 4, hipMemcpyDeviceToHost);
        int match = 0; unsigned seen = 0;
        for (int i = 0; i < gx * gy; ++i) { const unsigned x = h[i] & 0xf; seen |= 1u << x; if (x == (unsigned)(i % 8)) ++match; }
        printf("launch %d (lds %d KB): raw reg of block 0..15:", rep, rep ? 50 : 0);
        for (int i = 0; i < 16; ++i) printf(" %x", h[i]);
        printf("\n   xcc ids seen mask 0x%x, blocks with xcc == linear_id %% 8: %d of %d\n", seen, match, gx * gy);
    }
    return 0;
}
