// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes of the additive-joint kernels
// (VERDICT round 5, item 2b).  Every kernel reads (or writes) a buffer of KNOWN size exactly once -- 1 GiB, four times
// the Infinity Cache -- so counter / bytes is the calibration factor of that shape:
//   flat16      16 B per lane, a wavefront covers 1 KB contiguous (the materialised path's packet streams; guide: 0.5)
//   flat4        4 B per lane, a wavefront covers 256 B contiguous
//   flat8        8 B per lane
//   rowseg16    16 B per lane, EIGHT lanes cover 128 B of a row, eight rows (20 000 B apart) per instruction: joint_z_kernel
//   rowseg16x4  16 B per lane, 32 lanes cover 512 B of a row, two rows per instruction: joint_df / joint_dg (NK = 4)
//   write16     16 B per lane stores, flat (df)
// Run:  rocprofv3 --pmc FETCH_SIZE --kernel-trace -- ./fetch_calib   and   --pmc WRITE_SIZE ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename V>
__global__ __launch_bounds__(256) void flat_read(const V* __restrict__ in, float* __restrict__ out, size_t n) {
    float acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        V v = in[i];
        const float* p = reinterpret_cast<const float*>(&v);
        for (unsigned j = 0; j < sizeof(V) / 4; ++j) acc += p[j];
    }
    if (acc == 123.456f) out[0] = acc;
}
// rows of ROWB bytes; a wavefront reads LPR lanes x 16 B of 64/LPR rows per instruction and walks along the rows
template <int LPR>
__global__ __launch_bounds__(256) void rowseg_read(const float4* __restrict__ in, float* __restrict__ out, int rows, int row_f4) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RPI = 64 / LPR;                               // rows per instruction
    const int r0 = (blockIdx.x * 4 + wave) * RPI + lane / LPR;
    if (r0 >= rows) return;
    const float4* row = in + (size_t)r0 * row_f4;
    float acc = 0;
    for (int c = lane % LPR; c < row_f4; c += LPR) { float4 v = row[c]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void flat_write(float4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = make_float4(1, 2, 3, 4);
}

int main() {
    const int row_f4 = 1250;                                    // 5000 floats = 20 000 B per row (the c3 vocabulary)
    const int rows = 53687;                                     // ~1 GiB
    const size_t bytes = (size_t)rows * row_f4 * 16;
    float4* buf; float* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4096));
    CK(hipMemset(buf, 0, bytes));
    CK(hipDeviceSynchronize());
    printf("buffer %.3f MB\n", bytes / 1e6);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((flat_read<float4>), dim3(4096), dim3(256), 0, 0, buf, out, bytes / 16);
        hipLaunchKernelGGL((flat_read<float>), dim3(4096), dim3(256), 0, 0, (const float*)buf, out, bytes / 4);
        hipLaunchKernelGGL((flat_read<float2>), dim3(4096), dim3(256), 0, 0, (const float2*)buf, out, bytes / 8);
        hipLaunchKernelGGL((rowseg_read<8>), dim3((rows + 31) / 32), dim3(256), 0, 0, buf, out, rows, row_f4);
        hipLaunchKernelGGL((rowseg_read<32>), dim3((rows + 7) / 8), dim3(256), 0, 0, buf, out, rows, row_f4);
        hipLaunchKernelGGL(flat_write, dim3(4096), dim3(256), 0, 0, buf, bytes / 16);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
