// lin_chain.hip -- what one anti-diagonal of the lattice recursion costs a lone wavefront, by arithmetic:
//   mode 0  the log-domain step the library uses today (add, add, dpp, sub, max, v_exp_f32, add, v_log_f32, add), 16-step
//           chunks, fp32 results stored per chunk
//   mode 1  LINEAR-domain fp64 step  a' = fma(shr(a), pl, a * pb)  on probabilities: operands are fp32 probabilities
//           converted at use (v_cvt_f64_f32; the label operand shifted one lane first), results leave as fp32 base-2 logs
//           (exponent field + v_log_f32 of the mantissa), per-lane exponents re-normalised per chunk
//   mode 2  mode 1 with the results stored as raw fp64 (no logs)
//   mode 3  mode 1 without the per-step ldexp of the label operand (one exponent per wavefront)
//   mode 4  the bare fp64 chain on constant operands (latency floor)
//   mode 5  the bare fp32 chain  a' = fma(shr(a), pl, a * pb)
//   mode 6  the chain wavefront of a helper-wavefront design: fp64 operands read from LDS, raw fp64 results written to LDS
// One wavefront per block, one block per CU; reports shader cycles (s_memtime) and wall ns per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../warp-transducer_amd/csrc/rnnt_device.h"
using namespace rnnt;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ float log2_of_f64(double v, int delta) {
    const int e = __builtin_amdgcn_frexp_exp(v);
    const float m = static_cast<float>(__builtin_amdgcn_frexp_mant(v));
    return fmaxf(static_cast<float>(e + delta) + __builtin_amdgcn_logf(m), -1.0e30f);
}

template <int MODE, int C>
__global__ __launch_bounds__(64) void chain_kernel(const float2* __restrict__ cells, float* __restrict__ out32,
                                                   double* __restrict__ out64, int nchunks, unsigned long long* cyc) {
    const int lane = threadIdx.x;
    const size_t base = static_cast<size_t>(blockIdx.x) * (static_cast<size_t>(nchunks) + 1) * C * 64;
    const float2* src = cells + base + lane;
    float* dst32 = out32 + base + lane;
    double* dst64 = out64 + base + lane;
    float bx[2][C], by[2][C];
    auto fetch = [&](int j, float* x, float* y) {
#pragma unroll
        for (int k = 0; k < C; ++k) { const float2 v = src[(static_cast<size_t>(j) * C + k) * 64]; x[k] = v.x; y[k] = v.y; }
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (MODE == 0) {
        float a = lane == 0 ? 0.0f : -1.0e30f, up = -1.0e30f, hist[C];
        double off = 0;
        fetch(0, bx[0], by[0]);
        auto chunk = [&](int j, const float* x, const float* y, float* nx, float* ny) {
            if (j > 0) {
#pragma unroll
                for (int k = 0; k < C; ++k) dst32[(static_cast<size_t>(j - 1) * C + k) * 64] = hist[k];
            }
            fetch(j + 1, nx, ny);
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const float stay = a + x[k], emit = a + y[k];
                up = wave_shr1(up, emit);
                a = log2_add(stay, up);
                hist[k] = a;
            }
            const float m = wave_max_dpp(a);
            a -= m; off += m;
        };
        for (int j = 0; j < nchunks; j += 2) { chunk(j, bx[0], by[0], bx[1], by[1]); chunk(j + 1, bx[1], by[1], bx[0], by[0]); }
        out64[blockIdx.x * 64 + lane] = off + a + hist[0];
    } else if constexpr (MODE >= 1 && MODE <= 3) {
        double a = lane == 0 ? 1.0 : 0.0, hist[C], up = 0.0;
        int e_lane = 0, de = 0, e_hist = 0;
        fetch(0, bx[0], by[0]);
        auto chunk = [&](int j, const float* x, const float* y, float* nx, float* ny) {
            if (j > 0) {
#pragma unroll
                for (int k = 0; k < C; ++k) {
                    if constexpr (MODE == 2) dst64[(static_cast<size_t>(j - 1) * C + k) * 64] = hist[k];
                    else dst32[(static_cast<size_t>(j - 1) * C + k) * 64] = log2_of_f64(hist[k], e_hist);
                }
            }
            fetch(j + 1, nx, ny);
            e_hist = e_lane;
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const double pb = static_cast<double>(x[k]);
                double pl = static_cast<double>(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(y[k]), 0x138, 0xf, 0xf, true)));
                if constexpr (MODE != 3) pl = __builtin_ldexp(pl, de);
                const double t = a * pb;
                up = wave_shr1(up, a);
                a = __builtin_fma(up, pl, t);
                hist[k] = a;
            }
            // per-lane re-normalisation (zero lanes keep their exponent)
            const int e = a != 0.0 ? __builtin_amdgcn_frexp_exp(a) : 0;
            a = __builtin_ldexp(a, -e);
            e_lane += e;
            de = dpp_shr1(e_lane, e_lane) - e_lane;
            de = de > 100 ? 100 : de;
        };
        for (int j = 0; j < nchunks; j += 2) { chunk(j, bx[0], by[0], bx[1], by[1]); chunk(j + 1, bx[1], by[1], bx[0], by[0]); }
        out64[blockIdx.x * 64 + lane] = a + hist[0] + e_lane;
    } else if constexpr (MODE == 6) {
        // the chain wavefront of a helper-wavefront design: operands arrive as fp64 pairs in LDS (ds_read_b128), results leave
        // as raw fp64 into LDS (ds_write_b64); conversions would be done by other wavefronts of the block (not modelled: upper bound)
        __shared__ double2 opnd[2][C][64];
        __shared__ double res[2][C][64];
        for (int k = 0; k < C; ++k) { opnd[0][k][lane] = make_double2(0.4 + 1e-3 * lane, 0.1); opnd[1][k][lane] = make_double2(0.5, 0.12); }
        __syncthreads();
        double a = lane == 0 ? 1.0 : 0.0, up = 0.0;
        for (int j = 0; j < nchunks; ++j) {
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const double2 o = opnd[j & 1][k][lane];
                const double t = a * o.x;
                up = wave_shr1(up, a);
                a = __builtin_fma(up, o.y, t);
                res[j & 1][k][lane] = a;
            }
            const int e = a != 0.0 ? __builtin_amdgcn_frexp_exp(a) : 0;
            a = __builtin_ldexp(a, -e);
            __builtin_amdgcn_s_barrier();
        }
        out64[blockIdx.x * 64 + lane] = a + res[0][0][lane];
    } else if constexpr (MODE == 4) {
        double a = lane == 0 ? 1.0 : 0.0, up = 0.0;
        const double pb = 0.5 + 1e-3 * lane, pl = 0.5 - 1e-3 * lane;
        for (int j = 0; j < nchunks; ++j) {
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const double t = a * pb;
                up = wave_shr1(up, a);
                a = __builtin_fma(up, pl, t);
            }
        }
        out64[blockIdx.x * 64 + lane] = a;
    } else {
        float a = lane == 0 ? 1.0f : 0.0f, up = 0.0f;
        const float pb = 0.5f + 1e-3f * lane, pl = 0.5f - 1e-3f * lane;
        for (int j = 0; j < nchunks; ++j) {
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const float t = a * pb;
                up = wave_shr1(up, a);
                a = __builtin_fmaf(up, pl, t);
            }
        }
        out64[blockIdx.x * 64 + lane] = a;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int C> void run(const char* name, const float2* cells, float* o32, double* o64, unsigned long long* cyc, int nchunks, int blocks) {
    hipLaunchKernelGGL((chain_kernel<MODE, C>), dim3(blocks), dim3(64), 0, 0, cells, o32, o64, nchunks, cyc);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((chain_kernel<MODE, C>), dim3(blocks), dim3(64), 0, 0, cells, o32, o64, nchunks, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks);
    CK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    const double steps = static_cast<double>(nchunks) * C;
    printf("%-58s C=%2d blocks=%4d: %7.1f cycles/step  %6.1f ns/step (kernel %.3f ms)\n", name, C, blocks, avg / steps, ms * 1e6 / steps, ms);
}

int main(int argc, char** argv) {
    const int nchunks = argc > 1 ? atoi(argv[1]) : 128;      // even
    const int blocks = argc > 2 ? atoi(argv[2]) : 256;
    const size_t rows = static_cast<size_t>(blocks) * (nchunks + 1) * 16;
    std::vector<float2> lin(rows * 64), lg(rows * 64);
    for (size_t i = 0; i < lin.size(); ++i) {
        const float pb = 0.3f + 0.4f * (rand() % 1000) * 1e-3f, pl = 0.05f + 0.2f * (rand() % 1000) * 1e-3f;
        lin[i] = make_float2(pb, pl);
        lg[i] = make_float2(log2f(pb), log2f(pl));
    }
    float2 *dlin, *dlg; float* o32; double* o64; unsigned long long* cyc;
    CK(hipMalloc(&dlin, lin.size() * 8)); CK(hipMalloc(&dlg, lg.size() * 8));
    CK(hipMalloc(&o32, rows * 64 * 4)); CK(hipMalloc(&o64, rows * 64 * 8)); CK(hipMalloc(&cyc, blocks * 8));
    CK(hipMemcpy(dlin, lin.data(), lin.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dlg, lg.data(), lg.size() * 8, hipMemcpyHostToDevice));
    run<0, 16>("0 log-domain (today)", dlg, o32, o64, cyc, nchunks, blocks);
    run<1, 16>("1 linear fp64, cvt operands, log2 results, lane exps", dlin, o32, o64, cyc, nchunks, blocks);
    run<1, 8>("1 linear fp64, cvt operands, log2 results, lane exps", dlin, o32, o64, cyc, nchunks * 2, blocks);
    run<2, 16>("2 linear fp64, cvt operands, raw fp64 results", dlin, o32, o64, cyc, nchunks, blocks);
    run<3, 16>("3 linear fp64, cvt operands, log2 results, wave exp", dlin, o32, o64, cyc, nchunks, blocks);
    run<6, 16>("6 chain wavefront of a helper design (LDS in / out)", dlin, o32, o64, cyc, nchunks, blocks);
    run<4, 16>("4 bare fp64 chain (mul | dpp x2 -> fma)", dlin, o32, o64, cyc, nchunks, blocks);
    run<5, 16>("5 bare fp32 chain (mul | dpp -> fma)", dlin, o32, o64, cyc, nchunks, blocks);
    // check that mode 1 and mode 0 agree on a log-likelihood-like number
    return 0;
}
