// rnnt_device.h -- device-side building blocks for the gfx950 RNN-T kernels.
//
// Wave64 everywhere: one wavefront = 64 lanes, cross-lane traffic through DPP
// (v_mov_b32_dpp wave_shr/wave_shl) and ds_bpermute shuffles, no 32-lane
// assumptions (the reference's include/detail/reduce.h:5,22,32 hard-codes 32).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rnnt {

constexpr int kWave = 64;

// ----------------------------------------------------------------------------- dtypes
// Storage tags.  `store` is what sits in HBM, `comp` is the arithmetic type of the
// row passes and of the lattice (fp32 for 16/32-bit storage, fp64 for fp64 storage).
struct F32 { using store = float;    using comp = float;  };
struct F64 { using store = double;   using comp = double; };
struct BF16 { using store = uint16_t; using comp = float;  };
struct F16 { using store = uint16_t; using comp = float;  };

// 16-byte packet as a native vector (lowers to global_load/store_dwordx4).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ uint4 load_packet(const u32x4* p) {
    u32x4 v;
    if constexpr (NT) v = __builtin_nontemporal_load(p); else v = *p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <bool NT> __device__ __forceinline__ void store_packet(u32x4* p, const uint4& r) {
    u32x4 v = {r.x, r.y, r.z, r.w};
    if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

template <typename Tag> struct Vec {
    static constexpr int N = 16 / sizeof(typename Tag::store);   // elements per 16-byte access
};

__device__ __forceinline__ float bf16_to_f32(uint16_t h) {
    return __uint_as_float(static_cast<uint32_t>(h) << 16);
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {   // round-to-nearest-even
    uint32_t x = __float_as_uint(f);
    if ((x & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((x >> 16) | 0x40u);  // NaN
    x += 0x7fffu + ((x >> 16) & 1u);
    return static_cast<uint16_t>(x >> 16);
}
__device__ __forceinline__ float f16_to_f32(uint16_t h) {
    _Float16 v;
    __builtin_memcpy(&v, &h, 2);
    return static_cast<float>(v);
}
__device__ __forceinline__ uint16_t f32_to_f16(float f) {
    _Float16 v = static_cast<_Float16>(f);
    uint16_t h;
    __builtin_memcpy(&h, &v, 2);
    return h;
}

template <typename Tag> __device__ __forceinline__ typename Tag::comp load1(const typename Tag::store* p);
template <> __device__ __forceinline__ float load1<F32>(const float* p) { return *p; }
template <> __device__ __forceinline__ double load1<F64>(const double* p) { return *p; }
template <> __device__ __forceinline__ float load1<BF16>(const uint16_t* p) { return bf16_to_f32(*p); }
template <> __device__ __forceinline__ float load1<F16>(const uint16_t* p) { return f16_to_f32(*p); }

template <typename Tag> __device__ __forceinline__ void store1(typename Tag::store* p, typename Tag::comp v);
template <> __device__ __forceinline__ void store1<F32>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store1<F64>(double* p, double v) { *p = v; }
template <> __device__ __forceinline__ void store1<BF16>(uint16_t* p, float v) { *p = f32_to_bf16(v); }
template <> __device__ __forceinline__ void store1<F16>(uint16_t* p, float v) { *p = f32_to_f16(v); }

// 16-byte packet <-> Vec<Tag>::N compute values.
template <typename Tag> __device__ __forceinline__ void unpack(const uint4& raw, typename Tag::comp* v);
template <> __device__ __forceinline__ void unpack<F32>(const uint4& r, float* v) {
    v[0] = __uint_as_float(r.x); v[1] = __uint_as_float(r.y);
    v[2] = __uint_as_float(r.z); v[3] = __uint_as_float(r.w);
}
template <> __device__ __forceinline__ void unpack<F64>(const uint4& r, double* v) {
    v[0] = __hiloint2double(static_cast<int>(r.y), static_cast<int>(r.x));
    v[1] = __hiloint2double(static_cast<int>(r.w), static_cast<int>(r.z));
}
template <> __device__ __forceinline__ void unpack<BF16>(const uint4& r, float* v) {
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
    v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack<F16>(const uint4& r, float* v) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = f16_to_f32(static_cast<uint16_t>(w[i] & 0xffffu));
        v[2 * i + 1] = f16_to_f32(static_cast<uint16_t>(w[i] >> 16));
    }
}

// 8-byte half packet -> Vec<Tag>::N / 2 compute values (fp64: one value).
template <typename Tag> __device__ __forceinline__ void unpack_half(const uint2& raw, typename Tag::comp* v);
template <> __device__ __forceinline__ void unpack_half<F32>(const uint2& r, float* v) {
    v[0] = __uint_as_float(r.x); v[1] = __uint_as_float(r.y);
}
template <> __device__ __forceinline__ void unpack_half<F64>(const uint2& r, double* v) {
    v[0] = __hiloint2double(static_cast<int>(r.y), static_cast<int>(r.x));
}
template <> __device__ __forceinline__ void unpack_half<BF16>(const uint2& r, float* v) {
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack_half<F16>(const uint2& r, float* v) {
    v[0] = f16_to_f32(static_cast<uint16_t>(r.x & 0xffffu)); v[1] = f16_to_f32(static_cast<uint16_t>(r.x >> 16));
    v[2] = f16_to_f32(static_cast<uint16_t>(r.y & 0xffffu)); v[3] = f16_to_f32(static_cast<uint16_t>(r.y >> 16));
}

template <typename Tag> __device__ __forceinline__ uint4 pack(const typename Tag::comp* v);
template <> __device__ __forceinline__ uint4 pack<F32>(const float* v) {
    return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]),
                      __float_as_uint(v[3]));
}
template <> __device__ __forceinline__ uint4 pack<F64>(const double* v) {
    return make_uint4(static_cast<uint32_t>(__double2loint(v[0])), static_cast<uint32_t>(__double2hiint(v[0])),
                      static_cast<uint32_t>(__double2loint(v[1])), static_cast<uint32_t>(__double2hiint(v[1])));
}
// gfx950 converts two fp32 to packed bf16 (round-to-nearest-even) in one v_cvt_pk_bf16_f32.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    const f32x2_t f = {lo, hi};
    const bf16x2_t h = __builtin_convertvector(f, bf16x2_t);
    uint32_t u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}
template <> __device__ __forceinline__ uint4 pack<BF16>(const float* v) {
    return make_uint4(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]),
                      cvt_pk_bf16(v[6], v[7]));
}
template <> __device__ __forceinline__ uint4 pack<F16>(const float* v) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        w[i] = static_cast<uint32_t>(f32_to_f16(v[2 * i])) | (static_cast<uint32_t>(f32_to_f16(v[2 * i + 1])) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// ----------------------------------------------------------------------------- math
template <typename T> __device__ __forceinline__ T neg_inf();
template <> __device__ __forceinline__ float neg_inf<float>() { return -__builtin_huge_valf(); }
template <> __device__ __forceinline__ double neg_inf<double>() { return -__builtin_huge_val(); }

// exp / log for the streaming passes and the lattice chain.  fp32 uses the hardware
// v_exp_f32 / v_log_f32 paths; fp64 uses the full-precision library routines.
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ double fast_exp(double x) { return exp(x); }
__device__ __forceinline__ float fast_log(float x) { return __logf(x); }
__device__ __forceinline__ double fast_log(double x) { return log(x); }
__device__ __forceinline__ float acc_log(float x) { return logf(x); }
__device__ __forceinline__ double acc_log(double x) { return log(x); }
__device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double vmax(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float vmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double vmin(double a, double b) { return fmin(a, b); }

// log(e^a + e^b); -inf is the additive zero (reference include/detail/rnnt_helper.h:16-24).
// The lattice works in BASE-2 logs with a finite "log zero" sentinel:
//   log2_add(a, b) = log2(2^a + 2^b) = hi + log2(1 + 2^(lo - hi))
// is then add / max / min / sub / v_exp_f32 / add / v_log_f32 / add with no multiplies, and the
// sentinel (about -1e30; it absorbs every finite addend) needs no -inf / NaN special cases: the
// reference's `if (a == -inf) return b` short-circuits (include/detail/rnnt_helper.h:16-24)
// fall out of the arithmetic.  The argument of the log is in [1,2]: none of the denormal
// handling of the library wrappers is needed; absolute error ~1e-7 per call.
template <typename T> __device__ __forceinline__ T log_zero();
template <> __device__ __forceinline__ float log_zero<float>() { return -1.0e30f; }
template <> __device__ __forceinline__ double log_zero<double>() { return -1.0e300; }

// (-|a - b| instead of min - max: the negation and the absolute value are source modifiers of v_exp_f32,
// one instruction less on the lattice's dependent chain)
// (the maximum as the bare instruction: through fmaxf the compiler canonicalises an operand that came out of a DPP move --
//  a bit pattern it cannot prove quiet -- with an extra v_max x, x per step of the lattice's dependent chain)
__device__ __forceinline__ float log2_add(float a, float b) {
    const float d = a - b;
    float hi;
    asm("v_max_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    return hi + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-__builtin_fabsf(d)));
}
__device__ __forceinline__ double log2_add(double a, double b) {
    const double d = a - b, hi = fmax(a, b);
    return hi + log2(1.0 + exp2(-fabs(d)));
}
// TWO independent log2_add's with their instructions INTERLEAVED (the two lattice columns of a lane).  Left to the compiler the
// two chains came out one after the other -- sub, max, exp, add, log, add of the first, then the same of the second, sharing
// temporaries -- and an in-order SIMD with one resident wavefront then waits out every dependency twice: a two-column step took
// 1.7x a one-column one.  Interleaved, each instruction's producer is two issue slots back, which also covers the one wait state
// a reader of a transcendental's result needs (no s_nop inside).
__device__ __forceinline__ void log2_add_x2(float a0, float b0, float a1, float b1, float& r0, float& r1) {
    float d0, d1, h0, h1;
    asm("v_sub_f32 %2, %6, %7\n\t"
        "v_sub_f32 %3, %8, %9\n\t"
        "v_max_f32 %4, %6, %7\n\t"
        "v_max_f32 %5, %8, %9\n\t"
        "v_exp_f32_e64 %2, -|%2|\n\t"
        "v_exp_f32_e64 %3, -|%3|\n\t"
        "v_add_f32 %2, 1.0, %2\n\t"
        "v_add_f32 %3, 1.0, %3\n\t"
        "v_log_f32 %2, %2\n\t"
        "v_log_f32 %3, %3\n\t"
        "v_add_f32 %0, %4, %2\n\t"
        "v_add_f32 %1, %5, %3"
        : "=&v"(r0), "=&v"(r1), "=&v"(d0), "=&v"(d1), "=&v"(h0), "=&v"(h1)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
}
__device__ __forceinline__ void log2_add_x2(double a0, double b0, double a1, double b1, double& r0, double& r1) {
    r0 = log2_add(a0, b0);
    r1 = log2_add(a1, b1);
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ double fast_exp2(double x) { return exp2(x); }

// ----------------------------------------------------------------------------- cross-lane
// Whole-wave shift by one lane through DPP.  shr: lane i receives lane i-1; shl: lane i receives lane i+1.
// The lane without a source (0 for shr, 63 for shl) KEEPS what `keep` holds there, so a caller that carries
// `keep` from step to step (keep = wave_shr1(keep, v)) pays one v_mov_b32_dpp per shift and no re-initialisation.
__device__ __forceinline__ int dpp_shr1(int keep, int v) {
    return __builtin_amdgcn_update_dpp(keep, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ int dpp_shl1(int keep, int v) {
    return __builtin_amdgcn_update_dpp(keep, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ float wave_shr1(float keep, float v) {
    return __int_as_float(dpp_shr1(__float_as_int(keep), __float_as_int(v)));
}
__device__ __forceinline__ float wave_shl1(float keep, float v) {
    return __int_as_float(dpp_shl1(__float_as_int(keep), __float_as_int(v)));
}
__device__ __forceinline__ double wave_shr1(double keep, double v) {
    int lo = dpp_shr1(__double2loint(keep), __double2loint(v));
    int hi = dpp_shr1(__double2hiint(keep), __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_shl1(double keep, double v) {
    int lo = dpp_shl1(__double2loint(keep), __double2loint(v));
    int hi = dpp_shl1(__double2hiint(keep), __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Generic DPP move (lanes without a source keep their own value).
template <int CTRL, int ROWMASK> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROWMASK, 0xf, false));
}
template <int CTRL, int ROWMASK> __device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROWMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
// Shift inside the 16-lane rows by K lanes (K compile-time after unrolling): row_shl: lane l takes lane l + K of its row,
// row_shr: lane l takes lane l - K; lanes without a source get ZERO (bound_ctrl: no "old" operand, so no register copy in
// front of the move).  The lattice kernel's boundary hand-off: lane 0 <- lane k of row 0, lane 63 <- lane 63 - k of row 3,
// without a trip through the scalar unit; every other lane of the result is don't-care there.
template <int CTRL> __device__ __forceinline__ float dpp_mov0(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ double dpp_mov0(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <typename T> __device__ __forceinline__ T row_shl(T v, int k) {
    switch (k) {
        case 1: return dpp_mov0<0x101>(v); case 2: return dpp_mov0<0x102>(v); case 3: return dpp_mov0<0x103>(v);
        case 4: return dpp_mov0<0x104>(v); case 5: return dpp_mov0<0x105>(v); case 6: return dpp_mov0<0x106>(v);
        case 7: return dpp_mov0<0x107>(v); case 8: return dpp_mov0<0x108>(v); case 9: return dpp_mov0<0x109>(v);
        case 10: return dpp_mov0<0x10A>(v); case 11: return dpp_mov0<0x10B>(v); case 12: return dpp_mov0<0x10C>(v);
        case 13: return dpp_mov0<0x10D>(v); case 14: return dpp_mov0<0x10E>(v); case 15: return dpp_mov0<0x10F>(v);
        default: return v;
    }
}
template <typename T> __device__ __forceinline__ T row_shr(T v, int k) {
    switch (k) {
        case 1: return dpp_mov0<0x111>(v); case 2: return dpp_mov0<0x112>(v); case 3: return dpp_mov0<0x113>(v);
        case 4: return dpp_mov0<0x114>(v); case 5: return dpp_mov0<0x115>(v); case 6: return dpp_mov0<0x116>(v);
        case 7: return dpp_mov0<0x117>(v); case 8: return dpp_mov0<0x118>(v); case 9: return dpp_mov0<0x119>(v);
        case 10: return dpp_mov0<0x11A>(v); case 11: return dpp_mov0<0x11B>(v); case 12: return dpp_mov0<0x11C>(v);
        case 13: return dpp_mov0<0x11D>(v); case 14: return dpp_mov0<0x11E>(v); case 15: return dpp_mov0<0x11F>(v);
        default: return v;
    }
}
// Value of lane `k` (compile-time) as a wave-uniform scalar; write a scalar into lane `k`.
__device__ __forceinline__ float lane_get(float v, int k) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k));
}
__device__ __forceinline__ double lane_get(double v, int k) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), k),
                            __builtin_amdgcn_readlane(__double2loint(v), k));
}
// Write the wave-uniform scalar `s` into lane `k` (compile-time) of `dst`: one v_writelane_b32 per dword
// (inline assembly: this compiler has the readlane builtin but no writelane one; as a select on the lane
// index the compiler turned the boundary hand-off of the lattice kernel into a branch per step).
// v_writelane_b32 ignores EXEC: call it from wave-uniform control flow only (every caller is the body of an unrolled,
// unconditional step loop).  The lane index is an inline constant when the compiler can prove it constant (the
// unrolled loops of the optimised build) and a select on the lane index otherwise, so unoptimised / partially unrolled builds still assemble.
__device__ __forceinline__ int lane_set_b32(int dst, int s, int k) {
    if (__builtin_constant_p(k)) asm("v_writelane_b32 %0, %1, %2" : "+v"(dst) : "s"(s), "n"(k));
    else dst = (static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))) == k) ? s : dst;   // (select on the lane index)
    return dst;
}
__device__ __forceinline__ float lane_set(float dst, float s, int k) {
    return __int_as_float(lane_set_b32(__float_as_int(dst), __builtin_amdgcn_readfirstlane(__float_as_int(s)), k));
}
__device__ __forceinline__ double lane_set(double dst, double s, int k) {
    const int lo = lane_set_b32(__double2loint(dst), __builtin_amdgcn_readfirstlane(__double2loint(s)), k);
    const int hi = lane_set_b32(__double2hiint(dst), __builtin_amdgcn_readfirstlane(__double2hiint(s)), k);
    return __hiloint2double(hi, lo);
}
// Wave-wide maximum through DPP only (no LDS round trips): quad swaps, row mirrors, then the
// row broadcasts; the result is read from lane 63 and is wave-uniform.
template <typename T> __device__ __forceinline__ T wave_max_dpp(T v) {
    v = vmax(v, dpp_mov<0xB1, 0xf>(v));    // quad_perm [1,0,3,2]
    v = vmax(v, dpp_mov<0x4E, 0xf>(v));    // quad_perm [2,3,0,1]
    v = vmax(v, dpp_mov<0x141, 0xf>(v));   // row_half_mirror
    v = vmax(v, dpp_mov<0x140, 0xf>(v));   // row_mirror
    v = vmax(v, dpp_mov<0x142, 0xa>(v));   // row_bcast:15 -> rows 1,3
    v = vmax(v, dpp_mov<0x143, 0xc>(v));   // row_bcast:31 -> rows 2,3
    return lane_get(v, 63);
}

template <typename T> __device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = vmax(v, __shfl_xor(v, off, kWave));
    return v;
}
template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// Force a wave-uniform value into an SGPR so that address arithmetic built on it is scalar.
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// XCD-aware block order for launches whose logical blocks are (x, y, z) -- gx * gy * gz of them -- and whose gy blocks of
// one (x, z) pair read the SAME operand slab (the label rows' g columns under every time tile of a sample, ...).  HIP hands
// workgroup L of a launch to XCD L % 8, and each XCD has its own 4 MB L2: in the natural (x fastest) order the gy sharers
// land on different XCDs and every one of them fetches the slab from memory again (additive joint, c3 shape: g re-read five
// times, 1.9x the algorithmic bytes of the Z kernel by the memory-side counters).  Here the launch is one-dimensional,
// ceil(gx gz / 8) * 8 * gy blocks (xcd_shared_grid), and block L takes y = (L / 8) % gy of group (L / 8 / gy) * 8 + L % 8:
// the gy sharers of a group are consecutive workgroups of ONE XCD, so all but the first find the slab in that L2.
struct XcdBlock { int x, y, z; bool live; };
__device__ __forceinline__ XcdBlock xcd_shared_y(int gx, int gy, int gz) {
    const unsigned L = blockIdx.x, q = L >> 3;
    const unsigned grp = (q / static_cast<unsigned>(gy)) * 8u + (L & 7u);
    XcdBlock o;
    o.y = static_cast<int>(q % static_cast<unsigned>(gy));
    o.x = static_cast<int>(grp % static_cast<unsigned>(gx));
    o.z = static_cast<int>(grp / static_cast<unsigned>(gx));
    o.live = grp < static_cast<unsigned>(gx) * static_cast<unsigned>(gz);
    return o;
}
inline unsigned xcd_shared_grid(int gx, int gy, int gz) {
    return (static_cast<unsigned>(gx) * static_cast<unsigned>(gz) + 7u) / 8u * 8u * static_cast<unsigned>(gy);
}

}  // namespace rnnt
