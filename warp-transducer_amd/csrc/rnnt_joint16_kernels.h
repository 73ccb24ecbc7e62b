// rnnt_joint16_kernels.h -- the additive-joint GEMMs on the bf16 matrix cores, for 16-bit STORAGE (bf16 or fp16) of f, g,
// df, dg: the operands of the matrix cores are bf16 hi + lo pairs of fp32 values whatever the storage type is (fp16 has
// neither the range for W <= e^40 nor for the sampled references), the storage type only decides how a packet is
// unpacked and how a result is rounded.
//
// rnnt_joint_kernels.h runs the three contractions (Z = Ef Eg^T, DF = Ef .* (W Eg), DG = Eg .* (W^T Ef)) on
// v_mfma_f32_32x32x2_f32, which issues at the fp32 VALU rate and does not overlap with VALU work: with 16-bit
// storage those kernels are bound by instruction issue, not by bytes (EXPERIMENTS.md 8).  Here the operands are
// bf16 and the instruction is v_mfma_f32_32x32x16_bf16 (16x the multiply-accumulates per cycle), so the matrix
// work all but disappears behind the element-wise work (one exp per element of f or g per pass) and the
// kernels stream.  What makes that possible without an LDS transposition:
//   * a lane loads 16 bytes = EIGHT consecutive vocabulary columns of one row, and the MFMA wants eight
//     consecutive CONTRACTION indices per lane.  In Z the contraction runs over the vocabulary: the packet is
//     the fragment.  In DF / DG it runs over label rows / time rows: a lane loads the packets of eight rows,
//     owns an 8 (rows) x 8 (columns) block of exp values in fp32, and -- since two fp32 values have to be
//     packed into one bf16 pair anyway (v_cvt_pk_bf16_f32) -- packs them DOWN the block's columns instead of
//     along its rows: column m of the block is the B fragment of output tile m.  A wavefront therefore owns 32
//     output rows x 256 columns as eight 32 x 32 tiles with columns interleaved by eight, and in the epilogue
//     a lane holds eight ADJACENT columns of a row again: 16-byte loads of f / g and 16-byte stores of df / dg.
//   * precision: every operand is split into hi = bf16(x) and lo = bf16(x - hi) and the product is accumulated
//     as hi*hi + hi*lo + lo*hi in fp32 (relative error ~2^-17 per term: a single bf16 operand, 2^-9, would show
//     in the blank / label columns, where the GEMM term and its correction cancel, and -- in Z -- would walk the
//     lattice off by percents over 170 diagonals when the distributions are peaked).  Three MFMAs per step are
//     still a fraction of the element-wise instructions beside them.
// Same inputs, outputs, masking conventions (+inf sentinel as the row reference of masked rows: exp = 0, no
// select on loaded data) and epilogue corrections as the fp32 kernels; rows are whole 16-byte packets
// (A % 8 == 0, 16-byte aligned tensors), which the host checks.
#pragma once

#include <type_traits>

#include "rnnt_joint_kernels.h"

namespace rnnt {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8_t as_bf16x8(const u32x4& v) {
    bf16x8_t r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}
// Parts of the bf16 expansion of an fp32 operand: NS = 2 (hi + lo, ~2^-17 relative per product: enough for 16-bit
// storage, whose results are rounded to 2^-9 / 2^-12) or NS = 3 (hi + mid + lo, ~2^-25: fp32 storage keeps fp32-class
// results, "fp32 on the bf16 matrix cores").
template <typename Tag> struct JointSplit { static constexpr int NS = sizeof(typename Tag::store) == 4 ? 3 : 2; };

// two fp32 values -> NS packed pairs: their bf16 heads, the bf16 heads of the residuals, ...
template <int NS>
__device__ __forceinline__ void split_pair(float a, float b, uint32_t (&part)[NS]) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        part[i] = cvt_pk_bf16(a, b);
        if (i + 1 < NS) {
            a -= __uint_as_float(part[i] << 16);
            b -= __uint_as_float(part[i] & 0xffff0000u);
        }
    }
}
// One operand of the MFMA as NS fragments (eight contraction indices per lane, four packed pairs each).
template <int NS> struct JointFrag { u32x4 p[NS]; };
template <int NS>
__device__ __forceinline__ void set_pair(JointFrag<NS>& fr, int d, float a, float b) {    // contraction indices 2d, 2d + 1
    uint32_t part[NS];
    split_pair<NS>(a, b, part);
#pragma unroll
    for (int i = 0; i < NS; ++i) fr.p[i][d] = part[i];
}
__device__ __forceinline__ f32x16 mma_bf16(const u32x4& a, const u32x4& b, f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(b), acc, 0, 0, 0);
}
// acc += A B over the products whose order (sum of the two part indices) is below NS: hi*hi, hi*lo, lo*hi (NS = 2; the
// dropped lo*lo is 2^-18 relative) or all six products up to mid*mid (NS = 3; the dropped ones are 2^-26 and below)
template <int NS>
__device__ __forceinline__ f32x16 mma_split(const JointFrag<NS>& a, const JointFrag<NS>& b, f32x16 acc) {
#pragma unroll
    for (int o = 0; o < NS; ++o)                               // smallest terms last would be nicer numerically; fp32
#pragma unroll                                                  // accumulation of <= 6 terms per step does not care
        for (int i = 0; i <= o; ++i) acc = mma_bf16(a.p[i], b.p[o - i], acc);
    return acc;
}

// A lane's packet of NT adjacent columns: one 16-byte or 8-byte access (16-bit storage: NT = 8 -> a wavefront owns 256
// columns as eight tiles, NT = 4 -> 128 columns, half the accumulators and operand registers, two wavefronts per SIMD;
// fp32 storage: NT = 4, 16 bytes).
template <typename Tag, int NT> struct Packet16 {
    using ST = typename Tag::store;
    static constexpr int BYTES = NT * static_cast<int>(sizeof(ST));
    static_assert(BYTES == 16 || BYTES == 8, "a packet is one 16-byte or 8-byte access");
    typedef typename std::conditional<BYTES == 16, uint4, uint2>::type type;
    static __device__ __forceinline__ type load(const ST* p) {
        if constexpr (BYTES == 16) return load_packet<false>(reinterpret_cast<const u32x4*>(p));
        else return *reinterpret_cast<const uint2*>(p);
    }
    static __device__ __forceinline__ void unpack_to(const type& r, float* v) {
        if constexpr (BYTES == 16) unpack<Tag>(r, v); else unpack_half<Tag>(r, v);
    }
    static __device__ __forceinline__ void store(ST* p, const float* v) {
        if constexpr (BYTES == 16) {
            store_packet<false>(reinterpret_cast<u32x4*>(p), pack<Tag>(v));
        } else {
            const float w[8] = {v[0], v[1], v[2], v[3], 0.0f, 0.0f, 0.0f, 0.0f};
            const uint4 q = pack<Tag>(w);
            *reinterpret_cast<uint2*>(p) = make_uint2(q.x, q.y);
        }
    }
    static __device__ __forceinline__ void store_zero(ST* p) {
        if constexpr (BYTES == 16) store_packet<false>(reinterpret_cast<u32x4*>(p), make_uint4(0, 0, 0, 0));
        else *reinterpret_cast<uint2*>(p) = make_uint2(0, 0);
    }
};

// Operands of one contraction step of DF / DG: eight rows per lane half.
template <typename Tag, int NT> struct Joint16Operands { float w[8], m[8]; typename Packet16<Tag, NT>::type x[8]; };

// B fragments of the NT output tiles from eight packets (rows j = 0..7 of the lane's block, NT columns each):
// exp(x - reference) in fp32, split, packed down the columns; one split-MFMA group per tile.
template <typename Tag, int NT>
__device__ __forceinline__ void joint16_mma(const Joint16Operands<Tag, NT>& s, f32x16 (&acc)[NT]) {
    constexpr int NS = JointSplit<Tag>::NS;
    JointFrag<NS> a, bf[NT];
#pragma unroll
    for (int d = 0; d < 4; ++d) set_pair<NS>(a, d, s.w[2 * d], s.w[2 * d + 1]);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        float e0[NT], e1[NT];
        Packet16<Tag, NT>::unpack_to(s.x[2 * d], e0);
        Packet16<Tag, NT>::unpack_to(s.x[2 * d + 1], e1);
#pragma unroll
        for (int m = 0; m < NT; ++m) set_pair<NS>(bf[m], d, joint_exp(e0[m], s.m[2 * d]), joint_exp(e1[m], s.m[2 * d + 1]));
    }
#pragma unroll
    for (int m = 0; m < NT; ++m) acc[m] = mma_split<NS>(a, bf[m], acc[m]);
}

// ------------------------------------------------------------------------------------------
// DG[u,k] = eg[u,k] * sum_t W[t,u] ef[t,k] (joint_dg_kernel's contract).  A wavefront owns 32 label rows x 256
// columns; the contraction runs over the time rows in steps of sixteen (lane half h takes t = t2 + 8h + j):
// A operand = W[t][u0 + col] (coalesced along u), B operand = exp(f[t, k0 + 8 col ..] - mf[t]), the streaming read
// of f as whole 512-byte row segments.  grid = (ceil(A / (128 NT)), ceil(maxU / 32), N), block = 256.
template <typename Tag, int NT, bool PF>
__global__ __launch_bounds__(256, NT == 4 ? 2 : 1) void joint_dg16_kernel(
        const typename Tag::store* __restrict__ f, const typename Tag::store* __restrict__ g, const float* __restrict__ rowmax,
        const float* __restrict__ wmat, const float* __restrict__ scale, const int* __restrict__ xlen,
        const int* __restrict__ ylen, typename Tag::store* __restrict__ dg, int maxT, int maxU, int Upad, int A, int N,
        const int* __restrict__ labels, int blank, const float* __restrict__ sgb, const float* __restrict__ sgl) {
    const XcdBlock blk = xcd_shared_y((A + 128 * NT - 1) / (128 * NT), (maxU + 31) / 32, N);   // (block order: as joint_dg_kernel)
    if (!blk.live) return;
    const int b = blk.z;
    const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6), half = lane >> 5, col = lane & 31;
    using PK = Packet16<Tag, NT>;
    const int k0 = (blk.x * 4 + wave) * (32 * NT);
    if (k0 >= A) return;
    const int kc = k0 + NT * col;                          // first of this lane's NT columns
    const bool kin = kc < A;                               // A % NT == 0: all of them or none
    const int u0 = blk.y * 32;
    const int Tb = clamp_len(xlen[b], maxT), Ub = clamp_len(ylen[b] + 1, maxU);
    const float* mg = rowmax + static_cast<size_t>(N) * maxT + static_cast<size_t>(b) * maxU;
    f32x16 acc[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;

    if (u0 < Ub && Tb > 0) {
        // unconditional operand loads, masking through the +inf sentinel (joint_df_kernel); label rows past the sample
        // read column 0 and produce accumulator rows nobody stores
        const float* wcol = wmat + static_cast<size_t>(b) * maxT * Upad + (u0 + col < Ub ? u0 + col : 0);
        const typename Tag::store* fb = f + static_cast<size_t>(b) * maxT * A + (kin ? kc : A - NT);
        const unsigned Au = static_cast<unsigned>(A), Upu = static_cast<unsigned>(Upad);   // 32-bit offsets (host check)
        const unsigned mf0 = static_cast<unsigned>(b) * maxT;
        const unsigned sentinel = static_cast<unsigned>(N) * (maxT + maxU);
        auto load = [&](Joint16Operands<Tag, NT>& s, int t2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = t2 + 8 * half + j;
                const bool tin = t < Tb;
                const unsigned ts = static_cast<unsigned>(tin ? t : Tb - 1);
                s.w[j] = wcol[ts * Upu];
                s.m[j] = rowmax[tin ? mf0 + t : sentinel];
                s.x[j] = PK::load(fb + ts * Au);
            }
        };
        if constexpr (PF) {
            Joint16Operands<Tag, NT> s0, s1;
            load(s0, 0);
            int t2 = 0;
            while (t2 + 16 < Tb) {
                load(s1, t2 + 16);
                __builtin_amdgcn_sched_barrier(0);
                joint16_mma<Tag, NT>(s0, acc);
                __builtin_amdgcn_sched_barrier(0);
                load(s0, t2 + 32);
                __builtin_amdgcn_sched_barrier(0);
                joint16_mma<Tag, NT>(s1, acc);
                __builtin_amdgcn_sched_barrier(0);
                t2 += 32;
            }
            if (t2 < Tb) joint16_mma<Tag, NT>(s0, acc);
        } else {
            for (int t2 = 0; t2 < Tb; t2 += 16) {
                Joint16Operands<Tag, NT> s0;
                load(s0, t2);
                __builtin_amdgcn_sched_barrier(0);
                joint16_mma<Tag, NT>(s0, acc);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (!kin) return;
    const float sc = scale != nullptr ? scale[b] : 1.0f;   // per-sample factor, applied once per output
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int u = u0 + mfma_row(r, lane);
        if (u >= maxU) continue;
        const size_t at = (static_cast<size_t>(b) * maxU + u) * A + kc;
        float o[NT];
        if (u < Ub) {
            const float mu = mg[u];
            PK::unpack_to(PK::load(g + at), o);
            // corrections without atomics (joint_dg_kernel): row u has one label
            const float cb = sgb[static_cast<size_t>(b) * maxU + u] * sc;
            int lab = -1;
            float cl = 0.0f;
            if (u + 1 < Ub) {
                lab = labels[static_cast<size_t>(b) * (maxU - 1) + u];
                lab = lab < 0 ? 0 : (lab >= A ? A - 1 : lab);
                cl = sgl[static_cast<size_t>(b) * maxU + u] * sc;
            }
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                o[m] = joint_exp(o[m], mu) * (acc[m][r] * sc);
                o[m] -= (kc + m == blank) ? cb : 0.0f;
                o[m] -= (kc + m == lab) ? cl : 0.0f;
            }
        } else {
#pragma unroll
            for (int m = 0; m < NT; ++m) o[m] = 0.0f;
        }
        PK::store(dg + at, o);
    }
}

// ------------------------------------------------------------------------------------------
// DF[t,k] = ef[t,k] * sum_u W[t,u] eg[u,k] - corrections (joint_df_kernel's contract).  A wavefront owns 32 time rows
// x 256 columns; the contraction runs over the label rows in steps of sixteen (lane half h takes u = u2 + 8h + j):
// A operand = W[t0 + col][u ..] (two 16-byte loads of the dense weight row), B operand = exp(g[u, k0 + 8 col ..] - mg[u]).
// Epilogue, in the accumulator registers: times ef (one read of f as 16-byte packets), the blank column's row sum
// (joint_sums_kernel) in the lane that owns it, the label terms df[t, y_u] -= cl(t,u) through one-hot B fragments
// (exact in bf16) against -CL split into hi + lo -- built and issued only for the (step, tile) pairs a label of the
// step falls into --, the per-sample factor, one 16-byte store per row.  No atomics on the output.
// grid = (ceil(A / (128 NT)), ceil(maxT / 32), N), block = 256.
template <typename Tag, int NT, bool PF>
__global__ __launch_bounds__(256, NT == 4 ? 2 : 1) void joint_df16_kernel(
        const typename Tag::store* __restrict__ f, const typename Tag::store* __restrict__ g, const float* __restrict__ rowmax,
        const float* __restrict__ wmat, const float* __restrict__ scale, const int* __restrict__ labels,
        const int* __restrict__ xlen, const int* __restrict__ ylen, typename Tag::store* __restrict__ df, int maxT, int maxU,
        int Upad, int A, int N, int blank, const float* __restrict__ sfb) {
    const XcdBlock blk = xcd_shared_y((A + 128 * NT - 1) / (128 * NT), (maxT + 31) / 32, N);   // (block order: as joint_df_kernel)
    if (!blk.live) return;
    const int b = blk.z;
    const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6), half = lane >> 5, col = lane & 31;
    using PK = Packet16<Tag, NT>;
    const int k0 = (blk.x * 4 + wave) * (32 * NT);
    if (k0 >= A) return;
    const int kc = k0 + NT * col;
    const bool kin = kc < A;
    const int t0 = blk.y * 32;
    const int Tb = clamp_len(xlen[b], maxT), Ub = clamp_len(ylen[b] + 1, maxU);
    const float* mf = rowmax + static_cast<size_t>(b) * maxT;
    const typename Tag::store* fb = f + static_cast<size_t>(b) * maxT * A + (kin ? kc : A - NT);
    typename Tag::store* dfb = df + static_cast<size_t>(b) * maxT * A + kc;
    if (t0 >= Tb || Ub <= 0) {                             // time rows of the padding: zeros
        if (!kin) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = t0 + mfma_row(r, lane);
            if (t < maxT) PK::store_zero(dfb + static_cast<size_t>(t) * A);
        }
        return;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;

    const bool tin = t0 + col < Tb;
    const float sc = scale != nullptr ? scale[b] : 1.0f;
    // unconditional operand loads, nothing selects on a loaded value (joint_df_kernel): label rows past the sample take
    // the +inf sentinel as their reference (B operand exactly 0; W is finite everywhere), a step's second half that
    // would leave the weight row reads its last eight columns instead
    const float* wrow = wmat + (static_cast<size_t>(b) * maxT + (tin ? t0 + col : Tb - 1)) * Upad;
    const typename Tag::store* gb = g + static_cast<size_t>(b) * maxU * A + (kin ? kc : A - NT);
    const unsigned Au = static_cast<unsigned>(A);          // maxU * A < 2^31 (host check): 32-bit offsets
    const unsigned mg0 = static_cast<unsigned>(N) * maxT + static_cast<unsigned>(b) * maxU;
    const unsigned sentinel = static_cast<unsigned>(N) * (maxT + maxU);
    const size_t plane = static_cast<size_t>(N) * maxT * Upad;              // W | CB | CL
    const size_t labs0 = static_cast<size_t>(b) * (maxU > 1 ? maxU - 1 : 1);
    auto load = [&](Joint16Operands<Tag, NT>& s, int u2) {
        const int ub = u2 + 8 * half;
        const int ubc = ub + 8 <= Upad ? ub : Upad - 8;
        const float4 w0 = *reinterpret_cast<const float4*>(wrow + ubc), w1 = *reinterpret_cast<const float4*>(wrow + ubc + 4);
        s.w[0] = w0.x; s.w[1] = w0.y; s.w[2] = w0.z; s.w[3] = w0.w;
        s.w[4] = w1.x; s.w[5] = w1.y; s.w[6] = w1.z; s.w[7] = w1.w;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int u = ub + j;
            const bool uin = u < Ub;
            s.m[j] = rowmax[uin ? mg0 + u : sentinel];
            s.x[j] = PK::load(gb + static_cast<unsigned>(uin ? u : Ub - 1) * Au);
        }
    };
    if constexpr (PF) {
        Joint16Operands<Tag, NT> s0, s1;
        load(s0, 0);
        int u2 = 0;
        while (u2 + 16 < Ub) {
            load(s1, u2 + 16);
            __builtin_amdgcn_sched_barrier(0);
            joint16_mma<Tag, NT>(s0, acc);
            __builtin_amdgcn_sched_barrier(0);
            load(s0, u2 + 32);
            __builtin_amdgcn_sched_barrier(0);
            joint16_mma<Tag, NT>(s1, acc);
            __builtin_amdgcn_sched_barrier(0);
            u2 += 32;
        }
        if (u2 < Ub) joint16_mma<Tag, NT>(s0, acc);
    } else {
        for (int u2 = 0; u2 < Ub; u2 += 16) {
            Joint16Operands<Tag, NT> s0;
            load(s0, u2);
            __builtin_amdgcn_sched_barrier(0);
            joint16_mma<Tag, NT>(s0, acc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue 1: times ef, blank column
    const unsigned dblank = static_cast<unsigned>(blank - kc);             // < NT: the blank column is one of this lane's
#pragma unroll
    for (int r4 = 0; r4 < 16; r4 += 4) {
        typename PK::type fp[4];
        float mt[4], vb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = t0 + mfma_row(r4 + i, lane);
            const int ts = t < Tb ? t : Tb - 1;
            mt[i] = mf[ts];
            fp[i] = PK::load(fb + static_cast<unsigned>(ts) * Au);
            vb[i] = (t < Tb && dblank < static_cast<unsigned>(NT)) ? sfb[static_cast<size_t>(b) * maxT + t] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r4 + i;
            const bool live = t0 + mfma_row(r, lane) < Tb;
            float fv[NT];
            PK::unpack_to(fp[i], fv);
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                float o = live ? joint_exp(fv[m], mt[i]) * acc[m][r] : 0.0f;
                o -= (dblank == static_cast<unsigned>(m)) ? vb[i] : 0.0f;
                acc[m][r] = o;
            }
        }
    }
    // ---- epilogue 2: label terms
    for (int u2 = 0; u2 < Ub - 1; u2 += 16) {
        const int ub = u2 + 8 * half;
        const int ubc = ub + 8 <= Upad ? ub : Upad - 8;
        int d[8];                                              // label of row u = ub + j relative to the lane's first column
        bool any = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int u = ub + j;
            int l = labels[labs0 + (u < maxU - 1 ? u : (maxU > 1 ? maxU - 2 : 0))];
            l = l < 0 ? 0 : (l >= A ? A - 1 : l);
            d[j] = u < Ub - 1 ? l - kc : -1;                   // rows without a label transition never match
            any |= static_cast<unsigned>(d[j]) < static_cast<unsigned>(NT);
        }
        if (__ballot(any) == 0) continue;                      // no label of this step in the wavefront's columns
        const float4 c0 = *reinterpret_cast<const float4*>(wrow + 2 * plane + ubc), c1 = *reinterpret_cast<const float4*>(wrow + 2 * plane + ubc + 4);
        const float cl[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const bool shifted = ubc != ub;                        // (clamped second half: its rows lie past the sample, d = -1)
        constexpr int NS = JointSplit<Tag>::NS;
        JointFrag<NS> ac;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            set_pair<NS>(ac, q, (tin && !shifted) ? -cl[2 * q] : 0.0f, (tin && !shifted) ? -cl[2 * q + 1] : 0.0f);
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            uint32_t one[4];
            bool hit = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool lo = d[2 * q] == m, hi = d[2 * q + 1] == m;
                one[q] = (lo ? 0x3f80u : 0u) | (hi ? 0x3f800000u : 0u);
                hit |= lo || hi;
            }
            if (__ballot(hit) == 0) continue;
            const u32x4 bo = {one[0], one[1], one[2], one[3]};
#pragma unroll
            for (int i = 0; i < NS; ++i) acc[m] = mma_bf16(ac.p[i], bo, acc[m]);
        }
    }
    if (!kin) return;
    // ---- epilogue 3: per-sample factor, store
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int t = t0 + mfma_row(r, lane);
        if (t >= maxT) continue;
        float o[NT];
#pragma unroll
        for (int m = 0; m < NT; ++m) o[m] = acc[m][r] * sc;
        PK::store(dfb + static_cast<size_t>(t) * A, o);
    }
}

// ------------------------------------------------------------------------------------------
// Partition function (joint_z_kernel's contract: relative log Z into `logz`, blank / label log2-probs into `lp2`, skewed
// lattice layout; exact or SAMPLED row references with the guard and the gate).  A wavefront owns a 32 (t) x 32 (u) tile
// and contracts over its share of the vocabulary in chunks of 64 columns (32 for fp32 storage).  The contraction index IS the packet
// direction here: lane (row = lane & 31, half = lane >> 5) loads the 16-byte packets f[t0 + row][k + 8 half ..] and
// g[u0 + row][k + 8 half ..] and they become the A and B fragments as they are (exp, hi / lo split, packed pairwise
// along k) -- no LDS staging, no cross-lane traffic; four packets of each operand per chunk consume a 128-byte line of
// every row.  S wavefronts of a block split the chunks of ONE tile and add their fragments through LDS at the end.
// grid = (tiles or ceil(tiles/4) rounded up to 8, N), block = 64 * max(S, 4).
template <typename Tag, int S, bool SAMPLED>
__global__ __launch_bounds__(S == 1 ? 256 : S * 64) void joint_z16_kernel(
        const typename Tag::store* __restrict__ f, const typename Tag::store* __restrict__ g, float* rowmax,
        const int* __restrict__ labels, const int* __restrict__ xlen, const int* __restrict__ ylen,
        LogPair<float>* __restrict__ lp2, float* __restrict__ logz, int maxT, int maxU, int Up, int A,
        int blank, int tilesU, int tiles, int N, int* gate, int seq, int* __restrict__ poison) {   // poison: note_non_finite (rnnt_kernels.h)
    using ST = typename Tag::store;
    constexpr int WAVES = S == 1 ? 4 : S;
    __shared__ float xch[S == 1 ? 1 : S * 1024];           // S > 1: the wavefronts' fragments meet here
    __shared__ float refs[SAMPLED ? (S == 1 ? WAVES : 1) : 1][64];   // SAMPLED: the tile's 32 + 32 reference values (x log2 e)
    (void)refs; (void)xch;
    if constexpr (!SAMPLED)                                // the exact pass behind a sampled one: only when the gate is raised
        if (gate != nullptr && *gate != seq) return;
    const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6), half = lane >> 5, col = lane & 31;
    int b, group;                                          // (block order: as joint_z_kernel)
    if constexpr (S == 1) {
        b = blockIdx.y;
        group = static_cast<int>((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    } else {
        const XcdBlock o = xcd_shared_y(1, tiles, N);
        if (!o.live) return;
        b = o.z;
        group = o.y;
    }
    const int tile = S == 1 ? group * 4 + wave : group;
    if (tile >= tiles) return;                             // S == 1 only: a whole wavefront leaves
    const int Tb = clamp_len(xlen[b], maxT), Ub = clamp_len(ylen[b] + 1, maxU);
    const int t0 = (tile / tilesU) * 32, u0 = (tile % tilesU) * 32;
    if (t0 >= Tb || u0 >= Ub) return;                      // tile of padding (block-uniform when S > 1)
    const float* mf = rowmax + static_cast<size_t>(b) * maxT;
    const float* mg = rowmax + static_cast<size_t>(N) * maxT + static_cast<size_t>(b) * maxU;

    // operand role: this lane's row of f and of g (rows past the sample: any valid row, their cells are never stored)
    const int trow = t0 + col < Tb ? t0 + col : Tb - 1, urow = u0 + col < Ub ? u0 + col : Ub - 1;
    const ST* frow = f + (static_cast<size_t>(b) * maxT + trow) * A + 8 * half;
    const ST* grow = g + (static_cast<size_t>(b) * maxU + urow) * A + 8 * half;
    // eight consecutive columns of a row = one contraction half-step of this lane: PPS 16-byte packets
    constexpr int EPP = 16 / static_cast<int>(sizeof(ST)), PPS = 8 / EPP;     // elements per packet (8 | 4), packets per 8 columns (1 | 2)
    auto load8 = [&](const ST* p8, float* v) {
#pragma unroll
        for (int q = 0; q < PPS; ++q) unpack<Tag>(load_packet<false>(reinterpret_cast<const u32x4*>(p8 + q * EPP)), v + q * EPP);
    };
    float mfr, mgr;
    float tf = neg_inf<float>(), tg = neg_inf<float>();   // SAMPLED: the true maxima of what this lane streams
    if constexpr (!SAMPLED) {
        mfr = mf[trow];
        mgr = mg[urow];
    } else {
        // reference of a row = maximum of its first 32 columns (the two lane halves hold 16 each), clamped to a finite
        // value; S > 1: the first wavefront reads them, the others take the values from LDS (one block barrier)
        constexpr int RW = S == 1 ? WAVES : 1;
        const int rw = S == 1 ? wave : 0;
        if (S == 1 || wave == 0) {
            float a[8], c[8], a2[8], c2[8];
            load8(frow, a); load8(frow + 16, a2);
            load8(grow, c); load8(grow + 16, c2);
            float ma = neg_inf<float>(), mc = neg_inf<float>();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                ma = fmaxf(ma, fmaxf(a[i], a2[i]));
                mc = fmaxf(mc, fmaxf(c[i], c2[i]));
            }
            ma = fmaxf(ma, __shfl_xor(ma, 32));
            mc = fmaxf(mc, __shfl_xor(mc, 32));
            ma = fmaxf(ma, kJointMinMax) * static_cast<float>(kLog2e);
            mc = fmaxf(mc, kJointMinMax) * static_cast<float>(kLog2e);
            if (half == 0) {                               // one copy per row: for the block, and the arrays for the later kernels
                refs[rw % RW][col] = ma;
                refs[rw % RW][32 + col] = mc;
                const int t = t0 + col, u = u0 + col;
                if (u0 == 0 && t < Tb) rowmax[static_cast<size_t>(b) * maxT + t] = ma;
                if (t0 == 0 && u < Ub) rowmax[static_cast<size_t>(N) * maxT + static_cast<size_t>(b) * maxU + u] = mc;
            }
        }
        if constexpr (S == 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
        mfr = refs[rw % RW][col];
        mgr = refs[rw % RW][32 + col];
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    // a chunk = STEPS contraction steps of 16 columns = four 16-byte packets of each operand per lane (64 columns of
    // 16-bit storage, 32 of fp32)
    constexpr int STEPS = 4 / PPS, CH = 16 * STEPS;
    constexpr int NS = JointSplit<Tag>::NS;
    const int nchunk = (A + CH - 1) / CH;
    auto load = [&](uint4 (&fv)[4], uint4 (&gv)[4], int cc) {
#pragma unroll
        for (int j = 0; j < STEPS; ++j) {
            const int k = cc * CH + 16 * j + 8 * half;     // (columns past the end: the row's last eight, cancelled below)
            const int ko = k < A ? cc * CH + 16 * j : A - 8 - 8 * half;
#pragma unroll
            for (int q = 0; q < PPS; ++q) {
                fv[j * PPS + q] = load_packet<false>(reinterpret_cast<const u32x4*>(frow + ko + q * EPP));
                gv[j * PPS + q] = load_packet<false>(reinterpret_cast<const u32x4*>(grow + ko + q * EPP));
            }
        }
    };
    auto compute = [&](const uint4 (&fv)[4], const uint4 (&gv)[4], int cc) {
        const float pinf = -neg_inf<float>();
#pragma unroll
        for (int j = 0; j < STEPS; ++j) {
            const bool in = cc * CH + 16 * j + 8 * half < A;
            const float ra = in ? mfr : pinf, rb = in ? mgr : pinf;
            float x[8], y[8];
#pragma unroll
            for (int q = 0; q < PPS; ++q) {
                unpack<Tag>(fv[j * PPS + q], x + q * EPP);
                unpack<Tag>(gv[j * PPS + q], y + q * EPP);
            }
            if constexpr (SAMPLED) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { tf = fmaxf(tf, x[i]); tg = fmaxf(tg, y[i]); }
            }
            JointFrag<NS> fa, fb2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                set_pair<NS>(fa, q, joint_exp(x[2 * q], ra), joint_exp(x[2 * q + 1], ra));
                set_pair<NS>(fb2, q, joint_exp(y[2 * q], rb), joint_exp(y[2 * q + 1], rb));
            }
            acc = mma_split<NS>(fa, fb2, acc);
        }
    };
    {
        uint4 f0[4], g0[4], f1[4], g1[4];
        int c = S == 1 ? 0 : wave;
        load(f0, g0, c);
        while (c + S < nchunk) {                           // two chunks per trip, both unconditional (joint_z_kernel)
            load(f1, g1, c + S);
            __builtin_amdgcn_sched_barrier(0);
            compute(f0, g0, c);
            __builtin_amdgcn_sched_barrier(0);
            load(f0, g0, c + 2 * S);                       // past the end: reads the last packet, unused
            __builtin_amdgcn_sched_barrier(0);
            compute(f1, g1, c + S);
            __builtin_amdgcn_sched_barrier(0);
            c += 2 * S;
        }
        if (c < nchunk) compute(f0, g0, c);
    }

    if constexpr (SAMPLED) {
        // the guard: true maximum of every row this wavefront streamed against its reference
        const float a = fmaxf(tf, __shfl_xor(tf, 32)), c = fmaxf(tg, __shfl_xor(tg, 32));
        const bool trip = !(a * static_cast<float>(kLog2e) - mfr <= kJointGuard) || !(c * static_cast<float>(kLog2e) - mgr <= kJointGuard);
        if (__ballot(trip) != 0 && lane == 0) *gate = seq;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // refs[wave][..] written above, read below
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    auto ref_f = [&](int row, int t) -> float { if constexpr (SAMPLED) { (void)t; return refs[S == 1 ? wave : 0][row]; } else { (void)row; return mf[t]; } };
    auto ref_g = [&](int colx, int uu) -> float { if constexpr (SAMPLED) { (void)uu; return refs[S == 1 ? wave : 0][32 + colx]; } else { (void)colx; return mg[uu]; } };
    // per-lane constants of the epilogue: this lane's label row u = u0 + col
    const int u = u0 + col;
    const int ui = u < Ub ? u : Ub - 1;
    const bool has_lab = u < Ub - 1;
    int lab = blank;
    if (has_lab) {
        lab = labels[static_cast<size_t>(b) * (maxU - 1) + u];
        lab = lab < 0 ? 0 : (lab >= A ? A - 1 : lab);
    }
    const ST* gu = g + (static_cast<size_t>(b) * maxU + ui) * A;
    const float mgu = ref_g(col, ui);
    const float l2e = static_cast<float>(kLog2e), ln2 = static_cast<float>(kLn2);
    const float gbl = __builtin_fmaf(load1<Tag>(gu + blank), l2e, -mgu), glab = __builtin_fmaf(load1<Tag>(gu + lab), l2e, -mgu);   // base 2

    constexpr int PER = S == 1 ? 16 : 16 / S;              // fragment registers finished by this wavefront
    const int rbase = S == 1 ? 0 : wave * PER;
    float fbl[PER], flb[PER], mtv[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int t = t0 + mfma_row(rbase + i, lane);
        const int tc = t < Tb ? t : Tb - 1;
        const ST* ft = f + (static_cast<size_t>(b) * maxT + tc) * A;
        fbl[i] = load1<Tag>(ft + blank);
        flb[i] = load1<Tag>(ft + lab);
        mtv[i] = ref_f(mfma_row(rbase + i, lane), tc);
    }

    auto finish = [&](int i, float z) {                    // i-th register of this wavefront's share
        const int r = rbase + i;
        const int t = t0 + mfma_row(r, lane);
        const bool valid = t < Tb && u < Ub;
        float lz = acc_log(z);
        // cells whose two rows peak far apart: direct log-sum-exp over the vocabulary (joint_z_kernel)
        unsigned long long bad = __ballot(valid && !(z >= kJointFlagZ));
        while (bad) {
            const int src = __ffsll(static_cast<long long>(bad)) - 1;
            bad &= bad - 1;
            const int tt = t0 + mfma_row(r, src), uu = u0 + (src & 31);
            const ST* fr = f + (static_cast<size_t>(b) * maxT + tt) * A;
            const ST* gr = g + (static_cast<size_t>(b) * maxU + uu) * A;
            float m = neg_inf<float>();
            for (int k = lane; k < A; k += 64) m = fmaxf(m, load1<Tag>(fr + k) + load1<Tag>(gr + k));
            m = fmaxf(wave_max(m), kJointMinMax);
            float s = 0.0f;
            for (int k = lane; k < A; k += 64) s += fast_exp(load1<Tag>(fr + k) + load1<Tag>(gr + k) - m);
            s = wave_sum(s);
            const float v = (m - (ref_f(mfma_row(r, src), tt) + ref_g(src & 31, uu)) * ln2) + acc_log(s);
            if (lane == src) lz = v;
        }
        if (!valid) return;
        LogPair<float> rec;                               // lattice log-probs are kept in base 2
        const float lz2 = lz * l2e;
        rec.x = fmaxf(__builtin_fmaf(fbl[i], l2e, -mtv[i]) + gbl - lz2, log_zero<float>());
        rec.y = has_lab ? fmaxf(__builtin_fmaf(flb[i], l2e, -mtv[i]) + glab - lz2, log_zero<float>())
                        : log_zero<float>();
        const size_t idx = lat_index(b, t + u, u, maxT, maxU, Up);
        lp2[lat_pair_index(b, t + u, u, maxT, maxU, Up)] = rec;
        logz[idx] = lz;
        note_non_finite(poison, b, t + u, u, Up, lz);
    };

    if constexpr (S == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) finish(r, acc[r]);
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[wave * 1024 + r * 64 + lane] = acc[r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int r = rbase + i;
            float z = 0.0f;
#pragma unroll
            for (int s = 0; s < S; ++s) z += xch[s * 1024 + r * 64 + lane];
            finish(i, z);
        }
    }
}

}  // namespace rnnt
