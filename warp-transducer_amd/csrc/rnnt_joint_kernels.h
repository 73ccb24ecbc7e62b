// rnnt_joint_kernels.h -- additive-joint ("add network") fusion, SURVEY.md 8f rank 1.
//
// The reference's add_network branch (README.md:4, docs/rnnt_notes.tex:56-59,147-153,
// pytorch_binding/test/test_time.py:51-77) takes the transcription output f (N,T,A) and the
// prediction output g (N,U,A) of Graves' 2012 transducer, whose joint is
//     h(k,t,u) = f[t,k] + g[u,k],      Pr(k|t,u) = softmax_k h(k,t,u),
// and returns dL/df = sum_u dL/dh, dL/dg = sum_t dL/dh.  The (N,T,U,A) tensor is never formed,
// and neither are its T*U*A exponentials: an ADDITIVE joint factorises,
//     sum_k exp(f[t,k] + g[u,k]) = sum_k ef[t,k] * eg[u,k],
//         ef[t,k] = exp(f[t,k] - mf[t]),  eg[u,k] = exp(g[u,k] - mg[u]),  mf / mg = row maxima,
// so the partition function and both gradients are three small GEMMs per sample,
//     Z  (T x U) = Ef  Eg^T                           logZ(t,u) = mf[t] + mg[u] + log Z[t,u]
//     DF (T x A) = Ef .* (W  Eg)   - corrections      W[t,u] = exp(alpha + beta - ll) / Z[t,u]
//     DG (U x A) = Eg .* (W^T Ef)  - corrections
// with (T+U)*A exponentials per pass instead of T*U*A.  The contractions run on the fp32 matrix
// cores (v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fma chain, subnormals kept), operands go
// as 16-byte vectors.  Kernels:
//   joint_rowmax_kernel  mf, mg                                              (wave per row)
//   joint_z_kernel       Z tiles + blank / label log-probs into the skewed lattice arrays
//   (lattice_kernel, coef_kernel of rnnt_kernels.h run unchanged: the lattice array `logz` holds
//    the RELATIVE value log Z[t,u], which makes the coefficient record's c equal to log W[t,u],
//    and coef_kernel also writes the dense matrix W)
//   joint_df_kernel / joint_dg_kernel   the two gradient GEMMs with the exp(f) / exp(g) epilogue
//   joint_sums_kernel    row / column sums of the blank and label corrections (small fp32 side vectors), subtracted
//                        in the epilogues of the two GEMMs: no atomics on the outputs
//   joint_far_kernel     the far cells (rare), after the GEMMs
//
// Range: ef, eg are in (0,1], so Z[t,u] >= exp(-(separation of the two rows' peaks)).  A cell
// whose GEMM sum falls below kJointFlagZ (the rows peak at different symbols, > 41 nats apart) is
// recomputed in place with a direct log-sum-exp over k, and a cell whose log W exceeds kJointFarC
// is left out of the gradient GEMMs and added by joint_far_kernel from exp(f + g + c) directly,
// so any finite logit range is handled exactly; ordinary inputs never take either branch.
// Storage of f, g, df, dg: fp32, bf16 or fp16 (template tag); every kernel computes in fp32 (fp32 MFMA).
#pragma once

#include "rnnt_kernels.h"

namespace rnnt {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr float kJointFlagZ = 0x1p-60f;   // GEMM sums below this are recomputed directly
// kJointFarC (rnnt_kernels.h): log W above this -> the cell is handled by joint_far_kernel
constexpr float kJointMinMax = -3.0e38f;  // row maxima are clamped to a finite value

// C/D fragment of the 32x32 MFMA: register r of lane l holds row (r&3) + 8*(r>>2) + 4*(l>>5),
// column l&31.  A operand: lane l = A[row l&31][k = l>>5]; B operand: lane l = B[k = l>>5][col l&31].
__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// exp(x - m) with the row maximum kept as m2 = m * log2(e): one fma and one v_exp_f32.  Every
// kernel uses this same form, so the shift each row actually gets (m2 / log2 e, equal to m up to
// rounding) is consistent across Z, W and the gradients.  x = -inf or m2 = +inf -> 0.
__device__ __forceinline__ float joint_exp(float x, float m2) {
    return fast_exp2(__builtin_fmaf(x, static_cast<float>(kLog2e), -m2));
}

// ------------------------------------------------------------------------------------------
// Row maxima, stored TIMES log2(e).  rowmax[0, N*maxT) = mf, rowmax[N*maxT, N*(maxT+maxU)) = mg,
// rowmax[N*(maxT+maxU)] = +inf (sentinel).
// WPR wavefronts share a row (0: eight LANES per row, rows of at most 64 symbols; 1: wavefront per row; 4: the whole block, for rows >= 12 KB, so that the
// rows in flight form one contiguous window -- see row_stats_block_kernel); rows of the padding
// (t >= T_b, u > U_b) are skipped.  grid = ceil(rows * WPR / 4) (WPR = 0: ceil(rows / 32)), block = 256.
template <typename Tag, bool VEC, int WPR>
__global__ __launch_bounds__(256) void joint_rowmax_kernel(
        const typename Tag::store* __restrict__ f, const typename Tag::store* __restrict__ g, const int* __restrict__ xlen,
        const int* __restrict__ ylen, float* __restrict__ rowmax, int maxT, int maxU, int A, int N,
        float* __restrict__ side, unsigned nside, const int* __restrict__ gate, int seq) {
    __shared__ float red[4];
    // second, exact pass of the sampled-reference forward (joint_z_kernel, SAMPLED): only if a row tripped the guard
    if (gate != nullptr && *gate != seq) return;
    // the correction sums of joint_sums_kernel (N*(maxT + 2 maxU) floats + N flags) start at zero: the grid has at
    // least 8 threads per row of f and g, more than that many words
    {
        const unsigned long long gid = static_cast<unsigned long long>(blockIdx.x) * 256 + threadIdx.x;
        if (side != nullptr && gid < nside) side[gid] = 0.0f;
    }
    // one extra entry after the maxima holds +inf: operand loads of the gradient GEMMs point masked
    // rows at it, which makes their exp() exactly 0 without a select on loaded data
    if (blockIdx.x == 0 && threadIdx.x == 0) rowmax[static_cast<size_t>(N) * (maxT + maxU)] = -neg_inf<float>();
    if constexpr (WPR == 0) {
        // short rows (A <= 64): EIGHT lanes per row, eight rows per wavefront -- with a wavefront per row the c4 shape of the
        // additive joint (115 264 rows of 50 symbols) was a launch of 115 264 wavefronts with one load each, 27 us
        const long long row = static_cast<long long>(blockIdx.x) * 32 + (threadIdx.x >> 3);
        const long long rows_f = static_cast<long long>(N) * maxT;
        const int sub = threadIdx.x & 7;
        bool live = row < rows_f + static_cast<long long>(N) * maxU;
        const typename Tag::store* p = f;
        if (live) {
            if (row < rows_f) {
                const int b = static_cast<int>(row / maxT);
                live = static_cast<int>(row - static_cast<long long>(b) * maxT) < xlen[b];
                p = f + row * A;
            } else {
                const long long r = row - rows_f;
                const int b = static_cast<int>(r / maxU);
                live = static_cast<int>(r - static_cast<long long>(b) * maxU) <= ylen[b];
                p = g + r * A;
            }
        }
        float m = neg_inf<float>();
        if (live)
            for (int i = sub; i < A; i += 8) m = fmaxf(m, load1<Tag>(p + i));
        m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
        if (live && sub == 0) rowmax[row] = fmaxf(m, kJointMinMax) * static_cast<float>(kLog2e);
        return;
    }
    constexpr int TPR = 64 * (WPR > 0 ? WPR : 1);          // threads per row
    const int wave = threadIdx.x >> 6;
    const long long row = WPR == 4 ? static_cast<long long>(blockIdx.x)
                                   : static_cast<long long>(blockIdx.x) * 4 + wave;
    const long long rows_f = static_cast<long long>(N) * maxT;
    if (row >= rows_f + static_cast<long long>(N) * maxU) return;
    const int tid = WPR == 4 ? static_cast<int>(threadIdx.x) : static_cast<int>(threadIdx.x & 63);
    const typename Tag::store* p;
    if (row < rows_f) {
        const int b = static_cast<int>(row / maxT);
        if (static_cast<int>(row - static_cast<long long>(b) * maxT) >= xlen[b]) return;   // uniform per row
        p = f + row * A;
    } else {
        const long long r = row - rows_f;
        const int b = static_cast<int>(r / maxU);
        if (static_cast<int>(r - static_cast<long long>(b) * maxU) > ylen[b]) return;
        p = g + r * A;
    }
    float m = neg_inf<float>();
    if constexpr (VEC) {                                   // rows are whole 16-byte packets
        constexpr int VN = Vec<Tag>::N;
        const u32x4* p4 = reinterpret_cast<const u32x4*>(p);
        for (int i = tid; i < A / VN; i += TPR) {
            float v[VN];
            unpack<Tag>(load_packet<true>(p4 + i), v);
#pragma unroll
            for (int j = 0; j < VN; ++j) m = fmaxf(m, v[j]);
        }
    } else {
        for (int i = tid; i < A; i += TPR) m = fmaxf(m, load1<Tag>(p + i));
    }
    m = wave_max(m);
    if constexpr (WPR == 4) {
        if ((threadIdx.x & 63) == 0) red[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    }
    if (tid == 0) rowmax[row] = fmaxf(m, kJointMinMax) * static_cast<float>(kLog2e);
}

// Up to four consecutive elements as ONE access (the address is NK-element aligned): 4/8/16 bytes for fp32
// storage, 2/4/8 bytes for 16-bit storage; values in fp32.
template <typename Tag, int NK>
__device__ __forceinline__ void joint_loadv(const typename Tag::store* __restrict__ p, float (&v)[NK]) {
    if constexpr (sizeof(typename Tag::store) == 4) {
        if constexpr (NK == 4) {
            const float4 t = *reinterpret_cast<const float4*>(p);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else if constexpr (NK == 2) {
            const float2 t = *reinterpret_cast<const float2*>(p);
            v[0] = t.x; v[1] = t.y;
        } else {
            v[0] = p[0];
        }
    } else {
        if constexpr (NK == 4) {
            unpack_half<Tag>(*reinterpret_cast<const uint2*>(p), v);
        } else if constexpr (NK == 2) {
            float w[4];
            unpack_half<Tag>(make_uint2(*reinterpret_cast<const uint32_t*>(p), 0u), w);
            v[0] = w[0]; v[1] = w[1];
        } else {
            v[0] = load1<Tag>(p);
        }
    }
}
template <typename Tag, int NK>
__device__ __forceinline__ void joint_storev(typename Tag::store* __restrict__ p, const float (&v)[NK]) {
    if constexpr (sizeof(typename Tag::store) == 4) {
        if constexpr (NK == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        else if constexpr (NK == 2) *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
        else p[0] = v[0];
    } else {
        if constexpr (NK == 1) {
            store1<Tag>(p, v[0]);
        } else {
            float w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < NK; ++i) w[i] = v[i];
            const uint4 q = pack<Tag>(w);                  // two values per 32-bit word
            if constexpr (NK == 4) *reinterpret_cast<uint2*>(p) = make_uint2(q.x, q.y);
            else *reinterpret_cast<uint32_t*>(p) = q.x;
        }
    }
}

// Four consecutive columns k..k+3 of a row, loaded UNCONDITIONALLY: a column past the end is read from a
// valid one instead (VEC: the whole 4-element packet from the last packet, A % 4 == 0 and rows aligned to
// it; scalar form: each element from column A-1) and the CALLER cancels it (row maximum +inf ->
// exp(x - inf) = 0).  With guards instead, every scalar load sat in its own exec-masked branch.
template <typename Tag, bool VEC>
__device__ __forceinline__ float4 joint_load4(const typename Tag::store* __restrict__ row, int k, int A) {
    if constexpr (VEC) {
        float v[4];
        joint_loadv<Tag, 4>(row + (k < A ? k : A - 4), v);
        return make_float4(v[0], v[1], v[2], v[3]);
    } else {
        float4 v;
        v.x = load1<Tag>(row + (k < A ? k : A - 1));
        v.y = load1<Tag>(row + (k + 1 < A ? k + 1 : A - 1));
        v.z = load1<Tag>(row + (k + 2 < A ? k + 2 : A - 1));
        v.w = load1<Tag>(row + (k + 3 < A ? k + 3 : A - 1));
        return v;
    }
}

// ------------------------------------------------------------------------------------------
// Partition function.  A wavefront owns a 32 (t) x 32 (u) tile of Z and contracts over its share
// of the vocabulary in chunks of 32 columns.  Per chunk it reads the 32 x 32 pieces of f and g as
// full 128-byte row segments (eight lanes per row, eight rows per 16-byte load instruction), applies
// exp(x - rowmax) and parks the two pieces in its PRIVATE slice of LDS (row stride 36 floats); the
// MFMA operands come back as ds_read_b128 in fragment order (lane = row, the two lane halves take
// the two 16-column halves of the chunk: the k order inside a contraction is free as long as A and
// B agree).  No block barrier is involved: LDS operations of one wavefront execute in order.  The
// next chunk's global loads are in flight while the current one is in the matrix core.
// S wavefronts of a block split the chunks of ONE tile (large vocabulary, few tiles) and add their
// fragments through LDS at the end; with S = 1 the four wavefronts of a block own four tiles.
// Epilogue per cell: log Z -> logz (relative), blank / label log2-probs -> lp2, both in the skewed
// lattice layout.  grid = (tiles or ceil(tiles/4), N), block = 64 * max(S, 4).
// Sampled reference values (SAMPLED): the row-maximum pass is a full read of f and g (c3 shape: 438 MB, 66 us of a 0.6 ms
// step) whose only purpose is a safe exponent reference per row.  Any reference R with max - R <= kJointGuard (base 2)
// is as good: exp2((x - R) log2 e) <= 2^40, products of two <= 2^80, a sum over <= 2^23 symbols < 2^127; smaller elements
// underflow no earlier than with the exact maximum.  So the Z kernel takes R = the maximum of the row's FIRST 32 columns
// (every tile and every wavefront that needs row t derives the same value from the same 128 bytes), tracks the true
// maximum of what it streams anyway, and raises a gate word when some row exceeds its reference by more than the guard
// (or has no finite reference at all).  The exact pair -- joint_rowmax_kernel, then joint_z_kernel without SAMPLED -- is
// enqueued behind it in every call and returns at once unless the gate is raised: ordinary logits never raise it,
// arbitrary finite (and -inf-masked) logits are still handled exactly.  The gate holds the call's sequence number
// when raised and is reset by joint_prep_kernel, which also takes over the row-maximum kernel's housekeeping (zeroed
// correction sums, the +inf sentinel).
constexpr float kJointGuard = 40.0f;
template <int UNUSED = 0>                            // (a template so that the translation units of the three storage types share one definition)
static __global__ __launch_bounds__(256) void joint_prep_kernel(float* __restrict__ rowmax, float* __restrict__ side, unsigned nside,
                                                         int* __restrict__ gate, int seq, unsigned sentinel) {
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    if (side != nullptr && gid < nside) side[gid] = 0.0f;
    if (gid == 0) {
        rowmax[sentinel] = -neg_inf<float>();
        *gate = ~seq;
    }
}

constexpr int kJointZPad = 36;                              // LDS row stride in floats (conflict-free b128)
constexpr int kJointZSlice = 2 * 32 * kJointZPad;           // floats per wavefront: ef piece + eg piece

template <typename Tag, int S, bool VEC, bool SAMPLED = false>
__global__ __launch_bounds__(S == 1 ? 256 : S * 64) void joint_z_kernel(
        const typename Tag::store* __restrict__ f, const typename Tag::store* __restrict__ g, float* rowmax,
        const int* __restrict__ labels, const int* __restrict__ xlen, const int* __restrict__ ylen,
        LogPair<float>* __restrict__ lp2, float* __restrict__ logz, int maxT, int maxU, int Up, int A,
        int blank, int tilesU, int tiles, int N, int* gate, int seq, int* __restrict__ poison) {   // poison: note_non_finite (rnnt_kernels.h)
    constexpr int WAVES = S == 1 ? 4 : S;
    __shared__ float4 stage4[WAVES * kJointZSlice / 4];
    __shared__ float refs[SAMPLED ? (S == 1 ? WAVES : 1) : 1][64];   // SAMPLED: the tile's 32 + 32 reference values (x log2 e)
    (void)refs;
    float* stage = reinterpret_cast<float*>(stage4);
    if constexpr (!SAMPLED)                                // the exact pass behind a sampled one: only when the gate is raised
        if (gate != nullptr && *gate != seq) return;
    const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6), half = lane >> 5, col = lane & 31;
    // S == 1 (many tiles): XCD-aware order -- workgroup i runs on XCD i % 8, so every XCD is given one
    // contiguous range of tile groups and the partial lines of the skewed arrays that neighbouring tiles
    // write meet in one L2 (gridDim.x is a multiple of 8 then; see row_stats_tile_kernel)
    // S > 1 (one tile per block, few tiles): the tiles of ONE sample share its g rows (and, across label tiles, its f rows):
    // a one-dimensional launch in xcd_shared_y order keeps them on one XCD (rnnt_device.h)
    int b, group;
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

    // loader role: lane l moves columns 4*(l&7).. of rows (l>>3) + 8i, i = 0..3
    const int lrow = lane >> 3, lcol = (lane & 7) * 4;
    using ST = typename Tag::store;
    const ST* frow[4];
    const ST* grow4[4];
    float mfr[4], mgr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = lrow + 8 * i;
        const int t = t0 + r < Tb ? t0 + r : Tb - 1;      // rows past the sample: any valid row
        const int u = u0 + r < Ub ? u0 + r : Ub - 1;
        frow[i] = f + (static_cast<size_t>(b) * maxT + t) * A;
        grow4[i] = g + (static_cast<size_t>(b) * maxU + u) * A;
        if constexpr (!SAMPLED) {
            mfr[i] = mf[t];
            mgr[i] = mg[u];
        }
    }
    float tf[4], tg[4];                                    // SAMPLED: the true maxima of what this lane streams
    if constexpr (SAMPLED) {
        // reference of a row = maximum of its first 32 columns (eight lanes hold four each), clamped to a finite value;
        // S > 1: the wavefronts of the block share one tile -- the first one reads the 32 columns, the others take the
        // values from LDS (one block barrier)
        constexpr int RW = S == 1 ? WAVES : 1;             // copies of the reference table
        const int rw = S == 1 ? wave : 0;
        if (S == 1 || wave == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 a = joint_load4<Tag, VEC>(frow[i], lcol, A), c = joint_load4<Tag, VEC>(grow4[i], lcol, A);
                float ma = fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), mc = fmaxf(fmaxf(c.x, c.y), fmaxf(c.z, c.w));
#pragma unroll
                for (int d = 1; d < 8; d <<= 1) {
                    ma = fmaxf(ma, __shfl_xor(ma, d));
                    mc = fmaxf(mc, __shfl_xor(mc, d));
                }
                ma = fmaxf(ma, kJointMinMax) * static_cast<float>(kLog2e);
                mc = fmaxf(mc, kJointMinMax) * static_cast<float>(kLog2e);
                if ((lane & 7) == 0) {                     // one copy per row: for the block, and the arrays for the later kernels
                    refs[rw % RW][lrow + 8 * i] = ma;
                    refs[rw % RW][32 + lrow + 8 * i] = mc;
                    const int t = t0 + lrow + 8 * i, u = u0 + lrow + 8 * i;
                    if (u0 == 0 && t < Tb) rowmax[static_cast<size_t>(b) * maxT + t] = ma;
                    if (t0 == 0 && u < Ub) rowmax[static_cast<size_t>(N) * maxT + static_cast<size_t>(b) * maxU + u] = mc;
                }
            }
        }
        if constexpr (S == 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mfr[i] = refs[rw % RW][lrow + 8 * i];
            mgr[i] = refs[rw % RW][32 + lrow + 8 * i];
            tf[i] = neg_inf<float>();
            tg[i] = neg_inf<float>();
        }
    }
    (void)tf; (void)tg;
    float* fs = stage + wave * kJointZSlice;               // [32][kJointZPad]
    float* gs = fs + 32 * kJointZPad;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int nchunk = (A + 31) >> 5;
    auto load = [&](float4 (&fv)[4], float4 (&gv)[4], int cc) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fv[i] = joint_load4<Tag, VEC>(frow[i], cc * 32 + lcol, A);
            gv[i] = joint_load4<Tag, VEC>(grow4[i], cc * 32 + lcol, A);
        }
    };
    auto track = [&](const float4 (&fv)[4], const float4 (&gv)[4]) {     // (columns past the end repeat valid ones)
        if constexpr (SAMPLED) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                tf[i] = fmaxf(fmaxf(tf[i], fmaxf(fv[i].x, fv[i].y)), fmaxf(fv[i].z, fv[i].w));
                tg[i] = fmaxf(fmaxf(tg[i], fmaxf(gv[i].x, gv[i].y)), fmaxf(gv[i].z, gv[i].w));
            }
        }
    };
    auto compute = [&](const float4 (&fv)[4], const float4 (&gv)[4], int cc) {
        track(fv, gv);
        // columns past the end were read elsewhere: cancel them through the maximum (VEC: the whole packet)
        const int k = cc * 32 + lcol;
        const float pinf = -neg_inf<float>();
        const bool in0 = k < A, in1 = VEC ? in0 : k + 1 < A, in2 = VEC ? in0 : k + 2 < A, in3 = VEC ? in0 : k + 3 < A;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a = mfr[i], d = mgr[i];
            float4 e, h;
            e.x = joint_exp(fv[i].x, in0 ? a : pinf); e.y = joint_exp(fv[i].y, in1 ? a : pinf);
            e.z = joint_exp(fv[i].z, in2 ? a : pinf); e.w = joint_exp(fv[i].w, in3 ? a : pinf);
            h.x = joint_exp(gv[i].x, in0 ? d : pinf); h.y = joint_exp(gv[i].y, in1 ? d : pinf);
            h.z = joint_exp(gv[i].z, in2 ? d : pinf); h.w = joint_exp(gv[i].w, in3 ? d : pinf);
            *reinterpret_cast<float4*>(fs + (lrow + 8 * i) * kJointZPad + lcol) = e;
            *reinterpret_cast<float4*>(gs + (lrow + 8 * i) * kJointZPad + lcol) = h;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a4 = *reinterpret_cast<const float4*>(fs + col * kJointZPad + 16 * half + 4 * q);
            const float4 b4 = *reinterpret_cast<const float4*>(gs + col * kJointZPad + 16 * half + 4 * q);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // two register sets in ping-pong (a plain "current = next" copy would wait for the loads it
    // has just issued): while one chunk is computed, the loads of the following one are in flight
    {
        float4 f0[4], g0[4], f1[4], g1[4];
        int c = S == 1 ? 0 : wave;
        load(f0, g0, c);
        while (c + S < nchunk) {                           // two chunks per trip, both unconditional: a
            load(f1, g1, c + S);                           // branch around a compute lets the compiler sink
            __builtin_amdgcn_sched_barrier(0);             // that chunk's loads into it
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
        bool trip = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float a = tf[i], c = tg[i];
#pragma unroll
            for (int d = 1; d < 8; d <<= 1) {
                a = fmaxf(a, __shfl_xor(a, d));
                c = fmaxf(c, __shfl_xor(c, d));
            }
            trip |= !(a * static_cast<float>(kLog2e) - mfr[i] <= kJointGuard) || !(c * static_cast<float>(kLog2e) - mgr[i] <= kJointGuard);
        }
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

    // The per-cell gathers f[t,blank], f[t,label] and mf[t] of the registers this wavefront finishes are
    // requested together, BEFORE the cell loop: inside it they were sixteen dependent memory round trips
    // (the ballot / recompute loop between two cells keeps the compiler from batching them) -- most of a
    // tile's time when the vocabulary is small.
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
        // cells whose two rows peak far apart: direct log-sum-exp over the vocabulary, one cell at a
        // time by the whole wavefront (never taken for ordinary logits)
        unsigned long long bad = __ballot(valid && !(z >= kJointFlagZ));
        while (bad) {
            const int src = __ffsll(static_cast<long long>(bad)) - 1;
            bad &= bad - 1;
            const int tt = t0 + mfma_row(r, src), uu = u0 + (src & 31);
            const typename Tag::store* fr = f + (static_cast<size_t>(b) * maxT + tt) * A;
            const typename Tag::store* gr = g + (static_cast<size_t>(b) * maxU + uu) * A;
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
        // the wavefront's own (now idle) LDS slice carries its fragment to the others
#pragma unroll
        for (int r = 0; r < 16; ++r) fs[r * 64 + lane] = acc[r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int r = rbase + i;
            float z = 0.0f;
#pragma unroll
            for (int s = 0; s < S; ++s) z += stage[s * kJointZSlice + r * 64 + lane];
            finish(i, z);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Partition function, small vocabularies (A <= kJointZSmallA).  joint_z_kernel is bound there by the
// ADDRESS rate of the vector L1 and by instruction issue, not by bytes (PMC on the c4 shape, A = 50:
// 0.93 cache accesses per CU and cycle -- 4-byte operand loads that touch every 64-byte piece of a row
// four times, one write request per lattice cell because consecutive u of one t are Up+1 elements
// apart in the skewed arrays -- and ~1400 vector instructions per tile, most of them addresses).
// Here a wavefront reads the 32 rows of f and of g its tile needs as ONE flat run each (rows of a sample
// are adjacent in memory: 32*A consecutive floats, 256 contiguous bytes per buffer load, rows past the
// sample come back as zeros from the descriptor's range check; all loads are requested before the first
// is used) and keeps the RAW logits of the whole vocabulary in its private LDS slice (row stride AS odd:
// conflict-free column reads).  exp(x - rowmax) is applied when the MFMA operands are read back (lane =
// row, so the maximum is a per-lane constant), the blank / label logits of the epilogue come from the
// same slice instead of global gathers, and the finished tile is turned through LDS so that the stores
// run ALONG the anti-diagonals -- the direction in which the skewed arrays are contiguous (two diagonals
// per store instruction).
// grid = (ceil(tiles/4) rounded up to 8, N), block = 256, dynamic LDS = 4 * kJointZSmallSlice floats.
constexpr int kJointZSmallA = 56;
constexpr int kJointZSmallIt = kJointZSmallA / 2;          // flat loads per lane and operand (32 rows * A / 64)
constexpr int kJointZOutPad = 34;                          // row stride of the turned tile (diagonal reads hit distinct banks)
constexpr int kJointZSmallOp = 32 * (kJointZSmallA + 1);   // floats per operand
constexpr int kJointZSmallSlice = 2 * kJointZSmallOp + 64; // per wavefront at the LARGEST vocabulary; 4 slices < 64 KB
static_assert(kJointZSmallSlice >= 3 * 32 * kJointZOutPad && 4 * kJointZSmallSlice * 4 <= 65536, "LDS budget");
// The slice for THIS vocabulary: both raw operands at row stride A | 1 plus the spare words, and never less than the turned
// output tile.  Sized at run time because the block count per CU hangs on it: 4 x 59 KB allows two blocks, at A <= 50 the
// 4 x 13.3 KB of the real rows allow THREE (the kernel needs 148 registers: three wavefronts per SIMD fit).
__host__ __device__ inline int joint_z_small_slice(int A) {
    const int need = 2 * 32 * (A | 1) + 64, out = 3 * 32 * kJointZOutPad;
    return ((need > out ? need : out) + 3) & ~3;
}

template <typename Tag>
__global__ __launch_bounds__(256) void joint_z_small_kernel(
        const typename Tag::store* __restrict__ f, const typename Tag::store* __restrict__ g, const float* __restrict__ rowmax,
        const int* __restrict__ labels, const int* __restrict__ xlen, const int* __restrict__ ylen,
        LogPair<float>* __restrict__ lp2, float* __restrict__ logz, int maxT, int maxU, int Up, int A,
        int blank, int tilesU, int tiles, int N, int* __restrict__ poison, int slice) {   // slice: joint_z_small_slice(A) floats of LDS per wavefront
    extern __shared__ float4 zsmall4[];
    constexpr int IT = kJointZSmallIt;
    const int AS = A | 1;                                  // LDS row stride
    const int b = blockIdx.y;
    // the wavefront index as a SCALAR: everything derived from it (tile, row ranges, buffer descriptors)
    // stays in SGPRs -- as a function of threadIdx the compiler treats it as divergent and wraps every
    // buffer load in a waterfall loop
    const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
    const int half = lane >> 5, col = lane & 31;
    const int group = static_cast<int>((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3));   // XCD-aware, as joint_z_kernel
    const int tile = group * 4 + wave;
    if (tile >= tiles) return;
    const int Tb = clamp_len(xlen[b], maxT), Ub = clamp_len(ylen[b] + 1, maxU);
    const int t0 = (tile / tilesU) * 32, u0 = (tile % tilesU) * 32;
    if (t0 >= Tb || u0 >= Ub) return;
    const float* mf = rowmax + static_cast<size_t>(b) * maxT;
    const float* mg = rowmax + static_cast<size_t>(N) * maxT + static_cast<size_t>(b) * maxU;
    const float* sentinel = rowmax + static_cast<size_t>(N) * (maxT + maxU);      // +inf: exp(x - inf) = 0
    float* fs = reinterpret_cast<float*>(zsmall4) + wave * slice;                 // [32][AS] raw f
    float* gs = fs + 32 * AS;                                                     // [32][AS] raw g
    float* spare = gs + 32 * AS;

    // ---- every global load of the tile is requested here, before anything waits
    const int nf = (Tb - t0 < 32 ? Tb - t0 : 32) * A, ng = (Ub - u0 < 32 ? Ub - u0 : 32) * A;   // valid flat elements
    using ST = typename Tag::store;
    constexpr int ES = static_cast<int>(sizeof(ST));
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<ST*>(f) + (static_cast<size_t>(b) * maxT + t0) * A, 0, nf * ES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<ST*>(g) + (static_cast<size_t>(b) * maxU + u0) * A, 0, ng * ES, 0x00020000);
    float fv[IT], gv[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {                         // past the valid rows: 0
        if constexpr (ES == 4) {
            fv[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rf, lane * 4, i * 256, 0));
            gv[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, lane * 4, i * 256, 0));
        } else {                                           // one 16-bit element per lane, 128 contiguous bytes per load
            const uint16_t hf = static_cast<uint16_t>(__builtin_amdgcn_raw_buffer_load_b16(rf, lane * 2, i * 128, 0));
            const uint16_t hg = static_cast<uint16_t>(__builtin_amdgcn_raw_buffer_load_b16(rg, lane * 2, i * 128, 0));
            fv[i] = load1<Tag>(&hf);
            gv[i] = load1<Tag>(&hg);
        }
    }
    const float ma = *(t0 + col < Tb ? mf + t0 + col : sentinel);   // this lane's row of f as MFMA operand
    const float mb = *(u0 + col < Ub ? mg + u0 + col : sentinel);   // ... of g; also the label row of its cells
    const int u = u0 + col;
    const bool has_lab = u < Ub - 1;
    int lab = blank;
    if (has_lab) lab = labels[static_cast<size_t>(b) * (maxU - 1) + u];
    __builtin_amdgcn_sched_barrier(0);

    // ---- raw logits into LDS: element e = 64 i + lane of the flat run is (row e / A, column e % A)
    {
        int r = lane / A, k = lane - r * A;
        const int dr = 64 / A, dk = 64 - dr * A;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const bool in = r < 32;                        // branch-free: elements past the tile go to a spare word
            const int off = r * AS + k;
            (in ? fs + off : spare)[0] = fv[i];
            (in ? gs + off : spare)[0] = gv[i];
            k += dk; r += dr;
            const bool wrap = k >= A;
            k -= wrap ? A : 0;
            r += wrap ? 1 : 0;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- contraction: lane half h takes columns 2s + h; four steps per group, the next group's LDS
    //      reads are issued before the current group's exp / MFMA
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    {
        const float* fr = fs + col * AS + half;
        const float* gr = gs + col * AS + half;
        const int ngroup = (A + 7) >> 3;
        auto rd = [&](float (&x)[4], float (&y)[4], int j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { x[i] = fr[8 * j + 2 * i]; y[i] = gr[8 * j + 2 * i]; }
        };
        auto mm = [&](const float (&x)[4], const float (&y)[4], int j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool kin = 8 * j + 2 * i + half < A; // a column past the end reads the next row: cancel it
                const float a = kin ? joint_exp(x[i], ma) : 0.0f;
                const float d = kin ? joint_exp(y[i], mb) : 0.0f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, d, acc, 0, 0, 0);
            }
        };
        float x0[4], y0[4], x1[4], y1[4];
        rd(x0, y0, 0);
        int j = 0;
        while (j + 1 < ngroup) {
            rd(x1, y1, j + 1);
            mm(x0, y0, j);
            rd(x0, y0, j + 2);                             // past the last group: inside the slice, unused
            mm(x1, y1, j + 1);
            j += 2;
        }
        if (j < ngroup) mm(x0, y0, j);
    }

    // ---- epilogue: blank / label logits from the slice, the three values of every cell to LDS in tile order ...
    const float l2e = static_cast<float>(kLog2e), ln2 = static_cast<float>(kLn2);
    lab = lab < 0 ? 0 : (lab >= A ? A - 1 : lab);
    const float gbl = __builtin_fmaf(gs[col * AS + blank], l2e, -mb);     // base 2
    const float glab = __builtin_fmaf(gs[col * AS + lab], l2e, -mb);
    float fbl[16], flb[16], mtv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int tl = mfma_row(i, lane);
        fbl[i] = fs[tl * AS + blank];
        flb[i] = fs[tl * AS + lab];
        mtv[i] = __shfl(ma, tl);                           // lane tl holds the maximum of row tl
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                       // the logits are dead: the slice is reused
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float* px = fs;
    float* py = fs + 32 * kJointZOutPad;
    float* pz = fs + 64 * kJointZOutPad;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int tl = mfma_row(i, lane);
        const bool valid = t0 + tl < Tb && u < Ub;
        const float z = acc[i];
        float lz = acc_log(z);
        // rows peaking far apart: direct log-sum-exp, as in joint_z_kernel (never taken for ordinary logits)
        unsigned long long bad = __ballot(valid && !(z >= kJointFlagZ));
        while (bad) {
            const int src = __ffsll(static_cast<long long>(bad)) - 1;
            bad &= bad - 1;
            const int tt = t0 + mfma_row(i, src), uu = u0 + (src & 31);
            const typename Tag::store* fr = f + (static_cast<size_t>(b) * maxT + tt) * A;
            const typename Tag::store* gr = g + (static_cast<size_t>(b) * maxU + uu) * A;
            float m = neg_inf<float>();
            for (int k = lane; k < A; k += 64) m = fmaxf(m, load1<Tag>(fr + k) + load1<Tag>(gr + k));
            m = fmaxf(wave_max(m), kJointMinMax);
            float sum = 0.0f;
            for (int k = lane; k < A; k += 64) sum += fast_exp(load1<Tag>(fr + k) + load1<Tag>(gr + k) - m);
            sum = wave_sum(sum);
            const float v = (m - (mf[tt] + mg[uu]) * ln2) + acc_log(sum);
            if (lane == src) lz = v;
        }
        const float lz2 = lz * l2e;                        // lattice log-probs are kept in base 2
        px[tl * kJointZOutPad + col] = fmaxf(__builtin_fmaf(fbl[i], l2e, -mtv[i]) + gbl - lz2, log_zero<float>());
        py[tl * kJointZOutPad + col] = has_lab ? fmaxf(__builtin_fmaf(flb[i], l2e, -mtv[i]) + glab - lz2, log_zero<float>())
                                               : log_zero<float>();
        pz[tl * kJointZOutPad + col] = lz;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ... and leave along the anti-diagonals: diagonal d of the tile is cells (t0 + d - j, u0 + j), adjacent in j
    const size_t base = lat_index(b, t0 + u0, u0, maxT, maxU, Up), pbase = lat_pair_index(b, t0 + u0, u0, maxT, maxU, Up);
#pragma unroll 4
    for (int it = 0; it < 32; ++it) {
        const int d = 2 * it + half, tl = d - col;
        if (tl >= 0 && tl < 32 && t0 + tl < Tb && u < Ub) {
            const size_t idx = base + static_cast<size_t>(d) * Up + col;
            LogPair<float> rec;
            rec.x = px[tl * kJointZOutPad + col];
            rec.y = py[tl * kJointZOutPad + col];
            lp2[pbase + static_cast<size_t>(d) * Up + col] = rec;
            const float lz = pz[tl * kJointZOutPad + col];
            logz[idx] = lz;
            note_non_finite(poison, b, t0 + u0 + d, col + u0, Up, lz);
        }
    }
}

// ------------------------------------------------------------------------------------------
// DF[t,k] = ef[t,k] * sum_u W[t,u] eg[u,k].  A wavefront owns 32 time rows x 32*NK columns; lane
// `col` holds the NK ADJACENT columns k0 + NK*col + n (accumulator n), so every access to f, g and
// df is one 4*NK-byte vector per lane and a half-wavefront covers 128*NK contiguous bytes of a row.
// The contraction over the label rows runs in steps of eight: lane half h takes u = u2 + 4h + i,
// A operand = W[t0+col][u] (one 16-byte load of the dense weight row gives four MFMA steps), B
// operand = exp(g[u,k] - mg[u]).  Two operand sets alternate so the loads of step s+1 are in
// flight while step s is in the matrix core, and the f values of the epilogue (one read of f) are
// requested before the loop starts.  Epilogue: multiply by ef, store; rows of the padding are
// written as zeros.  The four wavefronts of a block take adjacent column groups of the same time
// rows.  grid = (ceil(A / (128 NK)), ceil(maxT/32), N), block = 256.
template <int NK> struct JointOperands { float w[4], m[4], x[4][NK], cb[4], cl[4]; int lab[4]; };   // (DG uses w, m, x only)

template <int NK>
__device__ __forceinline__ void joint_mma(const JointOperands<NK>& s, f32x16 (&acc)[NK]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < NK; ++n)
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(s.w[i], joint_exp(s.x[i][n], s.m[i]), acc[n], 0, 0, 0);
}

// SPLIT (small vocabularies): when the whole vocabulary fits one or two column groups, three of a block's four wavefronts
// have no columns of their own -- and the one that works runs the WHOLE contraction alone, a chain of dependent operand
// round trips (c4 shape, A = 50: DF 204 us with 3008 working wavefronts, DG 131 us with 640).  In this mode the four
// wavefronts share ONE column group and each takes every fourth step of the contraction; the partial accumulators meet in
// LDS (one wavefront's worth at a time, deterministic order 0 + 1 + 2 + 3) and wavefront 0 runs the epilogue.
// grid.x = ceil(A / (32 NK)).
template <int NV>
__device__ __forceinline__ void joint_reduce_waves(f32x16 (&acc)[NV], float* __restrict__ red, int wave, int lane) {
    for (int w = 1; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int n = 0; n < NV; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(n * 16 + r) * 64 + lane] = acc[n][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int n = 0; n < NV; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[n][r] += red[(n * 16 + r) * 64 + lane];
        }
        __syncthreads();
    }
}

// OH (small vocabularies): the blank / label corrections of df are accumulated here as well,
//     acc2 += CB[t][u] * [k == blank] + CL[t][u] * [k == y_u]       (two more MFMAs per step, one-hot B operands),
// from the dense CB / CL planes the coefficient kernel writes next to W.  It replaces one global atomic per
// lattice cell (the first version's fix-up kernel), which all land on the few cache lines of a short df row (c4 shape,
// A = 50: 340 us of atomics against ~100 us of extra matrix work); above a few hundred symbols the atomics
// are cheaper than the 3x contraction and the host keeps them.
template <typename Tag, int NK, bool PF, bool OH, bool SPLIT = false, bool BS = false>   // PF: operand ping-pong + epilogue values requested before the loop (more registers)
                                                                       // BS (with OH): the blank corrections come from the row sums sfb, no CB operand (no CB plane exists)
__global__ __launch_bounds__(256) void joint_df_kernel(
        const typename Tag::store* __restrict__ f, const typename Tag::store* __restrict__ g, const float* __restrict__ rowmax,
        const float* __restrict__ wmat, const float* __restrict__ scale, const int* __restrict__ labels,
        const int* __restrict__ xlen, const int* __restrict__ ylen, typename Tag::store* __restrict__ df, int maxT, int maxU,
        int Upad, int A, int N, int blank, const float* __restrict__ sfb) {
    __shared__ float red[SPLIT ? NK * 16 * 64 : 1];
    (void)red;
    // block order: the time tiles of one (column group, sample) share the g columns -- one XCD, consecutive (xcd_shared_y)
    const XcdBlock blk = xcd_shared_y((A + (SPLIT ? 32 : 128) * NK - 1) / ((SPLIT ? 32 : 128) * NK), (maxT + 31) / 32, N);
    if (!blk.live) return;
    const int b = blk.z;
    const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6), half = lane >> 5, col = lane & 31;
    const int k0 = SPLIT ? blk.x * (32 * NK) : (blk.x * 4 + wave) * (32 * NK);
    if (k0 >= A) return;                                   // (SPLIT: block-uniform)
    const int kc = k0 + NK * col;                          // first of this lane's NK columns
    const bool kin = kc < A;                               // A % NK == 0: all NK columns or none
    const int t0 = blk.y * 32;
    const int Tb = clamp_len(xlen[b], maxT), Ub = clamp_len(ylen[b] + 1, maxU);
    const float* mf = rowmax + static_cast<size_t>(b) * maxT;
    using ST = typename Tag::store;
    const ST* fb = f + static_cast<size_t>(b) * maxT * A + (kin ? kc : A - NK);   // lanes past the vocabulary read a valid column
    ST* dfb = df + static_cast<size_t>(b) * maxT * A + kc;
    f32x16 acc[NK];
#pragma unroll
    for (int n = 0; n < NK; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    f32x16 acc2[OH ? NK : 1];                              // corrections (OH only)
#pragma unroll
    for (int n = 0; n < (OH ? NK : 1); ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[n][r] = 0.0f;

    if (t0 >= Tb) {                                        // time rows of the padding: zeros
        if (!kin || (SPLIT && wave != 0)) return;
        float z[NK];
#pragma unroll
        for (int n = 0; n < NK; ++n) z[n] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = t0 + mfma_row(r, lane);
            if (t < maxT) joint_storev<Tag, NK>(dfb + static_cast<size_t>(t) * A, z);
        }
        return;
    }

    const bool tin = t0 + col < Tb;
    // per-sample factor (grad_output / N): applied in the epilogue -- a multiply on the W operand inside
    // the contraction loop sits in front of every MFMA group and cost 25 % of the kernel
    const float sc = scale != nullptr ? scale[b] : 1.0f;
    // Operand loads are UNCONDITIONAL and nothing selects on a loaded value (a select lets the compiler
    // turn the load into a branch with a full vmcnt(0) drain inside the prefetch phase): label rows past
    // the sample read the +inf sentinel as their maximum, so their B operand is exactly 0 (W is finite:
    // zero in the padding, zero in the pad columns, <= e^40 elsewhere); time rows past the sample and
    // lanes past the vocabulary compute on a valid neighbour and store nothing.
    const float* wrow = wmat + (static_cast<size_t>(b) * maxT + (tin ? t0 + col : Tb - 1)) * Upad;
    const ST* gb = g + static_cast<size_t>(b) * maxU * A + (kin ? kc : A - NK);
    const unsigned Au = static_cast<unsigned>(A);          // maxU * A < 2^31 (host check): 32-bit offsets
    const unsigned mg0 = static_cast<unsigned>(N) * maxT + static_cast<unsigned>(b) * maxU;
    const unsigned sentinel = static_cast<unsigned>(N) * (maxT + maxU);
    const size_t plane = static_cast<size_t>(N) * maxT * Upad;              // W | CB | CL
    const size_t labs0 = static_cast<size_t>(b) * (maxU > 1 ? maxU - 1 : 1);
    auto load = [&](JointOperands<NK>& s, int u2) {
        const int ub = u2 + 4 * half;                      // ub + 3 < Upad: Upad is a multiple of 8 and u2 < Ub
        const float4 w4 = *reinterpret_cast<const float4*>(wrow + ub);
        s.w[0] = w4.x; s.w[1] = w4.y; s.w[2] = w4.z; s.w[3] = w4.w;
        if constexpr (OH) {                                // zero outside the sample (padding, pad columns): no masks
            if constexpr (!BS) {
                const float4 b4 = *reinterpret_cast<const float4*>(wrow + plane + ub);
                s.cb[0] = b4.x; s.cb[1] = b4.y; s.cb[2] = b4.z; s.cb[3] = b4.w;
            }
            const float4 l4 = *reinterpret_cast<const float4*>(wrow + 2 * plane + ub);
            s.cl[0] = l4.x; s.cl[1] = l4.y; s.cl[2] = l4.z; s.cl[3] = l4.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = ub + i;
            const bool uin = u < Ub;
            s.m[i] = rowmax[uin ? mg0 + u : sentinel];
            joint_loadv<Tag, NK>(gb + static_cast<unsigned>(uin ? u : Ub - 1) * Au, s.x[i]);
            if constexpr (OH) s.lab[i] = labels[labs0 + (u < maxU - 1 ? u : (maxU > 1 ? maxU - 2 : 0))];
        }
    };
    // OH (small vocabularies: nearly every label lies in the wavefront's columns): the corrections of df ride along in
    // the contraction as one-hot B operands,  acc2 += CB[t][u] * [k == blank] + CL[t][u] * [k == y_u].
    auto corr = [&](const JointOperands<NK>& s) {
        if constexpr (OH) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int lab = s.lab[i] < 0 ? 0 : (s.lab[i] >= A ? A - 1 : s.lab[i]);
#pragma unroll
                for (int n = 0; n < NK; ++n) {
                    if constexpr (!BS) acc2[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(s.cb[i], kc + n == blank ? 1.0f : 0.0f, acc2[n], 0, 0, 0);
                    acc2[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(s.cl[i], kc + n == lab ? 1.0f : 0.0f, acc2[n], 0, 0, 0);
                }
            }
        }
    };

    float fv[16][NK], mt[16];
    auto load_f = [&]() {                                  // the epilogue's f values and row maxima
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = t0 + mfma_row(r, lane);
            const int ts = t < Tb ? t : Tb - 1;
            mt[r] = mf[ts];
            joint_loadv<Tag, NK>(fb + static_cast<unsigned>(ts) * Au, fv[r]);
        }
    };
    if constexpr (SPLIT) {
        // this wavefront's steps of the contraction: u2 = 8 (wave + 4 j); operand ping-pong as below
        constexpr int ST = 32;
        int u2 = 8 * wave;
        JointOperands<NK> s0, s1;
        if (u2 < Ub) load(s0, u2);
        if (wave == 0) load_f();                           // the epilogue is wavefront 0's
        __builtin_amdgcn_sched_barrier(0);
        while (u2 + ST < Ub) {
            load(s1, u2 + ST);
            __builtin_amdgcn_sched_barrier(0);
            joint_mma<NK>(s0, acc); corr(s0);
            __builtin_amdgcn_sched_barrier(0);
            if (u2 + 2 * ST < Ub) load(s0, u2 + 2 * ST);   // (wave-uniform)
            __builtin_amdgcn_sched_barrier(0);
            joint_mma<NK>(s1, acc); corr(s1);
            __builtin_amdgcn_sched_barrier(0);
            u2 += 2 * ST;
        }
        if (u2 < Ub) { joint_mma<NK>(s0, acc); corr(s0); }
        joint_reduce_waves<NK>(acc, red, wave, lane);
        if constexpr (OH) joint_reduce_waves<NK>(acc2, red, wave, lane);
        if (wave != 0) return;
    } else if constexpr (PF) {
        JointOperands<NK> s0, s1;
        load(s0, 0);
        load_f();                                          // requested now, consumed after the contraction
        __builtin_amdgcn_sched_barrier(0);
        int u2 = 0;
        while (u2 + 8 < Ub) {                              // two steps per trip, both unconditional (see joint_z_kernel)
            load(s1, u2 + 8);
            __builtin_amdgcn_sched_barrier(0);
            joint_mma<NK>(s0, acc); corr(s0);
            __builtin_amdgcn_sched_barrier(0);
            load(s0, u2 + 16);                             // past the sample: no loads, zero weights
            __builtin_amdgcn_sched_barrier(0);
            joint_mma<NK>(s1, acc); corr(s1);
            __builtin_amdgcn_sched_barrier(0);
            u2 += 16;
        }
        if (u2 < Ub) { joint_mma<NK>(s0, acc); corr(s0); }
    } else {
        for (int u2 = 0; u2 < Ub; u2 += 8) {
            JointOperands<NK> s0;
            load(s0, u2);
            __builtin_amdgcn_sched_barrier(0);             // all loads of the step first, then its MFMAs
            joint_mma<NK>(s0, acc); corr(s0);
            __builtin_amdgcn_sched_barrier(0);
        }
        load_f();
    }

    // ---- epilogue: df = ef * (W Eg) - corrections, finished IN the accumulator registers, then stored.  No atomics
    //      on the output.  OH: the corrections are in acc2.  Otherwise:
    //        df[t, y_u]   -= cl(t,u)         once the accumulators hold ef * (W Eg), the same one-hot MFMA with -CL as
    //                                        its A operand subtracts straight into them -- issued only for a label that
    //                                        falls into THIS wavefront's 128 NK columns (a ballot per label: about
    //                                        U * 128 NK / A of them hit), so the contraction loop above is untouched;
    //        df[t, blank] -= sum_u cb(t,u)   the row sum from joint_sums_kernel (sfb), in the lane that owns the column,
    //                                        folded into the scaling loop (as a loop of its own BEHIND the label
    //                                        MFMAs it sent the register allocator of the NK = 2 instantiation to
    //                                        256 VGPRs + 256 AGPRs + scratch, from 130 + 32).
    const unsigned dblank = static_cast<unsigned>(blank - kc);            // < NK: the blank column is one of this lane's
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int t = t0 + mfma_row(r, lane);
        const bool live = t < Tb;
        float vb = 0.0f;                                                  // (not OH, or BS) the blank column's row sum
        if constexpr (!OH || BS)
            if (live && dblank < static_cast<unsigned>(NK)) vb = sfb[static_cast<size_t>(b) * maxT + t];
#pragma unroll
        for (int n = 0; n < NK; ++n) {
            float o = live ? joint_exp(fv[r][n], mt[r]) * acc[n][r] : 0.0f;
            if constexpr (OH) o = live ? o - acc2[n][r] : 0.0f;
            if constexpr (!OH || BS) o -= (dblank == static_cast<unsigned>(n)) ? vb : 0.0f;
            acc[n][r] = o;
        }
    }
    if constexpr (!OH) {
        // (requesting these operands BEFORE the contraction loop hid their round trip but cost 32 live registers there:
        // c3 shape 186 -> 252 us for this kernel -- it is that sensitive to occupancy; they stay here)
        constexpr int CH = 4;                              // steps (of eight label rows) whose operands are requested together
        for (int u0c = 0; u0c < Ub; u0c += 8 * CH) {
            float cl[CH][4];
            int lab[CH][4];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                // (steps past the sample read the last step's operands again -- in range, Upad is a multiple of 8 -- and
                // are skipped below; a lane whose A-operand row lies past the sample reads row T_b - 1: forced to zero)
                const int u2 = u0c + 8 * j < Ub ? u0c + 8 * j : ((Ub - 1) & ~7);
                const int ub = u2 + 4 * half;
                const float4 l4 = *reinterpret_cast<const float4*>(wrow + 2 * plane + ub);   // zero outside the sample
                cl[j][0] = tin ? l4.x : 0.0f; cl[j][1] = tin ? l4.y : 0.0f; cl[j][2] = tin ? l4.z : 0.0f; cl[j][3] = tin ? l4.w : 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int u = ub + i;
                    const int l = labels[labs0 + (u < maxU - 1 ? u : (maxU > 1 ? maxU - 2 : 0))];
                    lab[j][i] = l < 0 ? 0 : (l >= A ? A - 1 : l);
                }
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                if (u0c + 8 * j >= Ub) break;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // the lane's label is the one-hot position of ITS half's contraction slot; a step is issued when either
                    // half's label lies in the wavefront's columns (cl of a row without a label transition is zero)
                    if (__ballot(static_cast<unsigned>(lab[j][i] - k0) < static_cast<unsigned>(32 * NK)) != 0) {
#pragma unroll
                        for (int n = 0; n < NK; ++n)
                            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(-cl[j][i], kc + n == lab[j][i] ? 1.0f : 0.0f, acc[n], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (!kin) return;
#pragma unroll
    for (int n = 0; n < NK; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] *= sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int t = t0 + mfma_row(r, lane);
        if (t >= maxT) continue;
        float o[NK];
#pragma unroll
        for (int n = 0; n < NK; ++n) o[n] = acc[n][r];
        joint_storev<Tag, NK>(dfb + static_cast<size_t>(t) * A, o);
    }
}

// DG[u,k] = eg[u,k] * sum_t W[t,u] ef[t,k]: the same with the roles of f and g exchanged.  The
// contraction runs over the time rows in steps of eight (lane half h takes t = t2 + 4h + i):
// A operand = W[t][u0+col] (coalesced along u), B operand = exp(f[t,k] - mf[t]), the streaming
// read of f, again with two alternating operand sets.
// grid = (ceil(A / (128 NK)), ceil(maxU/32), N), block = 256.
template <typename Tag, int NK, bool PF, bool SPLIT = false>      // SPLIT: see joint_df_kernel
__global__ __launch_bounds__(256) void joint_dg_kernel(
        const typename Tag::store* __restrict__ f, const typename Tag::store* __restrict__ g, const float* __restrict__ rowmax,
        const float* __restrict__ wmat, const float* __restrict__ scale, const int* __restrict__ xlen,
        const int* __restrict__ ylen, typename Tag::store* __restrict__ dg, int maxT, int maxU, int Upad, int A, int N,
        const int* __restrict__ labels, int blank, const float* __restrict__ sgb, const float* __restrict__ sgl) {
    __shared__ float red[SPLIT ? NK * 16 * 64 : 1];
    (void)red;
    // block order: the label tiles of one (column group, sample) share the f columns -- one XCD, consecutive (xcd_shared_y)
    const XcdBlock blk = xcd_shared_y((A + (SPLIT ? 32 : 128) * NK - 1) / ((SPLIT ? 32 : 128) * NK), (maxU + 31) / 32, N);
    if (!blk.live) return;
    const int b = blk.z;
    const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6), half = lane >> 5, col = lane & 31;
    const int k0 = SPLIT ? blk.x * (32 * NK) : (blk.x * 4 + wave) * (32 * NK);
    if (k0 >= A) return;                                   // (SPLIT: block-uniform)
    const int kc = k0 + NK * col;
    const bool kin = kc < A;
    const int u0 = blk.y * 32;
    const int Tb = clamp_len(xlen[b], maxT), Ub = clamp_len(ylen[b] + 1, maxU);
    const float* mg = rowmax + static_cast<size_t>(N) * maxT + static_cast<size_t>(b) * maxU;
    f32x16 acc[NK];
#pragma unroll
    for (int n = 0; n < NK; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;

    if (u0 < Ub) {
        // unconditional operand loads, masking through the +inf sentinel (see joint_df_kernel); label
        // rows past the sample read column 0 and produce accumulator rows nobody stores
        const float* wcol = wmat + static_cast<size_t>(b) * maxT * Upad + (u0 + col < Ub ? u0 + col : 0);
        const typename Tag::store* fb = f + static_cast<size_t>(b) * maxT * A + (kin ? kc : A - NK);
        const unsigned Au = static_cast<unsigned>(A), Upu = static_cast<unsigned>(Upad);   // 32-bit offsets (host check)
        const unsigned mf0 = static_cast<unsigned>(b) * maxT;
        const unsigned sentinel = static_cast<unsigned>(N) * (maxT + maxU);
        auto load = [&](JointOperands<NK>& s, int t2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = t2 + 4 * half + i;
                const bool tin = t < Tb;
                const unsigned ts = static_cast<unsigned>(tin ? t : Tb - 1);
                s.w[i] = wcol[ts * Upu];
                s.m[i] = rowmax[tin ? mf0 + t : sentinel];
                joint_loadv<Tag, NK>(fb + ts * Au, s.x[i]);
            }
        };
        if constexpr (SPLIT) {
            constexpr int ST = 32;                             // this wavefront's steps: t2 = 8 (wave + 4 j)
            int t2 = 8 * wave;
            JointOperands<NK> s0, s1;
            if (t2 < Tb) load(s0, t2);
            while (t2 + ST < Tb) {
                load(s1, t2 + ST);
                __builtin_amdgcn_sched_barrier(0);
                joint_mma<NK>(s0, acc);
                __builtin_amdgcn_sched_barrier(0);
                if (t2 + 2 * ST < Tb) load(s0, t2 + 2 * ST);
                __builtin_amdgcn_sched_barrier(0);
                joint_mma<NK>(s1, acc);
                __builtin_amdgcn_sched_barrier(0);
                t2 += 2 * ST;
            }
            if (t2 < Tb) joint_mma<NK>(s0, acc);
        } else if constexpr (PF) {
            JointOperands<NK> s0, s1;
            load(s0, 0);
            int t2 = 0;
            while (t2 + 8 < Tb) {
                load(s1, t2 + 8);
                __builtin_amdgcn_sched_barrier(0);
                joint_mma<NK>(s0, acc);
                __builtin_amdgcn_sched_barrier(0);
                load(s0, t2 + 16);
                __builtin_amdgcn_sched_barrier(0);
                joint_mma<NK>(s1, acc);
                __builtin_amdgcn_sched_barrier(0);
                t2 += 16;
            }
            if (t2 < Tb) joint_mma<NK>(s0, acc);
        } else {
            for (int t2 = 0; t2 < Tb; t2 += 16) {          // sixteen time rows per trip, all loads first
                JointOperands<NK> s0, s1;
                load(s0, t2);
                load(s1, t2 + 8);
                joint_mma<NK>(s0, acc);
                joint_mma<NK>(s1, acc);
            }
        }
    }
    if constexpr (SPLIT) {
        joint_reduce_waves<NK>(acc, red, wave, lane);          // (every wavefront of the block gets here: u0 < Ub is block-uniform)
        if (wave != 0) return;
    }
    if (!kin) return;
    const float sc = scale != nullptr ? scale[b] : 1.0f;   // per-sample factor, applied once per output
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int u = u0 + mfma_row(r, lane);
        if (u >= maxU) continue;
        const size_t at = (static_cast<size_t>(b) * maxU + u) * A + kc;
        float o[NK];
        if (u < Ub) {
            const float mu = mg[u];
            joint_loadv<Tag, NK>(g + at, o);
            // corrections without atomics: dg[u, blank] -= sum_t cb(t,u), dg[u, y_u] -= sum_t cl(t,u); both column
            // sums come from joint_sums_kernel, and row u has ONE label
            const float cb = sgb[static_cast<size_t>(b) * maxU + u] * sc;
            int lab = -1;
            float cl = 0.0f;
            if (u + 1 < Ub) {
                lab = labels[static_cast<size_t>(b) * (maxU - 1) + u];
                lab = lab < 0 ? 0 : (lab >= A ? A - 1 : lab);
                cl = sgl[static_cast<size_t>(b) * maxU + u] * sc;
            }
#pragma unroll
            for (int n = 0; n < NK; ++n) {
                o[n] = joint_exp(o[n], mu) * (acc[n][r] * sc);
                o[n] -= (kc + n == blank) ? cb : 0.0f;
                o[n] -= (kc + n == lab) ? cl : 0.0f;
            }
        } else {
#pragma unroll
            for (int n = 0; n < NK; ++n) o[n] = 0.0f;
        }
        joint_storev<Tag, NK>(dg + at, o);
    }
}

// ------------------------------------------------------------------------------------------
// Row and column sums of the correction terms, for the epilogues of the two gradient GEMMs (runs in the forward
// phase, right after the coefficient kernel):
//     sfb[b][t] = sum_u cb(t,u)        sgb[b][u] = sum_t cb(t,u)        sgl[b][u] = sum_t cl(t,u)
// (fp32 side vectors in the workspace, zeroed by the host) and one flag per sample: "has far cells" (log W above
// kJointFarC: joint_far_kernel handles them after the GEMMs).  Nothing here touches df / dg -- the first version
// applied every correction as an atomic on the outputs, one per lattice cell for the label terms.
// A block owns 64 label rows (one per lane) x kJointFixT time rows of one sample; wavefront w walks the time rows
// w, w+4, ...  grid = (ceil(maxU/64), ceil(maxT/kJointFixT), N), block = 256.
constexpr int kJointFixT = 32;

// the {c, cb, cl} of cell (t, u): from the record table, or (coefficient kernels with planes == 4) from the dense planes
__device__ __forceinline__ Cell<float> joint_cell(const Cell<float>* __restrict__ rowtab, const float* __restrict__ planes,
                                                  int b, int t, int u, int maxT, int maxU, int Upad, int N) {
    Cell<float> rec;
    if (planes != nullptr) {
        const size_t plane = static_cast<size_t>(N) * maxT * Upad;
        const size_t at = (static_cast<size_t>(b) * maxT + t) * Upad + u;
        rec.x = joint_is_far_mark(planes[at]) ? reinterpret_cast<const float*>(rowtab)[at] : 0.0f;   // (c itself is kept for far cells only)
        rec.y = planes[plane + at];
        rec.z = planes[2 * plane + at];
        rec.w = 0.0f;
    } else {
        rec = rowtab[(static_cast<size_t>(b) * maxT + t) * maxU + u];
    }
    return rec;
}

template <int UNUSED = 0>                            // (as joint_prep_kernel)
static __global__ __launch_bounds__(256) void joint_sums_kernel(
        const Cell<float>* __restrict__ rowtab, const int* __restrict__ xlen, const int* __restrict__ ylen,
        float* __restrict__ sfb, float* __restrict__ sgb, float* __restrict__ sgl, int* __restrict__ farflag,
        int maxT, int maxU, int N, const float* __restrict__ planes, int Upad) {
    __shared__ float red[2][4][64];
    const int b = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
    const int ub0 = blockIdx.x * 64, tb0 = blockIdx.y * kJointFixT;
    const int Tb = clamp_len(xlen[b], maxT), Ub = clamp_len(ylen[b] + 1, maxU);
    if (tb0 >= Tb || ub0 >= Ub) return;                    // block-uniform
    const int u = ub0 + lane;
    const bool uin = u < Ub;
    const int tend = tb0 + kJointFixT < Tb ? tb0 + kJointFixT : Tb;
    float dgb = 0.0f, dgl = 0.0f;
    bool any_far = false;
    for (int t = tb0 + wave; t < tend; t += 4) {
        Cell<float> rec;
        rec.x = log_zero<float>(); rec.y = 0.0f; rec.z = 0.0f; rec.w = 0.0f;
        if (uin) rec = joint_cell(rowtab, planes, b, t, u, maxT, maxU, Upad, N);
        dgb += rec.y;
        dgl += rec.z;
        const float rs = wave_sum(rec.y);
        if (lane == 0) unsafeAtomicAdd(sfb + static_cast<size_t>(b) * maxT + t, rs);
        any_far |= uin && rec.x > kJointFarC;
    }
    if (__ballot(any_far) != 0 && lane == 0) farflag[b] = 1;
    red[0][wave][lane] = dgb;
    red[1][wave][lane] = dgl;
    __syncthreads();
    if (wave == 0 && uin) {
        unsafeAtomicAdd(sgb + static_cast<size_t>(b) * maxU + u, red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]);
        unsafeAtomicAdd(sgl + static_cast<size_t>(b) * maxU + u, red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]);
    }
}

// The far cells (log W above kJointFarC: their weight was left out of the gradient GEMMs): exp(f + g + c) added to
// the df row and the dg row of the cell directly.  Runs after the GEMMs; a sample without far cells -- every
// ordinary input -- costs one flag read per block.  Same grid as joint_sums_kernel.
// fp32 outputs take the hardware float atomic; 16-bit outputs a compare-and-swap on the 32-bit word that holds the
// element (rare path: correctness, not speed).
template <typename Tag>
__device__ __forceinline__ void joint_atomic_add(typename Tag::store* p, float v) {
    if constexpr (sizeof(typename Tag::store) == 4) {
        unsafeAtomicAdd(p, v);
    } else {
        const uintptr_t a = reinterpret_cast<uintptr_t>(p);
        unsigned int* w = reinterpret_cast<unsigned int*>(a & ~static_cast<uintptr_t>(3));
        const int sh = (a & 2u) ? 16 : 0;
        unsigned int old = *w, seen;
        do {
            seen = old;
            const uint16_t cur = static_cast<uint16_t>(seen >> sh);
            uint16_t nxt;
            store1<Tag>(&nxt, load1<Tag>(&cur) + v);
            old = atomicCAS(w, seen, (seen & ~(0xffffu << sh)) | (static_cast<unsigned int>(nxt) << sh));
        } while (old != seen);
    }
}

template <typename Tag>
__global__ __launch_bounds__(256) void joint_far_kernel(
        const typename Tag::store* __restrict__ f, const typename Tag::store* __restrict__ g, const float* __restrict__ rowmax,
        const Cell<float>* __restrict__ rowtab, const float* __restrict__ scale, const int* __restrict__ xlen,
        const int* __restrict__ ylen, const int* __restrict__ farflag, typename Tag::store* __restrict__ df,
        typename Tag::store* __restrict__ dg,
        int maxT, int maxU, int A, int N, const float* __restrict__ planes, int Upad) {
    const int b = blockIdx.z;
    if (farflag[b] == 0) return;
    const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
    const int ub0 = blockIdx.x * 64, tb0 = blockIdx.y * kJointFixT;
    const int Tb = clamp_len(xlen[b], maxT), Ub = clamp_len(ylen[b] + 1, maxU);
    if (tb0 >= Tb || ub0 >= Ub) return;
    const int u = ub0 + lane;
    const bool uin = u < Ub;
    const float* mf = rowmax + static_cast<size_t>(b) * maxT;
    const float* mg = rowmax + static_cast<size_t>(N) * maxT + static_cast<size_t>(b) * maxU;
    const float sc = scale != nullptr ? scale[b] : 1.0f;
    const int tend = tb0 + kJointFixT < Tb ? tb0 + kJointFixT : Tb;
    for (int t = tb0 + wave; t < tend; t += 4) {
        float c = log_zero<float>();
        if (uin) {                                         // (only c: with planes == 5 the CB plane holds nothing)
            if (planes != nullptr) {                       // no records: the far mark in W, c stored for marked cells only
                const size_t at = (static_cast<size_t>(b) * maxT + t) * Upad + u;
                if (joint_is_far_mark(planes[at])) c = reinterpret_cast<const float*>(rowtab)[at];
            } else {
                c = rowtab[(static_cast<size_t>(b) * maxT + t) * maxU + u].x;
            }
        }
        typename Tag::store* dfrow = df + (static_cast<size_t>(b) * maxT + t) * A;
        unsigned long long far = __ballot(uin && c > kJointFarC);
        while (far) {
            const int src = __ffsll(static_cast<long long>(far)) - 1;
            far &= far - 1;
            const int uu = ub0 + src;
            const float shift = lane_get(c, src) - (mf[t] + mg[uu]) * static_cast<float>(kLn2);
            const typename Tag::store* fr = f + (static_cast<size_t>(b) * maxT + t) * A;
            const typename Tag::store* gr = g + (static_cast<size_t>(b) * maxU + uu) * A;
            typename Tag::store* dgrow = dg + (static_cast<size_t>(b) * maxU + uu) * A;
            for (int k = lane; k < A; k += 64) {
                const float p = fast_exp(load1<Tag>(fr + k) + load1<Tag>(gr + k) + shift) * sc;
                joint_atomic_add<Tag>(dfrow + k, p);
                joint_atomic_add<Tag>(dgrow + k, p);
            }
        }
    }
}

// 16-bit GRADIENT storage: every joint_atomic_add above rounds to the storage type, and joint_far_kernel adds one CELL at a time --
// a gradient element that collects hundreds of far cells drifted by ~1 % (fp16, T = 520: 1.7354 for 1.7591; tools/add_network_fuzz.py).
// Here the far cells of a whole row segment are summed in fp32 registers first and the stored element is touched ONCE per block:
// phase 0 (blockIdx.y < rows_f): a block owns 64 label rows (lanes) x kJointFixT time rows and adds  sum_u exp(f + g + c)  to df;
// phase 1: the transposed tiling, 64 time rows (lanes) x kJointFixT label rows, adds  sum_t  to dg.  At most ceil(U / 64) resp.
// ceil(T / 64) rounded additions per element instead of U resp. T.  grid = (ceil(max(maxT, maxU) / 64),
// ceil(maxT / kJointFixT) + ceil(maxU / kJointFixT), N), block = 256; a sample without far cells costs one flag read per block.
template <typename Tag>
__global__ __launch_bounds__(256) void joint_far16_kernel(
        const typename Tag::store* __restrict__ f, const typename Tag::store* __restrict__ g, const float* __restrict__ rowmax,
        const Cell<float>* __restrict__ rowtab, const float* __restrict__ scale, const int* __restrict__ xlen,
        const int* __restrict__ ylen, const int* __restrict__ farflag, typename Tag::store* __restrict__ df,
        typename Tag::store* __restrict__ dg,
        int maxT, int maxU, int A, int N, const float* __restrict__ planes, int Upad) {
    const int b = blockIdx.z;
    if (farflag[b] == 0) return;
    const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
    const int rows_f = (maxT + kJointFixT - 1) / kJointFixT;
    const bool for_dg = static_cast<int>(blockIdx.y) >= rows_f;                 // block-uniform
    const int Tb = clamp_len(xlen[b], maxT), Ub = clamp_len(ylen[b] + 1, maxU);
    const int Lb = for_dg ? Tb : Ub, Rb = for_dg ? Ub : Tb;                      // lanes run along L, the block's rows along R
    const int lb0 = blockIdx.x * 64, rb0 = (static_cast<int>(blockIdx.y) - (for_dg ? rows_f : 0)) * kJointFixT;
    if (rb0 >= Rb || lb0 >= Lb) return;
    const int l = lb0 + lane;
    const bool lin = l < Lb;
    const float* mf = rowmax + static_cast<size_t>(b) * maxT;
    const float* mg = rowmax + static_cast<size_t>(N) * maxT + static_cast<size_t>(b) * maxU;
    const float sc = scale != nullptr ? scale[b] : 1.0f;
    const int rend = rb0 + kJointFixT < Rb ? rb0 + kJointFixT : Rb;
    using ST = typename Tag::store;
    const ST* fb = f + static_cast<size_t>(b) * maxT * A;
    const ST* gb = g + static_cast<size_t>(b) * maxU * A;
    for (int r = rb0 + wave; r < rend; r += 4) {
        const int t = for_dg ? l : r, u = for_dg ? r : l;
        float c = log_zero<float>();
        if (lin) {
            if (planes != nullptr) {                       // no records: the far mark in W, c stored for marked cells only
                const size_t at = (static_cast<size_t>(b) * maxT + t) * Upad + u;
                if (joint_is_far_mark(planes[at])) c = reinterpret_cast<const float*>(rowtab)[at];
            } else {
                c = rowtab[(static_cast<size_t>(b) * maxT + t) * maxU + u].x;
            }
        }
        const unsigned long long far = __ballot(lin && c > kJointFarC);
        if (far == 0) continue;                            // (wave-uniform)
        const ST* own = for_dg ? gb + static_cast<size_t>(r) * A : fb + static_cast<size_t>(r) * A;   // the output row's own activations
        const ST* others = for_dg ? fb : gb;
        const float* mo = for_dg ? mf : mg;
        const float m_own = for_dg ? mg[r] : mf[r];
        ST* out = (for_dg ? dg + (static_cast<size_t>(b) * maxU + r) * A : df + (static_cast<size_t>(b) * maxT + r) * A);
        for (int k0 = 0; k0 < A; k0 += 64) {               // (wave-uniform trip count: lane_get below needs every lane there)
            const int k = k0 + lane;
            const bool kin = k < A;
            const float xo = kin ? load1<Tag>(own + k) : 0.0f;
            float acc = 0.0f;
            unsigned long long mask = far;
            while (mask) {
                const int src = __ffsll(static_cast<long long>(mask)) - 1;
                mask &= mask - 1;
                const int ll = lb0 + src;
                const float shift = lane_get(c, src) - (m_own + mo[ll]) * static_cast<float>(kLn2);
                if (kin) acc += fast_exp(xo + load1<Tag>(others + static_cast<size_t>(ll) * A + k) + shift);
            }
            if (kin) joint_atomic_add<Tag>(out + k, acc * sc);
        }
    }
}

}  // namespace rnnt
