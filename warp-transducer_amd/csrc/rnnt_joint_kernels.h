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
// global memory -> registers -> exp -> MFMA with no LDS staging.  Kernels:
//   joint_rowmax_kernel  mf, mg                                              (wave per row)
//   joint_z_kernel       Z tiles + blank / label log-probs into the skewed lattice arrays
//   (lattice_kernel, coef_kernel of rnnt_kernels.h run unchanged: the lattice array `logz` holds
//    the RELATIVE value log Z[t,u], which makes the coefficient record's c equal to log W[t,u])
//   joint_df_kernel / joint_dg_kernel   the two gradient GEMMs with the exp(f) / exp(g) epilogue
//   joint_fix_kernel     blank / label corrections (two columns per cell) and the far cells
//
// Range: ef, eg are in (0,1], so Z[t,u] >= exp(-(separation of the two rows' peaks)).  A cell
// whose GEMM sum falls below kJointFlagZ (the rows peak at different symbols, > 41 nats apart) is
// recomputed in place with a direct log-sum-exp over k, and a cell whose log W exceeds kJointFarC
// is left out of the gradient GEMMs and added by joint_fix_kernel from exp(f + g + c) directly,
// so any finite logit range is handled exactly; ordinary inputs never take either branch.
// fp32 only in this round.
#pragma once

#include "rnnt_kernels.h"

namespace rnnt {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr float kJointFlagZ = 0x1p-60f;   // GEMM sums below this are recomputed directly
constexpr float kJointFarC = 40.0f;       // log W above this: cell handled by joint_fix_kernel
constexpr float kJointMinMax = -3.0e38f;  // row maxima are clamped to a finite value

// C/D fragment of the 32x32 MFMA: register r of lane l holds row (r&3) + 8*(r>>2) + 4*(l>>5),
// column l&31.  A operand: lane l = A[row l&31][k = l>>5]; B operand: lane l = B[k = l>>5][col l&31].
__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float joint_exp(float x, float m) {   // exp(x - m), x <= m (or -inf)
    return fast_exp2((x - m) * static_cast<float>(kLog2e));
}

// ------------------------------------------------------------------------------------------
// Row maxima.  rowmax[0, N*maxT) = mf, rowmax[N*maxT, N*(maxT+maxU)) = mg.  One wavefront per row;
// rows of the padding (t >= T_b, u > U_b) are skipped.  grid = ceil(rows/4), block = 256.
template <bool VEC>
__global__ __launch_bounds__(256) void joint_rowmax_kernel(
        const float* __restrict__ f, const float* __restrict__ g, const int* __restrict__ xlen,
        const int* __restrict__ ylen, float* __restrict__ rowmax, int maxT, int maxU, int A, int N) {
    const long long row = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const long long rows_f = static_cast<long long>(N) * maxT;
    if (row >= rows_f + static_cast<long long>(N) * maxU) return;
    const int lane = threadIdx.x & 63;
    const float* p;
    if (row < rows_f) {
        const int b = static_cast<int>(row / maxT);
        if (static_cast<int>(row - static_cast<long long>(b) * maxT) >= xlen[b]) return;
        p = f + row * A;
    } else {
        const long long r = row - rows_f;
        const int b = static_cast<int>(r / maxU);
        if (static_cast<int>(r - static_cast<long long>(b) * maxU) > ylen[b]) return;
        p = g + r * A;
    }
    float m = neg_inf<float>();
    if constexpr (VEC) {
        const float4* p4 = reinterpret_cast<const float4*>(p);
        for (int i = lane; i < (A >> 2); i += 64) {
            const float4 v = p4[i];
            m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
        }
    } else {
        for (int i = lane; i < A; i += 64) m = fmaxf(m, p[i]);
    }
    m = wave_max(m);
    if (lane == 0) rowmax[row] = fmaxf(m, kJointMinMax);
}

// Eight consecutive columns k..k+7 of a row; columns >= A read as -inf (exp -> 0).
template <bool VEC>
__device__ __forceinline__ void joint_load8(const float* __restrict__ row, int k, int A, float (&v)[8]) {
    if constexpr (VEC) {                       // A % 4 == 0, row 16-byte aligned
        float4 a = {neg_inf<float>(), neg_inf<float>(), neg_inf<float>(), neg_inf<float>()}, c = a;
        if (k < A) a = *reinterpret_cast<const float4*>(row + k);
        if (k + 4 < A) c = *reinterpret_cast<const float4*>(row + k + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (k + j < A) ? row[k + j] : neg_inf<float>();
    }
}

// ------------------------------------------------------------------------------------------
// Partition function.  A wavefront owns a 32 (t) x 32 (u) tile of Z and contracts over its share
// of the vocabulary.  The k order inside a contraction is free as long as A and B agree, so the
// two lane halves take the two 8-column halves of a 16-column chunk (two 16-byte loads per row and
// operand) and MFMA step j pairs column j of both halves.  S wavefronts of a block split the chunks
// of ONE tile (large vocabulary, few tiles) and add their fragments through LDS; with S = 1 the
// four wavefronts of a block own four tiles.  The next chunk is loaded while the current one is
// in the matrix core.
// Epilogue per cell: log Z -> logz (relative), blank / label log2-probs -> lp2, both in the
// skewed lattice layout.  grid = (tiles or ceil(tiles/4), N), block = 64 * max(S, 4).
template <int S, bool VEC>
__global__ __launch_bounds__(S == 1 ? 256 : S * 64) void joint_z_kernel(
        const float* __restrict__ f, const float* __restrict__ g, const float* __restrict__ rowmax,
        const int* __restrict__ labels, const int* __restrict__ xlen, const int* __restrict__ ylen,
        LogPair<float>* __restrict__ lp2, float* __restrict__ logz, int maxT, int maxU, int Up, int A,
        int blank, int tilesU, int tiles, int N) {
    __shared__ float red[S == 1 ? 1 : S][S == 1 ? 1 : 16][64];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
    const int tile = S == 1 ? static_cast<int>(blockIdx.x) * 4 + wave : static_cast<int>(blockIdx.x);
    if (tile >= tiles) return;                             // S == 1 only: a whole wavefront leaves
    const int Tb = xlen[b], Ub = ylen[b] + 1;
    const int t0 = (tile / tilesU) * 32, u0 = (tile % tilesU) * 32;
    if (t0 >= Tb || u0 >= Ub) return;                      // tile of padding (block-uniform when S > 1)
    const float* mf = rowmax + static_cast<size_t>(b) * maxT;
    const float* mg = rowmax + static_cast<size_t>(N) * maxT + static_cast<size_t>(b) * maxU;
    const int ti = t0 + col < Tb ? t0 + col : Tb - 1;     // operand rows past the sample: any valid row
    const int ui = u0 + col < Ub ? u0 + col : Ub - 1;
    const float* frow = f + (static_cast<size_t>(b) * maxT + ti) * A;
    const float* grow = g + (static_cast<size_t>(b) * maxU + ui) * A;
    const float mft = mf[ti], mgu = mg[ui];

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int nchunk = (A + 15) >> 4;
    int c = S == 1 ? 0 : wave;
    float fa[8], ga[8];
    joint_load8<VEC>(frow, c * 16 + half * 8, A, fa);
    joint_load8<VEC>(grow, c * 16 + half * 8, A, ga);
    for (; c < nchunk; c += S) {
        float fn[8], gn[8];
        const int kn = (c + S) * 16 + half * 8;            // past the end: all -inf, never used
        joint_load8<VEC>(frow, kn, A, fn);
        joint_load8<VEC>(grow, kn, A, gn);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(joint_exp(fa[j], mft), joint_exp(ga[j], mgu), acc, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) { fa[j] = fn[j]; ga[j] = gn[j]; }
    }

    // per-lane constants of the epilogue: this lane's label row u = u0 + col
    const int u = u0 + col;
    const bool has_lab = u < Ub - 1;
    int lab = blank;
    if (has_lab) {
        lab = labels[static_cast<size_t>(b) * (maxU - 1) + u];
        lab = lab < 0 ? 0 : (lab >= A ? A - 1 : lab);
    }
    const float gbl = grow[blank] - mgu, glab = grow[lab] - mgu;

    auto finish = [&](int r, float z) {
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
            const float* fr = f + (static_cast<size_t>(b) * maxT + tt) * A;
            const float* gr = g + (static_cast<size_t>(b) * maxU + uu) * A;
            float m = neg_inf<float>();
            for (int k = lane; k < A; k += 64) m = fmaxf(m, fr[k] + gr[k]);
            m = fmaxf(wave_max(m), kJointMinMax);
            float s = 0.0f;
            for (int k = lane; k < A; k += 64) s += fast_exp(fr[k] + gr[k] - m);
            s = wave_sum(s);
            const float v = ((m - mf[tt]) - mg[uu]) + acc_log(s);
            if (lane == src) lz = v;
        }
        if (!valid) return;
        const float* ft = f + (static_cast<size_t>(b) * maxT + t) * A;
        const float mt = mf[t];
        LogPair<float> rec;                               // lattice log-probs are kept in base 2
        rec.x = fmaxf(((ft[blank] - mt) + gbl - lz) * static_cast<float>(kLog2e), log_zero<float>());
        rec.y = has_lab ? fmaxf(((ft[lab] - mt) + glab - lz) * static_cast<float>(kLog2e), log_zero<float>())
                        : log_zero<float>();
        const size_t idx = lat_index(b, t + u, u, maxT, maxU, Up);
        lp2[idx] = rec;
        logz[idx] = lz;
    };

    if constexpr (S == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) finish(r, acc[r]);
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
        __syncthreads();
        constexpr int per = 16 / S;                        // fragment registers finished by each wavefront
#pragma unroll
        for (int i = 0; i < per; ++i) {
            const int r = wave * per + i;
            float z = 0.0f;
#pragma unroll
            for (int s = 0; s < S; ++s) z += red[s][r][lane];
            finish(r, z);
        }
    }
}

// log W -> W for the gradient GEMMs: far cells (and the padding, c = log_zero) contribute nothing.
__device__ __forceinline__ float joint_weight(float c) { return c > kJointFarC ? 0.0f : fast_exp(c); }

// ------------------------------------------------------------------------------------------
// DF[t,k] = ef[t,k] * sum_u W[t,u] eg[u,k].  A wavefront owns 32 time rows x 32*NK columns and
// contracts over the label rows two at a time (lane half h takes u = 2s + h): A operand = W read
// from the coefficient table, B operand = exp(g[u,k] - mg[u]) (128-byte row segments).  The loop
// is unrolled kJointUnr steps with all loads first.  Epilogue: multiply by ef (one read of f),
// store; rows of the padding are written as zeros.  The four wavefronts of a block take adjacent
// column groups of the same time rows.  grid = (ceil(A / (128 NK)), ceil(maxT/32), N), block = 256.
constexpr int kJointUnr = 4;

template <int NK>
__global__ __launch_bounds__(256) void joint_df_kernel(
        const float* __restrict__ f, const float* __restrict__ g, const float* __restrict__ rowmax,
        const Cell<float>* __restrict__ rowtab, const int* __restrict__ xlen, const int* __restrict__ ylen,
        float* __restrict__ df, int maxT, int maxU, int A, int N) {
    const int b = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
    const int k0 = (static_cast<int>(blockIdx.x) * 4 + wave) * (32 * NK);
    if (k0 >= A) return;
    const int t0 = blockIdx.y * 32;
    const int Tb = xlen[b], Ub = ylen[b] + 1;
    const float* mf = rowmax + static_cast<size_t>(b) * maxT;
    const float* mg = rowmax + static_cast<size_t>(N) * maxT + static_cast<size_t>(b) * maxU;
    f32x16 acc[NK];
#pragma unroll
    for (int n = 0; n < NK; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;

    if (t0 < Tb) {
        const bool tin = t0 + col < Tb;
        const Cell<float>* wrow = rowtab + (static_cast<size_t>(b) * maxT + (tin ? t0 + col : Tb - 1)) * maxU;
        const float* gb = g + static_cast<size_t>(b) * maxU * A;
        for (int u2 = 0; u2 < Ub; u2 += 2 * kJointUnr) {
            float c[kJointUnr], m[kJointUnr], x[kJointUnr][NK];
#pragma unroll
            for (int i = 0; i < kJointUnr; ++i) {
                const int u = u2 + 2 * i + half;
                const bool uin = u < Ub;
                const int us = uin ? u : Ub - 1;
                c[i] = (uin && tin) ? wrow[us].x : log_zero<float>();
                m[i] = mg[us];
#pragma unroll
                for (int n = 0; n < NK; ++n) {
                    const int k = k0 + 32 * n + col;
                    x[i][n] = (uin && k < A) ? gb[static_cast<size_t>(us) * A + k] : neg_inf<float>();
                }
            }
#pragma unroll
            for (int i = 0; i < kJointUnr; ++i) {
                const float w = joint_weight(c[i]);
#pragma unroll
                for (int n = 0; n < NK; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, joint_exp(x[i][n], m[i]), acc[n], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int t = t0 + mfma_row(r, lane);
        if (t >= maxT) continue;
        const size_t row = (static_cast<size_t>(b) * maxT + t) * A;
        const float mt = t < Tb ? mf[t] : 0.0f;
#pragma unroll
        for (int n = 0; n < NK; ++n) {
            const int k = k0 + 32 * n + col;
            if (k >= A) continue;
            df[row + k] = t < Tb ? joint_exp(f[row + k], mt) * acc[n][r] : 0.0f;
        }
    }
}

// DG[u,k] = eg[u,k] * sum_t W[t,u] ef[t,k]: the same with the roles of f and g exchanged; the
// contraction runs over the time rows (A operand = W^T, 32 consecutive records of one time row).
// grid = (ceil(A / (128 NK)), ceil(maxU/32), N), block = 256.
template <int NK>
__global__ __launch_bounds__(256) void joint_dg_kernel(
        const float* __restrict__ f, const float* __restrict__ g, const float* __restrict__ rowmax,
        const Cell<float>* __restrict__ rowtab, const int* __restrict__ xlen, const int* __restrict__ ylen,
        float* __restrict__ dg, int maxT, int maxU, int A, int N) {
    const int b = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
    const int k0 = (static_cast<int>(blockIdx.x) * 4 + wave) * (32 * NK);
    if (k0 >= A) return;
    const int u0 = blockIdx.y * 32;
    const int Tb = xlen[b], Ub = ylen[b] + 1;
    const float* mf = rowmax + static_cast<size_t>(b) * maxT;
    const float* mg = rowmax + static_cast<size_t>(N) * maxT + static_cast<size_t>(b) * maxU;
    f32x16 acc[NK];
#pragma unroll
    for (int n = 0; n < NK; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;

    if (u0 < Ub) {
        const bool uin = u0 + col < Ub;
        const Cell<float>* wcol = rowtab + static_cast<size_t>(b) * maxT * maxU + (uin ? u0 + col : Ub - 1);
        const float* fb = f + static_cast<size_t>(b) * maxT * A;
        for (int t2 = 0; t2 < Tb; t2 += 2 * kJointUnr) {
            float c[kJointUnr], m[kJointUnr], x[kJointUnr][NK];
#pragma unroll
            for (int i = 0; i < kJointUnr; ++i) {
                const int t = t2 + 2 * i + half;
                const bool tin = t < Tb;
                const int ts = tin ? t : Tb - 1;
                c[i] = (uin && tin) ? wcol[static_cast<size_t>(ts) * maxU].x : log_zero<float>();
                m[i] = mf[ts];
#pragma unroll
                for (int n = 0; n < NK; ++n) {
                    const int k = k0 + 32 * n + col;
                    x[i][n] = (tin && k < A) ? fb[static_cast<size_t>(ts) * A + k] : neg_inf<float>();
                }
            }
#pragma unroll
            for (int i = 0; i < kJointUnr; ++i) {
                const float w = joint_weight(c[i]);
#pragma unroll
                for (int n = 0; n < NK; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, joint_exp(x[i][n], m[i]), acc[n], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int u = u0 + mfma_row(r, lane);
        if (u >= maxU) continue;
        const size_t row = (static_cast<size_t>(b) * maxU + u) * A;
        const float mu = u < Ub ? mg[u] : 0.0f;
#pragma unroll
        for (int n = 0; n < NK; ++n) {
            const int k = k0 + 32 * n + col;
            if (k >= A) continue;
            dg[row + k] = u < Ub ? joint_exp(g[row + k], mu) * acc[n][r] : 0.0f;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Corrections on top of the two GEMM results (runs after them on the same stream):
//     df[t,blank] -= sum_u cb(t,u)     df[t,y_u] -= cl(t,u)
//     dg[u,blank] -= sum_t cb(t,u)     dg[u,y_u] -= sum_t cl(t,u)
// and the far cells' exp(f + g + c) terms.  A block owns 64 label rows (one per lane) x kJointFixT
// time rows of one sample; wavefront w walks the time rows w, w+4, ...: the df terms go out as
// atomics (labels repeat), the dg terms accumulate in two registers per lane and leave as one
// atomic per (u, column) and block.  grid = (ceil(maxU/64), ceil(maxT/kJointFixT), N), block = 256.
constexpr int kJointFixT = 32;

__global__ __launch_bounds__(256) void joint_fix_kernel(
        const float* __restrict__ f, const float* __restrict__ g, const float* __restrict__ rowmax,
        const Cell<float>* __restrict__ rowtab, const int* __restrict__ labels,
        const int* __restrict__ xlen, const int* __restrict__ ylen, float* __restrict__ df,
        float* __restrict__ dg, int maxT, int maxU, int A, int blank, int N) {
    __shared__ float red[2][4][64];
    const int b = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ub0 = blockIdx.x * 64, tb0 = blockIdx.y * kJointFixT;
    const int Tb = xlen[b], Ub = ylen[b] + 1;
    if (tb0 >= Tb || ub0 >= Ub) return;                    // block-uniform
    const int u = ub0 + lane;
    const bool uin = u < Ub, has_lab = u < Ub - 1;
    int lab = 0;
    if (has_lab) {
        lab = labels[static_cast<size_t>(b) * (maxU - 1) + u];
        lab = lab < 0 ? 0 : (lab >= A ? A - 1 : lab);
    }
    const float* mf = rowmax + static_cast<size_t>(b) * maxT;
    const float* mg = rowmax + static_cast<size_t>(N) * maxT + static_cast<size_t>(b) * maxU;
    const int tend = tb0 + kJointFixT < Tb ? tb0 + kJointFixT : Tb;
    float dgb = 0.0f, dgl = 0.0f;
    for (int t = tb0 + wave; t < tend; t += 4) {
        Cell<float> rec;
        rec.x = log_zero<float>(); rec.y = 0.0f; rec.z = 0.0f; rec.w = 0.0f;
        if (uin) rec = rowtab[(static_cast<size_t>(b) * maxT + t) * maxU + u];
        dgb += rec.y;
        dgl += rec.z;
        float* dfrow = df + (static_cast<size_t>(b) * maxT + t) * A;
        const float rs = wave_sum(rec.y);
        if (lane == 0) unsafeAtomicAdd(dfrow + blank, -rs);
        if (has_lab && rec.z != 0.0f) unsafeAtomicAdd(dfrow + lab, -rec.z);
        unsigned long long far = __ballot(uin && rec.x > kJointFarC);
        while (far) {                                      // never taken for ordinary logits
            const int src = __ffsll(static_cast<long long>(far)) - 1;
            far &= far - 1;
            const int uu = ub0 + src;
            const float shift = lane_get(rec.x, src) - mf[t] - mg[uu];
            const float* fr = f + (static_cast<size_t>(b) * maxT + t) * A;
            const float* gr = g + (static_cast<size_t>(b) * maxU + uu) * A;
            float* dgrow = dg + (static_cast<size_t>(b) * maxU + uu) * A;
            for (int k = lane; k < A; k += 64) {
                const float p = fast_exp(fr[k] + gr[k] + shift);
                unsafeAtomicAdd(dfrow + k, p);
                unsafeAtomicAdd(dgrow + k, p);
            }
        }
    }
    red[0][wave][lane] = dgb;
    red[1][wave][lane] = dgl;
    __syncthreads();
    if (wave == 0 && uin) {
        const float sb = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
        const float sl = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
        float* dgrow = dg + (static_cast<size_t>(b) * maxU + u) * A;
        unsafeAtomicAdd(dgrow + blank, -sb);
        if (has_lab) unsafeAtomicAdd(dgrow + lab, -sl);
    }
}

}  // namespace rnnt
