// rnnt_joint_kernels.h -- additive-joint ("add network") fusion, SURVEY.md 8f rank 1.
//
// The reference's add_network branch (README.md:4, docs/rnnt_notes.tex:56-59,147-153,
// pytorch_binding/test/test_time.py:51-77) takes the transcription output f (N,T,A) and the
// prediction output g (N,U,A) of Graves' 2012 transducer, whose joint is
//     h(k,t,u) = f[t,k] + g[u,k],      Pr(k|t,u) = softmax_k h(k,t,u),
// and returns dL/df = sum_u dL/dh, dL/dg = sum_t dL/dh.  Here the (N,T,U,A) tensor is never
// materialised in HBM: both streaming passes of the materialised path are replaced by kernels
// that form f+g on the fly from LDS / register tiles,
//   joint_stats_kernel : logZ(t,u) and the blank / label log-probs      (replaces row_stats)
//   joint_grad_kernel  : df, dg reduced in-kernel from exp(f+g+c) terms (replaces grad_flat)
// while the lattice and coefficient kernels of rnnt_kernels.h run unchanged in between.
// fp32 only in this round.
#pragma once

#include "rnnt_kernels.h"

namespace rnnt {

// ------------------------------------------------------------------------------------------
// Joint statistics.  A 256-thread block owns 256 consecutive lattice cells q = t*U+u of one
// sample (flat order, so every lane has work whatever U is), i.e. a window of time rows
// [t_lo, t_lo+nT) and label rows [u_lo, u_lo+nU).  The vocabulary is walked in chunks of
// kJointVC columns: the f rows and g rows of the window are staged in LDS (coalesced 128-byte
// row segments), then every thread runs an ONLINE log-sum-exp over its own (t,u) pair, eight
// columns at a time (one rescale per eight exps), so any logit range is safe.
// grid = (ceil(T*U/256), N) when U <= 128; one time row per block, (T * ceil(U/256), N), above.
constexpr int kJointVC = 32;            // vocabulary columns per LDS chunk
constexpr int kJointPad = kJointVC + 4; // LDS row stride (floats): keeps 16-byte reads aligned, spreads banks
constexpr int kJointMaxRows = 400;      // f rows + g rows of one window (host checks)

__global__ __launch_bounds__(256) void joint_stats_kernel(
        const float* __restrict__ f, const float* __restrict__ g, const int* __restrict__ labels,
        const int* __restrict__ xlen, const int* __restrict__ ylen, LogPair<float>* __restrict__ lp2,
        float* __restrict__ logz, int maxT, int maxU, int Up, int A, int blank, int per_row) {
    extern __shared__ float4 joint_lds4[];               // 16-byte aligned base (rows are read as float4)
    float* joint_lds = reinterpret_cast<float*>(joint_lds4);
    const int b = blockIdx.y;
    const int Tb = xlen[b], Ub = ylen[b] + 1;
    // window of this block
    int t_lo, nT, u_lo, nU, t, u;
    bool in_range;
    if (per_row) {                       // U > 128: one time row, 256 consecutive label positions
        const int chunks = (maxU + 255) / 256;
        t_lo = blockIdx.x / chunks;
        u_lo = (blockIdx.x - t_lo * chunks) * 256;
        nT = 1;
        nU = maxU - u_lo < 256 ? maxU - u_lo : 256;
        t = t_lo;
        u = u_lo + threadIdx.x;
        in_range = threadIdx.x < nU;
    } else {                             // flat cells, all label rows staged
        const int q0 = blockIdx.x * 256;
        const int q = q0 + threadIdx.x;
        const int qmax = maxT * maxU;
        in_range = q < qmax;
        t = in_range ? q / maxU : maxT - 1;
        u = in_range ? q - t * maxU : 0;
        t_lo = q0 / maxU;
        int t_hi = (q0 + 255) / maxU;
        if (t_hi > maxT - 1) t_hi = maxT - 1;
        nT = t_hi - t_lo + 1;
        u_lo = 0;
        nU = maxU;
    }
    if (t_lo >= Tb) return;              // the whole window is padding (block-uniform)
    const bool valid = in_range && t < Tb && u < Ub;

    float* ftile = joint_lds;                         // [nT][kJointPad]
    float* gtile = joint_lds + nT * kJointPad;        // [nU][kJointPad]
    const float* fb = f + (static_cast<size_t>(b) * maxT + t_lo) * A;
    const float* gb = g + (static_cast<size_t>(b) * maxU + u_lo) * A;
    const float* frow = ftile + (t - t_lo) * kJointPad;
    const float* grow = gtile + (u - u_lo) * kJointPad;

    float m = neg_inf<float>(), s = 0.0f;
    const int rows = nT + nU;
    for (int v0 = 0; v0 < A; v0 += kJointVC) {
        const int vc = A - v0 < kJointVC ? A - v0 : kJointVC;
        // stage: element (row r, column c) <- f or g; consecutive threads take consecutive columns
        for (int i = threadIdx.x; i < rows * kJointVC; i += 256) {
            const int r = i / kJointVC, c = i - r * kJointVC;
            float x = neg_inf<float>();              // columns past A contribute exp(-inf) = 0
            if (c < vc) x = (r < nT) ? fb[static_cast<size_t>(r) * A + v0 + c]
                                     : gb[static_cast<size_t>(r - nT) * A + v0 + c];
            joint_lds[r * kJointPad + c] = x;
        }
        __syncthreads();
        if (valid) {
#pragma unroll
            for (int c = 0; c < kJointVC; c += 8) {
                const float4 a0 = *reinterpret_cast<const float4*>(frow + c);
                const float4 a1 = *reinterpret_cast<const float4*>(frow + c + 4);
                const float4 b0 = *reinterpret_cast<const float4*>(grow + c);
                const float4 b1 = *reinterpret_cast<const float4*>(grow + c + 4);
                float x[8] = {a0.x + b0.x, a0.y + b0.y, a0.z + b0.z, a0.w + b0.w,
                              a1.x + b1.x, a1.y + b1.y, a1.z + b1.z, a1.w + b1.w};
                float mx = x[0];
#pragma unroll
                for (int k = 1; k < 8; ++k) mx = fmaxf(mx, x[k]);
                const float mn = fmaxf(m, mx);
                if (mn != neg_inf<float>()) {
                    float acc = 0.0f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc += fast_exp(x[k] - mn);
                    s = s * fast_exp(m - mn) + acc;
                    m = mn;
                }
            }
        }
        __syncthreads();
    }
    if (!valid) return;
    const float logZ = m + acc_log(s);
    const bool has_lab = u < Ub - 1;
    int lab = blank;
    if (has_lab) {
        lab = labels[static_cast<size_t>(b) * (maxU - 1) + u];
        lab = lab < 0 ? 0 : (lab >= A ? A - 1 : lab);
    }
    const float* ft = f + (static_cast<size_t>(b) * maxT + t) * A;
    const float* gu = g + (static_cast<size_t>(b) * maxU + u) * A;
    LogPair<float> rec;                               // lattice log-probs are kept in base 2
    rec.x = fmaxf((ft[blank] + gu[blank] - logZ) * static_cast<float>(kLog2e), log_zero<float>());
    rec.y = has_lab ? fmaxf((ft[lab] + gu[lab] - logZ) * static_cast<float>(kLog2e), log_zero<float>())
                    : log_zero<float>();
    const size_t idx = lat_index(b, t + u, u, maxT, maxU, Up);
    lp2[idx] = rec;
    logz[idx] = logZ;
}

// ------------------------------------------------------------------------------------------
// Joint gradient.  With the row table {c, cb, cl, label} of coef_kernel,
//     dL/dh(k,t,u) = exp(f[t,k] + g[u,k] + c(t,u)) - [k=blank] cb(t,u) - [k=label] cl(t,u)
// (zero for padded cells), df[t,k] = sum_u, dg[u,k] = sum_t.
// A block owns 64 vocabulary columns (one per lane) and a slice of kJointTS = 32 time rows whose
// f values sit in 32 registers per lane.  Wavefront w takes the label rows u = w, w+4, ...: per
// row it loads g[u,k] (coalesced), fetches the 32 records (t0..t0+31, u) with one load (lane i
// holds record i) and runs the fully unrolled loop over the 32 time rows, handing c(t,u) round
// with v_readlane: dg[u,k] accumulates in ONE register (this wavefront is the only one that
// touches row u in this block) and goes to global memory as one atomic per (u,k) and block (dg is
// zero-filled by the host first; blocks of other time slices add to it), df[t,k] accumulates in
// 32 registers, the four wavefronts' partial sums are combined through LDS at the end and stored
// by their unique owner.  The blank / label corrections touch two columns per cell, so only
// blocks whose 64 columns contain the blank or one of the sample's labels run the longer loop.
// grid = (ceil(A/64), ceil(T/32), N), block = 256.
constexpr int kJointTS = 32;    // time rows per block

template <bool SPECIAL>
__device__ __forceinline__ void joint_grad_rows(
        const float (&fv)[kJointTS], float (&dfacc)[kJointTS], const float* __restrict__ g,
        const Cell<float>* __restrict__ tab, float* __restrict__ dg, int b, int t0, int maxT, int maxU,
        int Ub, int A, int k, bool kin, bool is_blank, int lane, int wave) {
    for (int u = wave; u < Ub; u += 4) {
        const float gv = kin ? g[(static_cast<size_t>(b) * maxU + u) * A + k] : 0.0f;
        Cell<float> rec;
        rec.x = log_zero<float>(); rec.y = 0; rec.z = 0; rec.w = static_cast<float>(kPadded);
        if (lane < kJointTS && t0 + lane < maxT) rec = tab[static_cast<size_t>(t0 + lane) * maxU + u];
        float dgacc = 0.0f;
#pragma unroll
        for (int i = 0; i < kJointTS; ++i) {
            const float c = lane_get(rec.x, i);                 // log_zero for padded cells -> p = 0
            float p = fast_exp(fv[i] + (gv + c));
            if constexpr (SPECIAL) {
                const float cb = lane_get(rec.y, i), cl = lane_get(rec.z, i);
                const int lab = static_cast<int>(lane_get(rec.w, i));
                if (is_blank) p -= cb;
                if (k == lab) p -= cl;
            }
            dfacc[i] += p;
            dgacc += p;
        }
        if (kin) atomicAdd(dg + (static_cast<size_t>(b) * maxU + u) * A + k, dgacc);
    }
}

__global__ __launch_bounds__(256) void joint_grad_kernel(
        const float* __restrict__ f, const float* __restrict__ g, const Cell<float>* __restrict__ rowtab,
        const int* __restrict__ labels, const int* __restrict__ xlen, const int* __restrict__ ylen,
        float* __restrict__ df, float* __restrict__ dg, int maxT, int maxU, int A, int blank) {
    __shared__ float dft[4][kJointTS][64];
    const int b = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k0 = blockIdx.x * 64;
    const int k = k0 + lane;                               // vocabulary column of this lane
    const bool kin = k < A;
    const int t0 = blockIdx.y * kJointTS;
    const int Tb = xlen[b], Ub = ylen[b] + 1;
    const bool is_blank = (k == blank);
    const Cell<float>* tab = rowtab + static_cast<size_t>(b) * maxT * maxU;

    // does this block's column range hold the blank or one of the sample's labels?
    int hit = (blank >= k0 && blank < k0 + 64) ? 1 : 0;
    for (int i = threadIdx.x; i < Ub - 1; i += 256) {
        const int lab = labels[static_cast<size_t>(b) * (maxU - 1) + i];
        hit |= (lab >= k0 && lab < k0 + 64) ? 1 : 0;
    }
    const bool special = __syncthreads_or(hit) != 0;

    float fv[kJointTS], dfacc[kJointTS];
#pragma unroll
    for (int i = 0; i < kJointTS; ++i) {
        const int t = t0 + i;
        fv[i] = (kin && t < Tb) ? f[(static_cast<size_t>(b) * maxT + t) * A + k] : 0.0f;
        dfacc[i] = 0.0f;
    }
    if (t0 < Tb) {                                         // block-uniform: time rows past T_b are padding
        if (special)
            joint_grad_rows<true>(fv, dfacc, g, tab, dg, b, t0, maxT, maxU, Ub, A, k, kin, is_blank, lane, wave);
        else
            joint_grad_rows<false>(fv, dfacc, g, tab, dg, b, t0, maxT, maxU, Ub, A, k, kin, is_blank, lane, wave);
    }
#pragma unroll
    for (int i = 0; i < kJointTS; ++i) dft[wave][i][lane] = dfacc[i];
    __syncthreads();
    for (int i = wave; i < kJointTS; i += 4) {
        const int t = t0 + i;
        if (kin && t < maxT)
            df[(static_cast<size_t>(b) * maxT + t) * A + k] = dft[0][i][lane] + dft[1][i][lane] + dft[2][i][lane] + dft[3][i][lane];
    }
}

}  // namespace rnnt
