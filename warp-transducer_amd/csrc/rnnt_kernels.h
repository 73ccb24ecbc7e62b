// rnnt_kernels.h -- the gfx950 kernels of the RNN-T loss hot path (four stages).
//
//   1 row_stats_block_kernel rows >= 12 KB: one 256-thread block per (b,t,u) row,
//     row_stats_kernel       rows > 2 KB: one wavefront per row, ONE read of the logits,
//     row_stats_tile_kernel  rows <= 2 KB: a block stages a contiguous tile of rows in LDS;
//                            online log-sum-exp per row + gather of the blank / label logits
//                                                                      [HBM-bound, E*s read]
//   2 lattice_kernel         alpha and beta recursions over the T x U lattice (concurrently),
//                            one wavefront lane per u sweeping the anti-diagonals, values in
//                            registers, neighbour exchange through DPP wave shifts, base-2
//                            scaled arithmetic                          [issue/latency-bound, O(R)]
//   3 coef_kernel            per lattice cell: the three numbers the gradient needs (row exponent
//                            offset, blank correction, label correction) into a natural-order
//                            row table                                                   [O(R)]
//   4 grad_flat_kernel       second read of the logits as one flat packet stream + dense gradient
//                            g_v = exp(x_v + c) - [v==blank] cb - [v==label] cl, zero-fill of the
//                            padding                              [HBM-bound, 2*E*s read+write]
//     grad_rows_kernel       the same per row, for tensors that are not 16-byte aligned
//
// What they replace in the reference (behaviour, not structure):
//   include/detail/reduce.h:45-104 + gpu_rnnt.h:73-80   (two-pass max / exp-sum denominators)
//   include/detail/gpu_rnnt_kernel.h:11-47, 79-113      (alpha / beta kernels)
//   include/detail/gpu_rnnt_kernel.h:143-179 + gpu_rnnt.h:107-110 (gradient kernel + memset)
//
// Lattice side data lives in the caller's workspace in a DIAGONAL-SKEWED layout:
//   cell(b, t, u) -> (kLatPad + (t + u)) * Up + u  inside sample b's array,  Dp = maxT + maxU - 1 + 2*kLatPad rows,
//   Up = 8*ceil(maxU/8)                                           (lat_index below)
// and, since round 6, in PER-SAMPLE BLOCKS [lp2 | logz | alpha | beta] (lat_block): a sample's side data is one contiguous
// piece of the workspace, which is dead as a whole once its coefficient records exist -- the record table of later samples
// overlays it (rnnt_host.h, make_layout).
// so that the lanes of the lattice wavefront (consecutive u on one anti-diagonal t+u = n)
// touch consecutive addresses.  All indices are 64-bit (the reference's are 32-bit int:
// gpu_rnnt_kernel.h:7-8,161,174).
#pragma once

#include "rnnt_device.h"

namespace rnnt {

// Skewed lattice arrays (per sample a structure of arrays, dense rows of Up values):
//   lp2   : {x = log2 p(blank|t,u), y = log2 p(y_u|t,u)}   written by the row-stats pass
//   logz  : natural-log partition function of the row        written by the row-stats pass
//   alpha : scaled forward variable (base 2)                 written by the lattice kernel
//   beta  : scaled backward variable (base 2)                written by the lattice kernel
// The gradient coefficients {c, cb, cl, label} use the 4-word record `Cell` in a separate
// natural-order row table (coef_kernel).
template <typename L> struct alignas(2 * sizeof(L)) LogPair { L x, y; };
template <typename L> struct alignas(4 * sizeof(L)) Cell { L x, y, z, w; };

constexpr double kLog2e = 1.4426950408889634;
constexpr double kLn2 = 0.6931471805599453;

// Additive-joint path (rnnt_joint_kernels.h): coef_kernel also emits the dense weight matrix
// W[b][t][u] = exp(c) of the gradient GEMMs; cells with c above kJointFarC ("far" cells, whose two
// logit rows peak at different symbols) and the padding get 0 and are handled outside the GEMMs.
constexpr float kJointFarC = 40.0f;
// ... and W holds this mark there: NEGATIVE zero.  As an operand of the gradient GEMMs it is a zero like any other (no
// clamp, no mask in DF / DG), and no exp() produces it, so joint_far_kernel / joint_sums_kernel recognise a far cell by the
// bit pattern -- with the one-hot planes (no records) only those cells' c is stored, into the record table's memory
// at the plane index, instead of a dense plane of c that nothing else read (117 MB per step on the c4 shape).
constexpr float kJointFarMark = -0.0f;
__device__ __forceinline__ bool joint_is_far_mark(float w) { return __float_as_uint(w) == 0x80000000u; }
// Row stride of those dense matrices: maxU rounded up to 8 (the GEMMs read them eight columns at a time;
// the pad columns are kept zero).  Three planes of N*maxT*Upad floats each: W, CB (blank corrections),
// CL (label corrections).
__host__ __device__ inline int joint_upad(int maxU) { return (maxU + 7) & ~7; }

// The skewed arrays carry kLatPad spare rows before diagonal 0 and after diagonal D-1 of every
// sample, so the last (partial) chunk of a sweep can run its full C steps without bounds checks.
// Cost written for a sample whose device-side lengths do not fit the tensor (T_b outside [1,maxT] or
// U_b outside [1,maxU]): a quiet NaN with a recognisable payload.
template <typename L> __host__ __device__ inline L cost_invalid();
template <> __host__ __device__ inline float cost_invalid<float>() {
    const unsigned int bits = 0x7fc0deadu; float f; __builtin_memcpy(&f, &bits, 4); return f;
}
template <> __host__ __device__ inline double cost_invalid<double>() {
    const unsigned long long bits = 0x7ff8dead00000000ull; double d; __builtin_memcpy(&d, &bits, 8); return d;
}
template <typename L> inline bool is_cost_invalid(L v) {
    const L m = cost_invalid<L>();
    return __builtin_memcmp(&v, &m, sizeof(L)) == 0;
}

// Device-side lengths clamped to the tensor (see lattice_kernel: such a sample is flagged through its cost;
// every kernel clamps so that nothing is read or written outside the caller's arrays).
__device__ __forceinline__ int clamp_len(int v, int hi) { return v > hi ? hi : v; }
// The coefficient kernels' view of a sample: lengths that do not fit the tensor (the lattice kernel has marked its
// cost) make EVERY cell of the sample a padded one, so its gradient is zero instead of a function of workspace
// cells nobody wrote (T_b <= 0: the statistics kernels skip the sample altogether).
__device__ __forceinline__ void coef_lens(const int* __restrict__ xlen, const int* __restrict__ ylen, int b, int maxT,
                                          int maxU, int& Tb, int& Ub) {
    const int T = xlen[b], U = ylen[b] + 1;
    const bool bad = T < 1 || U < 1 || T > maxT || U > maxU;
    Tb = bad ? 0 : T;
    Ub = bad ? 0 : U;
}

// Overlay guard of coef_kernel: per sample two words -- tiles that have finished reading the sample's lattice block, and
// "the sample may store" -- each in a 128-byte line of its own (ints apart).  Packed into four lines, the 18 240 announcements
// and as many coherent looks of a c4 call queued up at the memory side: the kernel took 1.1 ms instead of 0.28.
constexpr int kCoefDoneStride = 64;
// Kernels that put the samples on gridDim.y are launched in slices of at most kGridSamples samples (the hardware
// limit of that dimension is 65535; the reference puts the samples on gridDim.x, gpu_rnnt.h:127-128, and so takes
// any batch size): `b0` = first sample of the slice.
constexpr int kGridSamples = 65535;
constexpr int kLatPad = 16;
// Row stride of the skewed arrays: maxU rounded up to 8 cells (16-byte aligned rows for every element type; the
// lattice lanes past the row are parked on an out-of-range buffer offset, see lattice_kernel).  Until round 3 rows
// were padded to whole 64-lane wavefronts, which tripled the lattice workspace at U = 21.
__host__ __device__ inline int lat_stride(int maxU) { return (maxU + 7) & ~7; }
__host__ __device__ inline size_t lat_rows(int maxT, int maxU) { return static_cast<size_t>(maxT) + maxU - 1 + 2 * kLatPad; }
// Lattice VALUES in one sample's block [lp2: 2 per cell | logz | alpha | beta + the rows its readers overshoot]; a multiple of
// 64 values, so that every block (and every array inside it: Dp * Up is a multiple of 8) starts on a 256-byte boundary in fp32
__host__ __device__ inline size_t lat_block(int maxT, int maxU, int Up) {
    return (5 * lat_rows(maxT, maxU) * static_cast<size_t>(Up) + Up + 64 + 63) & ~static_cast<size_t>(63);
}
// offsets (in values) of the four arrays inside a block
__host__ __device__ inline size_t lat_block_logz(int maxT, int maxU, int Up) { return 2 * lat_rows(maxT, maxU) * static_cast<size_t>(Up); }
__host__ __device__ inline size_t lat_block_alpha(int maxT, int maxU, int Up) { return 3 * lat_rows(maxT, maxU) * static_cast<size_t>(Up); }
__host__ __device__ inline size_t lat_block_beta(int maxT, int maxU, int Up) { return 4 * lat_rows(maxT, maxU) * static_cast<size_t>(Up); }
// first element of sample b in a VALUE array (logz, alpha, beta: the pointer addresses sample 0's array) ...
__host__ __device__ inline size_t lat_sample(int b, int maxT, int maxU, int Up) { return static_cast<size_t>(b) * lat_block(maxT, maxU, Up); }
// ... and in the PAIR array lp2 (elements of two values: half the stride)
__host__ __device__ inline size_t lat_sample_pair(int b, int maxT, int maxU, int Up) { return static_cast<size_t>(b) * (lat_block(maxT, maxU, Up) >> 1); }
// element index of (b, n, u) in a value array / in lp2, row stride Up
__host__ __device__ inline size_t lat_index(int b, int n, int u, int maxT, int maxU, int Up) {
    return lat_sample(b, maxT, maxU, Up) + static_cast<size_t>(kLatPad + n) * Up + u;
}
__host__ __device__ inline size_t lat_pair_index(int b, int n, int u, int maxT, int maxU, int Up) {
    return lat_sample_pair(b, maxT, maxU, Up) + static_cast<size_t>(kLatPad + n) * Up + u;
}

// NON-FINITE ROW STATISTICS.  A NaN or +inf logit (or a row of -inf only) makes log Z of its row non-finite, and the
// reference then returns a NaN cost for that sample and NaN gradients on all its rows: its max / exp-sum reductions
// and log_sum_exp propagate (include/detail/reduce.h:85,103 -> gpu_rnnt_kernel.h:5-9 -> rnnt_helper.h:16-24).  Here the
// lattice works on log-probs clamped to [log zero, 0] (lat_clamp), which would turn such a row into "probability
// zero" and route the mass around it -- a finite, wrong cost.  So the statistics kernels leave a HINT: the lane that
// stores a non-finite log Z also stores (cell index inside the sample's skewed array) + 1 into poison[b]; the alpha
// block of the sample reads poison[b] when it starts and, if it is non-zero, LOOKS AT THE CELL: the sample is poisoned
// iff the hinted cell lies inside T_b x U_b and its log Z (written in THIS call: every in-lattice row is) is
// non-finite.  The workspace is undefined on entry, so a stale or random word is possible -- it costs that one
// look-up and is then zeroed; it can never poison a clean sample, and a poisoned sample always carries a valid hint
// (several bad rows race with plain stores; any of them will do).  The streaming kernels pay one compare per row.
// A poisoned sample: cost = NaN (a plain quiet NaN, not the marker of cost_invalid), llForward = NaN, hence every
// record of the coefficient kernels and every gradient of its in-lattice rows is NaN; padded rows stay zero; the
// other samples of the batch are untouched.
template <typename C> __device__ __forceinline__ bool non_finite(C v) { return !(v - v == C(0)); }
template <typename C>
__device__ __forceinline__ void note_non_finite(int* __restrict__ poison, int b, int n, int u, int Up, C logZ) {
    if (non_finite(logZ)) poison[b] = (kLatPad + n) * Up + u + 1;
}
// the alpha block's check of a non-zero hint (see above); Tb, Ub: the sample's clamped lengths
template <typename L>
__device__ __forceinline__ bool hint_is_poison(int hint, const L* __restrict__ logz, size_t sample0, int Up, int Tb, int Ub) {
    if (hint <= 0) return false;
    const int cell = hint - 1, row = cell / Up, u = cell - row * Up, t = row - kLatPad - u;
    if (u >= Ub || t < 0 || t >= Tb) return false;
    return non_finite(logz[sample0 + cell]);
}


// ------------------------------------------------------------------------------------------
// Online (max, sum-exp) accumulation of N values into a lane's running pair.
template <typename C, int N>
__device__ __forceinline__ void absorb(const C (&v)[N], C& m, C& s) {
    C mx = v[0];
#pragma unroll
    for (int i = 1; i < N; ++i) mx = vmax(mx, v[i]);
    const C mn = vmax(m, mx);
    const C shift = (mn == neg_inf<C>()) ? C(0) : mn;
    C acc = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) acc += fast_exp(v[i] - shift);
    s = s * fast_exp(m - shift) + acc;
    m = mn;
}

// Split of one row of A elements starting at byte address `addr` into
// [head scalars][nvec 16-byte packets][tail scalars] so that packets are 16-byte aligned.
template <typename S>
__device__ __forceinline__ void row_split(uintptr_t addr, int A, bool vec_ok, int& head, int& nvec, int& tail0) {
    constexpr int V = 16 / sizeof(S);
    head = static_cast<int>(((16u - static_cast<unsigned>(addr & 15u)) & 15u) / sizeof(S));
    if (!vec_ok || head > A) head = A;
    nvec = (A - head) / V;
    tail0 = head + nvec * V;
}

// ------------------------------------------------------------------------------------------
// Pass A.  grid = (ceil(maxT*maxU / WAVES), N), block = WAVES*64; one wavefront per row.
template <typename Tag, int WAVES, bool NT>
__global__ __launch_bounds__(WAVES * 64) void row_stats_kernel(
        const typename Tag::store* __restrict__ acts, const int* __restrict__ labels,
        const int* __restrict__ xlen, const int* __restrict__ ylen,
        LogPair<typename Tag::comp>* __restrict__ lp2, typename Tag::comp* __restrict__ logz,
        int maxT, int maxU, int Up, int A, int blank, int vec_ok, const long long* __restrict__ offsets,
        unsigned long long total_rows, int b0, int* __restrict__ poison) {   // packed layout: rows of the tensor (offsets are device data: never read past it); b0 = first sample of this launch; poison: note_non_finite
    using S = typename Tag::store;
    using C = typename Tag::comp;
    constexpr int V = Vec<Tag>::N;
    const int b = b0 + blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int q = uniform(blockIdx.x * WAVES + (threadIdx.x >> 6));
    if (q >= maxT * maxU) return;
    const int Tb = clamp_len(xlen[b], maxT), Ub = clamp_len(ylen[b] + 1, maxU);
    if (Tb <= 0 || Ub <= 0) return;
    // packed layout (offsets != nullptr, include/rnnt.h compute_rnnt_loss_packed): sample b is T_b * U_b
    // consecutive rows starting at row offsets[b], row (t, u) at t * U_b + u -- no padded rows exist
    const int ustride = offsets != nullptr ? Ub : maxU;
    const int t = q / ustride, u = q - t * ustride;
    if (t >= Tb || u >= Ub) return;   // padded cell: never read

    const size_t rix = offsets != nullptr ? static_cast<size_t>(offsets[b]) + q
                                          : static_cast<size_t>(b) * maxT * maxU + q;
    if (offsets != nullptr && rix >= total_rows) return;
    const S* row = acts + rix * A;
    const bool has_lab = u < Ub - 1;
    int lab = blank;
    if (has_lab) {
        lab = labels[static_cast<size_t>(b) * (maxU - 1) + u];
        lab = lab < 0 ? 0 : (lab >= A ? A - 1 : lab);
    }
    const C xb = load1<Tag>(row + blank);
    const C xl = load1<Tag>(row + lab);

    C m = neg_inf<C>(), s = 0;
    if (vec_ok) {
        // The aligned 16-byte packets that COVER the row (its first and last packet may reach into the neighbouring rows:
        // those elements are masked to -inf, which drops them from the maximum and from the sum; a packet never
        // leaves the 16-byte granule that holds an element of this row, so it stays inside mapped memory even at the
        // ends of the tensor).  Up to four packets per lane in flight.  A scalar head / tail per row, which this
        // replaces, made a 1025-symbol bf16 vocabulary 1.7x slower than a 1024-symbol one.
        const uintptr_t addr = reinterpret_cast<uintptr_t>(row);
        const int skip = static_cast<int>((addr & 15u) / sizeof(S));      // elements of the first packet before the row
        const u32x4* vp = reinterpret_cast<const u32x4*>(addr & ~static_cast<uintptr_t>(15));
        const int npk = (skip + A + V - 1) / V;
        auto finish = [&](const uint4& raw, int pidx, C* v) {            // unpack one packet, edges masked
            unpack<Tag>(raw, v);
            if (pidx == 0 || pidx == npk - 1) {
#pragma unroll
                for (int e = 0; e < V; ++e)
                    if (static_cast<unsigned>(pidx * V + e - skip) >= static_cast<unsigned>(A)) v[e] = neg_inf<C>();
            }
        };
        for (int base = 0; base < npk;) {                                  // rounds of (up to) four packets per lane
            const int remaining = npk - base;                              // wave-uniform
            const int i = base + lane;
            const int left = npk - i;                                      // this lane has packets i, i+64, ... below npk
            if (remaining > 256 && remaining <= 320) {
                // a row of 2^k + 1 symbols: the few packets past a full round ride along as a fifth load instead of taking
                // a round (= a memory round trip) of their own
                const bool has5 = i + 256 < npk;
                const uint4 r0 = load_packet<NT>(vp + i), r1 = load_packet<NT>(vp + i + 64),
                            r2 = load_packet<NT>(vp + i + 128), r3 = load_packet<NT>(vp + i + 192);
                uint4 r4 = make_uint4(0, 0, 0, 0);
                if (has5) r4 = load_packet<NT>(vp + i + 256);
                C v[4 * V];
                finish(r0, i, v); finish(r1, i + 64, v + V); finish(r2, i + 128, v + 2 * V); finish(r3, i + 192, v + 3 * V);
                absorb<C, 4 * V>(v, m, s);
                if (has5) {
                    C w[V];
                    finish(r4, i + 256, w);
                    absorb<C, V>(w, m, s);
                }
                base += 320;
                continue;
            }
            base += 256;
            // all loads of the round first (a load behind the masking branch of the previous packet is a round trip of its own)
            if (left > 192) {
                const uint4 r0 = load_packet<NT>(vp + i), r1 = load_packet<NT>(vp + i + 64),
                            r2 = load_packet<NT>(vp + i + 128), r3 = load_packet<NT>(vp + i + 192);
                C v[4 * V];
                finish(r0, i, v); finish(r1, i + 64, v + V); finish(r2, i + 128, v + 2 * V); finish(r3, i + 192, v + 3 * V);
                absorb<C, 4 * V>(v, m, s);
            } else if (left > 128) {
                const uint4 r0 = load_packet<NT>(vp + i), r1 = load_packet<NT>(vp + i + 64), r2 = load_packet<NT>(vp + i + 128);
                C v[3 * V];
                finish(r0, i, v); finish(r1, i + 64, v + V); finish(r2, i + 128, v + 2 * V);
                absorb<C, 3 * V>(v, m, s);
            } else if (left > 64) {
                const uint4 r0 = load_packet<NT>(vp + i), r1 = load_packet<NT>(vp + i + 64);
                C v[2 * V];
                finish(r0, i, v); finish(r1, i + 64, v + V);
                absorb<C, 2 * V>(v, m, s);
            } else if (left > 0) {
                const uint4 r0 = load_packet<NT>(vp + i);
                C v[V];
                finish(r0, i, v);
                absorb<C, V>(v, m, s);
            }
        }
    } else {
        for (int e = lane; e < A; e += 64) {                               // elements not even element-aligned packets: scalar
            C v[1] = {load1<Tag>(row + e)};
            absorb<C, 1>(v, m, s);
        }
    }

    const C M = wave_max(m);
    const C shift = (M == neg_inf<C>()) ? C(0) : M;
    const C S_ = wave_sum(s * fast_exp(m - shift));
    const C logZ = shift + acc_log(S_);

    if (lane == 0) {
        LogPair<C> rec;                                // lattice log-probs are kept in base 2
        rec.x = vmax((xb - logZ) * C(kLog2e), log_zero<C>());
        rec.y = has_lab ? vmax((xl - logZ) * C(kLog2e), log_zero<C>()) : log_zero<C>();
        const size_t idx = lat_index(b, t + u, u, maxT, maxU, Up);
        lp2[lat_pair_index(b, t + u, u, maxT, maxU, Up)] = rec;
        logz[idx] = logZ;
        note_non_finite(poison, b, t + u, u, Up, logZ);
    }
}

// ------------------------------------------------------------------------------------------
// Pass A, BLOCK form for long rows: the 256 threads of a block share ONE row (thread i takes packets
// i, i + 256, ...), so a row is consumed in one or two load rounds and the rows being read at any
// moment form one contiguous window of the tensor -- the access pattern of a flat copy -- instead of
// thousands of 1 KB pieces 20 KB apart.  Partial (max, sum) pairs meet in LDS.
// grid = (maxT*maxU, N), block = 256.
template <typename Tag, bool NT, int K>   // K = packets in flight per thread
__global__ __launch_bounds__(256) void row_stats_block_kernel(
        const typename Tag::store* __restrict__ acts, const int* __restrict__ labels,
        const int* __restrict__ xlen, const int* __restrict__ ylen,
        LogPair<typename Tag::comp>* __restrict__ lp2, typename Tag::comp* __restrict__ logz,
        int maxT, int maxU, int Up, int A, int blank, int vec_ok, const long long* __restrict__ offsets,
        unsigned long long total_rows, int b0, int* __restrict__ poison) {   // packed layout: rows of the tensor (offsets are device data: never read past it); b0 = first sample of this launch; poison: note_non_finite
    using S = typename Tag::store;
    using C = typename Tag::comp;
    constexpr int V = Vec<Tag>::N;
    __shared__ C red_m[4], red_s[4];
    const int b = b0 + blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x;
    const int Tb = clamp_len(xlen[b], maxT), Ub = clamp_len(ylen[b] + 1, maxU);
    if (Tb <= 0 || Ub <= 0) return;
    const int ustride = offsets != nullptr ? Ub : maxU;      // packed layout: see row_stats_kernel
    const int t = q / ustride, u = q - t * ustride;
    if (t >= Tb || u >= Ub) return;   // padded cell: never read (block-uniform)

    const size_t rix = offsets != nullptr ? static_cast<size_t>(offsets[b]) + q
                                          : static_cast<size_t>(b) * maxT * maxU + q;
    if (offsets != nullptr && rix >= total_rows) return;
    const S* row = acts + rix * A;
    int head, nvec, tail0;
    row_split<S>(reinterpret_cast<uintptr_t>(row), A, vec_ok != 0, head, nvec, tail0);

    C m = neg_inf<C>(), s = 0;
    for (int e = tid; e < head; e += 256) {
        C v[1] = {load1<Tag>(row + e)};
        absorb<C, 1>(v, m, s);
    }
    const u32x4* vp = reinterpret_cast<const u32x4*>(row + head);
    for (int i = tid; i < nvec; i += 256 * K) {
        uint4 r[K];
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (i + 256 * k < nvec) r[k] = load_packet<NT>(vp + i + 256 * k);
        C v[K * V];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (i + 256 * k < nvec) {
                unpack<Tag>(r[k], v + k * V);
            } else {
#pragma unroll
                for (int j = 0; j < V; ++j) v[k * V + j] = neg_inf<C>();
            }
        }
        absorb<C, K * V>(v, m, s);
    }
    for (int e = tail0 + tid; e < A; e += 256) {
        C v[1] = {load1<Tag>(row + e)};
        absorb<C, 1>(v, m, s);
    }

    const C Mw = wave_max(m);
    const C shw = (Mw == neg_inf<C>()) ? C(0) : Mw;
    const C Sw = wave_sum(s * fast_exp(m - shw));
    if (lane == 0) { red_m[wave] = Mw; red_s[wave] = Sw; }
    __syncthreads();
    if (tid == 0) {
        const C M = vmax(vmax(red_m[0], red_m[1]), vmax(red_m[2], red_m[3]));
        const C shift = (M == neg_inf<C>()) ? C(0) : M;
        C S_ = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) S_ += red_s[w] * fast_exp(((red_m[w] == neg_inf<C>()) ? shift : red_m[w]) - shift);
        const C logZ = shift + acc_log(S_);
        const bool has_lab = u < Ub - 1;
        int lab = blank;
        if (has_lab) {
            lab = labels[static_cast<size_t>(b) * (maxU - 1) + u];
            lab = lab < 0 ? 0 : (lab >= A ? A - 1 : lab);
        }
        const C xb = load1<Tag>(row + blank);
        const C xl = load1<Tag>(row + lab);
        LogPair<C> rec;                                // lattice log-probs are kept in base 2
        rec.x = vmax((xb - logZ) * C(kLog2e), log_zero<C>());
        rec.y = has_lab ? vmax((xl - logZ) * C(kLog2e), log_zero<C>()) : log_zero<C>();
        const size_t idx = lat_index(b, t + u, u, maxT, maxU, Up);
        lp2[lat_pair_index(b, t + u, u, maxT, maxU, Up)] = rec;
        logz[idx] = logZ;
        note_non_finite(poison, b, t + u, u, Up, logZ);
    }
}

// ------------------------------------------------------------------------------------------
// Pass A, TILE form for short rows (row bytes <= kTileMaxRowBytes).  A wavefront per row wastes
// lanes and pays a 12-shuffle reduction per row when a row is only tens to hundreds of bytes,
// so here a 256-thread block streams a contiguous tile of RT = 256/G rows (a flat, fully
// coalesced range of the tensor) into LDS with non-temporal 16-byte loads, then G lanes per row
// reduce it out of LDS: pass 1 max, pass 2 sum-exp.  Lane-to-element order is rotated by the
// row number when A is even so that the RT rows fall into different LDS banks.
// grid = ceil(R / RT), dynamic LDS = RT*A*s + 32 bytes.
// Reduce phase of the tile kernel for rows made of whole LDS words (W = uint4: 16-byte packets,
// uint2: 8-byte words; NE elements each).  Lane j of the row's G lanes owns words j, j+G, ...  When a
// lane owns at most kTileRegWords of them it reads them ONCE, back to back, into registers and runs
// both passes (max, then sum of exp2) from there; counters on c4 showed the two rolled passes over LDS
// at 12.8 VALU instructions per element (loop control + addressing per 8-byte read, twice).  A word
// index past the row is clamped to the last word (harmless for the max) and its sum is dropped.
constexpr int kTileRegWords = 16;   // one-shot kernel; halved for 16-bit storage (the unpacked values stay live)

template <typename Tag, typename W, int NE, int KW, typename Unpack>
__device__ __forceinline__ void tile_reduce_words(const W* __restrict__ words, int nw, int j, int G,
                                                  typename Tag::comp& m_out, typename Tag::comp& shift_out,
                                                  typename Tag::comp& sum_out, Unpack unpack_word) {
    using C = typename Tag::comp;
    C m = neg_inf<C>(), sum = 0, shift = 0;
    if (nw <= KW * G) {
        W r[KW];
#pragma unroll
        for (int i = 0; i < KW; ++i) {
            const int p = j + i * G;
            r[i] = words[p < nw ? p : nw - 1];
        }
#pragma unroll
        for (int i = 0; i < KW; ++i) {
            C v[NE];
            unpack_word(r[i], v, j + i * G);
#pragma unroll
            for (int e = 0; e < NE; ++e) m = vmax(m, v[e]);
        }
        for (int off = G / 2; off > 0; off >>= 1) m = vmax(m, __shfl_xor(m, off, kWave));
        shift = (m == neg_inf<C>()) ? C(0) : m;
        const C sh2 = -shift * C(kLog2e);
#pragma unroll
        for (int i = 0; i < KW; ++i) {
            C v[NE];
            unpack_word(r[i], v, j + i * G);
            C ps = 0;
#pragma unroll
            for (int e = 0; e < NE; ++e) ps += fast_exp2(v[e] * C(kLog2e) + sh2);
            sum += (j + i * G < nw) ? ps : C(0);
        }
    } else {
        for (int p = j; p < nw; p += G) {
            C v[NE];
            unpack_word(words[p], v, p);
#pragma unroll
            for (int e = 0; e < NE; ++e) m = vmax(m, v[e]);
        }
        for (int off = G / 2; off > 0; off >>= 1) m = vmax(m, __shfl_xor(m, off, kWave));
        shift = (m == neg_inf<C>()) ? C(0) : m;
        const C sh2 = -shift * C(kLog2e);
        for (int p = j; p < nw; p += G) {
            C v[NE];
            unpack_word(words[p], v, p);
#pragma unroll
            for (int e = 0; e < NE; ++e) sum += fast_exp2(v[e] * C(kLog2e) + sh2);
        }
    }
    m_out = m; shift_out = shift; sum_out = sum;
}

constexpr int kTileMaxRowBytes = 4096;   // (2048 until the reduce phase learned rows at any byte phase: A/B on 2-4 KB rows in EXPERIMENTS.md 6)

template <typename Tag, int G>
__global__ __launch_bounds__(256) void row_stats_tile_kernel(
        const typename Tag::store* __restrict__ acts, const int* __restrict__ labels,
        const int* __restrict__ xlen, const int* __restrict__ ylen,
        LogPair<typename Tag::comp>* __restrict__ lp2, typename Tag::comp* __restrict__ logz,
        unsigned long long R, int maxT, int maxU, int Up, int A, int blank, int xcd_remap,
        const long long* __restrict__ offsets, int N, int* __restrict__ poison) {    // offsets != nullptr: packed layout, R = offsets[N] rows; poison: note_non_finite
    using S = typename Tag::store;
    using C = typename Tag::comp;
    constexpr int V = Vec<Tag>::N;
    constexpr int RT = 256 / G;
    extern __shared__ uint4 tile_raw[];
    S* tile = reinterpret_cast<S*>(tile_raw);

    // XCD-aware tile order (workgroup i runs on XCD i % 8, each XCD has its own L2): every XCD walks
    // ONE contiguous range of tiles.  Neighbouring cells of a skewed row come from tensor rows maxU-1
    // apart, i.e. from tiles a few positions apart that run at the same time; in launch order they sit
    // on different XCDs and each L2 writes its partial line back, with this order they meet in one L2
    // and leave as full lines (the scattered result stores are what bounds this kernel on c4).
    const unsigned ntile = static_cast<unsigned>((R + RT - 1) / RT);
    const unsigned per = (ntile + 7u) >> 3;
    const unsigned tile_id = (xcd_remap & 1) ? (blockIdx.x & 7u) * per + (blockIdx.x >> 3) : blockIdx.x;
    if (tile_id >= ntile) return;                          // grid is rounded up to a multiple of 8
    const unsigned long long r0 = static_cast<unsigned long long>(tile_id) * RT;
    const int nrows = static_cast<int>(R - r0 < static_cast<unsigned long long>(RT) ? R - r0 : RT);
    const S* base = acts + r0 * static_cast<unsigned>(A);
    const int n_el = nrows * A;
    const int phase = static_cast<int>((reinterpret_cast<uintptr_t>(base) & 15u) / sizeof(S));
    int head = (V - phase) % V;
    if (head > n_el) head = n_el;
    const int nbody = (n_el - head) / V;
    const int tail0 = head + nbody * V;

    // ---- global -> LDS, the tile keeps the 16-byte phase of its global address.  ALL packets of the
    // tile are requested before the first one is stored (one memory round trip per block, not one per
    // four packets), then the row metadata, then everything drains into LDS.
    for (int e = threadIdx.x; e < head; e += 256) tile[phase + e] = base[e];
    constexpr int kTilePk = 12;                            // packets per thread: tiles up to 48 KB
    uint4 pk[kTilePk];
    const u32x4* src = reinterpret_cast<const u32x4*>(base + head);
#pragma unroll
    for (int i = 0; i < kTilePk; ++i)
        if (i * 256 < nbody) {                             // block-uniform
            const int pi = i * 256 + static_cast<int>(threadIdx.x);
            pk[i] = load_packet<true>(src + (pi < nbody ? pi : nbody - 1));
        }
    __builtin_amdgcn_sched_barrier(0);                     // the packets go out before the index arithmetic below
    // The row's position, lengths and label: requested right behind the tile's packets, by every lane,
    // as three independent loads (the label index is clamped instead of depending on the lengths), so
    // they have long arrived when the epilogue needs them.  Done at the end, as two dependent global
    // latencies, they held the block's LDS tile for 2.6 us of its 8.7 us life (c4).
    const int rl = threadIdx.x / G, j = threadIdx.x % G;
    int b, t, u, Tb, Ub;
    if (offsets == nullptr) {
        const unsigned TU = static_cast<unsigned>(maxT) * maxU;
        const unsigned long long b0 = r0 / TU;             // block-uniform
        unsigned q = static_cast<unsigned>(r0 - b0 * TU) + static_cast<unsigned>(rl);
        b = static_cast<int>(b0);
        while (q >= TU) { q -= TU; ++b; }
        const int nb = static_cast<int>(R / TU);
        b = b < nb ? b : nb - 1;                           // lanes past the last row: any valid sample
        t = static_cast<int>(q / static_cast<unsigned>(maxU));
        u = static_cast<int>(q) - t * maxU;
        Tb = clamp_len(xlen[b], maxT); Ub = clamp_len(ylen[b] + 1, maxU);
    } else {
        // Packed layout: the sample of the tile's first row by a block-uniform (scalar) binary search over the
        // cumulative row offsets, its neighbour's data as scalars too; a lane's row is in one of the two unless
        // the tile spans three or more samples (tiny samples), which takes the per-lane search.
        int lo = 0, hi = N;                                // offsets[lo] <= r0 < offsets[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (r0 >= static_cast<unsigned long long>(offsets[mid])) lo = mid; else hi = mid;
        }
        const int b1 = lo + 1 < N ? lo + 1 : lo;
        const unsigned long long o0 = static_cast<unsigned long long>(offsets[lo]);
        const unsigned long long o1 = static_cast<unsigned long long>(offsets[lo + 1]);
        const int T0 = clamp_len(xlen[lo], maxT), U0 = clamp_len(ylen[lo] + 1, maxU), T1 = clamp_len(xlen[b1], maxT), U1 = clamp_len(ylen[b1] + 1, maxU);
        const unsigned long long r = r0 + static_cast<unsigned>(rl);
        unsigned long long ob = o0;
        b = lo; Tb = T0; Ub = U0;
        if (r >= o1 && lo + 1 < N) {
            b = b1; Tb = T1; Ub = U1; ob = o1;
            if (b + 1 < N && r >= static_cast<unsigned long long>(offsets[b + 1])) {
                while (b + 1 < N && r >= static_cast<unsigned long long>(offsets[b + 1])) ++b;
                ob = static_cast<unsigned long long>(offsets[b]);
                Tb = clamp_len(xlen[b], maxT); Ub = clamp_len(ylen[b] + 1, maxU);
            }
        }
        const unsigned q = static_cast<unsigned>(r - ob);  // rows past the end: t >= Tb, never stored
        const unsigned us = Ub > 0 ? static_cast<unsigned>(Ub) : 1u;
        t = static_cast<int>(q / us);
        u = static_cast<int>(q - static_cast<unsigned>(t) * us);
    }
    int lab = labels[maxU > 1 ? static_cast<size_t>(b) * (maxU - 1) + (u < maxU - 1 ? u : maxU - 2) : 0];
    {
        uint4* dst = tile_raw + (phase + head) / V;
#pragma unroll
        for (int i = 0; i < kTilePk; ++i) {
            const int pi = i * 256 + static_cast<int>(threadIdx.x);
            if (i * 256 < nbody && pi < nbody) dst[pi] = pk[i];
        }
        for (int pi = kTilePk * 256 + threadIdx.x; pi < nbody; pi += 256) dst[pi] = load_packet<true>(src + pi);   // > 48 KB tiles
    }
    for (int e = tail0 + threadIdx.x; e < n_el; e += 256) tile[phase + e] = base[e];
    __syncthreads();

    // ---- G lanes per row
    if (rl >= nrows) return;                                   // whole lane groups leave together
    const S* rowp = tile + phase + rl * A;
    C m = neg_inf<C>(), sum = 0, shift = 0;
    constexpr int H = V / 2;                                   // elements per 8-byte LDS read
    // The reduce phase is instruction-bound (measured on c4: loads alone 0.82 ms, reduce alone
    // 1.06 ms before this form, 0.5 ms with it), so it reads LDS 16 or 8 bytes at a time where the
    // row allows it and forms exp(x - shift) as exp2(x*log2e - shift*log2e): one fma + one v_exp_f32.
    if (phase == 0 && A % V == 0) {
        // rows are whole 16-byte packets: ds_read_b128
        constexpr int KW = sizeof(S) == 2 ? kTileRegWords / 2 : kTileRegWords;
        tile_reduce_words<Tag, uint4, V, KW>(tile_raw + rl * (A / V), A / V, j, G, m, shift, sum,
                                             [](const uint4& w, C* v, int) { unpack<Tag>(w, v); });
    } else if (H > 1 && (phase % H) == 0 && A % H == 0) {
        // rows are whole 8-byte words: ds_read_b64
        constexpr int KW = sizeof(S) == 2 ? kTileRegWords / 2 : kTileRegWords;
        tile_reduce_words<Tag, uint2, (H > 1 ? H : 1), KW>(reinterpret_cast<const uint2*>(tile_raw) + (phase + rl * A) / H,
                                                       A / H, j, G, m, shift, sum,
                                                       [](const uint2& w, C* v, int) { unpack_half<Tag>(w, v); });
    } else if (H > 1) {
        // any other row position (odd vocabularies, 2- or 4-byte phases): the aligned 8-byte words that COVER the row, the
        // elements of the first and last word that belong to the neighbouring rows masked to -inf (they then drop out of
        // the maximum and of the sum alike).  An element-by-element loop here made a 1023-symbol bf16 vocabulary 2.4x
        // slower than a 1024-symbol one.
        constexpr int KW = sizeof(S) == 2 ? kTileRegWords / 2 : kTileRegWords;
        constexpr int HH = H > 1 ? H : 1;
        const int e0 = phase + rl * A;                         // the row's first element, counted from tile_raw
        const int w0 = e0 / HH, skip = e0 - w0 * HH;
        const int nw = (skip + A + HH - 1) / HH;
        // (8-byte words: with 16-byte words the same scheme measured 1.5-1.8x slower -- bf16 A = 1023 0.655 vs 0.450 ms,
        // fp32 A = 511 0.654 vs 0.358)
        tile_reduce_words<Tag, uint2, HH, KW>(reinterpret_cast<const uint2*>(tile_raw) + w0, nw, j, G, m, shift, sum,
                                              [skip, nw, A](const uint2& w, C* v, int p) {
                                                  unpack_half<Tag>(w, v);
                                                  if (p == 0 || p >= nw - 1) {
#pragma unroll
                                                      for (int e = 0; e < HH; ++e)
                                                          if (static_cast<unsigned>(p * HH + e - skip) >= static_cast<unsigned>(A))
                                                              v[e] = neg_inf<C>();
                                                  }
                                              });
    } else {
        const int rot = (A & 1) ? 0 : (rl % A);
        for (int e = j; e < A; e += G) {
            int pos = e + rot;
            if (pos >= A) pos -= A;
            m = vmax(m, load1<Tag>(rowp + pos));
        }
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) m = vmax(m, __shfl_xor(m, off, kWave));
        shift = (m == neg_inf<C>()) ? C(0) : m;
        const C sh2 = -shift * C(kLog2e);
        for (int e = j; e < A; e += G) {
            int pos = e + rot;
            if (pos >= A) pos -= A;
            sum += fast_exp2(load1<Tag>(rowp + pos) * C(kLog2e) + sh2);
        }
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, kWave);

    if (j == 0 && t < Tb && u < Ub) {
        const C logZ = shift + fast_log(sum);
        const bool has_lab = u < Ub - 1;
        lab = has_lab ? (lab < 0 ? 0 : (lab >= A ? A - 1 : lab)) : blank;
        LogPair<C> rec;                            // lattice log-probs are kept in base 2
        rec.x = vmax((load1<Tag>(rowp + blank) - logZ) * C(kLog2e), log_zero<C>());
        rec.y = has_lab ? vmax((load1<Tag>(rowp + lab) - logZ) * C(kLog2e), log_zero<C>()) : log_zero<C>();
        // two scattered stores per row: they combine into full lines in the XCD's L2 (tile order above)
        const size_t idx = lat_index(b, t + u, u, maxT, maxU, Up);
        const size_t pidx = lat_pair_index(b, t + u, u, maxT, maxU, Up);
#ifdef RNNT_DEV
        // development build only (RNNT_TUNE=xst=..): timing experiments on the result stores, results are WRONG
        const int xst = xcd_remap >> 4;
        if (xst == 1) { reinterpret_cast<Cell<C>*>(lp2)[pidx >> 1] = Cell<C>{rec.x, rec.y, logZ, C(0)}; return; }   // one 16-byte record
        if (xst == 2) { const size_t nat = r0 + rl; lp2[nat] = rec; logz[nat] = logZ; return; }              // natural order
        if (xst == 3) { if (rec.x == C(12345)) logz[idx] = logZ; return; }                                   // no stores
#endif
        lp2[pidx] = rec;
        logz[idx] = logZ;
        note_non_finite(poison, b, t + u, u, Up, logZ);
    }
}

// ------------------------------------------------------------------------------------------
// Pass A, 2-D CELL-TILE form for short rows under wide lattices (c4: 200-byte rows, U = 301).  The flat tile kernel
// above streams 256 consecutive rows -- less than one time row of such a lattice -- so its 256 results belong to 256
// different anti-diagonals and leave as two scattered stores per row, 64 distinct lines per store instruction: request
// rate, not bytes, is what bounds it there (EXPERIMENTS.md 3: 1.05 ms as shipped, 0.90 with coalesced stores, 0.78 with
// none).  Here a block owns TT time rows x TU label rows of ONE sample: TT contiguous pieces of TU rows each, loaded as
// the aligned 16-byte packets that cover them (all requested before the first is stored, as above), one lane per row
// for the reduction, and the 256 results are turned through LDS so that they leave ALONG the anti-diagonals: the
// cells (t0 + i, u0 + d - i), i = 0 .. TT-1, of a tile's diagonal d are TT consecutive elements of a skewed row.
// Rows are whole 8-byte words (A * s % 8 == 0, 8-byte aligned tensor); non-packed layout; LDS = TT * piece + results.
// grid = (8 * ceil(tiles / 8), 1, 1) with tiles = N * ceil(maxT / TT) * ceil(maxU / TU), XCD-aware tile order, block = 256.
#ifndef RNNT_TILE2D_KW
#define RNNT_TILE2D_KW 26
#endif
template <typename Tag, int TT, int TU>
__global__ __launch_bounds__(256) void row_stats_tile2d_kernel(
        const typename Tag::store* __restrict__ acts, const int* __restrict__ labels,
        const int* __restrict__ xlen, const int* __restrict__ ylen,
        LogPair<typename Tag::comp>* __restrict__ lp2, typename Tag::comp* __restrict__ logz,
        int maxT, int maxU, int Up, int A, int blank, int N, int tilesT, int tilesU, int piece_bytes, int* __restrict__ poison,
        int order) {                 // tile order, see below
    using S = typename Tag::store;
    using C = typename Tag::comp;
    static_assert(TT * TU == 256, "one lane per row");
    constexpr int H = 8 / sizeof(S);                           // elements per 8-byte word
    constexpr int PK = (TU * 208 + 30 + 4095) / 4096;          // packets per thread and piece (rows <= 208 bytes: host check)
    extern __shared__ uint4 tile2_raw[];
    // (the results overlay the tile once every lane has finished reading it: 3 KB less LDS, three blocks per CU at c4's size)
    LogPair<C> (*out_lp)[TU + 1] = reinterpret_cast<LogPair<C> (*)[TU + 1]>(tile2_raw);
    C (*out_lz)[TU + 1] = reinterpret_cast<C (*)[TU + 1]>(reinterpret_cast<char*>(tile2_raw) + sizeof(LogPair<C>) * TT * (TU + 1));
    // Tile order (workgroup i runs on XCD i % 8): 0 = every XCD owns one contiguous eighth of the BATCH's tiles (eight windows,
    // hundreds of MB apart); 1 = plain (the eight XCDs share one moving window); 2 = every XCD owns one contiguous eighth of
    // each SAMPLE's tiles (grid = N * 8 * ceil(tiles per sample / 8)).
    const unsigned ntile = static_cast<unsigned>(N) * tilesT * tilesU;
    unsigned tile_id;
    if ((order & 3) == 1) {
        tile_id = blockIdx.x;
    } else if ((order & 3) == 2) {
        const unsigned pts = static_cast<unsigned>(tilesT) * tilesU, per2 = (pts + 7u) >> 3;
        const unsigned bs = blockIdx.x / (8u * per2), r = blockIdx.x - bs * 8u * per2;
        const unsigned within = (r & 7u) * per2 + (r >> 3);
        if (within >= pts) return;
        tile_id = bs * pts + within;
    } else {
        const unsigned per = (ntile + 7u) >> 3;
        tile_id = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    }
    if (tile_id >= ntile) return;                              // grid is rounded up to a multiple of 8
    const int tu = static_cast<int>(tile_id % tilesU), tt = static_cast<int>((tile_id / tilesU) % tilesT);
    const int b = static_cast<int>(tile_id / (static_cast<unsigned>(tilesU) * tilesT));
    const int t0 = tt * TT, u0 = tu * TU;
    const int Tb = clamp_len(xlen[b], maxT), Ub = clamp_len(ylen[b] + 1, maxU);
    if (t0 >= Tb || u0 >= Ub) return;                          // tile of padding: never read
    const int nu = Ub - u0 < TU ? Ub - u0 : TU;                // label rows of this tile inside the sample
    const int nt = Tb - t0 < TT ? Tb - t0 : TT;
    const size_t row_bytes = static_cast<size_t>(A) * sizeof(S);
    const int span = nu * static_cast<int>(row_bytes);         // bytes of one piece

    // ---- global -> LDS: piece i = rows (t0 + i, u0 .. u0 + nu - 1), kept at the 16-byte phase of its global address
    const char* base0 = reinterpret_cast<const char*>(acts) + ((static_cast<size_t>(b) * maxT + t0) * maxU + u0) * row_bytes;
    const size_t tstride = static_cast<size_t>(maxU) * row_bytes;
    uint4 pk[TT][PK];
#pragma unroll
    for (int i = 0; i < TT; ++i) {
        const char* start = base0 + static_cast<size_t>(i < nt ? i : nt - 1) * tstride;
        const int phase = static_cast<int>(reinterpret_cast<uintptr_t>(start) & 15u);
        const u32x4* src = reinterpret_cast<const u32x4*>(start - phase);
        const int npk = (phase + span + 15) >> 4;              // <= piece_bytes / 16 <= 256 PK
#pragma unroll
        for (int k = 0; k < PK; ++k) {
            const int pi = k * 256 + static_cast<int>(threadIdx.x);
            pk[i][k] = load_packet<true>(src + (pi < npk ? pi : npk - 1));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const int i_row = threadIdx.x / TU, j_row = threadIdx.x % TU;     // this lane's row of the tile
    const int u = u0 + j_row, t = t0 + i_row;
    int lab = labels[maxU > 1 ? static_cast<size_t>(b) * (maxU - 1) + (u < maxU - 1 ? u : maxU - 2) : 0];
    char* lds = reinterpret_cast<char*>(tile2_raw);
#pragma unroll
    for (int i = 0; i < TT; ++i) {
        const char* start = base0 + static_cast<size_t>(i < nt ? i : nt - 1) * tstride;
        const int phase = static_cast<int>(reinterpret_cast<uintptr_t>(start) & 15u);
        const int npk = (phase + span + 15) >> 4;
        uint4* dst = reinterpret_cast<uint4*>(lds + static_cast<size_t>(i) * piece_bytes);
#pragma unroll
        for (int k = 0; k < PK; ++k) {
            const int pi = k * 256 + static_cast<int>(threadIdx.x);
            if (pi < npk) dst[pi] = pk[i][k];
        }
    }
    __syncthreads();

    // ---- one lane per row: rows are whole 8-byte words at any 8-byte phase
    const bool live = i_row < nt && j_row < nu;
    C m = neg_inf<C>(), sum = 0, shift = 0;
    {
        const char* start = base0 + static_cast<size_t>(i_row < nt ? i_row : nt - 1) * tstride;
        const int phase = static_cast<int>(reinterpret_cast<uintptr_t>(start) & 15u);
        const char* rowp = lds + static_cast<size_t>(i_row) * piece_bytes + phase + static_cast<size_t>(j_row < nu ? j_row : nu - 1) * row_bytes;
        // one lane per row: the whole row (<= 208 bytes = 26 words) is read ONCE, back to back, into registers (LDS limits
        // this kernel to three wavefronts per SIMD, so the 52 registers are free); with the flat tile kernel's 16 words the
        // 25-word rows of c4 fell into the two rolled passes over LDS (12.8 VALU instructions per element)
        constexpr int KW = RNNT_TILE2D_KW;
        tile_reduce_words<Tag, uint2, H, KW>(reinterpret_cast<const uint2*>(rowp), A / H, 0, 1, m, shift, sum,
                                             [](const uint2& w, C* v, int) { unpack_half<Tag>(w, v); });
        const C logZ = shift + fast_log(sum);
        const bool has_lab = u < Ub - 1;
        lab = has_lab ? (lab < 0 ? 0 : (lab >= A ? A - 1 : lab)) : blank;
        const S* rp = reinterpret_cast<const S*>(rowp);
        LogPair<C> rec;                                // lattice log-probs are kept in base 2
        rec.x = vmax((load1<Tag>(rp + blank) - logZ) * C(kLog2e), log_zero<C>());
        rec.y = has_lab ? vmax((load1<Tag>(rp + lab) - logZ) * C(kLog2e), log_zero<C>()) : log_zero<C>();
#ifdef RNNT_DEV
        // development build only (RNNT_TUNE=xst=2): timing experiment -- results stored in NATURAL row order (coalesced 32-cell
        // runs), which the lattice kernel cannot read: the numbers that come out are WRONG
        if (poison == nullptr) {
            if (live) { const size_t nat = (static_cast<size_t>(b) * maxT + t) * maxU + u; lp2[nat] = rec; logz[nat] = logZ; }
            return;
        }
#endif
        __syncthreads();                               // every lane is done with the tile: the results may overlay it
        if (live) {
            out_lp[i_row][j_row] = rec;
            out_lz[i_row][j_row] = logZ;
        }
    }
    __syncthreads();
    // ---- results leave along the anti-diagonals: slot = (diagonal d of the tile, time row i), TT consecutive lanes per diagonal
    constexpr int NSLOT = (TT + TU - 1) * TT;
#pragma unroll
    for (int s0 = 0; s0 < NSLOT; s0 += 256) {
        const int slot = s0 + static_cast<int>(threadIdx.x);
        const int d = slot / TT, i = slot % TT, j = d - i;
        if (slot < NSLOT && j >= 0 && j < nu && i < nt) {
            const size_t idx = lat_index(b, t0 + u0 + d, u0 + j, maxT, maxU, Up);
            const C lz = out_lz[i][j];
            lp2[lat_pair_index(b, t0 + u0 + d, u0 + j, maxT, maxU, Up)] = out_lp[i][j];
            logz[idx] = lz;
            note_non_finite(poison, b, t0 + u0 + d, u0 + j, Up, lz);
        }
    }
    (void)t;
}

// ------------------------------------------------------------------------------------------
// Lattice recursion.  grid = N * dirs (dirs = 2: the alpha block and the beta block of a sample
// run concurrently; dirs = 1: alpha only, forward scoring), block = Up = 64*ceil(maxU/64)
// threads; thread u owns lattice column u, a wavefront owns 64 columns.
//
// A wavefront walks the anti-diagonals with its values in registers (base-2 logs):
//   alpha: a(n,u) = log2add(a(n-1,u) + lp_blank(n-1,u), a(n-1,u-1) + lp_label(n-1,u-1))
//   beta : b(n,u) = log2add(b(n+1,u) + lp_blank(n,u),   b(n+1,u+1) + lp_label(n,u))
// (n = t+u; the skewed layout makes every per-diagonal access one coalesced row, addressed as
// buffer base + scalar row offset + per-lane column offset).  The neighbour value moves one lane
// through a DPP wave shift.  The step is stripped to its ten arithmetic instructions plus one load
// and one store (a lone wavefront pays ~2 ns per issued instruction of any kind, a dependent VALU op
// 3.5 ns; on long lattices the rest of the time is memory: EXPERIMENTS 13, the ablation table):
//   * NO validity tests.  All state starts at the finite "log zero" sentinel and every loaded
//     log-prob is clamped to [sentinel, 0] (one v_med3_f32), so columns that have not started
//     yet stay at "zero", and whatever is computed for cells outside the T_b x U_b lattice can
//     only flow further outside it (alpha moves to larger t,u; beta to smaller): probability
//     mass that leaves the lattice never comes back, and none exists outside it to begin with.
//   * Diagonals are processed in chunks of C with two register buffers: the log-probs of chunk
//     j+1 are fetched while chunk j computes, one row per step, and the C results of a chunk are
//     stored during the next chunk, one row per step; the fp32 lattice issues both by hand and waits
//     with the exact in-order count (LatIO, lattice_body: the compiler's own wait insertion made every
//     chunk wait for its own prefetch).
//   * At the end of a chunk the wavefront re-centres its values on their maximum (DPP reduction)
//     and adds the shift to its fp64 offset (off[b][wave][n]); fp32 values stay O(10) however
//     long T+U is, so the round-off of an un-scaled fp32 recursion (5e-3 on grads at T=1500,
//     U=301 in the reference's own fp32 CPU path) does not build up.
//   * U > 64: the wavefronts of a block form a skewed pipeline -- in time slot s wavefront w
//     works on chunk s-w (alpha; mirrored for beta) and reads the C boundary values its
//     neighbour produced in slot s-1 from a 2-deep LDS ring (converted between the two
//     wavefronts' offsets), so the block synchronises once per C diagonals instead of once per
//     diagonal (the reference barriers every diagonal: gpu_rnnt_kernel.h:26-40).
//   * COLS lattice columns per lane.  A lattice up to 64 columns wide is one wavefront with one column per
//     lane (no synchronisation at all).  Wider lattices give every lane TWO adjacent columns: half the
//     wavefronts (U = 301: three instead of five, so no SIMD hosts two of them and sets the pace of the
//     pipeline), one DPP shift per two cells (column 2l+1 takes its neighbour from the lane's own
//     registers), two independent log2_add chains per lane to fill the issue slots a single dependent
//     chain leaves empty, and 16-byte loads / 8-byte stores per lane.  maxU <= 1024 then needs at most
//     eight wavefronts, so no 1024-thread instantiation (and its 128-register ceiling) exists.
// Chunk length: 16 diagonals for an fp32 lattice, 8 for fp64; halved for two-column lanes in blocks of more
// than four wavefronts (two wavefronts per SIMD: 256 registers per lane; the two chunk buffers and the result
// history of a two-column lane are 10 values per diagonal).
template <typename L, int MAXW, int COLS> struct LatChunk {
    static constexpr int C = sizeof(L) == 4 ? (MAXW <= 4 ? 16 : 16 / (COLS > 1 ? 2 : 1)) : 8 / ((COLS > 1 && MAXW > 4) ? 2 : 1);
};

typedef unsigned int lat_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int lat_u32x4 __attribute__((ext_vector_type(4)));

// Row-addressed access to the skewed arrays through buffer instructions: the descriptor holds the
// sample's base, `soff` is the scalar byte offset of a row, `voff` the lane's byte offset in it (the only
// part the hardware range-checks: a lane parked at kLatOob neither loads nor stores).
constexpr int kLatOob = 0x7ffffff0;
__device__ __forceinline__ float lat_clamp(float v) { return __builtin_amdgcn_fmed3f(v, log_zero<float>(), 0.0f); }
__device__ __forceinline__ double lat_clamp(double v) { return fmin(fmax(v, log_zero<double>()), 0.0); }
__device__ __forceinline__ double lat_f64(unsigned lo, unsigned hi) { return __hiloint2double(static_cast<int>(hi), static_cast<int>(lo)); }

// The fp32 lattice issues its row loads and stores as inline assembly and counts them itself (kHand): the sweep keeps the NEXT
// chunk's operand rows in flight while it works on this one, and the compiler's own wait insertion cannot express that -- it
// merged the waits for the rows of a chunk into `s_waitcnt vmcnt(15 - k)` placed AFTER the stores and loads the chunk had just
// issued, so every chunk waited a full memory round trip for its own prefetch (round 6: 0.275 -> see EXPERIMENTS 13).  The
// counter is in order: with F younger accesses issued after the rows that are needed, `s_waitcnt vmcnt(F)` is exact.  `Raw` is a row
// as it arrives (the clamp to [log zero, 0] is applied when the value is used); the fp64 lattice keeps the compiler-tracked builtins.
typedef int lat_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ lat_i32x4 lat_desc(const void* p, int bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    return lat_i32x4{__builtin_amdgcn_readfirstlane(static_cast<int>(a)), __builtin_amdgcn_readfirstlane(static_cast<int>((a >> 32) & 0xffffu)),
                     __builtin_amdgcn_readfirstlane(bytes), 0x00020000};     // (wave-uniform by construction; said so for the "s" operands)
}
template <int N> __device__ __forceinline__ void lat_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N)); }
template <typename R> __device__ __forceinline__ void lat_pin(R& r) { asm volatile("" : "+v"(r)); }
// Scalar operands: a vector-memory instruction that reads an SGPR written by the VECTOR unit less than five wait states earlier
// reads the old value, and the compiler's hazard recogniser does not look into an asm block: when it keeps a row offset in a
// vector register (it does, when scalar registers run short) its v_readfirstlane lands right in front of the access.  A row
// access therefore ADVANCES a running offset with an s_add of its own, inside its asm block (the scalar unit's reads are
// interlocked, and a scalar write needs no wait states before a vector-memory read) -- which also spares the multiply and add per
// access the compiler spent on the offsets; the descriptors are checked in the ISA (tools/check_lattice_asm_hazards.py).
// "These registers are written by loads the compiler cannot see": every later use depends on this statement, so none moves
// above the wait that precedes it, and the registers stay allocated to the rows until here.
template <typename R> __device__ __forceinline__ void lat_pin(R (&r)[16]) {
    asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                      "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]));
}
template <typename R> __device__ __forceinline__ void lat_pin(R (&r)[8]) {
    asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
}
__device__ __forceinline__ void lat_put_f64(const lat_i32x4& d, int voff, int soff, double v) {
    int t;
    asm volatile("s_mov_b32 %0, %4\n\tbuffer_store_dwordx2 %1, %2, %3, %0 offen" : "=&s"(t) : "v"(v), "v"(voff), "s"(d), "s"(soff));
}

template <typename L, int COLS> struct LatIO;
template <> struct LatIO<float, 1> {
    static constexpr bool kHand = true;
    using Raw = lat_u32x2;
    static __device__ __forceinline__ void request(const lat_i32x4& d, __amdgpu_buffer_rsrc_t, int voff, int& run, int stride, Raw& r) {
        asm volatile("s_add_i32 %1, %1, %4\n\tbuffer_load_dwordx2 %0, %2, %3, %1 offen" : "+v"(r), "+s"(run) : "v"(voff), "s"(d), "s"(stride) : "scc");
    }
    static __device__ __forceinline__ void unpack(const Raw& v, float* x, float* y) {
        x[0] = lat_clamp(__uint_as_float(v.x)); y[0] = lat_clamp(__uint_as_float(v.y));
    }
    static __device__ __forceinline__ void put(const lat_i32x4& d, __amdgpu_buffer_rsrc_t, int voff, int& run, int stride, const float* v) {
        asm volatile("s_add_i32 %0, %0, %4\n\tbuffer_store_dword %1, %2, %3, %0 offen" : "+s"(run) : "v"(v[0]), "v"(voff), "s"(d), "s"(stride) : "scc");
    }
};
template <> struct LatIO<float, 2> {
    static constexpr bool kHand = true;
    using Raw = lat_u32x4;
    static __device__ __forceinline__ void request(const lat_i32x4& d, __amdgpu_buffer_rsrc_t, int voff, int& run, int stride, Raw& r) {
        asm volatile("s_add_i32 %1, %1, %4\n\tbuffer_load_dwordx4 %0, %2, %3, %1 offen" : "+v"(r), "+s"(run) : "v"(voff), "s"(d), "s"(stride) : "scc");
    }
    static __device__ __forceinline__ void unpack(const Raw& v, float* x, float* y) {
        x[0] = lat_clamp(__uint_as_float(v.x)); y[0] = lat_clamp(__uint_as_float(v.y));
        x[1] = lat_clamp(__uint_as_float(v.z)); y[1] = lat_clamp(__uint_as_float(v.w));
    }
    static __device__ __forceinline__ void put(const lat_i32x4& d, __amdgpu_buffer_rsrc_t, int voff, int& run, int stride, const float* v) {
        const lat_u32x2 w = {__float_as_uint(v[0]), __float_as_uint(v[1])};
        asm volatile("s_add_i32 %0, %0, %4\n\tbuffer_store_dwordx2 %1, %2, %3, %0 offen" : "+s"(run) : "v"(w), "v"(voff), "s"(d), "s"(stride) : "scc");
    }
};
template <> struct LatIO<double, 1> {
    static constexpr bool kHand = false;
    using Raw = lat_u32x4;
    static __device__ __forceinline__ void request(const lat_i32x4&, __amdgpu_buffer_rsrc_t r, int voff, int& run, int stride, Raw& v) {
        run += stride;
        v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, run, 0);
    }
    static __device__ __forceinline__ void unpack(const Raw& v, double* x, double* y) {
        x[0] = lat_clamp(lat_f64(v.x, v.y)); y[0] = lat_clamp(lat_f64(v.z, v.w));
    }
    static __device__ __forceinline__ void put(const lat_i32x4&, __amdgpu_buffer_rsrc_t r, int voff, int& run, int stride, const double* v) {
        const lat_u32x2 w = {static_cast<unsigned>(__double2loint(v[0])), static_cast<unsigned>(__double2hiint(v[0]))};
        run += stride;
        __builtin_amdgcn_raw_buffer_store_b64(w, r, voff, run, 0);
    }
};
template <> struct LatIO<double, 2> {
    static constexpr bool kHand = false;
    struct Raw { lat_u32x4 a, b; };
    static __device__ __forceinline__ void request(const lat_i32x4&, __amdgpu_buffer_rsrc_t r, int voff, int& run, int stride, Raw& v) {
        run += stride;
        v.a = __builtin_amdgcn_raw_buffer_load_b128(r, voff, run, 0);
        v.b = __builtin_amdgcn_raw_buffer_load_b128(r, voff, run + 16, 0);
    }
    static __device__ __forceinline__ void unpack(const Raw& v, double* x, double* y) {
        x[0] = lat_clamp(lat_f64(v.a.x, v.a.y)); y[0] = lat_clamp(lat_f64(v.a.z, v.a.w));
        x[1] = lat_clamp(lat_f64(v.b.x, v.b.y)); y[1] = lat_clamp(lat_f64(v.b.z, v.b.w));
    }
    static __device__ __forceinline__ void put(const lat_i32x4&, __amdgpu_buffer_rsrc_t r, int voff, int& run, int stride, const double* v) {
        run += stride;
        const lat_u32x4 w = {static_cast<unsigned>(__double2loint(v[0])), static_cast<unsigned>(__double2hiint(v[0])),
                             static_cast<unsigned>(__double2loint(v[1])), static_cast<unsigned>(__double2hiint(v[1]))};
        __builtin_amdgcn_raw_buffer_store_b128(w, r, voff, run, 0);
    }
};
// Single elements (the corner cells), addressed by their column's byte offset.
__device__ __forceinline__ float lat_load1(__amdgpu_buffer_rsrc_t r, int voff, int soff, float) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ double lat_load1(__amdgpu_buffer_rsrc_t r, int voff, int soff, double) {
    const lat_u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return lat_f64(w.x, w.y);
}
__device__ __forceinline__ void lat_store1(__amdgpu_buffer_rsrc_t r, int voff, int soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}
__device__ __forceinline__ void lat_store1(__amdgpu_buffer_rsrc_t r, int voff, int soff, double v) {
    const lat_u32x2 w = {static_cast<unsigned>(__double2loint(v)), static_cast<unsigned>(__double2hiint(v))};
    __builtin_amdgcn_raw_buffer_store_b64(w, r, voff, soff, 0);
}

// Wavefronts per block of the lattice kernel for a row stride Up and `cols` columns per lane, and the shift that
// maps a column to its wavefront (the per-wavefront offsets off[b][wave][n] are indexed by it; the coefficient
// kernels get both numbers from the host).
__host__ __device__ inline int lat_waves(int Up, int cols) { return (Up + 64 * cols - 1) / (64 * cols); }
__host__ __device__ inline int lat_col_shift(int cols) { return cols == 2 ? 7 : 6; }

// Block barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL access of the
// wavefront (s_waitcnt vmcnt(0)): here that would be the next chunk's prefetch and the acknowledgement of the last
// chunk's stores, neither of which the other wavefronts of the block ever look at (they meet in the LDS ring only).
// The two LDS-only fences make the ordering visible to the compiler as well (both builtins alone are "no memory" to
// LLVM, which could move the ring accesses across them); restricted to the local address space they emit nothing
// beyond the lgkmcnt wait.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0); vmcnt and expcnt left alone
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// The sweep of ONE (sample, direction) by the W wavefronts whose threads call this with tid = 0 .. 64 W - 1 (the kernel
// below: the whole block; the linear-domain kernel's fallback: its first wavefront, W = 1).
template <typename L, int MAXW, int COLS>
__device__ __forceinline__ void lattice_body(
        const LogPair<L>* __restrict__ lp2, L* __restrict__ alpha, L* __restrict__ beta,
        double* __restrict__ offa, double* __restrict__ offb, double* __restrict__ ll_fwd,
        double* __restrict__ ll_bwd, L* __restrict__ costs_dev, const int* __restrict__ xlen,
        const int* __restrict__ ylen, int maxT, int maxU, int Up, const int b, const int dir, const int tid, const int W,
        const L* __restrict__ logz, int* __restrict__ poison) {     // the statistics kernels' hint of a non-finite row: note_non_finite
    constexpr int C = LatChunk<L, MAXW, COLS>::C;
    constexpr bool MULTI = MAXW > 1;
    using IO = LatIO<L, COLS>;
    using Raw = typename IO::Raw;
    constexpr bool HAND = IO::kHand;
    static_assert(COLS <= 2 && C <= 16 && C <= kLatPad, "two columns per lane at most; a chunk's boundary values fit one DPP row; the beta sweep's overshoot stays in the front padding");
    __shared__ L ring[MAXW][2][C];
    __shared__ double ringoff[MAXW][2];
    __shared__ L dump[MULTI ? MAXW : 1][MULTI ? 64 + C : 1];      // where the lanes that are NOT the boundary lane put their copy of a step's hand-off value
    const int u0 = tid * COLS;                       // first of this lane's COLS adjacent columns
    const int lane = tid & 63;
    const int wave = uniform(tid >> 6);
    // Lengths live on the device, so the host cannot validate them (the CPU location does: rnnt_cpu.cpp).  A
    // sample whose lengths do not fit the tensor is run on clamped lengths (memory-safe) and its cost becomes
    // the marker NaN of cost_invalid(), which the synchronous entry points turn into RNNT_STATUS_INVALID_VALUE.
    const int Tb_raw = xlen[b], Ub_raw = ylen[b] + 1;
    const bool bad_len = Tb_raw < 1 || Ub_raw < 1 || Tb_raw > maxT || Ub_raw > maxU;
    const int Tb = Tb_raw < 1 ? 1 : (Tb_raw > maxT ? maxT : Tb_raw);
    const int Ub = Ub_raw < 1 ? 1 : (Ub_raw > maxU ? maxU : Ub_raw);
    const int Db = Tb + Ub - 1;
    const size_t Dp = lat_rows(maxT, maxU);
    // descriptors start at the first PAD row of this sample: row n lives at (n + kLatPad)
    const size_t sample0 = lat_sample(b, maxT, maxU, Up);             // (value arrays; lp2 has half the element stride)
    const int cell_row = Up * static_cast<int>(sizeof(LogPair<L>));   // bytes per row of `lp2`
    const int beta_row = Up * static_cast<int>(sizeof(L));            // bytes per row of `alpha` / `beta`
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<LogPair<L>*>(lp2) + lat_sample_pair(b, maxT, maxU, Up), 0, static_cast<int>(Dp * cell_row), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (dir == 0 ? alpha : beta) + sample0, 0, static_cast<int>(Dp * beta_row), 0x00020000);
    // lane offsets of column u0 in an lp2 row / an alpha-beta row; lanes past the row (COLS = 2 rounds the
    // block up to whole wavefronts) are parked out of range: they load zeros and store nothing
    const bool in_row = u0 < Up;
    const int vc = in_row ? u0 * static_cast<int>(sizeof(LogPair<L>)) : kLatOob;
    const int vb = in_row ? u0 * static_cast<int>(sizeof(L)) : kLatOob;
    double* off = (dir == 0 ? offa : offb) + (static_cast<size_t>(b) * W + wave) * Dp + kLatPad;
    // the same three ranges for the hand-issued accesses (LatIO; `off` from its first pad entry, like the rows)
    const lat_i32x4 dc = lat_desc(lp2 + lat_sample_pair(b, maxT, maxU, Up), static_cast<int>(Dp * cell_row));
    const lat_i32x4 db = lat_desc((dir == 0 ? alpha : beta) + sample0, static_cast<int>(Dp * beta_row));
    const lat_i32x4 dof = lat_desc(off - kLatPad, static_cast<int>(Dp * sizeof(double)));
    (void)dc; (void)db; (void)dof;
    const L NEG = log_zero<L>();
    unsigned Tb_eff[COLS];                           // cell (n-u,u) in the lattice <=> (unsigned)(n-u) < Tb_eff
#pragma unroll
    for (int c = 0; c < COLS; ++c) Tb_eff[c] = (u0 + c < Ub) ? static_cast<unsigned>(Tb) : 0u;
    const int nsteps = Db - 1;
    const int nchunks = (nsteps + C - 1) / C;
    const int nslots = nchunks + (MULTI ? W - 1 : 0);
    (void)ring; (void)ringoff; (void)dump;
    double Coff = 0.0, Cused = 0.0;
    Raw rawA[C], rawB[C];                // operand rows of this chunk and of the next, as they arrive (LatIO: clamped when used)
    L hist[C][COLS];
    int jprev = -1;                      // chunk whose results are still in `hist`
    int ld_run = 0, st_run = 0;          // running row offsets of the operand requests / the result stores (bytes)
    const int ulast = Ub - 1;            // the column of the terminal cell, its owner lane and slot
    const bool own_last = (ulast / COLS) == tid;
    // Memory traffic of a chunk (fp32: counted by hand, LatIO), SPREAD over its steps -- a burst of 2C + 1 accesses at the chunk's
    // start fills the CU's vector-memory queue and the wavefront, alone on its SIMD, sits in the issue stage until it drains
    // (round 6: 0.220 ms on config 4 against 0.141 with no traffic at all).  Step k of chunk j: store result row k of chunk
    // j - 1 (still in hist[k]), request operand row k of chunk j + 1, wait for operand row k of chunk j, compute.  That row was
    // requested one chunk ago; since then the wavefront has issued 2 (C - 1 - k) accesses in the steps after it, the offsets'
    // store, and 2 (k + 1) in this chunk: 2C + 1, the same for every step.  To keep the count the same for a wavefront's FIRST
    // chunk, its stores are issued too, parked out of range (dropped by the range check, counted like any other), and the
    // prologue that requests chunk 0 pairs every request with such a store.
    // (Measured against the burst on one box, alternating -- profiles/r06/lattice_spread_ab.log: never slower; -11 % at N = 16, -10 % at
    // U = 512, -2 % with two wavefronts per SIMD, U > 512; equal where the kernel runs at the memory system's rate, N = 64.)
    auto await = [&](Raw& row) {
        if constexpr (HAND) { lat_wait_vm<2 * C + 1>(); lat_pin(row); }
    };
    // Slots: wavefront w works on chunk j in slot j + lead (the wavefronts form a pipeline along the diagonal direction),
    // every wavefront passes every slot's barrier.  The two chunk buffers have FIXED roles per call site, so no value that
    // is still in flight ever crosses a control-flow merge (where the compiler might copy it).
    auto sweep = [&](int lead, auto& chunk) {
        if constexpr (MULTI) for (int i = 0; i < lead; ++i) lds_barrier();
        for (int j = 0; j < nchunks; j += 2) {
            chunk(lead + j, j, rawA, rawB);
            if constexpr (MULTI) lds_barrier();
            if (j + 1 < nchunks) {
                chunk(lead + j + 1, j + 1, rawB, rawA);
                if constexpr (MULTI) lds_barrier();
            }
        }
        if constexpr (MULTI) for (int i = lead + nchunks; i < nslots; ++i) lds_barrier();
    };
    auto drain = [&]() {                 // the last prefetch (nobody reads it) and the stores: before the registers go to other values / the read-back
        if constexpr (HAND) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); lat_pin(rawA); lat_pin(rawB); }
    };
    auto prologue = [&](auto& put, auto& request) {      // chunk 0's rows, in the rhythm of a chunk (see await)
#pragma unroll
        for (int k = 0; k < C; ++k) {
#pragma unroll
            for (int c = 0; c < COLS; ++c) hist[k][c] = L(0);
            if constexpr (HAND) { int parked = 0; IO::put(db, rb, kLatOob, parked, 0, hist[k]); }
            request(0, k, rawA[k]);
        }
    };
    auto epilogue = [&](auto& put, auto& put_off) {      // the last chunk's results
        if (jprev >= 0) {
            put_off(true);
#pragma unroll
            for (int k = 0; k < C; ++k) put(k, vb);
        }
        drain();
    };

    if (dir == 0) {
        // ------------------------------- alpha -------------------------------
        int hint = 0;                                // asked for now, looked at after the sweep
        if (own_last) hint = poison[b];
        L a[COLS];
#pragma unroll
        for (int c = 0; c < COLS; ++c) a[c] = (u0 + c == 0) ? L(0) : NEG;
        L up = NEG;                                  // shifted neighbour values; lane 0 stays "zero" (see the step)
        if (tid == 0) lat_store1(rb, 0, kLatPad * beta_row, L(0));
        if (lane == 0) off[0] = 0.0;
        // chunk j: diagonals j*C+1 .. j*C+C read SOURCE rows j*C .. j*C+C-1 and write result rows j*C+1 .. j*C+C
        // (row k of a chunk: the running offset is aimed one row short at k = 0 and advanced by the access itself -- LatIO)
        auto request = [&](int j, int k, Raw& dst) {
            if (k == 0) ld_run = (j * C + kLatPad - 1) * cell_row;
            IO::request(dc, rc, vc, ld_run, cell_row, dst);
        };
        auto put = [&](int k, int vput) {
            if (k == 0) st_run = (jprev * C + kLatPad) * beta_row;
            IO::put(db, rb, vput, st_run, beta_row, hist[k]);
        };
        auto put_off = [&](bool flushed) {
            if constexpr (HAND) lat_put_f64(dof, (flushed && lane < C) ? lane * 8 : kLatOob, (jprev * C + 1 + kLatPad) * 8, Cused);
            else if (flushed && lane < C) off[jprev * C + 1 + lane] = Cused;
        };
        auto chunk = [&](int s, int j, Raw (&cur)[C], Raw (&nxt)[C]) {
            const bool flushed = jprev >= 0;
            const int vput = flushed ? vb : kLatOob;         // a wavefront's first chunk has nothing to store yet
            const int jn = j + 1 < nchunks ? j + 1 : j;      // (stay inside the back padding)
            put_off(flushed);
            L inv = NEG;
            if constexpr (MULTI) {
                if (wave > 0) inv = ring[wave - 1][(s - 1) & 1][lane & (C - 1)] + static_cast<L>(ringoff[wave - 1][(s - 1) & 1] - Coff);
            }
            // Boundary hand-off between the wavefronts of a block, per step, WITHOUT the scalar unit (round 5: a readlane + writelane
            // pair each way per step, with their hazard nops): IN -- the C values of the chunk sit in the lanes of `inv` (lane l
            // holds value l mod C), a row shift by k brings value k to lane 0, and the wave shift below KEEPS lane 0 of its
            // first operand; OUT -- every lane stores its step value to LDS, the boundary lane into the ring slot, the others
            // into a scratch row nobody reads (one ds_write per step, no EXEC games).
            L* wr = nullptr;
            if constexpr (MULTI) wr = lane == 63 ? &ring[wave][s & 1][0] : &dump[wave][lane];
#pragma unroll
            for (int k = 0; k < C; ++k) {
                L pb[COLS], pl[COLS], stay[COLS], emit[COLS];
                put(k, vput);
                request(jn, k, nxt[k]);
                await(cur[k]);
                IO::unpack(cur[k], pb, pl);
#pragma unroll
                for (int c = 0; c < COLS; ++c) { stay[c] = a[c] + pb[c]; emit[c] = a[c] + pl[c]; }
                // the left neighbour of column u0 is the previous lane's last column; lane 0 of `up` is never
                // written by the shift: it keeps the "zero" it started with, or (MULTI) the neighbouring wavefront's boundary value
                if constexpr (MULTI) {
                    up = wave_shr1(row_shl(inv, k), emit[COLS - 1]);
                    wr[k] = emit[COLS - 1];
                } else {
                    up = wave_shr1(up, emit[COLS - 1]);
                }
                if constexpr (COLS == 2) {
                    log2_add_x2(stay[0], up, stay[1], emit[0], a[0], a[1]);    // the lane's two columns, interleaved
                } else {
                    a[0] = log2_add(stay[0], up);
                }
#pragma unroll
                for (int c = 0; c < COLS; ++c) hist[k][c] = a[c];
            }
            jprev = j;
            Cused = Coff;
            if constexpr (MULTI) {
                if (lane == 0) ringoff[wave][s & 1] = Coff;
            }
            if (j + 1 < nchunks) {                       // re-centre (not after the final diagonal)
                // maximum over the cells that are INSIDE the lattice on this diagonal only: whatever
                // sits in the others must not steer the offset
                L mine = NEG;
#pragma unroll
                for (int c = 0; c < COLS; ++c)
                    mine = vmax(mine, static_cast<unsigned>(j * C + C - (u0 + c)) < Tb_eff[c] ? a[c] : NEG);
                const L m = wave_max_dpp(mine);
                if (m > NEG * L(0.5)) {                  // skip all-"zero" waves
#pragma unroll
                    for (int c = 0; c < COLS; ++c) a[c] -= m;
                    Coff += static_cast<double>(m);
                }
            }
        };
        prologue(put, request);
        sweep(MULTI ? wave : 0, chunk);
        epilogue(put, put_off);
        if (own_last) {
            // alpha(T-1,U-1): read it back (this thread wrote it), with the offset that was current when its
            // diagonal was stored
            const int vb1 = ulast * static_cast<int>(sizeof(L)), vc1 = ulast * static_cast<int>(sizeof(LogPair<L>));
            const L a_last = (nsteps == 0) ? L(0) : lat_load1(rb, vb1, (Db - 1 + kLatPad) * beta_row, L(0));
            const double o_last = (nsteps == 0) ? 0.0 : Cused;   // the last chunk is never re-centred
            const L xb_ = lat_clamp(lat_load1(rc, vc1, (Db - 1 + kLatPad) * cell_row, L(0)));
            double ll2 = static_cast<double>(a_last) + o_last + static_cast<double>(xb_);
            if (hint != 0) {                                  // (never taken on a workspace this library has used before, unless a row IS non-finite)
                poison[b] = 0;
                if (hint_is_poison(hint, logz, sample0, Up, Tb, Ub)) ll2 = __builtin_nan("");
            }
            // NO alignment has non-zero probability (a required label or blank masked with -inf everywhere it could be emitted):
            // the sweep ends on the finite "log zero" sentinel.  The reference's arithmetic ends on -inf there -- cost +inf, and
            // exp(alpha + beta - ll) = exp(-inf + inf) = NaN for every in-lattice gradient (rnnt_helper.h:16-24,
            // gpu_rnnt_kernel.h:161-174); same here: +inf cost, and the NaN in ll_fwd reaches the gradients through the coefficients
            const bool impossible = ll2 < 0.5 * static_cast<double>(log_zero<L>());
            ll_fwd[b] = impossible ? __builtin_nan("") : ll2; // base 2, for the coefficient kernel
            costs_dev[b] = bad_len ? cost_invalid<L>() : impossible ? -neg_inf<L>() : static_cast<L>(-ll2 * kLn2);
        }
    } else {
        // ------------------------------- beta -------------------------------
        L bv[COLS];
#pragma unroll
        for (int c = 0; c < COLS; ++c) bv[c] = NEG;
        if (own_last) {
            const int vb1 = ulast * static_cast<int>(sizeof(L)), vc1 = ulast * static_cast<int>(sizeof(LogPair<L>));
            const L x = lat_clamp(lat_load1(rc, vc1, (Db - 1 + kLatPad) * cell_row, L(0)));
#pragma unroll
            for (int c = 0; c < COLS; ++c) bv[c] = (u0 + c == ulast) ? x : bv[c];
            lat_store1(rb, vb1, (Db - 1 + kLatPad) * beta_row, x);
        }
        L right = NEG;                               // shifted neighbour values; lane 63 stays "zero"
        if (lane == 0) off[Db - 1] = 0.0;
        // chunk j: steps i = j*C .. j*C+C-1, TARGET rows n = Db-2-i (i may run past nsteps - 1 in the last chunk: those rows lie in the front padding)
        auto request = [&](int j, int k, Raw& dst) {
            if (k == 0) ld_run = (Db - 1 - j * C + kLatPad) * cell_row;
            IO::request(dc, rc, vc, ld_run, -cell_row, dst);
        };
        auto put = [&](int k, int vput) {
            if (k == 0) st_run = (Db - 1 - jprev * C + kLatPad) * beta_row;
            IO::put(db, rb, vput, st_run, -beta_row, hist[k]);
        };
        auto put_off = [&](bool flushed) {
            if constexpr (HAND) lat_put_f64(dof, (flushed && lane < C) ? (C - 1 - lane) * 8 : kLatOob, (Db - 2 - jprev * C - (C - 1) + kLatPad) * 8, Cused);
            else if (flushed && lane < C) off[Db - 2 - (jprev * C + lane)] = Cused;
        };
        auto chunk = [&](int s, int j, Raw (&cur)[C], Raw (&nxt)[C]) {
            const bool flushed = jprev >= 0;
            const int vput = flushed ? vb : kLatOob;
            const int jn = j + 1 < nchunks ? j + 1 : j;      // (no rows below the front padding)
            put_off(flushed);
            L inv = NEG;
            if constexpr (MULTI) {
                if (wave + 1 < W) inv = ring[wave + 1][(s - 1) & 1][lane & (C - 1)] + static_cast<L>(ringoff[wave + 1][(s - 1) & 1] - Coff);
            }
            L* wr = nullptr;                                 // (hand-off as in the alpha sweep, mirrored: in at lane 63, out from lane 0)
            if constexpr (MULTI) wr = lane == 0 ? &ring[wave][s & 1][0] : &dump[wave][lane];
#pragma unroll
            for (int k = 0; k < C; ++k) {
                L pb[COLS], pl[COLS];
                put(k, vput);
                request(jn, k, nxt[k]);
                await(cur[k]);
                IO::unpack(cur[k], pb, pl);
                // the right neighbour of the lane's last column is the next lane's first; lane 63 of `right`
                // keeps its "zero" or (MULTI) takes the neighbouring wavefront's boundary value: value k sits in lane 48 + k
                // (C = 8: 56 + k) of `inv`, a row shift right by C - 1 - k brings it to lane 63
                if constexpr (MULTI) {
                    right = wave_shl1(row_shr(inv, C - 1 - k), bv[0]);
                    wr[k] = bv[0];
                } else {
                    right = wave_shl1(right, bv[0]);
                }
                L nv[COLS];
                if constexpr (COLS == 2) {
                    log2_add_x2(bv[0] + pb[0], bv[1] + pl[0], bv[1] + pb[1], right + pl[1], nv[0], nv[1]);   // interleaved
                } else {
                    nv[0] = log2_add(bv[0] + pb[0], right + pl[0]);
                }
#pragma unroll
                for (int c = 0; c < COLS; ++c) { bv[c] = nv[c]; hist[k][c] = nv[c]; }
            }
            jprev = j;
            Cused = Coff;
            if constexpr (MULTI) {
                if (lane == 0) ringoff[wave][s & 1] = Coff;
            }
            if (j + 1 < nchunks) {
                L mine = NEG;
#pragma unroll
                for (int c = 0; c < COLS; ++c)
                    mine = vmax(mine, static_cast<unsigned>(Db - 2 - (j * C + C - 1) - (u0 + c)) < Tb_eff[c] ? bv[c] : NEG);
                const L m = wave_max_dpp(mine);
                if (m > NEG * L(0.5)) {
#pragma unroll
                    for (int c = 0; c < COLS; ++c) bv[c] -= m;
                    Coff += static_cast<double>(m);
                }
            }
        };
        prologue(put, request);
        sweep(MULTI ? W - 1 - wave : 0, chunk);
        epilogue(put, put_off);
        if (tid == 0) {
            const L b0 = lat_load1(rb, 0, kLatPad * beta_row, L(0));
            ll_bwd[b] = (static_cast<double>(b0) + (nsteps == 0 ? 0.0 : Cused)) * kLn2;
        }
    }
}

template <typename L, int MAXW, int COLS>
static __global__ __launch_bounds__(MAXW * 64) void lattice_kernel(
        const LogPair<L>* __restrict__ lp2, L* __restrict__ alpha, L* __restrict__ beta,
        double* __restrict__ offa, double* __restrict__ offb, double* __restrict__ ll_fwd,
        double* __restrict__ ll_bwd, L* __restrict__ costs_dev, const int* __restrict__ xlen,
        const int* __restrict__ ylen, int maxT, int maxU, int Up, int dirs, int* __restrict__ padflag,
        const L* __restrict__ logz, int* __restrict__ poison, int* __restrict__ coef_done) {
    const int b = blockIdx.x / dirs;
    if (threadIdx.x == 0 && static_cast<int>(blockIdx.x) == b * dirs) { coef_done[kCoefDoneStride * b] = 0; coef_done[kCoefDoneStride * b + kCoefDoneStride / 2] = 0; }   // tiles of coef_kernel that have finished READING this sample's block
    if (blockIdx.x == 0 && threadIdx.x == 0) { padflag[0] = 0; padflag[2] = 0; }   // "has padding" (set again by the coefficient kernel) and "blocks overlaid by records" (make_layout: [0], [2]; a second half's private pair is [1], [3])
    lattice_body<L, MAXW, COLS>(lp2, alpha, beta, offa, offb, ll_fwd, ll_bwd, costs_dev, xlen, ylen, maxT, maxU, Up, b,
                                static_cast<int>(blockIdx.x) - b * dirs, static_cast<int>(threadIdx.x), static_cast<int>(blockDim.x >> 6),
                                logz, poison);
}

// ------------------------------------------------------------------------------------------
// Lattice recursion in the LINEAR domain, for lattices of one wavefront (maxU <= 64) with an fp32 lattice: a CHAIN
// wavefront and six helper wavefronts per (sample, direction).
//
// The log-domain step above is a dependent chain of nine instructions with two transcendentals (108 cycles per
// anti-diagonal for a lone wavefront); on probabilities the same step is  a' = fma(shr(a), pl, a * pb)  in fp64 -- 22
// cycles -- but a lone wavefront then drowns in the conversions around it (exp2 of the stored log-probs, and a log2 of
// every result, because the stored lattice has to stay in the range-free log form the coefficient kernel reads):
// 96 cycles, tools/microbench/lin_chain.hip.  A single-wavefront lattice block leaves three of the CU's four SIMDs idle,
// so the conversions move there (wavefront w runs on SIMD w & 3):
//   wavefront 0 (chain)        chunk j: the chunk's C operand pairs from LDS into registers, per diagonal the fp64 step and
//                              one ds_write_b64; per chunk one re-normalisation by a power of two (exact) on the
//                              wavefront's in-lattice maximum, the exponent accumulated in fp64 -- the same
//                              per-(wavefront, diagonal) offsets as above.  Alone on SIMD 0 (wavefront 4 only waits);
//   wavefronts 1-3 (operands)  chunk j+1, a third of its diagonals each: the rows of log-probs (requested 8 chunks ahead),
//                              cells outside the T_b x U_b lattice masked to probability 0 (so the chain needs no
//                              validity logic: unstarted and finished columns are exact zeros), exp2, the label operand
//                              shifted one lane (alpha), fp64 pairs into LDS;
//   wavefronts 5-7 (results)   chunk j-1, a third each: log2 of the results (exponent field + v_log_f32 of the top
//                              mantissa bits), stored in the skewed arrays exactly as lattice_kernel stores them, and
//                              the offsets.
// One block barrier per chunk of C = 12 diagonals.  Measured (tools/microbench/lattice_bench, MI355X): 35.7 ns per diagonal
// on long lattices against 56.2 for lattice_kernel; 10.6 us against 13.2 on N=16 T=150 U=41, 10.9 / 12.7 on N=128 T=150
// U=21, 13.6 / 17.2 on N=128 T=200 U=41.  Each role alone runs at 26-32 ns per diagonal, the chain being the slowest.
// RANGE GUARD: fp64 spans 2^-1022 below the wavefront's maximum and fp32 exp2 covers log-probs down to -126; a lattice
// cell that underflows (or is exactly zero: -inf logits), comes out negative / inf / NaN, a log-prob in (-1e29, -126), or
// a non-positive final value raises the block's flag, and the block's first wavefront then runs the log-domain sweep
// (lattice_body) for that (sample, direction) -- every input is handled exactly as before, ordinary inputs never take
// the fallback.  grid = N * dirs, block = 512.
constexpr int kLinC = 12;                           // diagonals per chunk: 4 per operand / result wavefront
constexpr int kLinThreads = 512;

__device__ __forceinline__ int wave_max_i32_dpp(int v) {
    auto mx = [](int a, int b) { return a > b ? a : b; };
    v = mx(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v = mx(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v = mx(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));   // row_half_mirror
    v = mx(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));   // row_mirror
    v = mx(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1,3
    v = mx(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2,3
    return __builtin_amdgcn_readlane(v, 63);
}

// LDS of one block of the linear-domain kernel (declared once in the kernel, shared by its two direction instantiations)
struct LinShared {
    double2 ops[2][kLinC][64];                       // {p_blank, p_label (alpha: of the lane's left neighbour)} per diagonal and lane
    double res[2][kLinC][64];                        // the chain's results, relative to chunk_off
    double chunk_off[2];
    int bad;
};

template <int DIR>                                   // 0 alpha, 1 beta: compile-time, so that no step carries a direction branch
__device__ __forceinline__ void lattice_lin_body(
        LinShared& sh, const LogPair<float>* __restrict__ lp2, float* __restrict__ alpha, float* __restrict__ beta,
        double* __restrict__ offa, double* __restrict__ offb, double* __restrict__ ll_fwd,
        double* __restrict__ ll_bwd, float* __restrict__ costs_dev, const int* __restrict__ xlen,
        const int* __restrict__ ylen, int maxT, int maxU, int Up, const int b, const int force_fallback,
        const float* __restrict__ logz, int* __restrict__ poison) {
    constexpr int C = kLinC;
    constexpr int KW = C / 3;                        // diagonals of a chunk per operand wavefront and per result wavefront
    constexpr int PFD = 8;                           // chunks of log-probs in flight per operand wavefront (96 rows ahead)
    using L = float;
    const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
    const int Tb_raw = xlen[b], Ub_raw = ylen[b] + 1;
    const bool bad_len = Tb_raw < 1 || Ub_raw < 1 || Tb_raw > maxT || Ub_raw > maxU;
    const int Tb = Tb_raw < 1 ? 1 : (Tb_raw > maxT ? maxT : Tb_raw);
    const int Ub = Ub_raw < 1 ? 1 : (Ub_raw > maxU ? maxU : Ub_raw);
    const int Db = Tb + Ub - 1;
    const size_t Dp = lat_rows(maxT, maxU);
    const size_t sample0 = lat_sample(b, maxT, maxU, Up), sample0p = lat_sample_pair(b, maxT, maxU, Up);   // value arrays / lp2
    const int cell_row = Up * static_cast<int>(sizeof(LogPair<L>));
    const int val_row = Up * static_cast<int>(sizeof(L));
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<LogPair<L>*>(lp2) + sample0p, 0, static_cast<int>(Dp * cell_row), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        (DIR == 0 ? alpha : beta) + sample0, 0, static_cast<int>(Dp * val_row), 0x00020000);
    const int u = lane;
    const bool in_row = u < Up;                      // lanes past the row: parked (load zeros, store nothing)
    const int vc = in_row ? u * static_cast<int>(sizeof(LogPair<L>)) : kLatOob;
    const int vv = in_row ? u * static_cast<int>(sizeof(L)) : kLatOob;
    const bool ucol = u < Ub;
    // cell (n - u, u) of diagonal n lies in the lattice  <=>  unsigned(n - ueff) < T_b  (columns past U_b: never)
    const int ueff = ucol ? u : u - (1 << 30);
    double* off = (DIR == 0 ? offa : offb) + static_cast<size_t>(b) * Dp + kLatPad;     // (one wavefront per block in the offsets' layout)
    const int nsteps = Db - 1;
    const int nchunks = (nsteps + C - 1) / C;
    const int ulast = Ub - 1;
    const unsigned Tbu = static_cast<unsigned>(Tb);
    const float NEG = log_zero<float>();
    // step i (0-based over the whole sweep): alpha computes diagonal i + 1 from row i; beta computes diagonal Db - 2 - i
    // (i may run past nsteps - 1 in the last chunk: those rows lie in the padding of the skewed arrays)
    auto row_of = [&](int i) { return DIR == 0 ? i : Db - 2 - i; };          // the row whose log-probs step i reads
    auto diag_of = [&](int i) { return DIR == 0 ? i + 1 : Db - 2 - i; };     // the diagonal step i produces
    if (threadIdx.x == 0) sh.bad = 0;
    float x_last = 0.0f;                             // beta: log2 p(blank) of the terminal cell
    int hint = 0;                                    // alpha: the statistics kernels' hint of a non-finite row (note_non_finite)
    if (wave == 0) {
        if (DIR == 0) {
            hint = poison[b];
            if (lane == 0) { lat_store1(rv, 0, kLatPad * val_row, 0.0f); off[0] = 0.0; }
        } else {
            x_last = lat_clamp(lat_load1(rc, ulast * static_cast<int>(sizeof(LogPair<L>)), (Db - 1 + kLatPad) * cell_row, 0.0f));
            if (lane == ulast) lat_store1(rv, ulast * static_cast<int>(sizeof(L)), (Db - 1 + kLatPad) * val_row, x_last);
            if (lane == 0) off[Db - 1] = 0.0;
        }
    }
    __syncthreads();

    // Roles by wavefront (w -> SIMD w & 3): 0 the chain, alone on its SIMD (4 waits at the barriers and nothing else);
    // 1..3 operands and 5..7 results, one of each per remaining SIMD.  Every role runs the same nchunks + 2 block barriers
    // (slot s: operands of chunk s, chain on chunk s - 1, results of chunk s - 2).  The barrier orders LDS traffic only
    // (lds_barrier): the operand wavefronts' prefetch and the result wavefronts' stores stay in flight across it.
    // Everything inside the per-diagonal loops is straight-line and lean: a wave64 VALU instruction is 4 issue cycles, the
    // chain needs ~44 cycles per diagonal, so each helper role may spend ~130 cycles per diagonal of ITS share -- the first
    // form (masks as bool logic, && / ||, a run-time direction) spent 250 per diagonal in the operand role alone.
    const int nslots = nchunks + 2;
    if (wave >= 1 && wave <= 3) {
        // ---- operands: diagonals [cw*KW, cw*KW + KW) of every chunk; a ring of PFD chunks of rows in registers
        const int cw = wave - 1;
        // A ring of PFD chunks of rows in registers, refilled slot by slot.  The requests and the waits are written out
        // (inline asm): the memory is ~1 us away, a chunk lasts ~0.4 us, and with compiler-tracked loads the vmcnt
        // bookkeeping across the loop's back edge collapsed to "wait for everything in flight" in every form tried
        // (measured: this role alone 25-29 ns per diagonal instead of ~8).  Nothing else in this wavefront uses vmcnt.
        typedef int lin_i32x4 __attribute__((ext_vector_type(4)));
        const unsigned long long cbase = reinterpret_cast<unsigned long long>(lp2 + sample0p);
        const lin_i32x4 rd = {static_cast<int>(cbase), static_cast<int>((cbase >> 32) & 0xffffu),
                              static_cast<int>(Dp * cell_row), 0x00020000};
        lat_u32x2 ring[PFD][KW];
        auto request = [&](int j, lat_u32x2 (&dst)[KW]) {
            const int jj = j < nchunks ? j : (nchunks > 0 ? nchunks - 1 : 0);      // (past the end: a valid chunk again, unused)
#pragma unroll
            for (int k = 0; k < KW; ++k)
                asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen"
                             : "=v"(dst[k]) : "v"(vc), "s"(rd), "s"((row_of(jj * C + cw * KW + k) + kLatPad) * cell_row) : "memory");
        };
        // RANGE GUARD, operand side: a log-prob strictly inside (-1e29, -126) has no normal fp32 exp2.  As bit patterns
        // (negative floats order as unsigned integers) that is one unsigned window; the minimum over the wavefront's
        // values of (bits - window start) falls below the window's width iff some value lies in it.  Values at or below
        // the sentinel (-1e30: masked cells, -inf logits) give exact zeros; NaN and +inf pass through to the chain and
        // are caught on the result side.
        constexpr unsigned kWin0 = 0xC2FC0001u;                        // one past bits(-126.0f)
        constexpr unsigned kWinW = 0xEFA18F08u - kWin0;                // up to bits(-1e29f), exclusive (log_zero is beyond it)
        unsigned win = 0xffffffffu;
#pragma unroll
        for (int r = 0; r < PFD; ++r) request(r, ring[r]);
        for (int s0 = 0; s0 < nslots; s0 += PFD) {
#pragma unroll
            for (int r = 0; r < PFD; ++r) {
                const int j = s0 + r;
                if (j < nslots) {
                    // chunk j's rows have landed once at most the (PFD - 1) * KW younger requests are outstanding
                    static_assert(KW == 4, "the wait below names the chunk's four registers");
                    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ring[r][0]), "+v"(ring[r][1]), "+v"(ring[r][2]), "+v"(ring[r][3])
                                 : "n"((PFD - 1) * KW) : "memory");
                    if (j < nchunks) {                                             // (the two slots past the last chunk write nothing)
#pragma unroll
                        for (int k = 0; k < KW; ++k) {
                            const int kk = cw * KW + k;
                            const bool inl = static_cast<unsigned>(row_of(j * C + kk) - ueff) < Tbu;
                            const float x = inl ? __uint_as_float(ring[r][k].x) : NEG, y = inl ? __uint_as_float(ring[r][k].y) : NEG;
                            win = min(win, min(__float_as_uint(x) - kWin0, __float_as_uint(y) - kWin0));
                            const float pb = __builtin_amdgcn_exp2f(x);
                            float pl = __builtin_amdgcn_exp2f(y);
                            if (DIR == 0) pl = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(pl), 0x138, 0xf, 0xf, true));   // wave_shr:1, lane 0 <- 0
                            sh.ops[j & 1][kk][lane] = make_double2(static_cast<double>(pb), static_cast<double>(pl));
                        }
                    }
                    request(j + PFD, ring[r]);
                    lds_barrier();
                }
            }
        }
        // The ring's last requests are still in flight and nobody reads them: without a use after the wait the compiler
        // hands their registers to other values (it did: the guard word below) and a late arrival overwrites those.
#pragma unroll
        for (int r = 0; r < PFD; ++r)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(ring[r][0]), "+v"(ring[r][1]), "+v"(ring[r][2]), "+v"(ring[r][3]) : : "memory");
        if (__ballot(win < kWinW) != 0 && lane == 0) sh.bad = 1;
    } else if (wave == 0) {
        // ---- the chain: chunk s - 1 in slot s; the chunk's operands are all requested before the first step
        double a, up = 0.0, Ecum = 0.0;
        if (DIR == 0) a = (u == 0) ? 1.0 : 0.0;
        else a = (u == ulast) ? exp2(static_cast<double>(x_last)) : 0.0;
        for (int s = 0; s < nslots; ++s) {
            if (s >= 1 && s <= nchunks) {
                const int j = s - 1;
                double2 o[C];
#pragma unroll
                for (int k = 0; k < C; ++k) o[k] = sh.ops[j & 1][k][lane];
                if (lane == 0) sh.chunk_off[j & 1] = Ecum;
#pragma unroll
                for (int k = 0; k < C; ++k) {
                    const double t = a * o[k].x;
                    up = DIR == 0 ? wave_shr1(up, a) : wave_shl1(up, a);       // the edge lane keeps its zero
                    a = __builtin_fma(up, o[k].y, t);
                    sh.res[j & 1][k][lane] = a;
                }
                if (j + 1 < nchunks) {
                    // re-normalise on the in-lattice maximum of the diagonal just computed: an exact power of two
                    const int n = diag_of(j * C + C - 1);
                    const bool inl = (static_cast<unsigned>(n - ueff) < Tbu) & (a > 0.0);
                    const int e = inl ? __builtin_amdgcn_frexp_exp(a) : -100000;
                    const int E = wave_max_i32_dpp(e);
                    if (E > -50000) {
                        a = __builtin_ldexp(a, -E);
                        Ecum += static_cast<double>(E);
                    }
                }
            }
            lds_barrier();
        }
    } else if (wave >= 5) {
        // ---- results of chunk s - 2 in slot s: base-2 logs into the skewed array, and the offsets.
        // RANGE GUARD, result side: the biased exponent field (with the sign bit above it) of an in-lattice result must be
        // 1..2046 -- zero, a denormal (underflow), a negative number, inf or NaN raise the flag.  The same field, minus
        // the bias, is the integer part of the logarithm; the fraction is v_log_f32 of the top 23 mantissa bits.
        const int fw = wave - 5;
        unsigned worst = 0;
        for (int s = 0; s < nslots; ++s) {
            if (s >= 2) {
                const int j = s - 2;
                const double Eo = sh.chunk_off[j & 1];
#pragma unroll
                for (int k = 0; k < KW; ++k) {
                    const int kk = fw * KW + k;
                    const int n = diag_of(j * C + kk);
                    const double v = sh.res[j & 1][kk][lane];
                    const unsigned hi = static_cast<unsigned>(__double2hiint(v)), lo = static_cast<unsigned>(__double2loint(v));
                    const unsigned field = (hi >> 20) - 1u;                                  // 0..2045 for a positive normal number
                    const bool inl = static_cast<unsigned>(n - ueff) < Tbu;
                    worst = max(worst, inl ? field : 0u);
                    const float mant = __uint_as_float((__builtin_amdgcn_alignbit(hi, lo, 29) & 0x007fffffu) | 0x3f800000u);
                    const float val = fmaxf((static_cast<float>(static_cast<int>(field)) - 1022.0f) + __builtin_amdgcn_logf(mant), NEG);
                    lat_store1(rv, vv, (n + kLatPad) * val_row, inl ? val : NEG);
                }
                if (lane < KW) off[diag_of(j * C + fw * KW + lane)] = Eo;                    // this wavefront's diagonals: one store
            }
            lds_barrier();
        }
        if (__ballot(worst >= 2046u) != 0 && lane == 0) sh.bad = 1;
    } else {
        for (int s = 0; s < nslots; ++s) lds_barrier();                                      // wavefront 4: leaves the chain's SIMD to the chain
    }
    __syncthreads();

    // ---- the likelihood, from the last diagonal the chain computed
    if (wave == 0) {
        const int il = nsteps - 1, jl = il >= 0 ? il / C : 0, kl = il >= 0 ? il % C : 0;
        const int who = DIR == 0 ? ulast : 0;
        double fin = 1.0, Eo = 0.0;
        if (nsteps > 0) { fin = sh.res[jl & 1][kl][who]; Eo = sh.chunk_off[jl & 1]; }
        if (lane == who) {
            if (!(fin > 0.0)) sh.bad = 1;
            if (DIR == 0) {
                const float xb = lat_clamp(lat_load1(rc, ulast * static_cast<int>(sizeof(LogPair<L>)), (Db - 1 + kLatPad) * cell_row, 0.0f));
                double ll2 = log2(fin) + Eo + static_cast<double>(xb);
                // (when the log-domain sweep is going to redo this (sample, direction) it reads the hint itself)
                if (hint != 0 && (sh.bad | force_fallback) == 0) {
                    poison[b] = 0;
                    if (hint_is_poison(hint, logz, sample0, Up, Tb, Ub)) ll2 = __builtin_nan("");
                }
                const bool impossible = ll2 < 0.5 * static_cast<double>(log_zero<float>());   // (see lattice_body; a zero probability takes the fallback anyway)
                ll_fwd[b] = impossible ? __builtin_nan("") : ll2;
                costs_dev[b] = bad_len ? cost_invalid<float>() : impossible ? -neg_inf<float>() : static_cast<float>(-ll2 * kLn2);
            } else {
                ll_bwd[b] = (nsteps > 0 ? log2(fin) + Eo : static_cast<double>(x_last)) * kLn2;
            }
        }
    }
    __syncthreads();
}

template <int UNUSED = 0>                            // (a template: defined in a header; static: every translation unit owns its copy, rnnt_gpu_impl.h)
static __global__ __launch_bounds__(kLinThreads) void lattice_lin_kernel(
        const LogPair<float>* __restrict__ lp2, float* __restrict__ alpha, float* __restrict__ beta,
        double* __restrict__ offa, double* __restrict__ offb, double* __restrict__ ll_fwd,
        double* __restrict__ ll_bwd, float* __restrict__ costs_dev, const int* __restrict__ xlen,
        const int* __restrict__ ylen, int maxT, int maxU, int Up, int dirs, int force_fallback, int* __restrict__ padflag,
        const float* __restrict__ logz, int* __restrict__ poison, int* __restrict__ coef_done) {
    __shared__ LinShared sh;
    const int b = blockIdx.x / dirs;
    if (threadIdx.x == 0 && static_cast<int>(blockIdx.x) == b * dirs) { coef_done[kCoefDoneStride * b] = 0; coef_done[kCoefDoneStride * b + kCoefDoneStride / 2] = 0; }   // (as lattice_kernel)
    if (blockIdx.x == 0 && threadIdx.x == 0) { padflag[0] = 0; padflag[2] = 0; }   // (as lattice_kernel)
    const int dir = static_cast<int>(blockIdx.x) - b * dirs;
    if (dir == 0) lattice_lin_body<0>(sh, lp2, alpha, beta, offa, offb, ll_fwd, ll_bwd, costs_dev, xlen, ylen, maxT, maxU, Up, b, force_fallback, logz, poison);
    else lattice_lin_body<1>(sh, lp2, alpha, beta, offa, offb, ll_fwd, ll_bwd, costs_dev, xlen, ylen, maxT, maxU, Up, b, force_fallback, logz, poison);
    if ((sh.bad | force_fallback) != 0 && threadIdx.x < 64)   // the range guard tripped (or RNNT_TUNE=latlin=2, the tests): the log-domain sweep
        lattice_body<float, 1, 1>(lp2, alpha, beta, offa, offb, ll_fwd, ll_bwd, costs_dev, xlen, ylen, maxT, maxU, Up, b, dir,
                                  static_cast<int>(threadIdx.x), 1, logz, poison);
}

// ------------------------------------------------------------------------------------------
// Gradient coefficients, one record per (b,t,u) ROW in natural row order r = (b*maxT + t)*maxU + u
// (the order the gradient pass streams the big tensor in).  grid = (8*ceil(D*Up/2048), N), block 256.
//   x = c  = alpha + beta - ll - logZ                   (g_v = exp(x_v + c) for every v)
//   y = cb = exp(alpha + lp_blank + beta(t+1,u) - ll)    (t < T-1)
//          = exp(alpha + lp_blank - ll)                  (t = T-1, u = U-1)
//   z = cl = exp(alpha + lp_label + beta(t,u+1) - ll)    (u < U-1)
//   w = label index of the row (-1: no label transition, kPadded: padded row -> zero gradient)
// Formulas: reference gpu_rnnt_kernel.h:161-174, docs/rnnt_notes.tex:138-145.  The sums are
// formed in fp64 from the scaled fp32 lattice values and their fp64 offsets.
constexpr int kPadded = -2;
// One-hot df corrections (additive joint, small vocabularies): planes written by the coefficient kernels.
// 4 = no records: a far cell's c is stored at its PLANE index in the records' memory (which must cover it: ceil8(maxU) <= 4 maxU), else 3.
__host__ __device__ inline int joint_planes_onehot(int maxU) { return joint_upad(maxU) <= 4 * maxU ? 4 : 3; }

// Everything the record of one lattice cell needs from memory.  coef_fetch() issues all of it UNCONDITIONALLY
// from addresses that are valid for any (n, u) of the skewed index space (clamped label index; the neighbour
// rows n+1 lie in the back padding at worst; `beta` carries Up+64 elements of slack), so the loads of a cell --
// and, in the tiled kernel, of all the cells a wavefront works on -- are ONE memory round trip.  Behind the
// `last_t` / `last_u` branches they were three dependent round trips per cell, and the tiled kernel on the c4
// lattice was bound by exactly that latency (block lifetime 24 us = 8 cells x 3 round trips per wavefront).
template <typename L> struct CoefRaw {
    LogPair<L> p; L lz, al, b0, b1, b2;
    double oa, ob, ob1, obr;
    int lab;
};

template <typename L>
__device__ __forceinline__ CoefRaw<L> coef_fetch(
        const LogPair<L>* __restrict__ lp2, const L* __restrict__ logz, const L* __restrict__ alpha_arr,
        const L* __restrict__ beta, const double* __restrict__ offa, const double* __restrict__ offb,
        const int* __restrict__ labels, int b, int n, int u, int maxT, int maxU, int Up, int lw, int lsh, int wu) {
    CoefRaw<L> r;
    const size_t Dp = lat_rows(maxT, maxU);
    const size_t idx = lat_index(b, n, u, maxT, maxU, Up);
    r.p = lp2[lat_pair_index(b, n, u, maxT, maxU, Up)];
    r.lz = logz[idx];
    r.al = alpha_arr[idx];
    const L* bp = beta + idx;
    r.b0 = bp[0]; r.b1 = bp[Up]; r.b2 = bp[Up + 1];
    // offsets are per wavefront of the lattice block: column u belongs to wave u >> lsh.  `wu` is that index
    // for the columns of THIS wavefront when they share it (the kernels below: 64 consecutive columns starting
    // at a multiple of 64, n wave-uniform): the three offsets are then scalar loads into SGPRs; -1 = per lane.
    const int wi = wu >= 0 ? wu : (u >> lsh);
    const double* oa = offa + (static_cast<size_t>(b) * lw + wi) * Dp + kLatPad;
    const double* ob = offb + (static_cast<size_t>(b) * lw + wi) * Dp + kLatPad;
    const double* ob_r = offb + (static_cast<size_t>(b) * lw + ((u + 1) >> lsh)) * Dp + kLatPad;   // column u+1
    r.oa = oa[n]; r.ob = ob[n]; r.ob1 = ob[n + 1]; r.obr = ob_r[n + 1];
    r.lab = maxU > 1 ? labels[static_cast<size_t>(b) * (maxU - 1) + (u < maxU - 1 ? u : maxU - 2)] : 0;
    return r;
}

// The record of lattice cell (t, u) from its fetched operands; ll2 = log2 P(y|x) of the sample.  Padded cells
// (t >= T_b or u >= U_b): c = "log zero" (exp(x + c) = 0 for any x), no corrections, flagged.
template <typename L>
__device__ __forceinline__ Cell<L> coef_eval(const CoefRaw<L>& r, double ll2, int t, int u, int Tb, int Ub, float fastemit) {
    Cell<L> o;
    o.x = log_zero<L>(); o.y = 0; o.z = 0; o.w = static_cast<L>(kPadded);
    if (t < Tb && u < Ub) {
        // everything below is in base-2 logs until the final conversion
        const double alpha = static_cast<double>(r.al) + r.oa - ll2;         // log2 alpha(t,u) - log2 P(y|x)
        const bool last_t = (t == Tb - 1), last_u = (u == Ub - 1);
        const double occ = alpha + static_cast<double>(r.b0) + r.ob;         // log2 of the cell's occupancy
        o.x = static_cast<L>(occ * kLn2 - static_cast<double>(r.lz));
        if (!last_t)
            o.y = fast_exp2(static_cast<L>(alpha + static_cast<double>(r.p.x) + static_cast<double>(r.b1) + r.ob1));
        else if (last_u)
            o.y = fast_exp2(static_cast<L>(alpha + static_cast<double>(r.p.x)));
        int lab = -1;
        if (!last_u) {
            const double arg_l = alpha + static_cast<double>(r.p.y) + static_cast<double>(r.b2) + r.obr;
            o.z = fast_exp2(static_cast<L>(arg_l));
            lab = r.lab;
            if (fastemit != 0.0f) {
                // FastEmit (Yu et al. 2021, the form NVIDIA NeMo's RNN-T loss uses): the label
                // transition's log-prob gradient is scaled by (1 + lambda).  In the record form
                //   g_v = p_v (gamma + lambda cl) - [v=blank] cb - [v=label] (1 + lambda) cl,   gamma = exp(alpha+beta-ll)
                // so c grows by log(1 + lambda cl/gamma) (cl/gamma <= 1, formed from the two exponents) and cl
                // by the factor (1 + lambda); the gradient kernel is unchanged.
                const double ratio = exp2(arg_l - occ);
                o.x += static_cast<L>(log1p(static_cast<double>(fastemit) * (ratio < 1.0 ? ratio : 1.0)));
                o.z *= static_cast<L>(1.0f + fastemit);
            }
        }
        o.w = static_cast<L>(lab);
    }
    return o;
}

// Gradient coefficients, one thread per cell of the SKEWED index space (diagonal n, column u): a wavefront
// covers 64 consecutive columns of one diagonal, so every read of the lattice arrays is one coalesced row
// segment; the record goes out as one scattered 16-byte store into the natural-order row table.  Used for
// small lattices (the tiled form below wastes most of its 64-column tiles when U is a few dozen).
// XCD-aware block order (workgroup i runs on XCD i % 8, each XCD has its own L2): every XCD owns one
// contiguous range of diagonals, so the partial lines of the row table combine in ONE L2.  gridDim.x is a
// multiple of 8.  grid = (8 * ceil(D*Up/2048), N), block = 256.
template <typename L>
static __global__ __launch_bounds__(256) void coef_cell_kernel(
        const LogPair<L>* __restrict__ lp2, const L* __restrict__ logz, const L* __restrict__ alpha_arr,
        const L* __restrict__ beta, const double* __restrict__ offa,
        const double* __restrict__ offb, const double* __restrict__ ll_fwd,
        const int* __restrict__ labels, const int* __restrict__ xlen, const int* __restrict__ ylen,
        Cell<L>* __restrict__ rowtab, int maxT, int maxU, int Up, float* __restrict__ wmat, int Upad, float fastemit,
        int planes, const long long* __restrict__ offsets, int lw, int lsh, int b0, int N, int* __restrict__ padflag, int recycled) {   // recycled: lattice blocks (from sample 0) under the records written up to this launch (launch_coef)   // offsets: packed row order (see row_stats_kernel); b0 = first sample of this launch, N = samples of the batch   // planes: 1 = W only; 2 = W and CL (third plane); 3 = W, CB, CL (one-hot df corrections); 4 = W, CB, CL, no records; the c of a FAR cell (W = kJointFarMark) is
                        // written INTO the record table's memory at the plane index (stride Upad <= 4 maxU floats) -- with the
                        // one-hot DF nothing reads cb / cl / label per record any more, and 12 instead of 28 bytes leave per cell
    const int b = b0 + blockIdx.y;
    if (recycled > 0 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicMax(padflag + 2, recycled);
    const unsigned per = gridDim.x >> 3;
    const unsigned blk = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    const int D = maxT + maxU - 1;
    // a wavefront takes one 64-column segment of one skewed row (rows are Up = 8 ceil(maxU / 8) cells long)
    const int segs = (Up + 63) >> 6;
    const long long w = static_cast<long long>(blk) * 4 + uniform(threadIdx.x >> 6);
    if (w >= static_cast<long long>(D) * segs) return;
    const int n = uniform(static_cast<int>(w / segs));
    const int useg = uniform(static_cast<int>(w - static_cast<long long>(n) * segs)) * 64;
    const int u = useg + static_cast<int>(threadIdx.x & 63);
    const int t = n - u;
    const size_t plane = static_cast<size_t>(N) * maxT * Upad;
    if (u >= maxU || t < 0 || t >= maxT) return;              // not a row of the tensor
    int Tb, Ub;
    coef_lens(xlen, ylen, b, maxT, maxU, Tb, Ub);
    const int wu = uniform(useg >> lsh);                      // this wavefront's 64 columns share it
    const CoefRaw<L> raw = coef_fetch<L>(lp2, logz, alpha_arr, beta, offa, offb, labels, b, n, u, maxT, maxU, Up, lw, lsh, wu);
    const Cell<L> o = coef_eval<L>(raw, ll_fwd[b], t, u, Tb, Ub, fastemit);
    if (offsets != nullptr) {
        const size_t at = static_cast<size_t>(offsets[b]) + static_cast<size_t>(t) * Ub + u;
        if (t < Tb && u < Ub && at < static_cast<size_t>(N) * maxT * maxU) rowtab[at] = o;   // (inside the table whatever the offsets say)
    } else if (planes != 4) {
        rowtab[(static_cast<size_t>(b) * maxT + t) * maxU + u] = o;
        if (t >= Tb || u >= Ub) *padflag = 1;                 // the batch has padded rows: the gradient kernel may skip their logits
    }
    if (wmat != nullptr) {                                    // additive joint only: W = exp(c), cb, cl; row stride Upad
        const float c = static_cast<float>(o.x);
        const size_t at = (static_cast<size_t>(b) * maxT + t) * Upad + u;
        wmat[at] = c > kJointFarC ? kJointFarMark : fast_exp(c);
        if (planes >= 3) wmat[plane + at] = static_cast<float>(o.y);
        if (planes >= 2) wmat[2 * plane + at] = static_cast<float>(o.z);
        if (planes == 4 && c > kJointFarC) reinterpret_cast<float*>(rowtab)[at] = c;   // (rare) the far cell's c, in the record table's memory
        if (u == maxU - 1)                                    // the row's pad columns [maxU, Upad) are zero
            for (int k = 1; k <= Upad - maxU; ++k)
                for (int pl = 0; pl < 3; ++pl)
                        if (pl == 0 || planes >= 3 || (pl == 2 && planes == 2)) wmat[pl * plane + at + k] = 0.0f;
    }
}

// Gradient coefficients over 2-D tiles of the skewed index space: DN diagonals x 64 columns per block.
//   compute: wavefront w takes the diagonals w, w+4, ... of the tile, lane = column -- every read of the
//            lattice arrays is a coalesced 256-byte row segment and the per-diagonal offsets are
//            wave-uniform, exactly as in the cell-per-thread form;
//   store:   the records meet in LDS and leave along NATURAL rows: for a time row t the tile holds the
//            columns u with n0 <= t + u < n0 + DN, a run of up to DN consecutive 16-byte records, written by
//            DN consecutive lanes (512-byte runs for fp32) -- instead of one scattered 16-byte store per
//            cell, which cost 0.225 of this kernel's 0.42 ms on c4 even with the partial lines meeting in
//            one L2.  The dense weight matrix of the additive-joint path leaves the same way.
// grid = (ceil(D/DN) * ceil(maxU/64), N), block = 256.
// Cells of a wavefront whose operands are requested together in the tiled kernel (COEF_KB) and the occupancy the kernel is
// compiled for (COEF_WAVES per SIMD).  Round 5 form (interleaved diagonals, 8 vector loads per cell): all eight 155 VGPRs, c4
// 0.30 ms; four 126 VGPRs, 0.27 ms.  Round 6 form (consecutive diagonals, shared beta rows), c4 / additive joint's c4 shape:
// four cells at 4 waves 0.234 / 0.246 ms, at 5 waves 0.239 / 0.250; EIGHT cells at 4 waves 0.227 / 0.235, at 5 waves
// 0.228 / 0.231, at 3 waves 0.229 / 0.234 (`profiles/r06/coef_variants.log`).
#ifndef COEF_KB
#define COEF_KB 8
#endif
#ifndef COEF_WAVES
#define COEF_WAVES 5
#endif
template <typename L, bool SUMS = false>   // SUMS: additive joint, the correction sums of the gradient GEMMs' epilogues (see below)
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(COEF_WAVES))) void coef_kernel(
        const LogPair<L>* __restrict__ lp2, const L* __restrict__ logz, const L* __restrict__ alpha_arr,
        const L* __restrict__ beta, const double* __restrict__ offa,
        const double* __restrict__ offb, const double* __restrict__ ll_fwd,
        const int* __restrict__ labels, const int* __restrict__ xlen, const int* __restrict__ ylen,
        Cell<L>* __restrict__ rowtab, int maxT, int maxU, int Up, float* __restrict__ wmat, int Upad, int tilesU,
        float fastemit, int planes, const long long* __restrict__ offsets, int lw, int lsh, int b0, int N,
        int* __restrict__ padflag, float* __restrict__ sfb, float* __restrict__ sgb, float* __restrict__ sgl,
        int* __restrict__ farflag, int recycled, int* coef_done, unsigned long long ov_rec, unsigned long long ov_head,
        unsigned long long ov_block, int first_sample) {      // recycled: as coef_cell_kernel; coef_done .. first_sample: the overlay guard below (nullptr: the record table overlays nothing)      // additive joint (sfb != nullptr): the correction sums of the gradient GEMMs' epilogues, see below
    constexpr int DN = sizeof(L) == 4 ? 32 : 16;           // diagonals per tile (LDS: DN * 64 records)
    __shared__ Cell<L> recs[DN][64];
    // Additive joint: sfb[b][t] = sum_u cb(t,u), sgb[b][u] = sum_t cb(t,u), sgl[b][u] = sum_t cl(t,u) and the "has far cells"
    // flag of the sample (rnnt_joint_kernels.h) are formed HERE, from the tile's records while they are in registers / LDS:
    // as a kernel of its own (joint_sums_kernel, still used behind the cell-per-thread form) they were a second pass over
    // the planes -- 0.11 ms of the 1.3 ms c4-shaped step.  Tile sums go to the side vectors as float atomics (as before).

    // ONE-dimensional launch, sample-major by construction: workgroup L is tile L % tiles of sample L / tiles (the overlay
    // guard below relies on the tiles of earlier samples having been DISPATCHED earlier: linear launch order)
    const int tiles = tilesU * ((maxT + maxU - 1 + DN - 1) / DN);
    const int tile = static_cast<int>(blockIdx.x % static_cast<unsigned>(tiles));
    const int b = b0 + static_cast<int>(blockIdx.x / static_cast<unsigned>(tiles));
    if (recycled > 0 && blockIdx.x == 0 && threadIdx.x == 0) atomicMax(padflag + 2, recycled);
    const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
    const int tu = tile % tilesU, tn = tile / tilesU;
    const int n0 = tn * DN, u0 = tu * 64;
    const int D = maxT + maxU - 1;
    int Tb, Ub;
    coef_lens(xlen, ylen, b, maxT, maxU, Tb, Ub);

    // ---- overlay guard (make_layout): the record table of this sample lies OVER the lattice blocks of earlier samples of
    // the batch.  All samples are in ONE launch (in groups of samples, launch after launch, the kernel lost a third of its
    // speed to launch gaps and tails: c4 0.26 -> 0.35 ms), so a block (a) announces -- after the barrier that ends its compute
    // phase -- that it has finished READING its sample's block, and (b) before it stores, makes sure that every tile of every
    // sample whose block its records may touch has announced the same.  Workgroups are dispatched in launch order
    // (sample-major: the launch is one-dimensional), so the tiles waited for were dispatched before this one: they are
    // running or done, and the wait cannot deadlock; it is also a formality -- those tiles are >= head / rec samples,
    // thousands of workgroups, ahead (counted on c4: 3 spin iterations in 54 720 workgroups).
    //   * Who looks, and when: tile 0 of a sample reads the counters of the samples its sample's records can touch and raises
    //     the sample's "clear" word; the other tiles read that one word.  Both looks are taken HERE, when the block starts, so
    //     the answer is back before the compute phase ends; only a block that saw "not yet" looks again before its barrier.
    //   * RELAXED atomics: this is a write-after-READ hazard -- nothing a tile has written is published to another tile here
    //     (records are read by the NEXT kernel), only "my loads have returned", which the block barrier implies for the whole
    //     block.  Release / acquire at agent scope write back and invalidate the XCD's L2 per tile: 0.26 -> 1.29 ms.
    //   * ONE 128-byte LINE PER WORD (kCoefDoneStride).  This is what the cost hung on: with the 2 N words of c4 packed into four
    //     lines, 18 240 announcements and as many coherent reads per call queued up at those lines' home in the memory
    //     system -- 0.74 to 1.16 ms whoever looked and whenever; a line per word: 0.275 ms (0.269 without any guard).
    // coef_done[64 s] = tiles of sample s that have finished reading, coef_done[64 s + 32] = sample s may store.
    auto ov_need_of = [&]() -> int {                       // lattice blocks of this plan that start below the end of sample b's records
        const unsigned long long end = ov_rec * static_cast<unsigned long long>(first_sample + b + 1);
        long long need = end > ov_head ? static_cast<long long>((end - ov_head + ov_block - 1) / ov_block) - first_sample : 0;
        return static_cast<int>(need > b ? b : (need < 0 ? 0 : need));      // (never its own block or a later one: head >= rec)
    };
    bool ov_pending = false;                               // wave-uniform, wavefront 0 only
    if (coef_done != nullptr && wave == 0) {
        const int need = ov_need_of();
        if (need > 0) {
            bool missing = false;
            if (tile == 0) {
                for (int s0 = 0; s0 < need; s0 += 64)
                    if (s0 + lane < need)
                        missing |= __hip_atomic_load(coef_done + kCoefDoneStride * static_cast<size_t>(s0 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < tiles;
            } else {
                missing = __hip_atomic_load(coef_done + kCoefDoneStride * static_cast<size_t>(b) + kCoefDoneStride / 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
            }
            ov_pending = __ballot(missing) != 0;           // (looked at again after this wavefront's compute phase)
            if (tile == 0 && !ov_pending && lane == 0)
                __hip_atomic_store(coef_done + kCoefDoneStride * static_cast<size_t>(b) + kCoefDoneStride / 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ---- compute, skewed order: wavefront w takes the K = DN/4 CONSECUTIVE diagonals n0 + w K ..., COEF_KB of them per round trip.
    // Consecutive, because the beta row a cell needs below it is the next diagonal's own row: a group of KB cells per lane reads
    // KB + 1 rows of beta instead of 3 KB values (its own, the one below, the one below-right), the below-right value is the
    // below value of the lane to the right (one DPP shift; lane 63 takes column u0 + 64 from a wave-uniform load), the right
    // column's offset is one of two wave-uniform values, and the label is read once per lane.  Vector loads per group of four
    // cells: 17 instead of 32 (round 5: lp2, log Z, alpha, three betas, a per-lane fp64 offset and the label per cell) -- the
    // kernel is bound by the rate of its L1 requests, not by bytes.  Same values into coef_eval, same records.
    // (Where the diagonal index is clamped at the lattice's end the shared row differs from the old "row + 1" -- only for the
    // terminal diagonal and beyond, whose cells use neither value: last_t / last_u in coef_eval, or padding.)
    {
        constexpr int K = DN / 4;                          // diagonals per wavefront
        constexpr int KB = COEF_KB < K ? COEF_KB : K;      // ... requested together
        const int u = u0 + lane;
        const int uc = u < maxU ? u : maxU - 1;            // columns past the lattice fetch a valid one, their record is padding
        const int uc63 = u0 + 63 < maxU ? u0 + 63 : maxU - 1;   // lane 63's column (wave-uniform)
        const double ll2 = ll_fwd[b];
        const size_t Dp = lat_rows(maxT, maxU);
        const int wi = u0 >> lsh;                          // the lattice wavefront these 64 columns belong to (offsets are per lattice wavefront)
        const double* oa = offa + (static_cast<size_t>(b) * lw + wi) * Dp + kLatPad;
        const double* ob = offb + (static_cast<size_t>(b) * lw + wi) * Dp + kLatPad;
        const double* obn = ob + Dp;                       // the next lattice wavefront's: column uc + 1 when it is that wavefront's first
        const bool right_next = ((uc + 1) >> lsh) != wi;
        const int lab = maxU > 1 ? labels[static_cast<size_t>(b) * (maxU - 1) + (uc < maxU - 1 ? uc : maxU - 2)] : 0;
        const LogPair<L>* pcol = lp2 + lat_pair_index(b, 0, uc, maxT, maxU, Up);
        const size_t vcol = lat_index(b, 0, uc, maxT, maxU, Up);
        const L* bedge = beta + lat_index(b, 0, uc63, maxT, maxU, Up) + Up + 1;       // [row * Up]: the value below-right of lane 63's cell
#pragma unroll 1
        for (int i0 = 0; i0 < K; i0 += KB) {
            LogPair<L> cp[KB];
            L lz[KB], al[KB], bb[KB + 1], be[KB];
            int row[KB];
#pragma unroll
            for (int i = 0; i < KB; ++i) {
                const int n = n0 + wave * K + i0 + i;
                row[i] = n < D ? n : D - 1;
                const size_t r = static_cast<size_t>(row[i]) * Up;
                cp[i] = pcol[r];
                lz[i] = logz[vcol + r];
                al[i] = alpha_arr[vcol + r];
                bb[i] = beta[vcol + r];
                be[i] = bedge[r];
            }
            bb[KB] = beta[vcol + static_cast<size_t>(row[KB - 1]) * Up + Up];
#pragma unroll
            for (int i = 0; i < KB; ++i) {
                const int dn = wave * K + i0 + i;
                const int n = n0 + dn, t = n - u;
                CoefRaw<L> raw;
                raw.p = cp[i]; raw.lz = lz[i]; raw.al = al[i];
                raw.b0 = bb[i]; raw.b1 = bb[i + 1];
                raw.b2 = wave_shl1(be[i], bb[i + 1]);      // lane l: the below value of lane l + 1; lane 63: the wave-uniform edge value
                raw.oa = oa[row[i]]; raw.ob = ob[row[i]]; raw.ob1 = ob[row[i] + 1];
                raw.obr = right_next ? obn[row[i] + 1] : raw.ob1;
                raw.lab = lab;
                Cell<L> o;
                o.x = log_zero<L>(); o.y = 0; o.z = 0; o.w = static_cast<L>(kPadded);
                if (n < D && u < maxU && t >= 0 && t < maxT) o = coef_eval<L>(raw, ll2, t, u, Tb, Ub, fastemit);
                recs[dn][lane] = o;
            }
        }
    }
    // (wavefront 0 arrives at the barrier only when the overlay guard lets it: `ov_pending` above)
    if (coef_done != nullptr && wave == 0 && ov_pending) {
        // the look taken when the block started did not show the sample clear: look again, and wait
        if (tile == 0) {
            const int need = ov_need_of();
            for (int s0 = 0; s0 < need; s0 += 64)
                if (s0 + lane < need)
                    while (__hip_atomic_load(coef_done + kCoefDoneStride * static_cast<size_t>(s0 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < tiles)
                        __builtin_amdgcn_s_sleep(8);
            if (lane == 0)
                __hip_atomic_store(coef_done + kCoefDoneStride * static_cast<size_t>(b) + kCoefDoneStride / 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(coef_done + kCoefDoneStride * static_cast<size_t>(b) + kCoefDoneStride / 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    // every load of this block has returned (its operands are in LDS): announce it
    if (coef_done != nullptr && threadIdx.x == 0)
        __hip_atomic_fetch_add(coef_done + kCoefDoneStride * static_cast<size_t>(b), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if constexpr (SUMS) {
        // passes of their own over the records in LDS (inside the compute phase or the store loop the sums cost the kernel
        // its fourth wavefront per SIMD: it sits at the 128-register line).  Wavefront 0: the tile's column sums of cb and
        // cl and the far test, one column per lane (padded cells carry zeros)
        if (wave == 0) {
            float cb = 0.0f, cl = 0.0f;
            bool far = false;
#pragma unroll 4
            for (int dn = 0; dn < DN; ++dn) {
                const Cell<L> o = recs[dn][lane];
                cb += static_cast<float>(o.y);
                cl += static_cast<float>(o.z);
                far |= static_cast<int>(o.w) != kPadded && static_cast<float>(o.x) > kJointFarC;
            }
            if (u0 + lane < maxU) {
                if (cb != 0.0f) unsafeAtomicAdd(sgb + static_cast<size_t>(b) * maxU + u0 + lane, cb);
                if (cl != 0.0f) unsafeAtomicAdd(sgl + static_cast<size_t>(b) * maxU + u0 + lane, cl);
            }
            if (__ballot(far) != 0 && lane == 0) farflag[b] = 1;
        }
        // wavefronts 1 .. 2: row sums of the blank corrections, one thread per time row that meets the tile walks its run
        const int r = static_cast<int>(threadIdx.x) - 64;
        const int t = n0 - (u0 + 63) + r;
        if (r >= 0 && r < DN + 63 && t >= 0 && t < maxT) {
            const int ulo = n0 - t > u0 ? n0 - t : u0;
            int uhi = n0 + DN - 1 - t;                                      // last column of the run
            if (uhi > u0 + 63) uhi = u0 + 63;
            if (uhi > maxU - 1) uhi = maxU - 1;
            if (uhi > D - 1 - t) uhi = D - 1 - t;
            float rs = 0.0f;
            for (int u = ulo; u <= uhi; ++u) rs += static_cast<float>(recs[t + u - n0][u - u0].y);
            if (rs != 0.0f) unsafeAtomicAdd(sfb + static_cast<size_t>(b) * maxT + t, rs);
        }
    }
    // ---- store, natural order: groups of DN lanes take one time row each
    const size_t plane = static_cast<size_t>(N) * maxT * Upad;
    constexpr int GROUPS = 256 / DN;
    const int grp = threadIdx.x / DN, c = threadIdx.x % DN;
    const int t_lo = n0 - (u0 + 63);                       // first time row that meets the tile
    for (int r = grp; r < DN + 63; r += GROUPS) {
        const int t = t_lo + r;
        if (t < 0 || t >= maxT) continue;
        const int ulo = n0 - t > u0 ? n0 - t : u0;         // columns of row t inside the tile
        const int u = ulo + c;
        if (u > u0 + 63 || u >= maxU || t + u >= n0 + DN || t + u >= D) continue;
        const Cell<L> o = recs[t + u - n0][u - u0];
        if (offsets != nullptr) {                          // packed row order: the run of a time row stays contiguous
            const size_t at = static_cast<size_t>(offsets[b]) + static_cast<size_t>(t) * Ub + u;
            if (t < Tb && u < Ub && at < static_cast<size_t>(N) * maxT * maxU) rowtab[at] = o;
        } else if (planes < 4) {
            rowtab[(static_cast<size_t>(b) * maxT + t) * maxU + u] = o;
            if (t >= Tb || u >= Ub) *padflag = 1;          // the batch has padded rows: the gradient kernel may skip their logits
        }
        if (wmat != nullptr) {                             // additive joint only: W = exp(c), cb, cl; row stride Upad
            // planes: 1 W | 2 W, CL | 3 W, CB, CL | 4 W, CB, CL, no records (a far cell's c goes where its record would start) | 5 (SUMS only) as 4 WITHOUT
            // the CB plane: the DF kernel takes its blank corrections from the row sums formed above, nothing reads cb per cell
            const float cc = static_cast<float>(o.x);
            const size_t at = (static_cast<size_t>(b) * maxT + t) * Upad + u;
            wmat[at] = cc > kJointFarC ? kJointFarMark : fast_exp(cc);
            if (planes == 3 || planes == 4) wmat[plane + at] = static_cast<float>(o.y);
            if (planes >= 2) wmat[2 * plane + at] = static_cast<float>(o.z);
            if (planes >= 4 && cc > kJointFarC) reinterpret_cast<float*>(rowtab)[at] = cc;   // (rare) the far cell's c, in the record table's memory
            if (u == maxU - 1)                             // the row's pad columns [maxU, Upad) are zero
                for (int k = 1; k <= Upad - maxU; ++k)
                    for (int pl = 0; pl < 3; ++pl)
                        if (pl == 0 || (pl == 1 && (planes == 3 || planes == 4)) || (pl == 2 && planes >= 2)) wmat[pl * plane + at + k] = 0.0f;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Debug aid (compute_rnnt_loss_lattice_dump): one sample's alpha / beta out of the skewed, base-2, re-centred workspace
// arrays into natural (t, u) order and natural logs; the same reconstruction the coefficient kernels do (coef_fetch).
template <typename L>
static __global__ __launch_bounds__(256) void lattice_dump_kernel(
        const L* __restrict__ alpha, const L* __restrict__ beta, const double* __restrict__ offa, const double* __restrict__ offb,
        const int* __restrict__ xlen, const int* __restrict__ ylen, int b, int maxT, int maxU, int Up, int lw, int lsh,
        double* __restrict__ a_out, double* __restrict__ b_out, const int* __restrict__ recycled) {
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= maxT * maxU) return;
    const int t = cell / maxU, u = cell - t * maxU, n = t + u;
    int Tb, Ub;
    coef_lens(xlen, ylen, b, maxT, maxU, Tb, Ub);
    // padding -- and a sample whose lattice block the record table of a gradient-computing call has overlaid (make_layout):
    // its alpha / beta no longer exist
    if (t >= Tb || u >= Ub || b < recycled[0]) { a_out[cell] = __builtin_nan(""); b_out[cell] = __builtin_nan(""); return; }
    const size_t Dp = lat_rows(maxT, maxU);
    const size_t idx = lat_index(b, n, u, maxT, maxU, Up);
    const size_t o = (static_cast<size_t>(b) * lw + (u >> lsh)) * Dp + kLatPad + n;
    a_out[cell] = (static_cast<double>(alpha[idx]) + offa[o]) * kLn2;
    b_out[cell] = (static_cast<double>(beta[idx]) + offb[o]) * kLn2;
}

// ------------------------------------------------------------------------------------------
// [summed loss, sample count] of a shard in fp64 -- the 16-byte payload of the batch-sharded step's one collective
// (compute_rnnt_loss_sharded).  One block of 256 threads; a marker NaN of an invalid sample propagates into the sum.
template <typename C>
static __global__ __launch_bounds__(256) void loss_sum_kernel(const C* __restrict__ costs, int N, double* __restrict__ out2) {
    __shared__ double part[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) s += static_cast<double>(costs[i]);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        out2[0] = part[0] + part[1] + part[2] + part[3];
        out2[1] = static_cast<double>(N);
    }
}

// ------------------------------------------------------------------------------------------
// Packed layout with a per-sample gradient scale: one scale per packed ROW, so that the gradient kernel's lookup
// needs no search (rowscale lives in lattice blocks of the workspace that are dead once the coefficients exist: make_layout).
// grid = (N, 8), block = 256.
template <typename C>
static __global__ __launch_bounds__(256) void fill_row_scale_kernel(
        const long long* __restrict__ offsets, const C* __restrict__ grad_scale, C* __restrict__ rowscale,
        long long total_rows) {
    const int b = blockIdx.x;
    const long long lo = offsets[b], hi = offsets[b + 1] < total_rows ? offsets[b + 1] : total_rows;
    const C s = grad_scale[b];
    for (long long r = lo + static_cast<long long>(blockIdx.y) * 256 + threadIdx.x; r < hi;
         r += static_cast<long long>(gridDim.y) * 256)
        rowscale[r] = s;
}

// ------------------------------------------------------------------------------------------
// Pass B, FLAT form (the production path).  The (N,T,U,A) tensor is streamed as one flat array
// of 16-byte packets: a block owns a contiguous, 16 KB-aligned chunk of kChunkPackets packets
// per iteration (each of its 4 wavefronts moves whole 1 KB lines) and grid-strides over the
// chunks; loads and stores carry the non-temporal hint.  The row of a packet is recovered
// arithmetically (row of the chunk start is carried incrementally in 64-bit; inside the chunk
// a 32-bit reciprocal division; chunk = PPT*256 packets) and its {c, cb, cl, label} record comes from the natural-order
// row table (L1/L2 hits: consecutive packets share rows).  Padded rows are written as zeros
// here, so no memset of the gradient tensor exists (the reference does one: gpu_rnnt.h:107-110).
// Measured on MI355X (tools/microbench/stream_variants.hip): this structure sustains
// 6.4-6.5 TB/s read+write with the exp included, the wavefront-per-row form 5.1 TB/s.
template <typename Tag, int SCALE, int PPT, int PADSKIP>   // SCALE: 0 none, 1 per sample (padded layout), 2 per row (packed); PPT = packets per thread and iteration;
                                                            // PADSKIP: the logits of padded rows are not read -- 0 never, 1 always, 2 when the batch has padding (padflag)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(typename Tag::comp) == 8 ? 1 : 8))) void grad_flat_kernel(
        const typename Tag::store* acts, typename Tag::store* grads,       // NOT __restrict__: gradients == activations is a supported call (rnnt.h)
        const Cell<typename Tag::comp>* __restrict__ rowtab, const typename Tag::comp* __restrict__ grad_scale,
        unsigned long long E, unsigned long long R, int A, int blank, int TU, float invA,
        unsigned long long dq, int drem, const typename Tag::comp* __restrict__ rowscale, const int* __restrict__ padflag) {
    using C = typename Tag::comp;
    constexpr bool SCALED = SCALE != 0;
    // Short rows (PADSKIP == 2): testing the record before the load is a dependent memory latency per block and costs a
    // batch WITHOUT padding 7 % (2 KB rows); whether there is any padding is one word the coefficient kernel has left
    // behind -- a scalar load that is back before the index arithmetic below is through.
    bool ps = PADSKIP == 1;
    if constexpr (PADSKIP == 2) ps = padflag[0] != 0;
    constexpr int V = Vec<Tag>::N;
    constexpr int kChunkPackets = PPT * 256;
    constexpr int CH = kChunkPackets * V;                 // elements per chunk
    const unsigned long long npk = E / V;
    const unsigned long long nchunks = (npk + kChunkPackets - 1) / kChunkPackets;
    const u32x4* in = reinterpret_cast<const u32x4*>(acts);
    u32x4* out = reinterpret_cast<u32x4*>(grads);

    unsigned long long c = blockIdx.x;
    unsigned long long r = (c * CH) / static_cast<unsigned>(A);                   // row of the chunk start
    int rem = static_cast<int>((c * CH) - r * static_cast<unsigned>(A));          // offset inside it

    // Per-sample scale of a row.  Padded layout: sample = row / (maxT*maxU).  Packed layout: the host has expanded
    // the per-sample scales into one value per packed row (`rowscale`, fill_row_scale_kernel) -- a stateless lookup
    // like the padded one.  (Tried first: the sample of a chunk kept as block-uniform state with the offsets in
    // LDS or global memory; every form of it ran the c3 gradient pass at 2.9-3.4 ms instead of 1.5.)
    // The chunk's scale as ONE block-uniform value whenever all its rows belong to one sample (a chunk is 2-8 KB, a
    // sample's slab megabytes: nearly always): a per-packet lookup is an integer division plus a dependent load in
    // front of every packet, which made the scaled form of this kernel -- the one every autograd caller gets,
    // grad_output / N folded in -- 10 % slower than the plain one on c3 (2.82 vs 2.55 ms; now 2.69) and the c5 step
    // through RNNTLoss 1.33 instead of 1.19 ms.  (Also measured: the scale folded into the exponent as exp(x + c + log s)
    // and a division-free sample index -- both no better, the former 7 % worse with 8-element bf16 packets.)
    // A chunk that does span two samples (or more: tiny lattices) goes through the plain per-element loop below, so the
    // packet code never divides: 68 -> 5x registers for the scaled form, i.e. the occupancy of the plain one.
    C chunk_scale = C(1);
    auto sample_scale = [&](unsigned long long row) -> C {      // (slow path and the tail elements only)
        // (a 64-bit division is ~5x the instructions of a 32-bit one; tensors below 2^32 rows take the latter)
        if (R <= 0xffffffffull) return grad_scale[static_cast<unsigned>(row) / static_cast<unsigned>(TU)];
        return grad_scale[row / static_cast<unsigned>(TU)];
    };
    auto scale_of = [&](unsigned long long row) -> C {
        if constexpr (SCALE == 2) return rowscale[row];
        else if constexpr (SCALE == 1) return chunk_scale;
        else return C(1);
    };
    // One element at position `pos` of a row with record `rec`.
    auto elem = [&](const Cell<C>& rec, int pos, C x, C gs) -> C {
        const int lab = static_cast<int>(rec.w);
        if (lab == kPadded) return C(0);
        C g = fast_exp(x + rec.x);
        if (pos == blank) g -= rec.y;
        if (pos == lab) g -= rec.z;
        if constexpr (SCALED) g *= gs;
        return g;
    };

    for (; c < nchunks; c += gridDim.x) {
        const unsigned long long pk0 = c * kChunkPackets;
        if constexpr (SCALE == 1) {
            {                                               // padded layout: sample = row / (maxT*maxU)
                const unsigned long long rl0 = r + static_cast<unsigned>(CH / A + 1);
                const unsigned long long rl = rl0 < R ? rl0 : R - 1;          // last row the chunk can touch
                unsigned long long b0, b1;
                if (R <= 0xffffffffull) {
                    b0 = static_cast<unsigned>(r) / static_cast<unsigned>(TU);
                    b1 = static_cast<unsigned>(rl) / static_cast<unsigned>(TU);
                } else {
                    b0 = r / static_cast<unsigned>(TU);
                    b1 = rl / static_cast<unsigned>(TU);
                }
                chunk_scale = grad_scale[b0];
                if (b0 != b1) {                             // block-uniform: the chunk crosses into another sample
                    // element by element, in 32-bit arithmetic relative to the chunk (no wide divisions: this rare path
                    // must not cost the packet code registers)
                    const unsigned long long e0 = c * static_cast<unsigned long long>(CH);
                    const unsigned len = static_cast<unsigned>(e0 + CH < npk * V ? CH : npk * V - e0);
                    for (unsigned i = threadIdx.x; i < len; i += 256) {
                        const unsigned idx = static_cast<unsigned>(rem) + i;      // offset from the start of row r (< 2^24)
                        unsigned q = static_cast<unsigned>(static_cast<float>(idx) * invA);
                        int pos = static_cast<int>(idx - q * static_cast<unsigned>(A));
                        if (pos < 0) { pos += A; --q; } else if (pos >= A) { pos -= A; ++q; }
                        const unsigned long long rw = r + q;
                        unsigned long long sb = b0, next = (b0 + 1) * static_cast<unsigned>(TU);
                        while (rw >= next) { ++sb; next += static_cast<unsigned>(TU); }
                        store1<Tag>(grads + e0 + i, elem(rowtab[rw], pos, load1<Tag>(acts + e0 + i), grad_scale[sb]));
                    }
                    r += dq;
                    rem += drem;
                    if (rem >= A) { rem -= A; ++r; }
                    continue;
                }
            }
        }
        uint4 raw[PPT];
        Cell<C> rec[PPT], rec2[PPT];                        // rec2: the NEXT row's record, for packets that straddle
        int v0[PPT];
        unsigned long long row[PPT];
        bool live[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int p = k * 256 + threadIdx.x;
            live[k] = pk0 + p < npk;
            const unsigned idx = static_cast<unsigned>(rem) + static_cast<unsigned>(p) * V;
            unsigned q = static_cast<unsigned>(static_cast<float>(idx) * invA);
            int rr = static_cast<int>(idx - q * static_cast<unsigned>(A));
            if (rr < 0) { rr += A; --q; } else if (rr >= A) { rr -= A; ++q; }
            v0[k] = rr;
            row[k] = r + q;
            if (live[k]) {
                rec[k] = rowtab[row[k]];
                if (PADSKIP == 0 || !ps) raw[k] = load_packet<true>(in + pk0 + p);
                // a packet that crosses into the next row (A % V != 0) needs that row's record too: asked
                // for here, with the other loads -- fetched inside the compute phase it was a dependent
                // global latency that nearly every wavefront paid on c4 (A = 50: 5.9 -> 6.4 TB/s)
                if (rr + V > A) rec2[k] = rowtab[row[k] + 1 < R ? row[k] + 1 : R - 1];
            }
        }
        if (PADSKIP != 0 && ps) {
            // The record comes first, and a packet that lies wholly inside a PADDED row is only zero-filled, its logits
            // are never read (with T_b, U_b spread over [max/2, max] that is ~40 % of the rows: c3 gradient pass 2.58 ->
            // 2.11 ms, c5 0.67 -> 0.60).  Rows from 8 KB on always go this way; shorter ones when the batch has padding.
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int p = k * 256 + threadIdx.x;
                const bool skip = (v0[k] + V <= A) && static_cast<int>(rec[k].w) == kPadded;
                raw[k] = make_uint4(0, 0, 0, 0);
                if (live[k] && !skip) raw[k] = load_packet<true>(in + pk0 + p);
            }
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            if (!live[k]) continue;
            const int p = k * 256 + threadIdx.x;
            C v[V];
            unpack<Tag>(raw[k], v);
            if (v0[k] + V <= A) {
                // whole packet inside one row (the common case)
                const int lab = static_cast<int>(rec[k].w);
                if (lab == kPadded) {
#pragma unroll
                    for (int j = 0; j < V; ++j) v[j] = 0;
                } else {
                    const C cc = rec[k].x;
#pragma unroll
                    for (int j = 0; j < V; ++j) v[j] = fast_exp(v[j] + cc);
                    if (static_cast<unsigned>(blank - v0[k]) < static_cast<unsigned>(V) ||
                        static_cast<unsigned>(lab - v0[k]) < static_cast<unsigned>(V)) {
#pragma unroll
                        for (int j = 0; j < V; ++j) {
                            if (v0[k] + j == blank) v[j] -= rec[k].y;
                            if (v0[k] + j == lab) v[j] -= rec[k].z;
                        }
                    }
                    if constexpr (SCALED) {
                        const C gs = scale_of(row[k]);
#pragma unroll
                        for (int j = 0; j < V; ++j) v[j] *= gs;
                    }
                }
            } else {
                // packet crosses a row boundary (A not a multiple of the packet, or A < packet)
                if (A >= V) {
                    // two rows at most: elements j < split belong to row[k], the rest to the next row
                    const int split = A - v0[k];
                    const unsigned long long rw2 = row[k] + 1 < R ? row[k] + 1 : R - 1;
                    const C gs1 = scale_of(row[k]), gs2 = scale_of(rw2);
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        const bool first = j < split;
                        v[j] = elem(first ? rec[k] : rec2[k], first ? v0[k] + j : j - split, v[j], first ? gs1 : gs2);
                    }
                } else {
                    unsigned long long rw = row[k];
                    int pos = v0[k];
                    Cell<C> cur = rec[k];
                    C gs = scale_of(rw);
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        while (pos >= A) {
                            pos -= A;
                            ++rw;
                            if (rw < R) cur = rowtab[rw];
                            gs = scale_of(rw < R ? rw : R - 1);
                        }
                        v[j] = elem(cur, pos, v[j], gs);
                        ++pos;
                    }
                }
            }
            store_packet<true>(out + pk0 + p, pack<Tag>(v));
        }
        r += dq;
        rem += drem;
        if (rem >= A) { rem -= A; ++r; }
    }

    // the E % V elements after the last whole packet
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (unsigned long long e = npk * V; e < E; ++e) {
            const unsigned long long rw = e / static_cast<unsigned>(A);
            const int pos = static_cast<int>(e - rw * static_cast<unsigned>(A));
            C gs = C(1);
            if constexpr (SCALE == 2) gs = rowscale[rw];
            else if constexpr (SCALE == 1) gs = sample_scale(rw);
            store1<Tag>(grads + e, elem(rowtab[rw], pos, load1<Tag>(acts + e), gs));
        }
    }
}

// ------------------------------------------------------------------------------------------
// Pass B, row form: one wavefront per row with a scalar head/tail, used only when the tensors
// are not 16-byte aligned (or differ in their 16-byte phase) and packets cannot be used flat.
// grid = (ceil(maxT*maxU / WAVES), N), block = WAVES*64.
template <typename Tag, int WAVES, bool SCALED>
__global__ __launch_bounds__(WAVES * 64) void grad_rows_kernel(
        const typename Tag::store* acts, typename Tag::store* grads,       // NOT __restrict__: in-place calls (see grad_flat_kernel)
        const Cell<typename Tag::comp>* __restrict__ rowtab, const typename Tag::comp* __restrict__ grad_scale,
        int maxT, int maxU, int A, int blank, int vec_ok, int b0) {
    using S = typename Tag::store;
    using C = typename Tag::comp;
    constexpr int V = Vec<Tag>::N;
    const int b = b0 + blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int q = uniform(blockIdx.x * WAVES + (threadIdx.x >> 6));
    if (q >= maxT * maxU) return;
    const size_t rix = static_cast<size_t>(b) * maxT * maxU + q;
    const size_t roff = rix * A;
    const S* row = acts + roff;
    S* grow = grads + roff;

    int head, nvec, tail0;
    row_split<S>(reinterpret_cast<uintptr_t>(row), A, vec_ok != 0, head, nvec, tail0);
    u32x4* gp = reinterpret_cast<u32x4*>(grow + head);
    const Cell<C> r = rowtab[rix];
    const int lab = static_cast<int>(r.w);

    if (lab == kPadded) {
        for (int e = lane; e < head; e += 64) store1<Tag>(grow + e, C(0));
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (int i = lane; i < nvec; i += 64) store_packet<false>(gp + i, z);
        for (int e = tail0 + lane; e < A; e += 64) store1<Tag>(grow + e, C(0));
        return;
    }
    const C c = r.x, cb = r.y, cl = r.z;
    C gs = 1;
    if constexpr (SCALED) gs = grad_scale[b];

    auto one = [&](int e, C x) -> C {
        C g = fast_exp(x + c);
        if (e == blank) g -= cb;
        if (e == lab) g -= cl;
        if constexpr (SCALED) g *= gs;
        return g;
    };
    for (int e = lane; e < head; e += 64) store1<Tag>(grow + e, one(e, load1<Tag>(row + e)));
    const u32x4* vp = reinterpret_cast<const u32x4*>(row + head);
    for (int i = lane; i < nvec; i += 64) {
        const uint4 r0 = load_packet<false>(vp + i);
        C v[V];
        unpack<Tag>(r0, v);
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = one(head + i * V + j, v[j]);
        store_packet<false>(gp + i, pack<Tag>(v));
    }
    for (int e = tail0 + lane; e < A; e += 64) store1<Tag>(grow + e, one(e, load1<Tag>(row + e)));
}

}  // namespace rnnt
