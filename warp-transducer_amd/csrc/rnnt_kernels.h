// rnnt_kernels.h -- the four gfx950 kernels of the RNN-T loss hot path.
//
//   row_stats_kernel   one read of the (N,T,U,A) logits: online log-sum-exp per (b,t,u)
//                      row + gather of the blank / label logits          [HBM-bound, E*s read]
//   lattice_kernel     alpha and beta recursions over the T x U lattice, one wavefront
//                      lane per u, sweeping anti-diagonals, values carried in registers,
//                      neighbour exchange through DPP wave shifts         [latency-bound, O(R)]
//   coef_kernel        per lattice cell: the three numbers the gradient needs
//                      (row exponent offset, blank correction, label correction)  [O(R)]
//   grad_kernel        second read of the logits + dense gradient write-back
//                      g_v = exp(x_v + c) - [v==blank] cb - [v==label] cl   [HBM-bound, 2*E*s]
//
// What they replace in the reference (behaviour, not structure):
//   include/detail/reduce.h:45-104 + gpu_rnnt.h:73-80   (two-pass max / exp-sum denominators)
//   include/detail/gpu_rnnt_kernel.h:11-47, 79-113      (alpha / beta kernels)
//   include/detail/gpu_rnnt_kernel.h:143-179 + gpu_rnnt.h:107-110 (gradient kernel + memset)
//
// Lattice side data lives in the caller's workspace in a DIAGONAL-SKEWED layout:
//   cell(b, t, u) -> ((b * D + (t + u)) * maxU + u),  D = maxT + maxU - 1
// so that the lanes of the lattice wavefront (consecutive u on one anti-diagonal t+u = n)
// touch consecutive addresses.  All indices are 64-bit (the reference's are 32-bit int:
// gpu_rnnt_kernel.h:7-8,161,174).
#pragma once

#include "rnnt_device.h"

namespace rnnt {

// One lattice cell record.  Written in three stages:
//   row_stats : x = log p(blank|t,u)   y = log p(y_u|t,u)   z = logZ(t,u)   w = (unused)
//   lattice   : w = scaled alpha(t,u)
//   coef      : x = c   y = cb   z = cl   w = label index (as number), overwriting in place
template <typename L> struct alignas(4 * sizeof(L)) Cell { L x, y, z, w; };

constexpr int kLatticeBlock = 8;   // diagonals per prefetch/renormalisation block

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
template <typename Tag, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void row_stats_kernel(
        const typename Tag::store* __restrict__ acts, const int* __restrict__ labels,
        const int* __restrict__ xlen, const int* __restrict__ ylen,
        Cell<typename Tag::comp>* __restrict__ cells, int maxT, int maxU, int A, int blank, int vec_ok) {
    using S = typename Tag::store;
    using C = typename Tag::comp;
    constexpr int V = Vec<Tag>::N;
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int q = uniform(blockIdx.x * WAVES + (threadIdx.x >> 6));
    if (q >= maxT * maxU) return;
    const int t = q / maxU, u = q - t * maxU;
    const int Tb = xlen[b], Ub = ylen[b] + 1;
    if (t >= Tb || u >= Ub) return;   // padded cell: never read

    const S* row = acts + (static_cast<size_t>(b) * maxT * maxU + q) * A;
    const bool has_lab = u < Ub - 1;
    int lab = blank;
    if (has_lab) {
        lab = labels[static_cast<size_t>(b) * (maxU - 1) + u];
        lab = lab < 0 ? 0 : (lab >= A ? A - 1 : lab);
    }
    const C xb = load1<Tag>(row + blank);
    const C xl = load1<Tag>(row + lab);

    int head, nvec, tail0;
    row_split<S>(reinterpret_cast<uintptr_t>(row), A, vec_ok != 0, head, nvec, tail0);

    C m = neg_inf<C>(), s = 0;
    for (int e = lane; e < head; e += 64) {
        C v[1] = {load1<Tag>(row + e)};
        absorb<C, 1>(v, m, s);
    }
    const uint4* vp = reinterpret_cast<const uint4*>(row + head);
    int i = lane;
    for (; i + 192 < nvec; i += 256) {
        const uint4 r0 = vp[i], r1 = vp[i + 64], r2 = vp[i + 128], r3 = vp[i + 192];
        C v[4 * V];
        unpack<Tag>(r0, v);
        unpack<Tag>(r1, v + V);
        unpack<Tag>(r2, v + 2 * V);
        unpack<Tag>(r3, v + 3 * V);
        absorb<C, 4 * V>(v, m, s);
    }
    for (; i < nvec; i += 64) {
        const uint4 r = vp[i];
        C v[V];
        unpack<Tag>(r, v);
        absorb<C, V>(v, m, s);
    }
    for (int e = tail0 + lane; e < A; e += 64) {
        C v[1] = {load1<Tag>(row + e)};
        absorb<C, 1>(v, m, s);
    }

    const C M = wave_max(m);
    const C shift = (M == neg_inf<C>()) ? C(0) : M;
    const C S_ = wave_sum(s * fast_exp(m - shift));
    const C logZ = shift + acc_log(S_);

    if (lane == 0) {
        const int D = maxT + maxU - 1;
        Cell<C> rec;
        rec.x = xb - logZ;
        rec.y = has_lab ? xl - logZ : C(0);
        rec.z = logZ;
        rec.w = 0;
        cells[(static_cast<size_t>(b) * D + (t + u)) * maxU + u] = rec;
    }
}

// ------------------------------------------------------------------------------------------
// Lattice recursion.  grid = N * dirs (dirs = 2: alpha block and beta block per sample run
// concurrently; dirs = 1: alpha only, forward scoring), block = ceil(maxU/64) wavefronts,
// thread u owns lattice column u and walks the anti-diagonals.
//
// Numerics: every kLatticeBlock diagonals the running values are re-centred on the block
// maximum and the shift is accumulated in an fp64 offset per sample and diagonal
// (offa / offb), so fp32 lattice values stay O(10) however long the utterance is; the
// fp32 round-off of the reference's un-scaled recursion (1 ulp of |alpha| ~ 6e3 is 5e-4 at
// T=1500,U=300: BASELINE.md section 3) does not build up.
template <typename L, bool MULTI>
__global__ __launch_bounds__(1024) void lattice_kernel(
        Cell<L>* __restrict__ cells, L* __restrict__ beta, double* __restrict__ offa,
        double* __restrict__ offb, double* __restrict__ ll_fwd, double* __restrict__ ll_bwd,
        L* __restrict__ costs_dev, const int* __restrict__ xlen, const int* __restrict__ ylen,
        int maxT, int maxU, int dirs) {
    constexpr int K = kLatticeBlock;
    __shared__ L edge[2][16];
    __shared__ L red[16];
    const int b = blockIdx.x / dirs;
    const int dir = blockIdx.x - b * dirs;
    const int u = threadIdx.x;
    const int lane = u & 63, wave = u >> 6, nwaves = blockDim.x >> 6;
    const int Tb = xlen[b], Ub = ylen[b] + 1;
    const int Db = Tb + Ub - 1;
    const int D = maxT + maxU - 1;
    const size_t base = static_cast<size_t>(b) * D * maxU;
    Cell<L>* c = cells + base;
    L* bt = beta + base;
    double* off = (dir == 0 ? offa : offb) + static_cast<size_t>(b) * D;
    const L NEG = neg_inf<L>();
    (void)nwaves; (void)lane; (void)wave; (void)edge; (void)red;

    auto block_max = [&](L v) -> L {
        L m = wave_max(v);
        if constexpr (MULTI) {
            if (lane == 0) red[wave] = m;
            __syncthreads();
            m = red[0];
            for (int w = 1; w < nwaves; ++w) m = vmax(m, red[w]);
            __syncthreads();
        }
        return m;
    };

    double Coff = 0.0;   // accumulated re-centring shift (block-uniform)

    if (dir == 0) {
        // ---------------- alpha: diagonal n is built from diagonal n-1 ----------------
        // lane u holds a = alpha~(n-1-u, u); it feeds (t+1,u) [blank] on its own lane and
        // (t,u+1) [label] on lane u+1; both use the SOURCE cell's log-probs.
        L a = (u == 0) ? L(0) : NEG;
        if (u == 0) { c[0].w = 0; off[0] = 0.0; }
        L pb[K], pl[K], nb[K], nl[K];
        auto fetch = [&](int n0, L* xb, L* xl) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int src = n0 + k - 1;          // source diagonal
                const int ts = src - u;
                const bool ok = (src < Db - 1) && (u < Ub) && (ts >= 0) && (ts < Tb);
                L x = 0, y = 0;
                if (ok) {
                    const Cell<L>* p = c + static_cast<size_t>(src) * maxU + u;
                    x = p->x; y = p->y;
                }
                xb[k] = x; xl[k] = y;
            }
        };
        fetch(1, pb, pl);
        for (int n0 = 1; n0 < Db; n0 += K) {
            fetch(n0 + K, nb, nl);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int n = n0 + k;
                if (n < Db) {                          // block-uniform
                    const L stay = a + pb[k];
                    const L emit = a + pl[k];
                    L up = wave_shr1(emit, NEG);
                    if constexpr (MULTI) {
                        if (lane == 63) edge[n & 1][wave] = emit;
                        __syncthreads();
                        if (lane == 0 && wave > 0) up = edge[n & 1][wave - 1];
                    }
                    const int t = n - u;
                    const bool valid = (u < Ub) && (t >= 0) && (t < Tb);
                    const L v = log_add(stay, up);
                    a = valid ? v : NEG;
                    if (valid) c[static_cast<size_t>(n) * maxU + u].w = a;
                    if (u == 0) off[n] = Coff;
                }
            }
            const L m = block_max(a);
            if (m != NEG) { a -= m; Coff += static_cast<double>(m); }
#pragma unroll
            for (int k = 0; k < K; ++k) { pb[k] = nb[k]; pl[k] = nl[k]; }
        }
        if (u == Ub - 1) {
            const double ll = static_cast<double>(a) + Coff +
                              static_cast<double>(c[static_cast<size_t>(Db - 1) * maxU + u].x);
            ll_fwd[b] = ll;
            costs_dev[b] = static_cast<L>(-ll);
        }
    } else {
        // ---------------- beta: diagonal n is built from diagonal n+1 ----------------
        // lane u holds bv = beta~(n+1-u, u); target (n-u, u) takes its own lane's value
        // [blank] and lane u+1's value [label], with the TARGET cell's log-probs.
        L bv = NEG;
        if (u == Ub - 1) {
            const size_t last = static_cast<size_t>(Db - 1) * maxU + u;
            bv = c[last].x;
            bt[last] = bv;
        }
        if (u == 0) off[Db - 1] = 0.0;
        L pb[K], pl[K], nb[K], nl[K];
        auto fetch = [&](int n0, L* xb, L* xl) {    // diagonals n0, n0-1, ... n0-K+1
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int n = n0 - k;
                const int t = n - u;
                const bool ok = (n >= 0) && (u < Ub) && (t >= 0) && (t < Tb);
                L x = 0, y = 0;
                if (ok) {
                    const Cell<L>* p = c + static_cast<size_t>(n) * maxU + u;
                    x = p->x; y = p->y;
                }
                xb[k] = x; xl[k] = y;
            }
        };
        fetch(Db - 2, pb, pl);
        for (int n0 = Db - 2; n0 >= 0; n0 -= K) {
            fetch(n0 - K, nb, nl);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int n = n0 - k;
                if (n >= 0) {                          // block-uniform
                    L right = wave_shl1(bv, NEG);
                    if constexpr (MULTI) {
                        if (lane == 0) edge[n & 1][wave] = bv;
                        __syncthreads();
                        if (lane == 63 && wave + 1 < nwaves) right = edge[n & 1][wave + 1];
                    }
                    const L stay = bv + pb[k];
                    const L emit = right + pl[k];
                    const int t = n - u;
                    const bool valid = (u < Ub) && (t >= 0) && (t < Tb);
                    const L v = log_add(stay, emit);
                    bv = valid ? v : NEG;
                    if (valid) bt[static_cast<size_t>(n) * maxU + u] = bv;
                    if (u == 0) off[n] = Coff;
                }
            }
            const L m = block_max(bv);
            if (m != NEG) { bv -= m; Coff += static_cast<double>(m); }
#pragma unroll
            for (int k = 0; k < K; ++k) { pb[k] = nb[k]; pl[k] = nl[k]; }
        }
        if (u == 0) ll_bwd[b] = static_cast<double>(bv) + Coff;
    }
}

// ------------------------------------------------------------------------------------------
// Gradient coefficients per lattice cell (skewed index space, fully coalesced).
// grid = (ceil(D*maxU/256), N), block = 256.  Overwrites the cell record in place.
//   c  = alpha + beta - ll - logZ                       (g_v = exp(x_v + c) for every v)
//   cb = exp(alpha + lp_blank + beta(t+1,u) - ll)        (t < T-1)
//      = exp(alpha + lp_blank - ll)                      (t = T-1, u = U-1)
//   cl = exp(alpha + lp_label + beta(t,u+1) - ll)        (u < U-1)
// Formulas: reference gpu_rnnt_kernel.h:161-174, docs/rnnt_notes.tex:138-145.
template <typename L>
__global__ __launch_bounds__(256) void coef_kernel(
        Cell<L>* __restrict__ cells, const L* __restrict__ beta, const double* __restrict__ offa,
        const double* __restrict__ offb, const double* __restrict__ ll_fwd,
        const int* __restrict__ labels, const int* __restrict__ xlen, const int* __restrict__ ylen,
        int maxT, int maxU) {
    const int b = blockIdx.y;
    const int D = maxT + maxU - 1;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= D * maxU) return;
    const int n = idx / maxU, u = idx - n * maxU;
    const int t = n - u;
    const int Tb = xlen[b], Ub = ylen[b] + 1;
    if (u >= Ub || t < 0 || t >= Tb) return;
    const size_t base = static_cast<size_t>(b) * D * maxU;
    Cell<L>* cp = cells + base + idx;
    const L* bp = beta + base + idx;
    const Cell<L> r = *cp;
    const double* oa = offa + static_cast<size_t>(b) * D;
    const double* ob = offb + static_cast<size_t>(b) * D;
    const double ll = ll_fwd[b];
    const double alpha = static_cast<double>(r.w) + oa[n] - ll;     // alpha(t,u) - ll
    const bool last_t = (t == Tb - 1), last_u = (u == Ub - 1);
    const double ob1 = (last_t && last_u) ? 0.0 : ob[n + 1];

    Cell<L> o;
    o.x = static_cast<L>(alpha + static_cast<double>(bp[0]) + ob[n] - static_cast<double>(r.z));
    L cb = 0, cl = 0;
    if (!last_t)
        cb = fast_exp(static_cast<L>(alpha + static_cast<double>(r.x) + static_cast<double>(bp[maxU]) + ob1));
    else if (last_u)
        cb = fast_exp(static_cast<L>(alpha + static_cast<double>(r.x)));
    int lab = -1;
    if (!last_u) {
        cl = fast_exp(static_cast<L>(alpha + static_cast<double>(r.y) + static_cast<double>(bp[maxU + 1]) + ob1));
        lab = labels[static_cast<size_t>(b) * (maxU - 1) + u];
    }
    o.y = cb;
    o.z = cl;
    o.w = static_cast<L>(lab);
    *cp = o;
}

// ------------------------------------------------------------------------------------------
// Pass B.  grid = (ceil(maxT*maxU / WAVES), N), block = WAVES*64; one wavefront per row.
// Padded rows (t >= T_b or u >= U_b) are zero-filled here, so no memset of the gradient
// tensor is needed (the reference does one: gpu_rnnt.h:107-110).
template <typename Tag, int WAVES, bool SCALED>
__global__ __launch_bounds__(WAVES * 64) void grad_kernel(
        const typename Tag::store* __restrict__ acts, typename Tag::store* __restrict__ grads,
        const Cell<typename Tag::comp>* __restrict__ cells, const int* __restrict__ xlen,
        const int* __restrict__ ylen, const typename Tag::comp* __restrict__ grad_scale,
        int maxT, int maxU, int A, int blank, int vec_ok) {
    using S = typename Tag::store;
    using C = typename Tag::comp;
    constexpr int V = Vec<Tag>::N;
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int q = uniform(blockIdx.x * WAVES + (threadIdx.x >> 6));
    if (q >= maxT * maxU) return;
    const int t = q / maxU, u = q - t * maxU;
    const int Tb = xlen[b], Ub = ylen[b] + 1;
    const size_t roff = (static_cast<size_t>(b) * maxT * maxU + q) * A;
    const S* row = acts + roff;
    S* grow = grads + roff;

    int head, nvec, tail0;
    row_split<S>(reinterpret_cast<uintptr_t>(row), A, vec_ok != 0, head, nvec, tail0);
    uint4* gp = reinterpret_cast<uint4*>(grow + head);

    if (t >= Tb || u >= Ub) {
        for (int e = lane; e < head; e += 64) store1<Tag>(grow + e, C(0));
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (int i = lane; i < nvec; i += 64) gp[i] = z;
        for (int e = tail0 + lane; e < A; e += 64) store1<Tag>(grow + e, C(0));
        return;
    }

    const int D = maxT + maxU - 1;
    const Cell<C> r = cells[(static_cast<size_t>(b) * D + (t + u)) * maxU + u];
    const C c = r.x, cb = r.y, cl = r.z;
    const int lab = static_cast<int>(r.w);
    C gs = 1;
    if constexpr (SCALED) gs = grad_scale[b];

    auto one = [&](int e, C x) -> C {
        C g = fast_exp(x + c);
        if (e == blank) g -= cb;
        if (e == lab) g -= cl;
        if constexpr (SCALED) g *= gs;
        return g;
    };
    auto packet = [&](int e0, C* v) {
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = fast_exp(v[j] + c);
        if (static_cast<unsigned>(blank - e0) < static_cast<unsigned>(V) ||
            static_cast<unsigned>(lab - e0) < static_cast<unsigned>(V)) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                if (e0 + j == blank) v[j] -= cb;
                if (e0 + j == lab) v[j] -= cl;
            }
        }
        if constexpr (SCALED) {
#pragma unroll
            for (int j = 0; j < V; ++j) v[j] *= gs;
        }
    };

    for (int e = lane; e < head; e += 64) store1<Tag>(grow + e, one(e, load1<Tag>(row + e)));
    const uint4* vp = reinterpret_cast<const uint4*>(row + head);
    int i = lane;
    for (; i + 192 < nvec; i += 256) {
        const uint4 r0 = vp[i], r1 = vp[i + 64], r2 = vp[i + 128], r3 = vp[i + 192];
        C v0[V], v1[V], v2[V], v3[V];
        unpack<Tag>(r0, v0); unpack<Tag>(r1, v1); unpack<Tag>(r2, v2); unpack<Tag>(r3, v3);
        packet(head + i * V, v0);
        packet(head + (i + 64) * V, v1);
        packet(head + (i + 128) * V, v2);
        packet(head + (i + 192) * V, v3);
        gp[i] = pack<Tag>(v0);
        gp[i + 64] = pack<Tag>(v1);
        gp[i + 128] = pack<Tag>(v2);
        gp[i + 192] = pack<Tag>(v3);
    }
    for (; i < nvec; i += 64) {
        const uint4 r0 = vp[i];
        C v0[V];
        unpack<Tag>(r0, v0);
        packet(head + i * V, v0);
        gp[i] = pack<Tag>(v0);
    }
    for (int e = tail0 + lane; e < A; e += 64) store1<Tag>(grow + e, one(e, load1<Tag>(row + e)));
}

}  // namespace rnnt
