// rnnt_cpu.cpp -- host (RNNT_CPU) path of libwarprnnt.
//
// Contract of the reference's CPU location (include/detail/cpu_rnnt.h:253-304, README.md:38-39):
//   * activations are LOG-PROBABILITIES (the caller applied log_softmax);
//   * the gradient is the sparse d(loss)/d(log-probs): only the blank and label columns of
//     each valid (t,u) row are written, the rest of the slab is zeroed (batch_first);
//   * one OpenMP task per sample; all scratch comes from the caller's workspace.
// Both tensor layouts of the reference are addressed through strides:
//   batch_first  (B,T,U,V): row(b,t,u) = ((b*maxT + t)*maxU + u) * A
//   !batch_first (T,U,B,V): row(b,t,u) = ((t*maxU + u)*B + b) * A      (cpu_rnnt.h:140-144,294-295)
#include "rnnt_cpu.h"

#include <omp.h>

#include <cmath>
#include <cstring>
#include <limits>

namespace rnnt {

namespace {

template <typename R> inline R lse(R a, R b) {
    // -inf is the additive identity (reference include/detail/rnnt_helper.h:16-24)
    const R ninf = -std::numeric_limits<R>::infinity();
    if (a == ninf) return b;
    if (b == ninf) return a;
    const R hi = a > b ? a : b, lo = a > b ? b : a;
    return hi + std::log1p(std::exp(lo - hi));
}

template <typename R> struct SampleView {
    const R* lp;       // base of this sample's log-probs
    R* grad;           // base of this sample's gradient (or nullptr)
    size_t t_stride;   // elements between (t,u) and (t+1,u)
    size_t u_stride;   // elements between (t,u) and (t,u+1)
};

// Forward/backward over one sample.  Scratch rows: alpha, beta (T*U each) and the two gathered
// transition tables stay (T*U) / emit (T*U).
template <typename R>
R one_sample(const SampleView<R>& s, const int* y, int T, int U, int blank, R* scratch) {
    const size_t cells = static_cast<size_t>(T) * U;
    R* alpha = scratch;
    R* beta = alpha + cells;
    R* stay = beta + cells;   // log p(blank | t,u)
    R* emit = stay + cells;   // log p(y_u   | t,u), u < U-1

    for (int t = 0; t < T; ++t) {
        const R* row = s.lp + t * s.t_stride;
        R* st = stay + static_cast<size_t>(t) * U;
        R* em = emit + static_cast<size_t>(t) * U;
        for (int u = 0; u < U; ++u, row += s.u_stride) {
            st[u] = row[blank];
            em[u] = (u + 1 < U) ? row[y[u]] : R(0);
        }
    }

    // alpha, one time row at a time; row 0 is a running sum of label emissions.
    alpha[0] = 0;
    for (int u = 1; u < U; ++u) alpha[u] = alpha[u - 1] + emit[u - 1];
    for (int t = 1; t < T; ++t) {
        const R* prev = alpha + static_cast<size_t>(t - 1) * U;
        const R* pst = stay + static_cast<size_t>(t - 1) * U;
        R* cur = alpha + static_cast<size_t>(t) * U;
        const R* em = emit + static_cast<size_t>(t) * U;
        cur[0] = prev[0] + pst[0];
        for (int u = 1; u < U; ++u) cur[u] = lse(prev[u] + pst[u], cur[u - 1] + em[u - 1]);
    }
    const R loglik = alpha[cells - 1] + stay[cells - 1];
    if (!s.grad) return loglik;

    // beta, last time row first.
    {
        R* cur = beta + static_cast<size_t>(T - 1) * U;
        const R* st = stay + static_cast<size_t>(T - 1) * U;
        const R* em = emit + static_cast<size_t>(T - 1) * U;
        cur[U - 1] = st[U - 1];
        for (int u = U - 2; u >= 0; --u) cur[u] = cur[u + 1] + em[u];
    }
    for (int t = T - 2; t >= 0; --t) {
        const R* nxt = beta + static_cast<size_t>(t + 1) * U;
        R* cur = beta + static_cast<size_t>(t) * U;
        const R* st = stay + static_cast<size_t>(t) * U;
        const R* em = emit + static_cast<size_t>(t) * U;
        cur[U - 1] = nxt[U - 1] + st[U - 1];
        for (int u = U - 2; u >= 0; --u) cur[u] = lse(nxt[u] + st[u], cur[u + 1] + em[u]);
    }
    const R norm = beta[0];   // the reference normalises by the backward likelihood (cpu_rnnt.h:251)

    for (int t = 0; t < T; ++t) {
        R* grow = s.grad + t * s.t_stride;
        const R* a = alpha + static_cast<size_t>(t) * U;
        const R* bcur = beta + static_cast<size_t>(t) * U;
        const R* bnxt = bcur + U;
        const R* st = stay + static_cast<size_t>(t) * U;
        const R* em = emit + static_cast<size_t>(t) * U;
        for (int u = 0; u < U; ++u, grow += s.u_stride) {
            if (t + 1 < T) grow[blank] = -std::exp(a[u] + st[u] + bnxt[u] - norm);
            if (u + 1 < U) grow[y[u]] = -std::exp(a[u] + em[u] + bcur[u + 1] - norm);
        }
    }
    s.grad[(T - 1) * s.t_stride + (U - 1) * s.u_stride + blank] =
        -std::exp(alpha[cells - 1] + stay[cells - 1] - norm);
    return loglik;
}

// offsets != nullptr: PACKED layout (include/rnnt.h compute_rnnt_loss_packed) -- sample b is T_b x U_b rows
// starting at row offsets[b] (host array of N+1 cumulative row counts), row (t,u) at t*U_b + u.
template <typename R>
rnntStatus_t run(const R* log_probs, R* grads, const int* labels, const int* label_lengths,
                 const int* input_lengths, int A, int N, R* costs, void* workspace, const rnntOptions& opt,
                 const long long* offsets = nullptr) {
    const int maxT = opt.maxT, maxU = opt.maxU, blank = opt.blank_label;
    if (blank < 0 || blank >= A) return RNNT_STATUS_INVALID_VALUE;
    const size_t per_sample = static_cast<size_t>(maxT) * maxU * 4;   // reals of scratch per sample
    const size_t slab = static_cast<size_t>(maxT) * maxU * A;
    const bool bf = opt.batch_first;
    R* scratch_all = static_cast<R*>(workspace);
    int bad = 0;

    // num_threads == 0 -> the OpenMP runtime default (reference cpu_rnnt.h:28-32), without the
    // reference's process-global omp_set_num_threads side effect.
    const int threads = opt.num_threads > 0 ? static_cast<int>(opt.num_threads) : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic) num_threads(threads) if (N > 1)
    for (int b = 0; b < N; ++b) {
        const int T = input_lengths[b], U = label_lengths[b] + 1;
        if (T <= 0 || T > maxT || U <= 0 || U > maxU) {
#pragma omp atomic write
            bad = 1;
            continue;
        }
        SampleView<R> s;
        if (offsets != nullptr) {
            if (offsets[b + 1] - offsets[b] != static_cast<long long>(T) * U) {
#pragma omp atomic write
                bad = 1;
                continue;
            }
            const size_t base = static_cast<size_t>(offsets[b]) * A;
            s.lp = log_probs + base;
            s.grad = grads ? grads + base : nullptr;
            s.u_stride = static_cast<size_t>(A);
            s.t_stride = s.u_stride * U;
            if (grads) std::memset(s.grad, 0, sizeof(R) * static_cast<size_t>(T) * U * A);
        } else {
        const size_t base = bf ? b * slab : static_cast<size_t>(b) * A;
        s.lp = log_probs + base;
        s.grad = grads ? grads + base : nullptr;
        s.u_stride = bf ? static_cast<size_t>(A) : static_cast<size_t>(N) * A;
        s.t_stride = s.u_stride * maxU;
        if (grads && bf) std::memset(s.grad, 0, sizeof(R) * slab);   // cpu_rnnt.h:155-158
        }
        const R ll = one_sample<R>(s, labels + static_cast<size_t>(b) * (maxU - 1), T, U, blank,
                                   scratch_all + b * per_sample);
        costs[b] = -ll;
    }
    return bad ? RNNT_STATUS_INVALID_VALUE : RNNT_STATUS_SUCCESS;
}

}  // namespace

size_t cpu_workspace_bytes(int maxT, int maxU, int minibatch, size_t lat) {
    return static_cast<size_t>(maxT) * maxU * 4 * lat * minibatch;
}

rnntStatus_t cpu_rnnt_f32(const float* log_probs, float* grads, const int* labels, const int* label_lengths,
                          const int* input_lengths, int A, int N, float* costs, void* workspace,
                          const rnntOptions& opt) {
    return run<float>(log_probs, grads, labels, label_lengths, input_lengths, A, N, costs, workspace, opt);
}

rnntStatus_t cpu_rnnt_f64(const double* log_probs, double* grads, const int* labels, const int* label_lengths,
                          const int* input_lengths, int A, int N, double* costs, void* workspace,
                          const rnntOptions& opt) {
    return run<double>(log_probs, grads, labels, label_lengths, input_lengths, A, N, costs, workspace, opt);
}

rnntStatus_t cpu_rnnt_packed(const void* log_probs, void* grads, const int* labels, const int* label_lengths,
                             const int* input_lengths, const long long* offsets, int A, int N, void* costs,
                             void* workspace, const rnntOptions& opt, bool fp64) {
    if (fp64)
        return run<double>(static_cast<const double*>(log_probs), static_cast<double*>(grads), labels, label_lengths,
                           input_lengths, A, N, static_cast<double*>(costs), workspace, opt, offsets);
    return run<float>(static_cast<const float*>(log_probs), static_cast<float*>(grads), labels, label_lengths,
                      input_lengths, A, N, static_cast<float*>(costs), workspace, opt, offsets);
}

}  // namespace rnnt
