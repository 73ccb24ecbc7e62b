/*
 * oracle/rnnt_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C CPU restatement of the RNN-T loss algorithm of the reference
 * (HawkAaron/warp-transducer, CPU path).  It exists only as the checker for
 * the HIP path: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product library (libwarprnnt.so) never
 * links, loads or calls anything in this directory.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this restatement against
 *   (1) every literal golden vector the reference's own tests hold
 *       (tests/test_cpu.cpp:18-26,79-109, tests/test_gpu.cu:117-133,
 *        pytorch_binding/test/test.py:52-161), and
 *   (2) the reference itself: oracle/_ref/libwarprnnt_ref.so is compiled from
 *       the reference sources where they lie (oracle/Makefile) and the
 *       committed fixtures the .npz files under tests/golden/ were produced by it
 *       (tests/golden/make_golden.py).
 *
 * Each function cites the reference file:line whose behaviour it restates.
 * The code is written from the maths, not transcribed.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>
#include <stdio.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------------------
 * Generic bodies, instantiated for float and double through the preprocessor.
 * ------------------------------------------------------------------------- */

#define DEFINE_ORACLE(SUFFIX, real, EXP, LOG, LOG1P, FABS)                                          \
                                                                                                     \
/* log(exp(a)+exp(b)) with the -inf short-circuits of                                                \
 * include/detail/rnnt_helper.h:16-24 (log_sum_exp). */                                              \
static real lse2_##SUFFIX(real a, real b) {                                                          \
    if (a == -(real)INFINITY) return b;                                                              \
    if (b == -(real)INFINITY) return a;                                                              \
    return (a > b) ? LOG1P(EXP(b - a)) + a : LOG1P(EXP(a - b)) + b;                                  \
}                                                                                                    \
                                                                                                     \
/* Row-wise log-softmax.  The reference CPU path expects its caller to have done                     \
 * this (README.md:38-39; tests/test.h:35-60 `softmax(..., applylog=true)`;                          \
 * pytorch_binding/warprnnt_pytorch/__init__.py:95-98).  The GPU path does it                        \
 * internally as max / exp-sum (include/detail/gpu_rnnt.h:73-80,                                     \
 * include/detail/reduce.h:45-104). */                                                               \
ORACLE_API void oracle_log_softmax_##SUFFIX(const real* x, size_t rows, int A, real* out) {          \
    _Pragma("omp parallel for schedule(static)")                                                     \
    for (long long r = 0; r < (long long)rows; ++r) {                                                \
        const real* xr = x + (size_t)r * A;                                                          \
        real* o = out + (size_t)r * A;                                                               \
        real m = -(real)INFINITY;                                                                    \
        for (int v = 0; v < A; ++v) if (xr[v] > m) m = xr[v];                                        \
        real s = 0;                                                                                  \
        for (int v = 0; v < A; ++v) s += EXP(xr[v] - m);                                             \
        real lz = m + LOG(s);                                                                        \
        for (int v = 0; v < A; ++v) o[v] = xr[v] - lz;                                               \
    }                                                                                                \
}                                                                                                    \
                                                                                                     \
/* One sample: alphas, (optionally) betas + sparse gradient wrt LOG-PROBS.                           \
 * lp      : this sample's (maxT,maxU,A) slab (only t<T,u<U is touched)                              \
 * grad    : same-shaped slab or NULL                                                                \
 * Follows include/detail/cpu_rnnt.h:146-173 (cost_and_grad_kernel),                                 \
 * :175-212 (compute_alphas), :214-270 (compute_betas_and_grad), and the                             \
 * blank/label cache of :115-128 (setup_probs).  Lattice arrays are compact                          \
 * (T x U), like :135-137. */                                                                        \
static real sample_##SUFFIX(const real* lp, real* grad, const int* labels, int T, int U,             \
                             int maxT, int maxU, int A, int blank, real* llb_out) {                  \
    size_t cells = (size_t)T * U;                                                                    \
    real* alpha = (real*)malloc(sizeof(real) * cells * 4);                                           \
    real* beta = alpha + cells;                                                                      \
    real* pb = beta + cells;   /* log p(blank | t,u)  */                                             \
    real* pl = pb + cells;     /* log p(y_u   | t,u)  */                                             \
    for (int t = 0; t < T; ++t)                                                                      \
        for (int u = 0; u < U; ++u) {                                                                \
            const real* row = lp + ((size_t)t * maxU + u) * A;                                       \
            pb[t * U + u] = row[blank];                                                              \
            pl[t * U + u] = (u < U - 1) ? row[labels[u]] : 0;                                        \
        }                                                                                            \
    /* forward variables: cpu_rnnt.h:181-209 */                                                      \
    alpha[0] = 0;                                                                                    \
    for (int t = 1; t < T; ++t) alpha[t * U] = alpha[(t - 1) * U] + pb[(t - 1) * U];                 \
    for (int u = 1; u < U; ++u) alpha[u] = alpha[u - 1] + pl[u - 1];                                 \
    for (int t = 1; t < T; ++t)                                                                      \
        for (int u = 1; u < U; ++u) {                                                                \
            real stay = alpha[(t - 1) * U + u] + pb[(t - 1) * U + u];                                \
            real emit = alpha[t * U + u - 1] + pl[t * U + u - 1];                                    \
            alpha[t * U + u] = lse2_##SUFFIX(emit, stay);                                            \
        }                                                                                            \
    real ll = alpha[cells - 1] + pb[cells - 1];                                                      \
    if (grad) {                                                                                      \
        /* zero the whole padded slab: cpu_rnnt.h:155-158 */                                         \
        memset(grad, 0, sizeof(real) * (size_t)maxT * maxU * A);                                     \
        /* backward variables: cpu_rnnt.h:222-236 */                                                 \
        beta[cells - 1] = pb[cells - 1];                                                             \
        for (int t = T - 2; t >= 0; --t)                                                             \
            beta[t * U + U - 1] = beta[(t + 1) * U + U - 1] + pb[t * U + U - 1];                     \
        for (int u = U - 2; u >= 0; --u)                                                             \
            beta[(T - 1) * U + u] = beta[(T - 1) * U + u + 1] + pl[(T - 1) * U + u];                 \
        for (int t = T - 2; t >= 0; --t)                                                             \
            for (int u = U - 2; u >= 0; --u) {                                                       \
                real stay = beta[(t + 1) * U + u] + pb[t * U + u];                                   \
                real emit = beta[t * U + u + 1] + pl[t * U + u];                                     \
                beta[t * U + u] = lse2_##SUFFIX(emit, stay);                                         \
            }                                                                                        \
        real llb = beta[0];                                                                          \
        if (llb_out) *llb_out = llb;                                                                 \
        /* sparse gradient wrt log-probs, normalised by the BACKWARD                                 \
         * likelihood as the reference does: cpu_rnnt.h:251-267.  Assignment                         \
         * (not accumulation): when labels[u]==blank the label write wins for                        \
         * t<T-1 exactly as in the reference's statement order. */                                   \
        for (int t = 0; t < T; ++t)                                                                  \
            for (int u = 0; u < U; ++u) {                                                            \
                real* g = grad + ((size_t)t * maxU + u) * A;                                         \
                real a = alpha[t * U + u];                                                           \
                if (t < T - 1)                                                                       \
                    g[blank] = -EXP(pb[t * U + u] + a + beta[(t + 1) * U + u] - llb);                \
                if (u < U - 1)                                                                       \
                    g[labels[u]] = -EXP(pl[t * U + u] + a + beta[t * U + u + 1] - llb);              \
            }                                                                                        \
        grad[((size_t)(T - 1) * maxU + (U - 1)) * A + blank] =                                       \
            -EXP(pb[cells - 1] + alpha[cells - 1] - llb);                                            \
    }                                                                                                \
    free(alpha);                                                                                     \
    return ll;                                                                                       \
}                                                                                                    \
                                                                                                     \
/* The reference's CPU contract (batch_first): input LOG-PROBS, output per-sample                    \
 * cost = -loglik and (if grads != NULL) the sparse gradient wrt log-probs.                          \
 * include/detail/cpu_rnnt.h:272-304 (cost_and_grad), :306-338 (score_forward);                      \
 * labels are padded (N, maxU-1): :299.  Returns 0. */                                               \
ORACLE_API int oracle_rnnt_logprobs_##SUFFIX(const real* log_probs, real* grads,                     \
        const int* labels, const int* label_lengths, const int* input_lengths,                       \
        int A, int N, int maxT, int maxU, int blank, real* costs) {                                  \
    size_t slab = (size_t)maxT * maxU * A;                                                           \
    _Pragma("omp parallel for schedule(dynamic)")                                                    \
    for (int b = 0; b < N; ++b) {                                                                    \
        int T = input_lengths[b], U = label_lengths[b] + 1;                                          \
        real ll = sample_##SUFFIX(log_probs + b * slab, grads ? grads + b * slab : NULL,             \
                                  labels + (size_t)b * (maxU - 1), T, U, maxT, maxU, A, blank,       \
                                  NULL);                                                             \
        costs[b] = -ll;                                                                              \
    }                                                                                                \
    return 0;                                                                                        \
}                                                                                                    \
                                                                                                     \
/* The reference's GPU contract: input raw LOGITS, output cost and the DENSE                         \
 * gradient wrt logits (include/detail/gpu_rnnt_kernel.h:143-179,                                    \
 * docs/rnnt_notes.tex:119-146).  Built as the bindings build it on CPU:                             \
 * log_softmax -> CPU loss -> chain rule through log_softmax                                         \
 *   g_logit[v] = g_lp[v] - softmax[v] * sum_v' g_lp[v']                                             \
 * (pytorch_binding/warprnnt_pytorch/__init__.py:67-68,95-98).                         \
 * scratch: caller-provided buffer of N*maxT*maxU*A reals (the log-probs). */                        \
ORACLE_API int oracle_rnnt_logits_##SUFFIX(const real* acts, real* grads, const int* labels,         \
        const int* label_lengths, const int* input_lengths, int A, int N, int maxT, int maxU,        \
        int blank, real* costs, real* scratch) {                                                     \
    size_t rows = (size_t)N * maxT * maxU;                                                           \
    oracle_log_softmax_##SUFFIX(acts, rows, A, scratch);                                             \
    oracle_rnnt_logprobs_##SUFFIX(scratch, grads, labels, label_lengths, input_lengths, A, N,        \
                                  maxT, maxU, blank, costs);                                         \
    if (!grads) return 0;                                                                            \
    _Pragma("omp parallel for schedule(static)")                                                     \
    for (long long r = 0; r < (long long)rows; ++r) {                                                \
        real* g = grads + (size_t)r * A;                                                             \
        const real* lp = scratch + (size_t)r * A;                                                    \
        real s = 0;                                                                                  \
        for (int v = 0; v < A; ++v) s += g[v];                                                       \
        if (s != 0)                                                                                  \
            for (int v = 0; v < A; ++v) g[v] -= EXP(lp[v]) * s;                                      \
    }                                                                                                \
    return 0;                                                                                        \
}                                                                                                    \
                                                                                                     \
/* The same, plus the SIZE OF THE TERMS each dense gradient element is made of:                      \
 *   mag[v] = |g_lp[v]| + softmax[v] * |sum_v' g_lp[v']|  >=  |g_logit[v]|.                          \
 * The blank / label entries are differences of two terms (include/detail/gpu_rnnt_kernel.h:159-176: \
 * exp(logpk + ...) minus the emission term); a rounding error of the arithmetic scales with the     \
 * terms, not with their difference, so a per-element tolerance relative to the result alone would   \
 * be unfair exactly there.  Everywhere else mag == |g_logit|.  Used by oracle.grad_check(). */      \
ORACLE_API int oracle_rnnt_logits_mag_##SUFFIX(const real* acts, real* grads, const int* labels,     \
        const int* label_lengths, const int* input_lengths, int A, int N, int maxT, int maxU,        \
        int blank, real* costs, real* scratch, real* mag) {                                          \
    size_t rows = (size_t)N * maxT * maxU;                                                           \
    oracle_log_softmax_##SUFFIX(acts, rows, A, scratch);                                             \
    oracle_rnnt_logprobs_##SUFFIX(scratch, grads, labels, label_lengths, input_lengths, A, N,        \
                                  maxT, maxU, blank, costs);                                         \
    _Pragma("omp parallel for schedule(static)")                                                     \
    for (long long r = 0; r < (long long)rows; ++r) {                                                \
        real* g = grads + (size_t)r * A;                                                             \
        real* m = mag + (size_t)r * A;                                                               \
        const real* lp = scratch + (size_t)r * A;                                                    \
        real s = 0;                                                                                  \
        for (int v = 0; v < A; ++v) s += g[v];                                                       \
        for (int v = 0; v < A; ++v) {                                                                \
            real soft = (s != 0) ? EXP(lp[v]) * s : 0;                                               \
            m[v] = FABS(g[v]) + FABS(soft);                                                          \
            g[v] -= soft;                                                                            \
        }                                                                                            \
    }                                                                                                \
    return 0;                                                                                        \
}

DEFINE_ORACLE(f32, float, expf, logf, log1pf, fabsf)
DEFINE_ORACLE(f64, double, exp, log, log1p, fabs)

ORACLE_API int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORACLE_API void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
