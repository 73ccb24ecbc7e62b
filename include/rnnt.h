/* rnnt.h -- C-ABI of the MI355X-native RNN-T loss library (libwarprnnt.so).
 *
 * Source-compatible with the reference's public header
 * (HawkAaron/warp-transducer include/rnnt.h:1-147): the same enums, the same
 * by-value `rnntOptions`, and the same five exported symbols with the same
 * argument meaning and status codes, so existing bindings compile and link
 * against this library unchanged.  Each declaration cites the reference
 * interface it replaces.  Everything under "Extensions" is new and optional.
 *
 * Contract kept from the reference:
 *   - the library never allocates device or host memory: the caller asks
 *     get_workspace_size() and passes a workspace in the same memory space as
 *     the activations (reference src/rnnt_entrypoint.cpp:96-128, README.md:36-37);
 *     the one opt-in exception (host memory, off by default) is rnnt_host_staging() below;
 *   - tensors are dense row-major (B, T, U, V), U = max_label_len + 1, labels
 *     are a padded (B, U-1) int32 array (reference include/rnnt.h:69-80,
 *     include/detail/gpu_rnnt_kernel.h:19);
 *   - RNNT_GPU: activations are raw LOGITS (log-softmax is done inside),
 *     gradients are dense d(loss)/d(logits); activations, gradients, workspace,
 *     flat_labels, label_lengths and input_lengths are DEVICE pointers; costs is
 *     a HOST pointer; the call returns after the stream has been synchronised
 *     (reference include/detail/gpu_rnnt.h:75-80,107-110,208-213,
 *     include/detail/gpu_rnnt_kernel.h:17-19);
 *   - RNNT_CPU: activations are LOG-PROBS (caller applied log_softmax),
 *     gradients are the sparse d(loss)/d(log-probs); all pointers are host
 *     (reference include/detail/cpu_rnnt.h:253-304);
 *   - gradients == NULL means score only (reference src/rnnt_entrypoint.cpp:65-72).
 */
#pragma once

#ifdef __cplusplus
#include <cstddef>
#include <cstdint>
extern "C" {
#else
#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>
#endif

/* Opaque stream handle.  The reference forward-declares CUDA's CUstream
 * (include/rnnt.h:13-14); the typedef name is kept so callers compile, and the
 * value is interpreted as a hipStream_t (NULL = the default stream). */
typedef struct CUstream_st* CUstream;

/* reference include/rnnt.h:16-22 */
typedef enum {
    RNNT_STATUS_SUCCESS = 0,
    RNNT_STATUS_MEMOPS_FAILED = 1,
    RNNT_STATUS_INVALID_VALUE = 2,
    RNNT_STATUS_EXECUTION_FAILED = 3,
    RNNT_STATUS_UNKNOWN_ERROR = 4
} rnntStatus_t;

/* API version of the library; returns 1 (reference include/rnnt.h:25,
 * src/rnnt_entrypoint.cpp:14-16). */
int get_warprnnt_version();

/* Text for a status code, same strings as the reference
 * (include/rnnt.h:31, src/rnnt_entrypoint.cpp:18-35). */
const char* rnntGetStatusString(rnntStatus_t status);

/* reference include/rnnt.h:33-36 */
typedef enum {
    RNNT_CPU = 0,
    RNNT_GPU = 1
} rnntComputeLocation;

/* Options, passed BY VALUE; zero-initialise before filling
 * (reference include/rnnt.h:43-64; 32 bytes on LP64). */
struct rnntOptions {
    rnntComputeLocation loc;   /* where to compute: RNNT_CPU | RNNT_GPU          */
    unsigned int num_threads;  /* RNNT_CPU: OpenMP threads (0 = runtime default)  */
    CUstream stream;           /* RNNT_GPU: HIP stream the work is enqueued on    */
    int blank_label;           /* index of the blank symbol                       */
    int maxT;                  /* time dimension of the activation tensor         */
    int maxU;                  /* label dimension of the tensor (max_label_len+1) */
    bool batch_first;          /* RNNT_CPU layout flag; GPU is always (B,T,U,V)   */
};
#ifndef __cplusplus
typedef struct rnntOptions rnntOptions;
#endif

/* RNN-T loss (negative log-likelihood per sample) and, when gradients != NULL,
 * its gradient with respect to the activations.  fp32.
 * Replaces reference include/rnnt.h:104-113 / src/rnnt_entrypoint.cpp:38-93.
 * Returns RNNT_STATUS_INVALID_VALUE for NULL activations / flat_labels /
 * label_lengths / input_lengths / costs / workspace, for alphabet_size,
 * minibatch, options.maxT or options.maxU <= 0, and for an unknown options.loc.
 * NON-FINITE LOGITS (RNNT_GPU; every entry point, also the packed and the additive-joint ones): a NaN or +inf logit -- or a
 * row of -inf only -- in a row INSIDE a sample's T_b x U_b lattice makes that sample's cost NaN and the gradient of each of
 * its in-lattice rows NaN, as the reference's arithmetic does (include/detail/reduce.h:85,103 -> gpu_rnnt_kernel.h:5-9 ->
 * rnnt_helper.h:16-24); its padded rows stay zero, the other samples of the batch are not affected, and the status is
 * RNNT_STATUS_SUCCESS (a NaN loss is a result, as in the reference).  Values in PADDED rows are never read.  Single -inf
 * logits are ordinary (probability zero) -- unless they leave a sample NO alignment of non-zero probability (a label it must
 * emit, or the blank, masked wherever it could be emitted): that sample's cost is +inf and its in-lattice gradients are NaN,
 * which is where the reference's arithmetic ends too (ll = -inf; exp(alpha + beta - ll) = exp(-inf + inf)).
 * A LABEL EQUAL TO THE BLANK SYMBOL is legal and each location keeps its reference's answer -- they differ:
 *   RNNT_GPU (every entry point, packed and additive-joint included): both the blank and the label correction are
 *     subtracted from the blank column of such a cell (reference include/detail/gpu_rnnt_kernel.h:161-174: independent
 *     `if`s) -- the true derivative; pinned against fp64 autograd by tests/test_gpu_label_equals_blank.py;
 *   RNNT_CPU: the label term is ASSIGNED after the blank term and overwrites it (reference include/detail/cpu_rnnt.h:253-267);
 *     kept as is -- it is what a caller of the reference's CPU location gets (tests/test_cpu_location.py).
 * The costs agree in both.
 * IN PLACE (RNNT_GPU, extension revision 5; every materialised entry point, padded and packed, one-call and two-phase):
 * `gradients == activations` is a supported call -- the gradient overwrites the logits it was computed from, which halves
 * the activation footprint of a training step (8 GB at N=128,T=150,U=21,A=5000 fp32).  The gradient pass reads an element
 * and writes the same element from the same thread, after the statistics pass has finished with the logits; the two
 * pointers are not declared __restrict__ for that reason.  The reference cannot do this: it memsets the gradient tensor
 * before it reads the activations (include/detail/gpu_rnnt.h:107-110).  Partial overlap of the two tensors is NOT supported
 * (RNNT_STATUS_INVALID_VALUE).  Results are bit-identical to the out-of-place call (tests/test_gpu_parity.py). */
rnntStatus_t compute_rnnt_loss(const float* const activations,
                               float* gradients,
                               const int* const flat_labels,
                               const int* const label_lengths,
                               const int* const input_lengths,
                               int alphabet_size,
                               int minibatch,
                               float* costs,
                               void* workspace,
                               rnntOptions options);

/* fp64 twin (reference include/rnnt.h:115-124, src/rnnt_entrypoint.cpp:130-185).
 * On RNNT_GPU the lattice recursion is carried in fp64 as well. */
rnntStatus_t compute_rnnt_loss_fp64(const double* const activations,
                                    double* gradients,
                                    const int* const flat_labels,
                                    const int* const label_lengths,
                                    const int* const input_lengths,
                                    int alphabet_size,
                                    int minibatch,
                                    double* costs,
                                    void* workspace,
                                    rnntOptions options);

/* Bytes of workspace compute_rnnt_loss* needs for these dimensions, in the
 * memory space of the activations (gpu = true: device).  dtype_size is the
 * element size of the activations (2, 4 or 8).  The reference's formula
 * (src/rnnt_entrypoint.cpp:96-128: (3 T U + 2) N values) is private to it; callers always query,
 * so only the signature and the no-internal-malloc contract are kept.  Here: five lattice values per
 * cell of the diagonal-skewed lattice in per-sample blocks, + the part of the 16-byte-per-row coefficient
 * table that does not overlay blocks already consumed (all of it up to 32 MB, an eighth beyond):
 * N=64,T=1500,U=301 fp32 0.78 GB (reference 0.35), N=128,T=150,U=21 19 MB (4.8).  Monotone in every argument.
 * Replaces reference include/rnnt.h:139-143.  INVALID_VALUE on dims <= 0. */
rnntStatus_t get_workspace_size(int maxT, int maxU,
                                int minibatch,
                                bool gpu,
                                size_t* size_bytes,
#ifdef __cplusplus
                                size_t dtype_size = sizeof(float));
#else
                                size_t dtype_size);
#endif

/* ------------------------------------------------------------------------- *
 * Extensions (not in the reference).  GPU only.                              *
 * ------------------------------------------------------------------------- */

/* bf16 / fp16 activations and gradients (raw 16-bit storage), fp32 lattice and
 * fp32 costs.  Same contract as compute_rnnt_loss with options.loc == RNNT_GPU;
 * size the workspace with dtype_size = 2.  RNNT_CPU -> INVALID_VALUE. */
rnntStatus_t compute_rnnt_loss_bf16(const uint16_t* const activations,
                                    uint16_t* gradients,
                                    const int* const flat_labels,
                                    const int* const label_lengths,
                                    const int* const input_lengths,
                                    int alphabet_size,
                                    int minibatch,
                                    float* costs,
                                    void* workspace,
                                    rnntOptions options);

rnntStatus_t compute_rnnt_loss_fp16(const uint16_t* const activations,
                                    uint16_t* gradients,
                                    const int* const flat_labels,
                                    const int* const label_lengths,
                                    const int* const input_lengths,
                                    int alphabet_size,
                                    int minibatch,
                                    float* costs,
                                    void* workspace,
                                    rnntOptions options);

/* Asynchronous form: identical work, but `costs_device` is a DEVICE array of
 * `minibatch` elements (float for dtype_size 2/4, double for 8), nothing is
 * copied to the host and the stream is NOT synchronised -- the call only
 * enqueues (hipGraph-capturable).  `grad_scale_device`, when not NULL, points
 * at `minibatch` device floats (doubles for fp64) that multiply each sample's
 * gradient inside the gradient kernel (folds the autograd `grads.mul_(grad_out)`
 * of reference pytorch_binding/warprnnt_pytorch/__init__.py:47-50 into the
 * write-back).  dtype_code: 0 = fp32, 1 = fp64, 2 = bf16, 3 = fp16. */
rnntStatus_t compute_rnnt_loss_async(const void* activations,
                                     void* gradients,
                                     const int* const flat_labels,
                                     const int* const label_lengths,
                                     const int* const input_lengths,
                                     int alphabet_size,
                                     int minibatch,
                                     void* costs_device,
                                     const void* grad_scale_device,
                                     void* workspace,
                                     rnntOptions options,
                                     int dtype_code);

/* Batch-sharded step for one rank of a multi-GPU job (samples are independent, every rank runs the path on
 * its own slab of the batch, gradients stay local, and the data path needs exactly ONE collective): compute_rnnt_loss_async
 * on this rank's shard, then [summed loss, sample count] of the shard as two fp64 values into `loss_sum_count_device`, then --
 * when `rccl_comm` is not NULL -- one in-place ncclAllReduce(sum) of those 16 bytes over the communicator (an `ncclComm_t` of
 * RCCL, passed as void*; shards may be ragged, the mean is sum / count of the reduced pair), all enqueued on options.stream:
 * no host copy, no synchronisation.  rccl_comm == NULL leaves the local pair (single GPU, or a caller with its own
 * collective).  Not in the reference (no multi-device layer).
 * WHICH RCCL: the library has no link dependency on RCCL and calls ncclAllReduce through a pointer -- which must belong to
 * the SAME RCCL copy that created `rccl_comm` (a process can hold two: PyTorch ships torch/lib/librccl.so next to
 * /opt/rocm/lib/librccl.so; a communicator handed to the other copy is undefined behaviour).  The pointer is, in order:
 * the one registered with rnnt_set_rccl_all_reduce(); else the ncclAllReduce of the ONE librccl already mapped into the
 * process (opened with RTLD_NOLOAD: nothing new is loaded); else -- none mapped -- librccl.so.1 / librccl.so by name.
 * Two different copies mapped and none registered: EXECUTION_FAILED and a line on stderr naming them, never a guess --
 * from rnnt_sharded_prepare(), NOT from this call (extension revision 5): a communicator must be introduced once with
 * rnnt_sharded_prepare(comm) before its first step; RCCL is resolved there, on every rank, before any collective can be
 * entered, so a rank that cannot resolve it fails where its launcher can still stop the job.  A non-NULL `rccl_comm` that
 * was not prepared is INVALID_VALUE (nothing enqueued) -- a programming error every rank of a job makes alike.
 * EXECUTION_FAILED from this call means the collective itself failed.
 * ALL RANKS OR NONE, without exception: once the arguments every rank shares are accepted (loss_sum_count_device, loc, a
 * prepared communicator), a rank whose LOCAL part fails (INVALID_VALUE for its shard's shape, a launch error) still joins
 * the collective, with a NaN pair, and then returns its own status: its peers are not left blocked in ncclAllReduce, and
 * every rank's reduced loss is NaN (tests/test_gpu_sharded_rccl.py runs this with two ranks, the all-reduce carried by gloo
 * through a registered function).
 * STATUS OF THIS ENTRY: run on one-rank RCCL communicators, with two processes up to RCCL's refusal of two ranks on one
 * device, and with two ranks over a registered (gloo-carried) all-reduce; no box with two GPUs has executed it yet. */
rnntStatus_t compute_rnnt_loss_sharded(const void* activations,
                                       void* gradients,
                                       const int* const flat_labels,
                                       const int* const label_lengths,
                                       const int* const input_lengths,
                                       int alphabet_size,
                                       int minibatch,
                                       void* costs_device,
                                       const void* grad_scale_device,
                                       double* loss_sum_count_device,
                                       void* rccl_comm,
                                       void* workspace,
                                       rnntOptions options,
                                       int dtype_code);

/* The ncclAllReduce compute_rnnt_loss_sharded calls (see there): the address of `ncclAllReduce` in the RCCL copy the caller's
 * communicators come from, e.g. dlsym(handle_of_that_librccl, "ncclAllReduce").  NULL un-registers and makes the next
 * sharded call look again.  rnnt_rccl_source() says where the pointer in use came from ("registered by the caller", the
 * path of the mapped library, the name it was opened by, or "" when there is none) -- for logs and tests. */
void rnnt_set_rccl_all_reduce(void* nccl_all_reduce_fn);
const char* rnnt_rccl_source(void);

/* Introduce a communicator to compute_rnnt_loss_sharded (extension revision 5): resolves the ncclAllReduce that will be
 * called for it (see WHICH RCCL above) and remembers the pair.  Call it on EVERY rank right after the communicator is
 * made (and after rnnt_set_rccl_all_reduce, if that is used): RNNT_STATUS_EXECUTION_FAILED here -- no RCCL found, or two
 * copies mapped and none registered -- comes before any collective exists, so the ranks can still agree not to step.
 * INVALID_VALUE for NULL.  Preparing a communicator again re-resolves it.  rnnt_sharded_release() forgets it (call it
 * before ncclCommDestroy: a later communicator may reuse the address).  Host-side bookkeeping only: no device work, no
 * allocation on the device. */
rnntStatus_t rnnt_sharded_prepare(void* rccl_comm);
void rnnt_sharded_release(void* rccl_comm);

/* Two-phase form for autograd frameworks ("fused backward").
 * compute_rnnt_loss_fwd enqueues the row statistics, the lattice and -- with prepare_backward != 0 --
 * the gradient-coefficient table, and writes the costs to `costs_device`; compute_rnnt_loss_bwd,
 * called later with the SAME activations, workspace and options, enqueues only the gradient kernel,
 * multiplying sample b's gradient by grad_scale_device[b] (float; double for fp64; NULL = 1).
 * Between the two calls only the workspace has to stay alive and untouched -- not a gradient tensor
 * the size of the activations, which the reference's binding keeps in its autograd context and
 * then rescales twice (pytorch_binding/warprnnt_pytorch/__init__.py:24,36-50).  Enqueue only, no
 * synchronisation, DEVICE costs; dtype_code as compute_rnnt_loss_async. */
rnntStatus_t compute_rnnt_loss_fwd(const void* activations,
                                   const int* const flat_labels,
                                   const int* const label_lengths,
                                   const int* const input_lengths,
                                   int alphabet_size,
                                   int minibatch,
                                   void* costs_device,
                                   void* workspace,
                                   rnntOptions options,
                                   int dtype_code,
                                   int prepare_backward);

rnntStatus_t compute_rnnt_loss_bwd(const void* activations,
                                   void* gradients,
                                   const void* grad_scale_device,
                                   int alphabet_size,
                                   int minibatch,
                                   void* workspace,
                                   rnntOptions options,
                                   int dtype_code);

/* The forward and the backward log-likelihood of every sample, log P(y|x) read off the end of the alpha recursion and
 * off the start of the beta recursion (the reference's llForward / llBackward, include/detail/gpu_rnnt.h:92-105,
 * whose CPU path compares them as a sanity check: include/detail/cpu_rnnt.h:167-170).  Reads them out of a workspace
 * that a gradient-computing call (any materialised-path entry above, same maxT / maxU / minibatch / dtype_code) has
 * filled: two HOST arrays of `minibatch` doubles, natural logs; synchronises options.stream.  A score-only call runs
 * no beta recursion and leaves ll_backward undefined.  Their difference is the cheapest whole-lattice numerical guard
 * there is; tests/ bound it. */
rnntStatus_t compute_rnnt_loss_likelihoods(const void* workspace,
                                           int minibatch,
                                           rnntOptions options,
                                           int dtype_code,
                                           double* ll_forward_host,
                                           double* ll_backward_host);

/* Debug aid -- the counterpart of the reference's -DDEBUG_KERNEL dumps of the alpha / beta tables
 * (include/detail/gpu_rnnt.h:136-156,175-191; CPU: include/detail/cpu_rnnt.h:197-207,238-248), as a call instead of a build
 * flag.  Writes the forward and backward variables of ONE sample of a workspace that a gradient-computing materialised-path call
 * (same maxT / maxU / minibatch / dtype_code) has filled into two DEVICE arrays of maxT * maxU doubles each: natural (t, u) order,
 * natural logs, alpha(t,u) = log P(y_1..u emitted by time t), beta(t,u) as in include/detail/gpu_rnnt_kernel.h:79-113
 * (beta(0,0) = log P(y|x)); cells outside the sample's T_b x U_b lattice are NaN.  The workspace's private layout (diagonal-skewed,
 * base-2, re-centred per chunk with fp64 offsets) is undone here so that no caller has to know it.  input_lengths /
 * label_lengths: the device arrays of the call.  Enqueue only (one small kernel on options.stream), nothing allocated.
 * WHICH SAMPLES ARE STILL THERE.  The record table of the gradient stage overlays the lattice data of the samples in front
 * of the ones being processed (that is what keeps the workspace at 2.2x the reference's instead of 3.4x on long utterances).
 * After a call that computed the coefficient table (gradients != NULL, or compute_rnnt_loss_fwd with prepare_backward != 0)
 * whose table exceeds 32 MB, the FIRST samples of the batch have lost their alpha / beta: for those the two arrays come back
 * all NaN.  Every sample is intact after a score-only call (gradients == NULL / prepare_backward == 0: same alpha, and a
 * score-only call runs no beta sweep: beta is undefined then) -- and in any call whose record table is at most 32 MB
 * (T * U * minibatch <= 2 M cells for an fp32 lattice): debugging shapes.  The likelihoods of
 * compute_rnnt_loss_likelihoods are kept for every sample in every case. */
rnntStatus_t compute_rnnt_loss_lattice_dump(const void* workspace,
                                            const int* const label_lengths,
                                            const int* const input_lengths,
                                            int minibatch,
                                            int sample,
                                            rnntOptions options,
                                            int dtype_code,
                                            double* alpha_device,
                                            double* beta_device);

/* FastEmit regularisation (Yu et al., "FastEmit", ICASSP 2021, in the form NVIDIA
 * NeMo's RNN-T loss uses): the gradient of every LABEL transition's log-probability is scaled by
 * (1 + fastemit_lambda), which pushes the model to emit earlier; the returned costs are the plain
 * negative log-likelihoods.  fastemit_lambda = 0 is exactly compute_rnnt_loss_async /
 * compute_rnnt_loss_fwd; negative or NaN -> INVALID_VALUE.  compute_rnnt_loss_fwd_fastemit pairs with
 * the unchanged compute_rnnt_loss_bwd (the factor lives in the coefficient table).  GPU only. */
rnntStatus_t compute_rnnt_loss_fastemit(const void* activations,
                                        void* gradients,
                                        const int* const flat_labels,
                                        const int* const label_lengths,
                                        const int* const input_lengths,
                                        int alphabet_size,
                                        int minibatch,
                                        void* costs_device,
                                        const void* grad_scale_device,
                                        void* workspace,
                                        rnntOptions options,
                                        int dtype_code,
                                        float fastemit_lambda);

rnntStatus_t compute_rnnt_loss_fwd_fastemit(const void* activations,
                                            const int* const flat_labels,
                                            const int* const label_lengths,
                                            const int* const input_lengths,
                                            int alphabet_size,
                                            int minibatch,
                                            void* costs_device,
                                            void* workspace,
                                            rnntOptions options,
                                            int dtype_code,
                                            int prepare_backward,
                                            float fastemit_lambda);

/* PACKED ("compact") activations (variable-length compaction).  The reference pads every
 * sample to (maxT, maxU) and its gradient kernel writes zeros over the padding
 * (include/detail/gpu_rnnt_kernel.h:159); here sample b is only its T_b x U_b real rows, stored back to back:
 *   activations / gradients : (total_rows, alphabet_size), total_rows = sum_b T_b * U_b,  U_b = label_lengths[b] + 1
 *   row (b, t, u)           : row_offsets[b] + t * U_b + u
 *   row_offsets             : DEVICE int64[minibatch + 1], row_offsets[0] = 0, row_offsets[b+1] = row_offsets[b] + T_b * U_b
 *   total_rows              : the same total on the HOST (grid sizes)
 * Labels, lengths, options.maxT / maxU (the maxima over the batch: lattice and workspace are sized by them, the
 * workspace query is unchanged), costs, grad_scale, dtype codes, stream semantics and FastEmit are as for
 * compute_rnnt_loss_async / _fwd / _bwd / _fastemit.  No byte of padding is read or written.  Both tensors must
 * be 16-byte aligned when gradients are requested (INVALID_VALUE otherwise); total_rows must be in
 * (0, minibatch * maxT * maxU].  compute_rnnt_loss_packed also accepts options.loc == RNNT_CPU with the CPU
 * location's contract (every array incl. row_offsets and costs on the HOST, log-probabilities in, sparse
 * log-prob gradients out, fp32 / fp64, no scale, lambda = 0); the _fwd / _bwd pair is GPU only. */
rnntStatus_t compute_rnnt_loss_packed(const void* activations,
                                      void* gradients,
                                      const int* const flat_labels,
                                      const int* const label_lengths,
                                      const int* const input_lengths,
                                      const long long* const row_offsets,
                                      long long total_rows,
                                      int alphabet_size,
                                      int minibatch,
                                      void* costs_device,
                                      const void* grad_scale_device,
                                      void* workspace,
                                      rnntOptions options,
                                      int dtype_code,
                                      float fastemit_lambda);

rnntStatus_t compute_rnnt_loss_packed_fwd(const void* activations,
                                          const int* const flat_labels,
                                          const int* const label_lengths,
                                          const int* const input_lengths,
                                          const long long* const row_offsets,
                                          long long total_rows,
                                          int alphabet_size,
                                          int minibatch,
                                          void* costs_device,
                                          void* workspace,
                                          rnntOptions options,
                                          int dtype_code,
                                          int prepare_backward,
                                          float fastemit_lambda);

rnntStatus_t compute_rnnt_loss_packed_bwd(const void* activations,
                                          void* gradients,
                                          const void* grad_scale_device,
                                          const long long* const row_offsets,
                                          long long total_rows,
                                          int alphabet_size,
                                          int minibatch,
                                          void* workspace,
                                          rnntOptions options,
                                          int dtype_code);

/* Additive joint ("add network", the reference's add_network branch: README.md:4,
 * docs/rnnt_notes.tex:56-59,147-153, pytorch_binding/test/test_time.py:51-77).  The joint logits
 * are h(k,t,u) = trans_acts[b,t,k] + pred_acts[b,u,k]; the (B,T,U,V) tensor is never formed.
 * trans_acts (B,maxT,V) and pred_acts (B,maxU,V) are dense fp32 DEVICE tensors (bf16 / fp16: the _dt entries below); trans_grads /
 * pred_grads receive dL/d(trans_acts) = sum_u dL/dh and dL/d(pred_acts) = sum_t dL/dh (both NULL:
 * score only).  Same conventions as compute_rnnt_loss_async: `costs_device` is a DEVICE array of
 * `minibatch` floats, the call only enqueues on options.stream; size the workspace with
 * get_workspace_size_add(maxT, maxU, minibatch, &bytes) -- the lattice workspace of
 * get_workspace_size plus the row maxima and the dense weight planes only this path uses (a
 * workspace of that size also serves every other GPU entry point). */
rnntStatus_t get_workspace_size_add(int maxT, int maxU,
                                    int minibatch,
                                    size_t* size_bytes);

rnntStatus_t compute_rnnt_loss_add(const float* const trans_acts,
                                   const float* const pred_acts,
                                   float* trans_grads,
                                   float* pred_grads,
                                   const int* const flat_labels,
                                   const int* const label_lengths,
                                   const int* const input_lengths,
                                   int alphabet_size,
                                   int minibatch,
                                   float* costs_device,
                                   void* workspace,
                                   rnntOptions options);

/* Two-phase form of compute_rnnt_loss_add for autograd frameworks, as compute_rnnt_loss_fwd / _bwd
 * above: the forward call writes the costs and (with prepare_backward != 0) leaves the weight matrix
 * and the correction table in the workspace; the backward call, with the SAME activations, labels,
 * lengths, workspace and options, enqueues only the two gradient GEMMs and the corrections, with
 * sample b's gradients multiplied by grad_scale_device[b] (NULL = 1) -- no pass over d(trans_acts) /
 * d(pred_acts) is left for the framework.  Enqueue only, no synchronisation. */
rnntStatus_t compute_rnnt_loss_add_fwd(const float* const trans_acts,
                                       const float* const pred_acts,
                                       const int* const flat_labels,
                                       const int* const label_lengths,
                                       const int* const input_lengths,
                                       int alphabet_size,
                                       int minibatch,
                                       float* costs_device,
                                       void* workspace,
                                       rnntOptions options,
                                       int prepare_backward);

/* compute_rnnt_loss_add_fwd with a FastEmit lambda (see compute_rnnt_loss_fastemit); pairs with the
 * unchanged compute_rnnt_loss_add_bwd. */
rnntStatus_t compute_rnnt_loss_add_fwd_fastemit(const float* const trans_acts,
                                                const float* const pred_acts,
                                                const int* const flat_labels,
                                                const int* const label_lengths,
                                                const int* const input_lengths,
                                                int alphabet_size,
                                                int minibatch,
                                                float* costs_device,
                                                void* workspace,
                                                rnntOptions options,
                                                int prepare_backward,
                                                float fastemit_lambda);

rnntStatus_t compute_rnnt_loss_add_bwd(const float* const trans_acts,
                                       const float* const pred_acts,
                                       float* trans_grads,
                                       float* pred_grads,
                                       const float* grad_scale_device,
                                       const int* const flat_labels,
                                       const int* const label_lengths,
                                       const int* const input_lengths,
                                       int alphabet_size,
                                       int minibatch,
                                       void* workspace,
                                       rnntOptions options);

/* The two-phase additive-joint entries with the STORAGE type of the activations as an argument:
 * dtype_code 0 = fp32 (identical to compute_rnnt_loss_add_fwd_fastemit / compute_rnnt_loss_add_bwd),
 * 2 = bf16, 3 = fp16 (raw 16-bit storage of trans_acts, pred_acts and both gradients; every kernel
 * computes in fp32 -- element-wise work and accumulation in fp32; with 16-bit storage, rows of whole 16-byte packets and 512
 * symbols or more the three contractions run on the bf16 matrix cores with every operand split into a bf16 hi + lo pair,
 * ~2^-17 relative per product, otherwise on the fp32 ones --; costs and grad_scale stay float).  Same workspace query, same
 * conventions.  (The rare "far" cells -- rows of trans_acts and pred_acts peaking more than 40 nats apart, not reached by ordinary
 * logits -- are summed in fp32 per row segment before they are added to a stored 16-bit gradient: at most ceil(maxU / 64) resp.
 * ceil(maxT / 64) rounded additions per element.)  */
rnntStatus_t compute_rnnt_loss_add_fwd_dt(const void* trans_acts,
                                          const void* pred_acts,
                                          const int* const flat_labels,
                                          const int* const label_lengths,
                                          const int* const input_lengths,
                                          int alphabet_size,
                                          int minibatch,
                                          float* costs_device,
                                          void* workspace,
                                          rnntOptions options,
                                          int dtype_code,
                                          int prepare_backward,
                                          float fastemit_lambda);

rnntStatus_t compute_rnnt_loss_add_bwd_dt(const void* trans_acts,
                                          const void* pred_acts,
                                          void* trans_grads,
                                          void* pred_grads,
                                          const float* grad_scale_device,
                                          const int* const flat_labels,
                                          const int* const label_lengths,
                                          const int* const input_lengths,
                                          int alphabet_size,
                                          int minibatch,
                                          void* workspace,
                                          rnntOptions options,
                                          int dtype_code);

/* Two-half schedule for long lattices (extension).  The alpha/beta recursion is a dependent chain: on long utterances (T + U in the
 * thousands) it takes hundreds of microseconds during which a few small blocks hold the device and HBM idles, between the two
 * streaming stages.  Given a SECOND stream, the one-call gradient-computing entry points above (compute_rnnt_loss and its fp64 /
 * bf16 / fp16 / async / fastemit / sharded forms; not the two-phase pair, not score-only calls, not the packed ones) split the batch into two
 * halves of samples and run the lattice kernel of one half on that stream while options.stream streams the other half's
 * statistics / gradient kernels (fork and join through events; still no synchronisation, still capturable into a HIP graph, still
 * nothing allocated: the stream is the caller's).  Results are bit-identical to the one-stream schedule.  Measured on N=64,T=1500,U=301,A=50
 * (lattice 0.27 ms of a 3.6 ms call): 3.4-3.6 -> 3.3-3.5 ms through the one-call entries -- the lattice disappears from the critical
 * path, the streaming kernels slow by 0.1 ms beside it -- but the forward half of a two-phase pair gets SLOWER (its second lattice
 * has only the first half's coefficient kernel to hide behind): opt-in, for the one-call entries.  Applies to calls from the
 * thread that set it, for lattices of 768 anti-diagonals and more and batches of two samples and more; NULL (the default)
 * switches it off.  The stream must belong to the device of the call and must outlive the calls that use it.  A thread that moves to
 * another GPU hands over a stream of THAT device: the four fork / join events are the calling thread's, made on the device of the call,
 * destroyed and made again when the stream is replaced or the device changes (NULL releases them); if they cannot be made or recorded, the
 * call runs the one-stream schedule.  The split falls on a 16-byte boundary of the tensors, and kernel forms that depend on the batch
 * size are chosen for the WHOLE batch, so the bits are the one-stream schedule's for any N (tests/test_gpu_parity_large.py). */
void rnnt_set_aux_stream(CUstream stream);

/* Revision of the extension entry points below (the reference's get_warprnnt_version() stays 1).
 *   3: the additive-joint entries (compute_rnnt_loss_add*) must be given a workspace sized by
 *      get_workspace_size_add(); get_workspace_size() covers the materialised entries only.
 *      Host staging of pageable costs became opt-in (rnnt_host_staging).
 *   4: compute_rnnt_loss_lattice_dump exists.
 *   5: rnnt_sharded_prepare / rnnt_sharded_release exist and compute_rnnt_loss_sharded REQUIRES a prepared communicator;
 *      gradients == activations (in place) is supported by every materialised entry. */
int get_warprnnt_extension_version(void);

/* Memory the library itself allocates.  By DEFAULT: none, on the host or on the device, in any entry point -- as the
 * reference (README.md:36-37); everything lives in the caller's workspace, and pageable host `costs` are copied behind
 * the last kernel exactly as the reference does (include/detail/gpu_rnnt.h:208-213).  `costs` in PINNED host memory are
 * written by the kernel directly (no copy, no allocation).
 * OPT-IN: rnnt_host_staging(1) (or the environment variable WARPRNNT_HOST_STAGING=1 read at the first GPU call) lets
 * compute_rnnt_loss / _fp64 / _bf16 / _fp16 route PAGEABLE costs through a pinned staging buffer of the calling thread
 * (about 9 us of a 50 us call): host memory only, 4 KB grown by doubling to at most 1 MB per calling thread; larger
 * batches use the copy.  rnnt_host_staging(0) turns it off again, any other value only queries; all return the
 * previous setting (0 | 1).  rnnt_host_staging_bytes() = pinned bytes currently held;
 * rnnt_host_staging_release() frees every buffer that is not inside a call at that moment and returns the bytes freed. */
int rnnt_host_staging(int mode);
long long rnnt_host_staging_bytes(void);
long long rnnt_host_staging_release(void);

/* Stage timing and stage markers.  rnnt_profile_enable(on): bit 0 = stage timers, bit 1 = roctx RANGES around the
 * enqueue of each stage ("warprnnt:row_stats", ":lattice", ":coefficients", ":gradient"; the additive-joint path:
 * ":joint_partition", ":lattice", ":coefficients", ":joint_gradients") for `rocprofv3 --marker-trace --kernel-trace` --
 * the counterpart of the reference's DEBUG_TIME stage timers (include/detail/gpu_rnnt.h:112-122).  The marker library
 * is looked up at run time (no link dependency); WARPRNNT_ROCTX=1 in the environment switches the ranges on without
 * code.  rnnt_profile_enable(1) makes every following
 * GPU call record HIP events around its kernels on options.stream (no extra
 * synchronisation); rnnt_profile_read() fills `ms` with the accumulated
 * milliseconds per stage since the last rnnt_profile_reset() and returns the
 * number of calls accumulated:
 *   ms[0] row-stats kernel (log-softmax denominators + blank/label gather)
 *   ms[1] lattice kernel   (alpha/beta recursion)
 *   ms[2] coefficient kernel (per-cell gradient coefficients)
 *   ms[3] gradient kernel  (dense gradient write-back)
 *   ms[4] whole enqueue, first kernel start -> last kernel end
 * compute_rnnt_loss_async does not synchronise, so its events are read by an explicit
 * rnnt_profile_collect() once the caller has synchronised the stream.  A two-phase step
 * (compute_rnnt_loss_fwd ... compute_rnnt_loss_bwd, one collect after both) counts as ONE call:
 * ms[0..2] come from the forward call, ms[3] from the backward call, and what the caller
 * enqueued between the two is inside ms[4] only.
 * Process-global: ONE set of timers and events for the library.  Calls from several threads are safe (a profiled call
 * holds the library's profiling mutex while it records, so profiled calls are serialised), but the accumulated times only
 * mean something with one calling thread at a time; meant for bench.py. */
void rnnt_profile_enable(int on);
void rnnt_profile_collect(void);   /* after synchronising a compute_rnnt_loss_async call: add its times */
void rnnt_profile_reset(void);
int rnnt_profile_read(double* ms, int n);

#ifdef __cplusplus
}
#endif
