// rnnt_cpu.h -- host (RNNT_CPU) path of libwarprnnt: declarations.
// Selected only by an explicit options.loc == RNNT_CPU, exactly as in the reference C-ABI
// (src/rnnt_entrypoint.cpp:61-72); the GPU path never falls back to it.
#pragma once
#include <cstddef>
#include "../../include/rnnt.h"

namespace rnnt {
size_t cpu_workspace_bytes(int maxT, int maxU, int minibatch, size_t lat);
rnntStatus_t cpu_rnnt_f32(const float* log_probs, float* grads, const int* labels, const int* label_lengths,
                          const int* input_lengths, int A, int N, float* costs, void* workspace,
                          const rnntOptions& opt);
rnntStatus_t cpu_rnnt_f64(const double* log_probs, double* grads, const int* labels, const int* label_lengths,
                          const int* input_lengths, int A, int N, double* costs, void* workspace,
                          const rnntOptions& opt);
// packed layout (compute_rnnt_loss_packed with options.loc == RNNT_CPU): host arrays, `offsets` on the host
rnntStatus_t cpu_rnnt_packed(const void* log_probs, void* grads, const int* labels, const int* label_lengths,
                             const int* input_lengths, const long long* offsets, int A, int N, void* costs,
                             void* workspace, const rnntOptions& opt, bool fp64);
}  // namespace rnnt
