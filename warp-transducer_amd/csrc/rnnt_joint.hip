// rnnt_joint.hip -- host driver and C entry points of the additive-joint ("add network") path, SURVEY.md 8f rank 1
// (kernels: rnnt_joint_kernels.h; lattice and coefficient stages: the launchers of rnnt_host.h).  Its own translation
// unit, hence its own code object: see rnnt_host.h; the 16-bit storage types have theirs (rnnt_joint_impl.h).
#define RNNT_JOINT_INSTANTIATE_F32 1
#include "rnnt_joint_impl.h"

namespace rnnt {
template rnntStatus_t run_gpu_joint<F32>(const float*, const float*, float*, float*, const int*, const int*, const int*, int, int, float*, const float*, void*, const rnntOptions&, int, bool, float);
}  // namespace rnnt

using namespace rnnt;

#pragma GCC visibility push(default)
extern "C" {

rnntStatus_t compute_rnnt_loss_add(const float* const trans_acts, const float* const pred_acts,
                                   float* trans_grads, float* pred_grads, const int* const flat_labels,
                                   const int* const label_lengths, const int* const input_lengths,
                                   int alphabet_size, int minibatch, float* costs_device, void* workspace,
                                   rnntOptions options) {
    if (bad_args(trans_acts, flat_labels, label_lengths, input_lengths, costs_device, workspace, alphabet_size,
                 minibatch, options) || pred_acts == nullptr || loc_of(options) != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    if ((trans_grads == nullptr) != (pred_grads == nullptr)) return RNNT_STATUS_INVALID_VALUE;
    const bool training = trans_grads != nullptr;
    return run_gpu_joint<F32>(trans_acts, pred_acts, trans_grads, pred_grads, flat_labels, label_lengths,
                         input_lengths, alphabet_size, minibatch, costs_device, nullptr, workspace, options,
                         training ? 3 : 1, training);
}

rnntStatus_t compute_rnnt_loss_add_fwd(const float* const trans_acts, const float* const pred_acts,
                                       const int* const flat_labels, const int* const label_lengths,
                                       const int* const input_lengths, int alphabet_size, int minibatch,
                                       float* costs_device, void* workspace, rnntOptions options,
                                       int prepare_backward) {
    if (bad_args(trans_acts, flat_labels, label_lengths, input_lengths, costs_device, workspace, alphabet_size,
                 minibatch, options) || pred_acts == nullptr || loc_of(options) != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    return run_gpu_joint<F32>(trans_acts, pred_acts, nullptr, nullptr, flat_labels, label_lengths, input_lengths,
                         alphabet_size, minibatch, costs_device, nullptr, workspace, options, 1,
                         prepare_backward != 0);
}

rnntStatus_t compute_rnnt_loss_add_fwd_fastemit(const float* const trans_acts, const float* const pred_acts,
                                                const int* const flat_labels, const int* const label_lengths,
                                                const int* const input_lengths, int alphabet_size, int minibatch,
                                                float* costs_device, void* workspace, rnntOptions options,
                                                int prepare_backward, float fastemit_lambda) {
    if (bad_args(trans_acts, flat_labels, label_lengths, input_lengths, costs_device, workspace, alphabet_size,
                 minibatch, options) || pred_acts == nullptr || loc_of(options) != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    return run_gpu_joint<F32>(trans_acts, pred_acts, nullptr, nullptr, flat_labels, label_lengths, input_lengths,
                         alphabet_size, minibatch, costs_device, nullptr, workspace, options, 1,
                         prepare_backward != 0, fastemit_lambda);
}

rnntStatus_t compute_rnnt_loss_add_bwd(const float* const trans_acts, const float* const pred_acts,
                                       float* trans_grads, float* pred_grads, const float* grad_scale_device,
                                       const int* const flat_labels, const int* const label_lengths,
                                       const int* const input_lengths, int alphabet_size, int minibatch,
                                       void* workspace, rnntOptions options) {
    if (trans_acts == nullptr || pred_acts == nullptr || trans_grads == nullptr || pred_grads == nullptr ||
        flat_labels == nullptr || label_lengths == nullptr || input_lengths == nullptr || workspace == nullptr ||
        alphabet_size <= 0 || minibatch <= 0 || options.maxT <= 0 || options.maxU <= 0 || loc_of(options) != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    return run_gpu_joint<F32>(trans_acts, pred_acts, trans_grads, pred_grads, flat_labels, label_lengths,
                         input_lengths, alphabet_size, minibatch, nullptr, grad_scale_device, workspace, options, 2,
                         true);
}

// Additive joint with the activations' storage type as an argument (0 fp32, 2 bf16, 3 fp16; fp64 is not offered).
rnntStatus_t compute_rnnt_loss_add_fwd_dt(const void* trans_acts, const void* pred_acts, const int* const flat_labels,
                                          const int* const label_lengths, const int* const input_lengths,
                                          int alphabet_size, int minibatch, float* costs_device, void* workspace,
                                          rnntOptions options, int dtype_code, int prepare_backward,
                                          float fastemit_lambda) {
    if (bad_args(trans_acts, flat_labels, label_lengths, input_lengths, costs_device, workspace, alphabet_size,
                 minibatch, options) || pred_acts == nullptr || loc_of(options) != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    const bool pb = prepare_backward != 0;
    switch (dtype_code) {
        case 0: return run_gpu_joint<F32>(static_cast<const float*>(trans_acts), static_cast<const float*>(pred_acts), nullptr,
                                          nullptr, flat_labels, label_lengths, input_lengths, alphabet_size, minibatch,
                                          costs_device, nullptr, workspace, options, 1, pb, fastemit_lambda);
        case 2: return run_gpu_joint<BF16>(static_cast<const uint16_t*>(trans_acts), static_cast<const uint16_t*>(pred_acts),
                                           nullptr, nullptr, flat_labels, label_lengths, input_lengths, alphabet_size, minibatch,
                                           costs_device, nullptr, workspace, options, 1, pb, fastemit_lambda);
        case 3: return run_gpu_joint<F16>(static_cast<const uint16_t*>(trans_acts), static_cast<const uint16_t*>(pred_acts),
                                          nullptr, nullptr, flat_labels, label_lengths, input_lengths, alphabet_size, minibatch,
                                          costs_device, nullptr, workspace, options, 1, pb, fastemit_lambda);
        default: return RNNT_STATUS_INVALID_VALUE;
    }
}

rnntStatus_t compute_rnnt_loss_add_bwd_dt(const void* trans_acts, const void* pred_acts, void* trans_grads,
                                          void* pred_grads, const float* grad_scale_device,
                                          const int* const flat_labels, const int* const label_lengths,
                                          const int* const input_lengths, int alphabet_size, int minibatch,
                                          void* workspace, rnntOptions options, int dtype_code) {
    if (trans_acts == nullptr || pred_acts == nullptr || trans_grads == nullptr || pred_grads == nullptr ||
        flat_labels == nullptr || label_lengths == nullptr || input_lengths == nullptr || workspace == nullptr ||
        alphabet_size <= 0 || minibatch <= 0 || options.maxT <= 0 || options.maxU <= 0 || loc_of(options) != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    switch (dtype_code) {
        case 0: return run_gpu_joint<F32>(static_cast<const float*>(trans_acts), static_cast<const float*>(pred_acts),
                                          static_cast<float*>(trans_grads), static_cast<float*>(pred_grads), flat_labels,
                                          label_lengths, input_lengths, alphabet_size, minibatch, nullptr, grad_scale_device,
                                          workspace, options, 2, true);
        case 2: return run_gpu_joint<BF16>(static_cast<const uint16_t*>(trans_acts), static_cast<const uint16_t*>(pred_acts),
                                           static_cast<uint16_t*>(trans_grads), static_cast<uint16_t*>(pred_grads), flat_labels,
                                           label_lengths, input_lengths, alphabet_size, minibatch, nullptr, grad_scale_device,
                                           workspace, options, 2, true);
        case 3: return run_gpu_joint<F16>(static_cast<const uint16_t*>(trans_acts), static_cast<const uint16_t*>(pred_acts),
                                          static_cast<uint16_t*>(trans_grads), static_cast<uint16_t*>(pred_grads), flat_labels,
                                          label_lengths, input_lengths, alphabet_size, minibatch, nullptr, grad_scale_device,
                                          workspace, options, 2, true);
        default: return RNNT_STATUS_INVALID_VALUE;
    }
}

}  // extern "C"
#pragma GCC visibility pop
