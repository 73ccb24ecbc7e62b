// rnnt_joint_bf16.hip -- the additive-joint path for bf16 storage: run_gpu_joint<BF16> and its kernels, a code object of its own
// (rnnt_joint_impl.h says why).
#define RNNT_JOINT_INSTANTIATE_BF16 1
#include "rnnt_joint_impl.h"

namespace rnnt {
template rnntStatus_t run_gpu_joint<BF16>(const uint16_t*, const uint16_t*, uint16_t*, uint16_t*, const int*, const int*, const int*, int, int, float*, const float*, void*, const rnntOptions&, int, bool, float);
}  // namespace rnnt
