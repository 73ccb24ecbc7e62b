// rnnt_gpu_h16.hip -- the materialised path for 16-bit activations (bf16, fp16): run_gpu<BF16>, run_gpu<F16> and their kernels,
// a code object of its own (rnnt_gpu_impl.h says why).
#define RNNT_GPU_INSTANTIATE_H16 1
#include "rnnt_gpu_impl.h"

namespace rnnt {
template rnntStatus_t run_gpu<BF16>(const uint16_t*, uint16_t*, const int*, const int*, const int*, int, int, float*, float*, const float*, void*,
                                    const rnntOptions&, int, int, float, const long long*, long long);
template rnntStatus_t run_gpu<F16>(const uint16_t*, uint16_t*, const int*, const int*, const int*, int, int, float*, float*, const float*, void*,
                                   const rnntOptions&, int, int, float, const long long*, long long);
}  // namespace rnnt
