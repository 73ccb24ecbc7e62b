// rnnt_gpu_f64.hip -- the materialised path for fp64 activations: run_gpu<F64> and its kernels, a code object of its own
// (rnnt_gpu_impl.h says why).
#define RNNT_GPU_INSTANTIATE_F64 1
#include "rnnt_gpu_impl.h"

namespace rnnt {
template rnntStatus_t run_gpu<F64>(const double*, double*, const int*, const int*, const int*, int, int, double*, double*, const double*, void*,
                                   const rnntOptions&, int, int, float, const long long*, long long);
}  // namespace rnnt
