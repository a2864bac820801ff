// amt_kernel_march.hip -- AMT_VARIANT_MARCH (placeholder until the kernel lands).
#include <hip/hip_runtime.h>
#include "amt_params.h"

template <typename T> bool amt_march_supported(const AmtParams<T> &) { return false; }
template <typename T> hipError_t amt_launch_march(hipStream_t, const AmtParams<T> &) { return hipErrorNotSupported; }

template bool amt_march_supported<float>(const AmtParams<float> &);
template bool amt_march_supported<double>(const AmtParams<double> &);
template hipError_t amt_launch_march<float>(hipStream_t, const AmtParams<float> &);
template hipError_t amt_launch_march<double>(hipStream_t, const AmtParams<double> &);
