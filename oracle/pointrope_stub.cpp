// TEST INFRASTRUCTURE (oracle).  libs/pointrope/pointrope.cpp:11 forward-declares pointrope_cuda (defined in kernels.cu,
// which needs nvcc); the CPU reference path never calls it.  This definition only satisfies the linker.
#include <torch/extension.h>

void pointrope_cuda(torch::Tensor, const torch::Tensor, const float, const float) {
  TORCH_CHECK(false, "oracle build of libs/pointrope: CPU reference only");
}
