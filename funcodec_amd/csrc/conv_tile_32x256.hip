// Instantiations of the implicit-GEMM conv kernel (conv_kernel.h) for the 32 x 256 tile: every prologue mode and staging variant.
#include "conv_kernel.h"

namespace fc {
template hipError_t launch_conv_tile<32, 256, 1, 4>(const ConvLaunch&, const ConvArgs&, dim3, size_t, hipStream_t);
}  // namespace fc
