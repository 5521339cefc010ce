// Instantiations of the implicit-GEMM conv kernel (conv_kernel.h) for the 32 x 128 tile in the quad-k operand layout (round 5).
#include "conv_kernel.h"

namespace fc {
template hipError_t launch_conv_tile_q<32, 128, 1, 4>(const ConvLaunch&, const ConvArgs&, dim3, size_t, hipStream_t);
}  // namespace fc
