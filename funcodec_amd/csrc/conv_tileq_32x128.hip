// Instantiations of the implicit-GEMM conv kernel (conv_kernel.h) for the 32 x 128 tile in the quad-k operand layout (round 5).
#include "conv_kernel.h"

namespace fc {
// prologue modes 0, 1 (conv_tileq_32x128_m01.hip) and 2 (_m2.hip) are instantiated in their own units; 5, 3, 4 here
FC_CONVQ_ELSEWHERE(32, 128, 1, 4, 0)
FC_CONVQ_ELSEWHERE(32, 128, 1, 4, 1)
FC_CONVQ_ELSEWHERE(32, 128, 1, 4, 2)
template hipError_t launch_conv_tile_q<32, 128, 1, 4>(const ConvLaunch&, const ConvArgs&, dim3, size_t, hipStream_t);
}  // namespace fc
