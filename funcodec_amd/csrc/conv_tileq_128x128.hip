// Instantiations of the implicit-GEMM conv kernel (conv_kernel.h) for the 128 x 128 tile in the quad-k operand layout (round 5).
#include "conv_kernel.h"

namespace fc {
// prologue modes 0, 1 (conv_tileq_128x128_m01.hip) and 2 (_m2.hip) are instantiated in their own units; 5, 3, 4 here
FC_CONVQ_ELSEWHERE(128, 128, 2, 2, 0)
FC_CONVQ_ELSEWHERE(128, 128, 2, 2, 1)
FC_CONVQ_ELSEWHERE(128, 128, 2, 2, 2)
template hipError_t launch_conv_tile_q<128, 128, 2, 2>(const ConvLaunch&, const ConvArgs&, dim3, size_t, hipStream_t);

// copy of the timeline stamps of a profiling build (zeros otherwise): [role][item][slot]
hipError_t debug_timeline_128q(unsigned long long* dst) {
#ifdef FC_TIMELINE
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_timeline), sizeof(unsigned long long) * 2 * 24 * 8);
#else
    for (int i = 0; i < 2 * 24 * 8; ++i) dst[i] = 0ull;
    return hipSuccess;
#endif
}
}  // namespace fc
