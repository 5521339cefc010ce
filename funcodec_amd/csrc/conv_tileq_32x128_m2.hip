// Quad-layout instantiations of the 32 x 128 tile, prologue mode 2 (affine + ELU): see conv_kernel.h launch_conv_mq.
#include "conv_kernel.h"

namespace fc {
FC_CONVQ_HERE(32, 128, 1, 4, 2)
}  // namespace fc
