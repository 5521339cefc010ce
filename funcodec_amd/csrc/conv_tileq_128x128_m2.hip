// Quad-layout instantiations of the 128 x 128 tile, prologue mode 2 (affine + ELU): see conv_kernel.h launch_conv_mq.
#include "conv_kernel.h"

namespace fc {
FC_CONVQ_HERE(128, 128, 2, 2, 2)
}  // namespace fc
