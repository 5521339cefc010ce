// Quad-layout instantiations of the 128 x 128 tile, prologue modes 0 (plain) and 1 (affine): see conv_kernel.h launch_conv_mq.
#include "conv_kernel.h"

namespace fc {
FC_CONVQ_HERE(128, 128, 2, 2, 0)
FC_CONVQ_HERE(128, 128, 2, 2, 1)
}  // namespace fc
