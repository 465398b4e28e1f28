// step_amd/csrc/conv_tap_f16.hip -- conv_tap_kernel instantiations for f16_t storage (see conv_tap_kernel.h)
#include "conv_tap_kernel.h"

namespace step {
template int conv_tap_launch<f16_t>(const ConvPlan&, const ConvParams&, int, dim3, step_stream_t);
}  // namespace step
