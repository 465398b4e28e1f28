// step_amd/csrc/conv_tap_f32.hip -- conv_tap_kernel instantiations for float storage (see conv_tap_kernel.h)
#include "conv_tap_kernel.h"

namespace step {
template int conv_tap_launch<float>(const ConvPlan&, const ConvParams&, int, dim3, step_stream_t);
}  // namespace step
