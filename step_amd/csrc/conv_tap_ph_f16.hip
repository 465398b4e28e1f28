// step_amd/csrc/conv_tap_ph_f16.hip -- the two-phase (anti-phase wave groups) instantiations of conv_tap_kernel for f16 storage
#include "conv_tap_kernel.h"
namespace step {
template <> int conv_tap_ph_launch<f16_t>(const ConvPlan& pl, const ConvParams& p, int kd, dim3 grid, step_stream_t stream) {
    return conv_tap_ph_launch_impl<f16_t>(pl, p, kd, grid, stream);
}
template <> int conv_tap_group_launch<f16_t>(int twl, int NB, const ConvGroupParams& g, dim3 grid, step_stream_t stream) {
    return conv_tap_group_launch_impl<f16_t>(twl, NB, g, grid, stream);
}
}  // namespace step
