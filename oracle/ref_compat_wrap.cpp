// Build recipe TU for oracle/_ref/_C.so : the reference's OWN C++ CPU operators
// (nms_cpu, ROIAlign_forward_cpu and the pybind module of vision.cpp), compiled
// from the sources WHERE THEY LIE under /root/reference -- nothing is copied.
//
// TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// The only adaptation is the macro below.  The reference calls
//   AT_DISPATCH_FLOATING_TYPES(tensor.type(), ...)      (cpu/ROIAlign_cpu.cpp:266,
//                                                        cpu/nms_cpu.cpp:95)
// i.e. it passes an at::DeprecatedTypeProperties, which PyTorch >= 2.x no longer
// accepts in that macro (it wants an at::ScalarType).  We re-state the macro so that
// it accepts either, exactly as PyTorch 1.x did; no reference line is modified and no
// header, library or generated file is substituted.
#include <torch/extension.h>

namespace step_oracle_compat {
inline at::ScalarType scalar_type_of(const at::DeprecatedTypeProperties& t) { return t.scalarType(); }
inline at::ScalarType scalar_type_of(at::ScalarType t) { return t; }
}  // namespace step_oracle_compat

#undef AT_DISPATCH_FLOATING_TYPES
#define AT_DISPATCH_FLOATING_TYPES(TYPE, NAME, ...) \
  AT_DISPATCH_SWITCH(step_oracle_compat::scalar_type_of(TYPE), NAME, AT_DISPATCH_CASE_FLOATING_TYPES(__VA_ARGS__))

// reference sources, included from where they lie (-I <reference>/external/maskrcnn_benchmark/csrc)
#include "cpu/ROIAlign_cpu.cpp"
#include "cpu/nms_cpu.cpp"
#include "vision.cpp"
