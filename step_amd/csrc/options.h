// step_amd/csrc/options.h -- host-side accessor of the planner options (include/step_amd.h, step_set_option).
#pragma once
#include "../../include/step_amd.h"

namespace step {
int opt(int id);      // current value of STEP_OPT_<id> (relaxed atomic load)
}
