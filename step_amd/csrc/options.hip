// step_amd/csrc/options.hip -- the planner options of include/step_amd.h (step_set_option / step_get_option).
// The only process-wide state of the library: a table of relaxed atomics.  No environment variable is read anywhere.
#include <atomic>
#include <limits.h>

#include "options.h"

namespace step {

namespace {
struct OptSpec { const char* name; int def, lo, hi; };
constexpr OptSpec SPEC[STEP_OPT_COUNT_] = {
    {"conv_impl", -1, -1, 5},
    {"conv_nb", 0, 0, 3},
    {"conv_waves", 0, 0, 8},
    {"conv_phased", 2, 0, 2},
    {"conv_gen", 93, 0, 100},
    {"conv_gmode", 1, 0, 1},
    {"conv_pws", -1, -1, 1},
    {"conv_splitk", 1, 0, 1},
    {"conv_tail", 1, 0, 1},
    {"conv_slots", 0, 0, 1 << 20},
    {"pool_direct", 0, 0, 1},
    {"wgrad_minpix", 0, 0, 1 << 24},
    {"wgrad16_lds", 1, 0, 2},
    {"conv_group_pw", 1 << 20, 0, 1 << 20},
    {"clip_vec", 1, 0, 1},
    {"conv_nb_rule", 0, 0, 1},
    {"throughput", 0, 0, 1},
    {"conv_persist", 1, 0, 1},
    {"conv_pws_waves", 0, 0, 16},
};
std::atomic<int> g_delta[STEP_OPT_COUNT_];      // value - default: zero-initialised static storage IS the default table
}  // namespace

int opt(int id) { return SPEC[id].def + g_delta[id].load(std::memory_order_relaxed); }

}  // namespace step

using namespace step;

extern "C" {

int step_set_option(int option, int value) {
    if (option < 0 || option >= STEP_OPT_COUNT_) return STEP_E_SHAPE;
    if (value < SPEC[option].lo || value > SPEC[option].hi) return STEP_E_SHAPE;
    if (option == STEP_OPT_CONV_WAVES && value != 0 && value != 4 && value != 8) return STEP_E_SHAPE;
    if (option == STEP_OPT_CONV_PWS_WAVES && value != 0 && value != 8 && value != 16) return STEP_E_SHAPE;
    if (option == STEP_OPT_CONV_IMPL && (value == 3 || value == 4)) return STEP_E_SHAPE;
    g_delta[option].store(value - SPEC[option].def, std::memory_order_relaxed);
    return STEP_OK;
}

int step_get_option(int option, int* value) {
    if (option < 0 || option >= STEP_OPT_COUNT_) return STEP_E_SHAPE;
    if (!value) return STEP_E_NULL;
    *value = opt(option);
    return STEP_OK;
}

void step_reset_options(void) {
    for (int i = 0; i < STEP_OPT_COUNT_; ++i) g_delta[i].store(0, std::memory_order_relaxed);
}

const char* step_option_name(int option) {
    return (option < 0 || option >= STEP_OPT_COUNT_) ? nullptr : SPEC[option].name;
}

}  // extern "C"
