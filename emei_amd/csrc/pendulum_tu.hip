// pendulum_tu.hip — one translation unit per (env, precision) so that the kernel instantiations
// compile in parallel.  Built 12 times by the Makefile with
//   -DEMEI_TU_FAMILY=CartPole|InvPend -DEMEI_TU_VARIANT=<int> -DEMEI_TU_REAL=double|float -DEMEI_TU_NAME=<symbol>
#include "pendulum_kernels.h"

namespace emei {
int EMEI_TU_NAME(const PendLaunch& L) { return launch_env<EMEI_TU_FAMILY<EMEI_TU_VARIANT, EMEI_TU_REAL>>(L); }
}  // namespace emei

#if defined(EMEI_NEWTON_STATS) || defined(EMEI_CLOCK_PROBE)
// variant builds only (tools/pend_stats.py, tools/clock_probe.py): copy out and clear this translation unit's event counters (emei_device.h)
#define EMEI_CAT2(a, b) a##b
#define EMEI_CAT(a, b) EMEI_CAT2(a, b)
extern "C" __attribute__((visibility("default"))) int EMEI_CAT(emei_debug_stats_, EMEI_TU_NAME)(unsigned long long* out) {
    unsigned long long zero[32] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(emei::g_debug_stats), sizeof(zero)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(emei::g_debug_stats), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

