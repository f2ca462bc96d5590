// pendulum_tu.hip — one translation unit per (env, precision) so that the kernel instantiations
// compile in parallel.  Built 12 times by the Makefile with
//   -DEMEI_TU_FAMILY=CartPole|InvPend -DEMEI_TU_VARIANT=<int> -DEMEI_TU_REAL=double|float -DEMEI_TU_NAME=<symbol>
#include "pendulum_kernels.h"

namespace emei {
int EMEI_TU_NAME(const PendLaunch& L) { return launch_env<EMEI_TU_FAMILY<EMEI_TU_VARIANT, EMEI_TU_REAL>>(L); }
}  // namespace emei
