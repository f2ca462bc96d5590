// body_tu.hip — one translation unit per (body, precision), built by the Makefile with
//   -DEMEI_BODY_TYPE='CheetahBody<double>' | 'DPendBody<2, float>' ...  -DEMEI_TU_NAME=<symbol>
#include "body_kernels.h"
#include "cheetah_model.h"
#include "dpend_model.h"

namespace emei {
int EMEI_TU_NAME(const BodyLaunch& L) { return launch_body<EMEI_BODY_TYPE>(L); }
}  // namespace emei
