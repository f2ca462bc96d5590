// body_tu.hip — one translation unit per (body, precision), built by the Makefile with
//   -DEMEI_BODY_FAM_{ch,chs,hp,hps,dp,ip} -DEMEI_BODY_VARIANT=<K> -DEMEI_TU_REAL={double,float} -DEMEI_TU_NAME=<symbol>
#include "body_kernels.h"
#if defined(EMEI_BODY_FAM_ch)
#include "cheetah_model.h"
#define EMEI_BODY_TYPE CheetahBody<EMEI_TU_REAL, EMEI_SOLVER_NEWTON>
#elif defined(EMEI_BODY_FAM_chs)
#include "cheetah_model.h"
#define EMEI_BODY_TYPE CheetahBody<EMEI_TU_REAL, EMEI_SOLVER_SWEEP1>
#elif defined(EMEI_BODY_FAM_hps)
#include "hopper_model.h"
#define EMEI_BODY_TYPE HopperBody<EMEI_TU_REAL, EMEI_SOLVER_SWEEP1>
#elif defined(EMEI_BODY_FAM_dp)
#include "dpend_model.h"
#define EMEI_BODY_TYPE DPendBody<EMEI_BODY_VARIANT, EMEI_TU_REAL>
#elif defined(EMEI_BODY_FAM_ip)
#include "ipend_model.h"
#define EMEI_BODY_TYPE InvPendBody<EMEI_BODY_VARIANT, EMEI_TU_REAL>
#elif defined(EMEI_BODY_FAM_hp)
#include "hopper_model.h"
#define EMEI_BODY_TYPE HopperBody<EMEI_TU_REAL, EMEI_SOLVER_NEWTON>
#else
#error "body_tu.hip: no EMEI_BODY_FAM_* given"
#endif

namespace emei {
int EMEI_TU_NAME(const BodyLaunch& L) { return launch_body<EMEI_BODY_TYPE>(L); }
}  // namespace emei

#ifdef EMEI_CYCLE_PROFILE
// variant builds only (tools/cycle_profile.py): copy out and clear this translation unit's per-region cycle sums
#define EMEI_CATP2(a, b) a##b
#define EMEI_CATP(a, b) EMEI_CATP2(a, b)
extern "C" __attribute__((visibility("default"))) int EMEI_CATP(emei_cycle_stats_, EMEI_TU_NAME)(unsigned long long* out) {
    unsigned long long zero[32] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(emei::g_cycle_stats), sizeof(zero)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(emei::g_cycle_stats), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

#if defined(EMEI_NEWTON_STATS) || defined(EMEI_CLOCK_PROBE)
// variant builds only (tools/newton_stats.py): copy out and clear this translation unit's solver counters
#define EMEI_CAT2(a, b) a##b
#define EMEI_CAT(a, b) EMEI_CAT2(a, b)
extern "C" __attribute__((visibility("default"))) int EMEI_CAT(emei_debug_stats_, EMEI_TU_NAME)(unsigned long long* out) {
    unsigned long long zero[32] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(emei::g_debug_stats), sizeof(zero)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(emei::g_debug_stats), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif
