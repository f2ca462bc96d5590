// body_dispatch.hip — routes a BodyLaunch to the translation unit that owns its (body, precision).
#include "body_kernels.h"

namespace emei {

#define EMEI_DECL(name) int name(const BodyLaunch&);
EMEI_DECL(body_tu_ch_f64) EMEI_DECL(body_tu_ch_f32) EMEI_DECL(body_tu_hp_f64) EMEI_DECL(body_tu_hp_f32)
EMEI_DECL(body_tu_chs_f64) EMEI_DECL(body_tu_chs_f32) EMEI_DECL(body_tu_hps_f64) EMEI_DECL(body_tu_hps_f32)
EMEI_DECL(body_tu_dp0_f64) EMEI_DECL(body_tu_dp0_f32) EMEI_DECL(body_tu_dp1_f64) EMEI_DECL(body_tu_dp1_f32)
EMEI_DECL(body_tu_dp2_f64) EMEI_DECL(body_tu_dp2_f32) EMEI_DECL(body_tu_dp3_f64) EMEI_DECL(body_tu_dp3_f32)
EMEI_DECL(body_tu_ip0_f64) EMEI_DECL(body_tu_ip0_f32) EMEI_DECL(body_tu_ip1_f64) EMEI_DECL(body_tu_ip1_f32)
EMEI_DECL(body_tu_ip2_f64) EMEI_DECL(body_tu_ip2_f32) EMEI_DECL(body_tu_ip3_f64) EMEI_DECL(body_tu_ip3_f32)
#undef EMEI_DECL

int body_launch(const BodyLaunch& L) {
    const bool f32 = L.precision == EMEI_PRECISION_F32;
    switch (L.env_id) {
        // ch / hp: MuJoCo's constraint formulation (Newton, the default); chs / hps: round 1's single sweep
        case EMEI_HALFCHEETAH_RUNNING:
            if (L.solver == EMEI_SOLVER_SWEEP1) return f32 ? body_tu_chs_f32(L) : body_tu_chs_f64(L);
            return f32 ? body_tu_ch_f32(L) : body_tu_ch_f64(L);
        case EMEI_HOPPER_RUNNING:
            if (L.solver == EMEI_SOLVER_SWEEP1) return f32 ? body_tu_hps_f32(L) : body_tu_hps_f64(L);
            return f32 ? body_tu_hp_f32(L) : body_tu_hp_f64(L);
        case EMEI_IDP_REBOUND_BALANCING: return f32 ? body_tu_dp0_f32(L) : body_tu_dp0_f64(L);
        case EMEI_IDP_BOUNDARY_BALANCING: return f32 ? body_tu_dp1_f32(L) : body_tu_dp1_f64(L);
        case EMEI_IDP_REBOUND_SWINGUP: return f32 ? body_tu_dp2_f32(L) : body_tu_dp2_f64(L);
        case EMEI_IDP_BOUNDARY_SWINGUP: return f32 ? body_tu_dp3_f32(L) : body_tu_dp3_f64(L);
        // InvertedPendulum with a non-default integrator or observation noise (ipend_model.h)
        case EMEI_IP_REBOUND_BALANCING: return f32 ? body_tu_ip0_f32(L) : body_tu_ip0_f64(L);
        case EMEI_IP_BOUNDARY_BALANCING: return f32 ? body_tu_ip1_f32(L) : body_tu_ip1_f64(L);
        case EMEI_IP_REBOUND_SWINGUP: return f32 ? body_tu_ip2_f32(L) : body_tu_ip2_f64(L);
        case EMEI_IP_BOUNDARY_SWINGUP: return f32 ? body_tu_ip3_f32(L) : body_tu_ip3_f64(L);
        default: return EMEI_ERR_UNSUPPORTED;
    }
}

}  // namespace emei
