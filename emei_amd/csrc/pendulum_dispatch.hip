// pendulum_dispatch.hip — routes a PendLaunch to the translation unit that owns its (env, precision).
#include "launch.h"

namespace emei {

#define EMEI_DECL(name) int name(const PendLaunch&);
EMEI_DECL(pend_tu_cp0_f64) EMEI_DECL(pend_tu_cp0_f32) EMEI_DECL(pend_tu_cp1_f64) EMEI_DECL(pend_tu_cp1_f32)
EMEI_DECL(pend_tu_cr0_f64) EMEI_DECL(pend_tu_cr0_f32) EMEI_DECL(pend_tu_cr1_f64) EMEI_DECL(pend_tu_cr1_f32)
EMEI_DECL(pend_tu_ip0_f64) EMEI_DECL(pend_tu_ip0_f32) EMEI_DECL(pend_tu_ip1_f64) EMEI_DECL(pend_tu_ip1_f32)
EMEI_DECL(pend_tu_ip2_f64) EMEI_DECL(pend_tu_ip2_f32) EMEI_DECL(pend_tu_ip3_f64) EMEI_DECL(pend_tu_ip3_f32)
#undef EMEI_DECL

int pend_launch(const PendLaunch& L) {
    const bool f32 = L.precision == EMEI_PRECISION_F32;
    if (L.ode_method == EMEI_ODE_RK4) {  // ODE_approximation(method="rk4"), base_control.py:165-170: CartPoleRK4's own translation units
        if (L.env_id == EMEI_CARTPOLE_SWINGUP) return f32 ? pend_tu_cr0_f32(L) : pend_tu_cr0_f64(L);
        if (L.env_id == EMEI_CARTPOLE_BALANCING) return f32 ? pend_tu_cr1_f32(L) : pend_tu_cr1_f64(L);
    }
    switch (L.env_id) {
        case EMEI_CARTPOLE_SWINGUP: return f32 ? pend_tu_cp0_f32(L) : pend_tu_cp0_f64(L);
        case EMEI_CARTPOLE_BALANCING: return f32 ? pend_tu_cp1_f32(L) : pend_tu_cp1_f64(L);
        case EMEI_IP_REBOUND_BALANCING: return f32 ? pend_tu_ip0_f32(L) : pend_tu_ip0_f64(L);
        case EMEI_IP_BOUNDARY_BALANCING: return f32 ? pend_tu_ip1_f32(L) : pend_tu_ip1_f64(L);
        case EMEI_IP_REBOUND_SWINGUP: return f32 ? pend_tu_ip2_f32(L) : pend_tu_ip2_f64(L);
        case EMEI_IP_BOUNDARY_SWINGUP: return f32 ? pend_tu_ip3_f32(L) : pend_tu_ip3_f64(L);
        default: return EMEI_ERR_UNSUPPORTED;
    }
}

}  // namespace emei
