// ipend_model.h — InvertedPendulum (emei/envs/mujoco/inverted_pendulum.py on mujoco_env.py; model
// emei/envs/mujoco/assets/inverted_pendulum.xml) as a `Body` of body_kernels.h.
//
// The default configuration (integrator "euler", no observation noise) runs in the staged 4-state
// kernel of pendulum_kernels.h (InvPend<> in pendulum_envs.h), whose carried-trig trick relies on the
// forward-Euler position override.  Every other configuration of mujoco_env.py:70-79,98-104
// (semi_implicit_euler, rk4, obs_noise_params != 0) steps through this Body: the same closed-form
// 2-DoF model written as a forward-dynamics function.  VARIANT 0 ReboundBalancing (:52-79),
// 1 BoundaryBalancing (:82-111), 2 ReboundSwingUp (:114-146), 3 BoundarySwingUp (:149-183).
// Parity with libmujoco is unpinned; the oracle is oracle/emei_oracle.c:ip_accel.
#pragma once
#include <cmath>
#include <cstring>

#include "emei_device.h"
#include "pendulum_envs.h"

namespace emei {
namespace ipend {

struct Model {
    double M11, M22, mpr, mgr, gear, ctrl_lo, ctrl_hi, x_lo, x_hi;
    double phi_off, sin_off, cos_off;  // phi = theta + phi_off is the angle of the pole's com from +z
    double invw, K, B, dmin, dmax, width;
    double dt;
};

namespace host {
inline double capsule_mass(double rho, double r, double half) { return rho * (M_PI * r * r * 2 * half + 4.0 / 3.0 * M_PI * r * r * r); }
inline double capsule_inertia_perp(double rho, double r, double half) {
    double h = 2 * half, mcyl = rho * M_PI * r * r * h, msph = rho * 4.0 / 3.0 * M_PI * r * r * r;
    return mcyl * (3 * r * r + h * h) / 12 + msph * (2 * r * r / 5 + h * h / 4 + 3 * h * r / 8);
}
}  // namespace host

// constants from assets/inverted_pendulum.xml (gravity :8; slider range :14; cart capsule :15;
// hinge :17; pole capsule :18; motor :23), capsule mass / inertia by MuJoCo's inertiafromgeom forms
inline Model make_model(bool swingup, double dt) {
    Model m;
    memset(&m, 0, sizeof(m));
    constexpr IpModel x = ip_make_model(false);  // pendulum_envs.h: the one typed copy of the XML's numbers
    const double g = x.gravity, mc = x.mc, mp = x.mp, Icom = x.Icom, r = x.r, phi0 = x.phi0;
    m.M11 = mc + mp, m.M22 = Icom + mp * r * r, m.mpr = mp * r, m.mgr = mp * g * r;
    m.gear = x.gear, m.ctrl_lo = x.ctrl_lo, m.ctrl_hi = x.ctrl_hi, m.x_lo = x.x_lo, m.x_hi = x.x_hi;
    m.phi_off = phi0 + (swingup ? M_PI : 0.0);  // _update_model: pole body turned by pi about y (:135-137)
    m.sin_off = std::sin(m.phi_off), m.cos_off = std::cos(m.phi_off);
    const double M12 = m.mpr * std::cos(phi0);
    m.invw = m.M22 / (m.M11 * m.M22 - M12 * M12);  // dof_invweight0 of the slider at qpos0
    const double tc = x.solref_tc < 2 * dt ? 2 * dt : x.solref_tc, dmax = x.dmax;  // default solref (.02 1) refsafe'd, solimp (.9 .95 .001)
    m.K = 1.0 / (dmax * dmax * tc * tc), m.B = 2.0 / (dmax * tc);
    m.dmin = x.dmin, m.dmax = dmax, m.width = x.width;
    m.dt = dt;
    return m;
}

}  // namespace ipend

template <int VARIANT, typename R>
struct InvPendBody {
    using real = R;
    using Model = ipend::Model;
    static constexpr int kMinWavesPerEU = 1;
    static constexpr bool kUnrollRK4 = true;
    static constexpr int kScratchPerLane = 0;
    static constexpr bool kHasCtrlCost = false;
    static constexpr bool kObsIsState = true;
    static constexpr bool kSpareReset = true;
    static constexpr bool kStreamOutputs = true;  // emei_device.h:store_body_out
    static constexpr int NS = 4, NO = 4, NA = 1;
    static Model make_model(double dt, const EnvParams&) { return ipend::make_model(VARIANT >= 2, dt); }

    // q = (x, theta), v = (xdot, omega); no joint damping in this model (`hd` unused)
    struct Warm {};  // a single constraint row: solved exactly in one step
    __device__ __forceinline__ static void begin_stages(Warm&) {}
    __device__ __forceinline__ static void accel(const R (&q)[2], const R (&v)[2], const R (&ctrl)[NA], const Model& m, R,
                                                 R (&qacc)[2], const TrigCtx& trig, Warm&) {
        R sn, cs;
        sincos_ctx(trig, q[1] + (R)m.phi_off, sn, cs);
        const R M11 = (R)m.M11, M22 = (R)m.M22, M12 = (R)m.mpr * cs;
        const R u = ctrl[0] < (R)m.ctrl_lo ? (R)m.ctrl_lo : (ctrl[0] > (R)m.ctrl_hi ? (R)m.ctrl_hi : ctrl[0]);
        const R f1 = (R)m.gear * u + (R)m.mpr * sn * v[1] * v[1];
        const R f2 = (R)m.mgr * sn;
        const R idet = rcp_r(fma_r(-M12, M12, M11 * M22));
        R a0 = fma_r(M22, f1, -(M12 * f2)) * idet;
        R a1 = fma_r(M11, f2, -(M12 * f1)) * idet;
        // Balancing variants: slider range AND the hinge's +-90 degree stop, in the general two-row form (pendulum_envs.h:
        // ip_limit_rows); SwingUp variants (free hinge): the slider's one-row closed form below
        if constexpr (VARIANT < 2) {
            constexpr IpModel x = ip_make_model(false);
            ip_limit_rows(x, q[0], q[1], v[0], v[1], M12, idet, (R)m.K, (R)m.B, a0, a1);
            qacc[0] = a0, qacc[1] = a1;
            return;
        }
        // soft slider limit; x_lo < x_hi: at most one side is violated, the smaller distance is it
        const R dlo = q[0] - (R)m.x_lo, dhi = (R)m.x_hi - q[0];
        const bool lower = dlo < dhi;
        const R dist = lower ? dlo : dhi, J = lower ? R(1) : R(-1);
        if (dist < R(0)) {
            const R xx = div_r(fabs(dist), (R)m.width);
            const R y = xx >= R(1) ? R(1) : (xx <= R(0.5) ? R(2) * xx * xx : R(1) - R(2) * (R(1) - xx) * (R(1) - xx));
            const R imp = (R)m.dmin + y * ((R)m.dmax - (R)m.dmin);
            const R aref = -(R)m.B * (J * v[0]) - (R)m.K * imp * dist;
            const R Rr = div_r(R(1) - imp, imp) * (R)m.invw;
            const R force = div_r(aref - J * a0, M22 * idet + Rr);
            if (force > R(0)) {
                a0 += (M22 * idet) * J * force;
                a1 += (-M12 * idet) * J * force;
            }
        }
        qacc[0] = a0, qacc[1] = a1;
    }

    template <typename T>
    __device__ __forceinline__ static T wrap(T th) {  // inverted_pendulum.py:45-49
        const T pi = T(3.141592653589793);
        return pymod_pos(th + pi, T(2) * pi, T(1.0 / (2 * 3.141592653589793))) - pi;
    }
    // `sc(x, s, c)`: the caller's sincos (LDS table inside the rollout, polynomial in the stateless kernels)
    template <typename T, typename SC>
    __device__ __forceinline__ static void reward_terminal(const T (&o)[NO], const Model& m, T& rew, bool& term, SC sc) {
        T sn, y;
        sc(o[1], sn, y);
        const bool fin = finite_r(o[0]) & finite_r(o[1]) & finite_r(o[2]) & finite_r(o[3]);
        const bool inx = ((T)m.x_lo < o[0]) & (o[0] < (T)m.x_hi);
        if (VARIANT == 0) rew = T(1), term = !((y >= T(0.9)) & fin);          // :73-79
        else if (VARIANT == 1) rew = T(1), term = !((y >= T(0)) & inx & fin);  // :103-111
        else if (VARIANT == 2) rew = (T(1) - y) / T(2), term = !fin;           // :139-146
        else rew = (T(1) - y) / T(2), term = !(inx & fin);                     // :174-183
    }
    __device__ __forceinline__ static void outputs(const R (&s)[NS], const R (&)[NS], const R (&)[NA], const Model& m, int,
                                                   float (&o)[NO], R& rew, bool& term, const TrigCtx& trig) {
        R ob[NO] = {s[0], wrap(s[1]), s[2], s[3]};
        reward_terminal(ob, m, rew, term, [&](R x, R& sn, R& cs) { sincos_ctx(trig, x, sn, cs); });
#pragma unroll
        for (int k = 0; k < NO; ++k) o[k] = (float)ob[k];
    }
    __device__ __forceinline__ static void init_base(R (&)[NS]) {}  // init_qpos = init_qvel = 0
    // the state of the padding lanes of a ragged last wave (body_kernels.h): at rest in the middle of the rail
    __device__ __forceinline__ static void park(R (&s)[NS]) {
#pragma unroll
        for (int k = 0; k < NS; ++k) s[k] = R(0);
    }
    __device__ __forceinline__ static void obs_of(const R (&s)[NS], double (&o)[NO], const Model&) {
        o[0] = (double)s[0], o[1] = (double)wrap(s[1]), o[2] = (double)s[2], o[3] = (double)s[3];
    }
    template <typename T>
    __device__ __forceinline__ static double batch_reward(const T* obs, const T*, const T*, const Model& m, int) {
        double o[NO] = {obs[0], obs[1], obs[2], obs[3]}, rew;
        bool term;
        reward_terminal(o, m, rew, term, [](double x, double& sn, double& cs) { sincos_r(x, sn, cs); });
        return rew;
    }
    template <typename T>
    __device__ __forceinline__ static bool batch_terminal(const T* obs, const Model& m) {
        double o[NO] = {obs[0], obs[1], obs[2], obs[3]}, rew;
        bool term;
        reward_terminal(o, m, rew, term, [](double x, double& sn, double& cs) { sincos_r(x, sn, cs); });
        return term;
    }
};

}  // namespace emei
