// pendulum_envs.h — per-env device arithmetic for the 4-state cart/pole family:
//   CartPoleSwingUp / CartPoleBalancing           (classic control, first-party ODE)
//   {Rebound,Boundary}InvertedPendulum{Balancing,SwingUp}  (MuJoCo-backed 2-DoF body)
// Each Env type exposes the same static interface used by the generic step/rollout kernel in
// pendulum_kernels.hip:
//   Carry            trig of the current angle, carried from one (sub)step to the next so that one
//                    sincos per substep serves both the dynamics and the reward
//   prime(s, c, p)   compute the carry for state s
//   step(s, c, act, p, freq_rate, obs, reward, terminal)
//   obs_of(s, o)     observation of a state
//   init(s, seed, env, episode, p)   device-side reset
// R is the arithmetic type: double (EMEI_PRECISION_REF) or float (EMEI_PRECISION_F32).
#pragma once
#include "emei_device.h"

namespace emei {

// Kernel-argument block with every model constant the family needs (filled on the host by
// pend_params(); InvertedPendulum constants are derived from assets/inverted_pendulum.xml).
struct PendParams {
    double dt;   // real_time_scale: the time step of ONE substep (base_control.py:73; mujoco_env.py:69)
    float dt32;  // float32(dt): the weak-scalar promotion of `derivs(y) * dt` (base_control.py:164)
    float init_sigma;
    // InvertedPendulum model (see oracle/emei_oracle.c:emei_oracle_ip_model for the derivation)
    double M11, M22, mpr, mgr, gear, ctrl_lo, ctrl_hi, x_lo, x_hi;
    double sin_off, cos_off, phi_off;  // phi = theta + phi_off is the com angle from +z
    double invw, tc, dampratio, dmin, dmax, width;
};

// =============================================================================================
// CartPole — emei/envs/classic_control/cartpole.py
// VARIANT 0 = SwingUp (:135-156), 1 = Balancing (:115-132)
// =============================================================================================
template <int VARIANT, typename R>
struct CartPole {
    using real = R;
    static constexpr bool kDiscrete = true;
    static constexpr int kActDim = 1;
    struct Carry {
        R sn, cs;
    };
    using Action = R;  // the force

    __device__ __forceinline__ static Action load_action(const void* p, int dtype, int64_t idx) {
        // _extract_action, cartpole.py:121-122,142-143: +force_mag if action == 1 else -force_mag
        return load_discrete_action(p, dtype, idx) == 1 ? R(10.0) : R(-10.0);
    }

    __device__ __forceinline__ static void prime(const R s[4], Carry& c, const PendParams&) {
        sincos_r(s[2], c.sn, c.cs);
    }

    // One explicit-Euler substep: cartpole.py:48-60 (_dsdt) + base_control.py:162-164.
    // The expression order is the reference's; the build compiles with -ffp-contract=off so that
    // no product/sum is fused behind its back.
    __device__ __forceinline__ static void substep(R s[4], Carry& c, R force, const PendParams& p) {
        const R gravity = R(9.8), mass_pole = R(0.1), total_mass = R(0.1) + R(1.0), length = R(0.5);
        const R pole_mass_length = mass_pole * length;
        R x_dot = s[1], theta_dot = s[3];
        R temp = (force + pole_mass_length * (theta_dot * theta_dot) * c.sn) / total_mass;
        R theta_acc = (gravity * c.sn - c.cs * temp) /
                      (length * (R(4.0) / R(3.0) - mass_pole * (c.cs * c.cs) / total_mass));
        R x_acc = temp - pole_mass_length * theta_acc * c.cs / total_mass;
        // derivative rounded to float32 (cartpole.py:60), float32 product with float32(dt), then
        // accumulated in R (float64 in the reference)
        s[0] += (R)__fmul_rn((float)x_dot, p.dt32);
        s[1] += (R)__fmul_rn((float)x_acc, p.dt32);
        s[2] += (R)__fmul_rn((float)theta_dot, p.dt32);
        s[3] += (R)__fmul_rn((float)theta_acc, p.dt32);
        sincos_r(s[2], c.sn, c.cs);
    }

    __device__ __forceinline__ static void obs_of(const R s[4], R o[4]) {
        o[0] = s[0], o[1] = s[1], o[2] = s[2], o[3] = s[3];
    }

    // reward / terminal from the carry of the NEW state
    __device__ __forceinline__ static R reward(const R o[4], const Carry& c, const PendParams&) {
        if (VARIANT == 0) return (c.cs + R(1)) / R(2);  // cartpole.py:149-151
        return R(1);                                     // cartpole.py:128-129
    }
    __device__ __forceinline__ static bool terminal(const R o[4], const Carry&, const PendParams&) {
        if (VARIANT == 0) {
            bool notdone = fabs(o[0]) < R(5);  // cartpole.py:140,145-147
            return !notdone;
        }
        const R theta_thr = R(12 * 2 * 3.141592653589793 / 360);  // cartpole.py:30
        bool notdone = (fabs(o[2]) < theta_thr) & (fabs(o[0]) < R(2.4));  // cartpole.py:124-126
        return !notdone;
    }

    __device__ __forceinline__ static void step(R s[4], Carry& c, Action force, const PendParams& p, int freq_rate,
                                                R o[4], R& rew, bool& term) {
        for (int k = 0; k < freq_rate; ++k) substep(s, c, force, p);  // base_control.py:73,162-164
        obs_of(s, o);
        rew = reward(o, c, p);
        term = terminal(o, c, p);
    }

    // device reset: U(-0.05,0.05)^4, SwingUp theta += pi (cartpole.py:131-132,153-156)
    __device__ __forceinline__ static void init(R s[4], uint64_t seed, uint64_t env, uint32_t episode,
                                                const PendParams&) {
        u32x4 r = philox4x32_10(seed, env, episode, 0);
        float f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __fmaf_rn(0.1f, u01(r.v[i]), -0.05f);
        if (VARIANT == 0) f[2] = __fadd_rn(f[2], 3.14159265358979323846f);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = (R)f[i];
    }
};

// =============================================================================================
// InvertedPendulum — emei/envs/mujoco/inverted_pendulum.py on emei/envs/mujoco/mujoco_env.py
// VARIANT 0 ReboundBalancing, 1 BoundaryBalancing, 2 ReboundSwingUp, 3 BoundarySwingUp.
// state = (x, theta_unwrapped, v, omega) = (qpos, qvel); obs wraps theta (:45-49).
// Dynamics: MuJoCo's 2-DoF model in closed form + emei's forward-Euler position override
// (mujoco_env.py:91-97,169-195).  Parity with libmujoco is unpinned.
// =============================================================================================
template <int VARIANT, typename R>
struct InvPend {
    using real = R;
    static constexpr bool kDiscrete = false;
    static constexpr int kActDim = 1;
    struct Carry {
        R sn, cs;  // of phi = theta + phi_off
    };
    using Action = R;  // clipped ctrl

    __device__ __forceinline__ static Action load_action(const void* p, int dtype, int64_t idx) {
        R a = (dtype == EMEI_ACT_F32) ? (R)((const float*)p)[idx] : (R)load_discrete_action(p, dtype, idx);
        return a;
    }

    __device__ __forceinline__ static void prime(const R s[4], Carry& c, const PendParams& p) {
        sincos_r(s[1] + (R)p.phi_off, c.sn, c.cs);
    }

    __device__ __forceinline__ static void substep(R s[4], Carry& c, R u, const PendParams& p) {
        const R M11 = (R)p.M11, M22 = (R)p.M22;
        R M12 = (R)p.mpr * c.cs;
        R ctrl = u < (R)p.ctrl_lo ? (R)p.ctrl_lo : (u > (R)p.ctrl_hi ? (R)p.ctrl_hi : u);  // ctrllimited
        R f1 = (R)p.gear * ctrl + (R)p.mpr * c.sn * s[3] * s[3];
        R f2 = (R)p.mgr * c.sn;
        R det = M11 * M22 - M12 * M12;
        R a0 = (M22 * f1 - M12 * f2) / det;
        R a1 = (M11 * f2 - M12 * f1) / det;
        // soft slider-limit constraint (MuJoCo joint limit, default solref/solimp)
        R dist = R(0), J = R(0);
        if (s[0] - (R)p.x_lo < R(0)) {
            dist = s[0] - (R)p.x_lo, J = R(1);
        } else if ((R)p.x_hi - s[0] < R(0)) {
            dist = (R)p.x_hi - s[0], J = R(-1);
        }
        if (J != R(0)) {
            R tc = (R)p.tc;
            R xx = fabs(dist) / (R)p.width;
            R y = xx >= R(1) ? R(1) : (xx <= R(0.5) ? R(2) * xx * xx : R(1) - R(2) * (R(1) - xx) * (R(1) - xx));
            R imp = (R)p.dmin + y * ((R)p.dmax - (R)p.dmin);
            R K = R(1) / ((R)p.dmax * (R)p.dmax * tc * tc * (R)p.dampratio * (R)p.dampratio);
            R B = R(2) / ((R)p.dmax * tc);
            R aref = -B * (J * s[2]) - K * imp * dist;
            R A = M22 / det;
            R Rr = (R(1) - imp) / imp * (R)p.invw;
            R force = (aref - J * a0) / (A + Rr);
            if (force > R(0)) {
                a0 += (M22 / det) * J * force;
                a1 += (-M12 / det) * J * force;
            }
        }
        R dt = (R)p.dt;
        R q0 = s[0] + dt * s[2], q1 = s[1] + dt * s[3];  // get_euler_pos, mujoco_env.py:189-191
        s[2] += dt * a0;                                  // MuJoCo Euler on qvel (no damping here)
        s[3] += dt * a1;
        s[0] = q0, s[1] = q1;
        sincos_r(s[1] + (R)p.phi_off, c.sn, c.cs);
    }

    __device__ __forceinline__ static void obs_of(const R s[4], R o[4]) {
        const R pi = R(3.141592653589793);
        o[0] = s[0];
        o[1] = pymod(s[1] + pi, R(2) * pi) - pi;  // inverted_pendulum.py:45-49
        o[2] = s[2], o[3] = s[3];
    }

    // cos(theta) from the carry of phi = theta + off
    __device__ __forceinline__ static R cos_theta(const Carry& c, const PendParams& p) {
        return c.cs * (R)p.cos_off + c.sn * (R)p.sin_off;
    }
    __device__ __forceinline__ static R reward(const R o[4], const Carry& c, const PendParams& p) {
        if (VARIANT >= 2) return (R(1) - cos_theta(c, p)) / R(2);  // inverted_pendulum.py:139-142,174-177
        return R(1);                                                // :73-74,103-104
    }
    __device__ __forceinline__ static bool terminal(const R o[4], const Carry& c, const PendParams& p) {
        bool fin = finite_r(o[0]) & finite_r(o[1]) & finite_r(o[2]) & finite_r(o[3]);
        bool inx = ((R)p.x_lo < o[0]) & (o[0] < (R)p.x_hi);
        R y = cos_theta(c, p);
        bool notdone;
        if (VARIANT == 0) notdone = (y >= R(0.9)) & fin;           // :76-79
        else if (VARIANT == 1) notdone = (y >= R(0)) & inx & fin;  // :106-111
        else if (VARIANT == 2) notdone = fin;                      // :144-146
        else notdone = inx & fin;                                  // :179-183
        return !notdone;
    }

    __device__ __forceinline__ static void step(R s[4], Carry& c, Action u, const PendParams& p, int freq_rate,
                                                R o[4], R& rew, bool& term) {
        for (int k = 0; k < freq_rate; ++k) substep(s, c, u, p);  // mujoco_env.py:88-97
        obs_of(s, o);
        rew = reward(o, c, p);
        term = terminal(o, c, p);
    }

    // device reset: init_qpos/qvel (zeros) + sigma * N(0,1) per coordinate (mujoco_env.py:137-140)
    __device__ __forceinline__ static void init(R s[4], uint64_t seed, uint64_t env, uint32_t episode,
                                                const PendParams& p) {
        u32x4 r = philox4x32_10(seed, env, episode, 0);
        float z[4];
        boxmuller(r.v[0], r.v[1], z[0], z[1]);
        boxmuller(r.v[2], r.v[3], z[2], z[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = (R)__fmul_rn(p.init_sigma, z[i]);
    }
};

}  // namespace emei
