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
    float init_sigma[4];               // Gaussian init noise per coordinate (x, theta, v, omega) (mujoco_env.py:137-140)
    int32_t noise_shared;              // EMEI_NOISE_SHARED: the reference's B = 1 row-slicing layout (:243-244)
    // InvertedPendulum model (see oracle/emei_oracle.c:emei_oracle_ip_model for the derivation)
    double M11, M22, mpr, mgr, gear, ctrl_lo, ctrl_hi, x_lo, x_hi;
    double sin_off, cos_off, phi_off;  // phi = theta + phi_off is the com angle from +z
    double invw, tc, dampratio, dmin, dmax, width;
    double limK, limB;  // 1 / (dmax^2 tc^2 dampratio^2), 2 / (dmax tc)
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
        TrigCtx trig;
    };
    using Action = R;  // force / total_mass
    // only the float32 time step reaches the kernel: a small argument block leaves the scalar
    // registers to the polynomial constants of sincos (otherwise they are copied through VGPRs)
    struct Params {
        float dt32;
    };
    static Params make_params(const PendParams& p) { return Params{p.dt32}; }

    using RawAction = int;
    __device__ __forceinline__ static RawAction load_raw(const void* p, int dtype, int64_t idx) {
        return load_discrete_action(p, dtype, idx);
    }
    template <typename T>
    __device__ __forceinline__ static Action decode_t(T a) {
        return a == T(1) ? R(10.0 / 1.1) : R(-10.0 / 1.1);
    }
    __device__ __forceinline__ static Action decode(RawAction a) {
        // _extract_action, cartpole.py:121-122,142-143: +force_mag if action == 1 else -force_mag
        // folded with the division by total_mass (cartpole.py:54): +-force_mag / total_mass
        return a == 1 ? R(10.0 / 1.1) : R(-10.0 / 1.1);
    }
    __device__ __forceinline__ static Action load_action(const void* p, int dtype, int64_t idx) {
        return decode(load_raw(p, dtype, idx));
    }

    __device__ __forceinline__ static void prime(const R s[4], Carry& c, const Params&) {
        sincos_ctx(c.trig, s[2], c.sn, c.cs);
    }

    // One explicit-Euler substep: cartpole.py:48-60 (_dsdt) + base_control.py:162-164.
    //   temp      = (force + pml*thd^2*sin) / M
    //   theta_acc = (g*sin - cos*temp) / (l*(4/3 - mp*cos^2/M))
    //   x_acc     = temp - pml*theta_acc*cos / M
    // The three divisions by the constant total mass are folded into constants and the one true
    // division uses a refined hardware reciprocal: a few ulp of the working precision away from the
    // reference's expression order, far below the float32 rounding the reference applies to the
    // derivative next (cartpole.py:60).
    __device__ __forceinline__ static void substep(R s[4], Carry& c, R force_over_m, const Params& p) {
        const R A = R(0.1 * 0.5 / 1.1);        // pole_mass_length / total_mass
        const R B = R(0.5 * 0.1 / 1.1);        // length * mass_pole / total_mass
        const R L43 = R(0.5 * 4.0 / 3.0);      // length * 4/3
        const R gravity = R(9.8);
        const R x_dot = s[1], theta_dot = s[3];
        // theta advances with the OLD theta_dot only, so its new sin/cos does not wait for the
        // accelerations: issue it first and let it overlap the dynamics below
        s[0] += (R)__fmul_rn((float)x_dot, p.dt32);
        s[2] += (R)__fmul_rn((float)theta_dot, p.dt32);
        R sn = c.sn, cs = c.cs;
        auto pending = sincos_begin_ctx(c.trig, s[2]);  // table read in flight under the dynamics
        sincos_pin(pending, sn, cs);
        R temp = fma_r(A * (theta_dot * theta_dot), sn, force_over_m);
        R num = fma_r(gravity, sn, -(cs * temp));
        R den = fma_r(-B, cs * cs, L43);
        R theta_acc = div_r(num, den);
        R x_acc = fma_r(-(A * theta_acc), cs, temp);
        // derivative rounded to float32 (cartpole.py:60), float32 product with float32(dt)
        // (base_control.py:164, weak-scalar promotion), accumulated in R (float64 in the reference)
        s[1] += (R)__fmul_rn((float)x_acc, p.dt32);
        s[3] += (R)__fmul_rn((float)theta_acc, p.dt32);
        sincos_end_ctx(pending, x_acc, theta_acc, c.sn, c.cs);
        sincos_repair_r(s[2], c.sn, c.cs);  // |theta| > 1e6 only; after the straight-line block
    }

    __device__ __forceinline__ static void obs_of(const R s[4], R o[4]) {
        o[0] = s[0], o[1] = s[1], o[2] = s[2], o[3] = s[3];
    }

    // reward / terminal from the carry of the NEW state
    __device__ __forceinline__ static R reward(const R o[4], const Carry& c, const Params&) {
        if (VARIANT == 0) return (R)__builtin_fmaf((float)c.cs, 0.5f, 0.5f);  // (cos+1)/2, cartpole.py:149-151
        return R(1);                                     // cartpole.py:128-129
    }
    __device__ __forceinline__ static bool terminal(const R o[4], const Carry&, const Params&) {
        if (VARIANT == 0) {
            bool notdone = fabs(o[0]) < R(5);  // cartpole.py:140,145-147
            return !notdone;
        }
        const R theta_thr = R(12 * 2 * 3.141592653589793 / 360);  // cartpole.py:30
        bool notdone = (fabs(o[2]) < theta_thr) & (fabs(o[0]) < R(2.4));  // cartpole.py:124-126
        return !notdone;
    }

    __device__ __forceinline__ static void step(R s[4], Carry& c, Action force, const Params& p, int freq_rate,
                                                R o[4], R& rew, bool& term) {
        for (int k = 0; k < freq_rate; ++k) substep(s, c, force, p);  // base_control.py:73,162-164
        obs_of(s, o);
        rew = reward(o, c, p);
        term = terminal(o, c, p);
    }

    // device reset: U(-0.05,0.05)^4, SwingUp theta += pi (cartpole.py:131-132,153-156)
    __device__ __forceinline__ static void init(R s[4], uint64_t seed, uint64_t env, uint32_t episode,
                                                const Params&) {
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
        TrigCtx trig;
    };
    using Action = R;  // clipped ctrl
    using Params = PendParams;
    static Params make_params(const PendParams& p) { return p; }

    using RawAction = float;
    __device__ __forceinline__ static RawAction load_raw(const void* p, int dtype, int64_t idx) {
        return (dtype == EMEI_ACT_F32) ? ((const float*)p)[idx] : (float)load_discrete_action(p, dtype, idx);
    }
    __device__ __forceinline__ static Action decode(RawAction a) { return (R)a; }
    template <typename T>
    __device__ __forceinline__ static Action decode_t(T a) {
        return (R)a;
    }
    __device__ __forceinline__ static Action load_action(const void* p, int dtype, int64_t idx) {
        return decode(load_raw(p, dtype, idx));
    }

    __device__ __forceinline__ static void prime(const R s[4], Carry& c, const PendParams& p) {
        sincos_ctx(c.trig, s[1] + (R)p.phi_off, c.sn, c.cs);
    }

    __device__ __forceinline__ static void substep(R s[4], Carry& c, R u, const PendParams& p) {
        const R M11 = (R)p.M11, M22 = (R)p.M22;
        const R dt = (R)p.dt;
        // the new angle needs only the OLD angular velocity (get_euler_pos, mujoco_env.py:189-191), so its
        // sin/cos is started first and lands under the dynamics (see CartPole::substep)
        const R x_old = s[0], v_old = s[2], om_old = s[3];
        s[0] = fma_r(dt, v_old, x_old);
        s[1] = fma_r(dt, om_old, s[1]);
        R sn = c.sn, cs = c.cs;
        auto pending = sincos_begin_ctx(c.trig, s[1] + (R)p.phi_off);
        sincos_pin(pending, sn, cs);
        R M12 = (R)p.mpr * cs;
        R ctrl = u < (R)p.ctrl_lo ? (R)p.ctrl_lo : (u > (R)p.ctrl_hi ? (R)p.ctrl_hi : u);  // ctrllimited
        R f1 = (R)p.gear * ctrl + (R)p.mpr * sn * om_old * om_old;
        R f2 = (R)p.mgr * sn;
        R det = fma_r(-M12, M12, M11 * M22);
        R idet = rcp_r(det);
        R a0 = fma_r(M22, f1, -(M12 * f2)) * idet;
        R a1 = fma_r(M11, f2, -(M12 * f1)) * idet;
        // soft slider-limit constraint (MuJoCo joint limit, default solref/solimp)
        // x_lo < x_hi: at most one side is violated, the smaller of the two distances is it (branch-free pick)
        const R dlo = x_old - (R)p.x_lo, dhi = (R)p.x_hi - x_old;
        const bool lower = dlo < dhi;
        const R dist = lower ? dlo : dhi, J = lower ? R(1) : R(-1);
        if (dist < R(0)) {
            R xx = div_r(fabs(dist), (R)p.width);
            R y = xx >= R(1) ? R(1) : (xx <= R(0.5) ? R(2) * xx * xx : R(1) - R(2) * (R(1) - xx) * (R(1) - xx));
            R imp = (R)p.dmin + y * ((R)p.dmax - (R)p.dmin);
            R aref = -(R)p.limB * (J * v_old) - (R)p.limK * imp * dist;  // solref stiffness / damping, host constants
            R A = M22 * idet;
            R Rr = div_r(R(1) - imp, imp) * (R)p.invw;
            R force = div_r(aref - J * a0, A + Rr);
            if (force > R(0)) {
                a0 += (M22 * idet) * J * force;
                a1 += (-M12 * idet) * J * force;
            }
        }
        s[2] = fma_r(dt, a0, v_old);   // MuJoCo Euler on qvel (no joint damping in this model)
        s[3] = fma_r(dt, a1, om_old);
        sincos_end_ctx(pending, a0, a1, c.sn, c.cs);
        sincos_repair_r(s[1] + (R)p.phi_off, c.sn, c.cs);
    }

    __device__ __forceinline__ static void obs_of(const R s[4], R o[4]) {
        const R pi = R(3.141592653589793);
        o[0] = s[0];
        o[1] = pymod_pos(s[1] + pi, R(2) * pi, R(1.0 / (2 * 3.141592653589793))) - pi;  // inverted_pendulum.py:45-49
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

    // device reset: init_qpos/qvel (zeros) + sigma * N(0,1) (mujoco_env.py:137-140); same draws as
    // body_kernels.h:gauss_state so both rollout paths of this env reset identically
    __device__ __forceinline__ static void init(R s[4], uint64_t seed, uint64_t env, uint32_t episode,
                                                const PendParams& p) {
        u32x4 r = philox4x32_10(seed, env, episode, 0);
        float z[4];
        boxmuller(r.v[0], r.v[1], z[0], z[1]);
        boxmuller(r.v[2], r.v[3], z[2], z[3]);
        if (p.noise_shared) {
            s[0] = s[1] = (R)__fmul_rn(p.init_sigma[0], z[0]);
            s[2] = s[3] = (R)__fmul_rn(p.init_sigma[2], z[1]);
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = (R)__fmul_rn(p.init_sigma[i], z[i]);
    }
};

}  // namespace emei
