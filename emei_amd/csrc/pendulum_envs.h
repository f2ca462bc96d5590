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
#include "constexpr_math.h"
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
#ifndef EMEI_SPLIT_MAX_ROUNDS
#define EMEI_SPLIT_MAX_ROUNDS 3
#endif
template <int VARIANT, typename R>
struct CartPole {
    using real = R;
    static constexpr bool kDiscrete = true;
    static constexpr int kActDim = 1;
    struct Carry {
        R sn, cs;
        // float32(x_dot), float32(theta_dot) of the CURRENT state: the position update multiplies exactly these by float32(dt)
        // (base_control.py:164) and the observation of the previous step stored exactly these — converted once, at the end of the
        // substep that produced them, instead of once for the observation and again for the next update (2 of a step's 74
        // vector instructions)
        float xd32, td32;
        TrigCtx trig;
    };
    using Action = R;  // force / total_mass
    __host__ __device__ static constexpr double trig_rot_c() { return 1.0; }  // the plain table (emei_device.h:stage_trig_table)
    __host__ __device__ static constexpr double trig_rot_s() { return 0.0; }
    // BASELINE configs[1] / [4] give the staged rollout 1-2 waves per SIMD: registers are free, the spare initial state
    // of the reset path lives in them
    static constexpr bool kSpareInLds = false;
    static constexpr bool kSpareFlagInVgpr = false;
    static constexpr bool kRotatePriority = false;  // pendulum_kernels.h: rotate_priority
    // pendulum_kernels.h:launch_rollout_full — bound by its write stream: shards of 2 .. kSplitMaxRounds waves per SIMD run as consecutive
    // launches of one wave per SIMD (set from the sweep in profiles/r05_split_launch.txt)
    static constexpr bool kSplitLaunch = true;
    static constexpr unsigned kSplitMaxRounds = EMEI_SPLIT_MAX_ROUNDS;
    static constexpr bool kXcdContiguous = true;  // pendulum_kernels.h: the env_block map of large shards
    static constexpr bool kPeerWrite = true;      // pendulum_kernels.h: pend_rollout_staged_peers_kernel is instantiated (emei_set_obs_peers)
    static constexpr bool kTileBarrier = false;   // pendulum_kernels.h: the block's waves meet once per tile — superseded by the non-temporal stores
    // Balancing under random actions: the pole falls within ~20 steps, some lane of a wave resets in 95 % of its env-steps
    // (SwingUp: 20 %) — the reset block in line (pendulum_kernels.h:maybe_reset)
    static constexpr bool kResetLikely = VARIANT == 1;
    static constexpr int kMinWavesPerEU = 1;
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
        after_reset(s, c);
    }
    // the part of the carry that is not trigonometry: refreshed wherever the state is replaced (reset from a spare state)
    __device__ __forceinline__ static void after_reset(const R s[4], Carry& c) { c.xd32 = (float)s[1], c.td32 = (float)s[3]; }
    // ... kept with a spare initial state, so that a reset copies it (two 32-bit selects) instead of converting again
    __device__ __forceinline__ static void save_extra(const Carry& c, float& e0, float& e1) { e0 = c.xd32, e1 = c.td32; }
    __device__ __forceinline__ static void load_extra(Carry& c, float e0, float e1) { c.xd32 = e0, c.td32 = e1; }

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
        const R theta_dot = s[3];
        // theta advances with the OLD theta_dot only, so its new sin/cos does not wait for the
        // accelerations: issue it first and let it overlap the dynamics below
        // the two float32 products with float32(dt) as ONE v_pk_mul_f32 (IEEE products per half, no contraction: the same bits;
        // hipcc's own pairing of the scalar form left 17 more register moves in the unrolled loop: -0.7 % SwingUp, -1.3 % Balancing)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 dpos = f32x2{c.xd32, c.td32} * p.dt32;
        s[0] += (R)dpos.x;
        s[2] += (R)dpos.y;
        R sn = c.sn, cs = c.cs;
        auto pending = sincos_begin_ctx(c.trig, s[2]);  // table read in flight under the dynamics
        sincos_pin(pending, sn, cs);
        // three-address FMAs where the addend outlives the product (force / M: its low half is shared by both actions; L43: a
        // constant): hipcc's two-address v_fmac_f64 copies the addend first (3 v_mov_b64 of a step's 75 vector instructions,
        // with the angle reduction of emei_device.h:sincos_begin_ctx)
        R temp = fma_vvv(A * (theta_dot * theta_dot), sn, force_over_m);
        R num = fma_r(gravity, sn, -(cs * temp));
        R den = fma_vsv(cs * cs, -B, L43);
        R theta_acc = div_r(num, den);
        R x_acc = fma_r(-(A * theta_acc), cs, temp);
        // derivative rounded to float32 (cartpole.py:60), float32 product with float32(dt)
        // (base_control.py:164, weak-scalar promotion), accumulated in R (float64 in the reference)
        const f32x2 dvel = f32x2{(float)x_acc, (float)theta_acc} * p.dt32;
        s[1] += (R)dvel.x;
        s[3] += (R)dvel.y;
        c.xd32 = (float)s[1], c.td32 = (float)s[3];  // the same conversions the observation store makes: one instruction each
        sincos_end_ctx(pending, x_acc, theta_acc, c.sn, c.cs);
        sincos_post_ctx(s[2], c.sn, c.cs);
    }

    __device__ __forceinline__ static void obs_of(const R s[4], R o[4]) {
        o[0] = s[0], o[1] = s[1], o[2] = s[2], o[3] = s[3];
    }

    // reward / terminal from the carry of the NEW state
    __device__ __forceinline__ static R reward(const R o[4], const Carry& c, const Params&) {
        // (cos+1)/2, cartpole.py:149-151, in the working precision: near the hanging position cos -> -1 and the sum cancels —
        // formed in float32 (round 2) the reward lost up to 3e-8 ABSOLUTE, i.e. 2e-5 relative at reward 5e-4
        if (VARIANT == 0) return fma_r(c.cs, R(0.5), R(0.5));
        return R(1);                                     // cartpole.py:128-129
    }
    __device__ __forceinline__ static R reward_exact(const R o[4], const Carry& c, const Params& p) { return reward(o, c, p); }
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
// CartPole integrated by ODE_approximation(method="rk4") — base_control.py:165-170.  The reference's step() never reaches
// this branch (:73 passes no `method`); a handle selects it explicitly (emei_config.ode_method = EMEI_ODE_RK4).  Everything
// but the substep is CartPole's.  A separate Env type (and translation unit) on purpose: the Euler kernels are the benchmarked
// ones and sit on an instruction-cache edge (pendulum_kernels.h: EMEI_GROUP_UNROLL) — not one instruction of theirs changes.
//
// The promotion chain is the reference's own (NumPy 2; pinned bit for bit by oracle/emei_oracle.c:cp_ode_rk4 against
// tests/golden/cartpole_rk4_golden.npz): _dsdt evaluates in float64 and returns float32; `dt * k` and `/ 2` are float32 (dt a
// weak Python scalar), the stage state `y + dt * k / 2` a float64 sum; `(k1 + 2 k2 + 2 k3 + k4) * dt / 6` a chain of float32
// roundings added to the float64 accumulator.  x itself enters no derivative (cartpole.py:48-60), so the stage states carry
// x_dot, theta, theta_dot only.
// =============================================================================================
template <int VARIANT, typename R>
struct CartPoleRK4 : CartPole<VARIANT, R> {
    using Base = CartPole<VARIANT, R>;
    using real = R;
    using Carry = typename Base::Carry;
    using Params = typename Base::Params;
    using Action = R;

    // (x_acc, theta_acc) of cartpole.py:48-60 at (theta_dot, sin theta, cos theta), rounded to float32 (:60); the same
    // arithmetic as CartPole::substep (constants folded, refined reciprocal: a few ulp of float64 from the reference's
    // expression order, far below the float32 rounding that follows)
    __device__ __forceinline__ static void accel32(R theta_dot, R sn, R cs, R force_over_m, float& x_acc, float& theta_acc) {
        const R A = R(0.1 * 0.5 / 1.1), B = R(0.5 * 0.1 / 1.1), L43 = R(0.5 * 4.0 / 3.0), gravity = R(9.8);
        const R temp = fma_r(A * (theta_dot * theta_dot), sn, force_over_m);
        const R num = fma_r(gravity, sn, -(cs * temp));
        const R den = fma_r(cs * cs, -B, L43);
        const R ta = div_r(num, den);
        const R xa = fma_r(-(A * ta), cs, temp);
        x_acc = (float)xa, theta_acc = (float)ta;
    }

    __device__ __forceinline__ static void substep(R s[4], Carry& c, R force_over_m, const Params& p) {
        const float dt = p.dt32;
        // k = (x_dot, x_acc, theta_dot, theta_acc) as float32; k1 at the current state (its sin / cos is the carry)
        float kxd[4], kxa[4], ktd[4], kta[4];
        kxd[0] = c.xd32, ktd[0] = c.td32;
        accel32(s[3], c.sn, c.cs, force_over_m, kxa[0], kta[0]);
#pragma unroll
        for (int st = 1; st < 4; ++st) {
            // y + dt * k / 2 (stages 2, 3) or y + dt * k (stage 4): float32 product, float32 halving, float64 sum
            const float hxa = __fmul_rn(dt, kxa[st - 1]), htd = __fmul_rn(dt, ktd[st - 1]), hta = __fmul_rn(dt, kta[st - 1]);
            const float half = st < 3 ? 0.5f : 1.0f;  // x / 2 == x * 0.5f exactly, for every float
            const R xd = s[1] + (R)__fmul_rn(hxa, half), th = s[2] + (R)__fmul_rn(htd, half), td = s[3] + (R)__fmul_rn(hta, half);
            R sn, cs;
            sincos_ctx(c.trig, th, sn, cs);
            kxd[st] = (float)xd, ktd[st] = (float)td;
            accel32(td, sn, cs, force_over_m, kxa[st], kta[st]);
        }
        // y += (k1 + 2 k2 + 2 k3 + k4) * dt / 6: float32 throughout, a true division by 6
        auto comb = [&](const float (&k)[4]) __attribute__((always_inline)) {
            const float a = __fadd_rn(k[0], __fmul_rn(2.0f, k[1]));
            const float b = __fadd_rn(a, __fmul_rn(2.0f, k[2]));
            return __fdiv_rn(__fmul_rn(__fadd_rn(b, k[3]), dt), 6.0f);
        };
        s[0] += (R)comb(kxd), s[1] += (R)comb(kxa), s[2] += (R)comb(ktd), s[3] += (R)comb(kta);
        c.xd32 = (float)s[1], c.td32 = (float)s[3];
        sincos_ctx(c.trig, s[2], c.sn, c.cs);
    }

    __device__ __forceinline__ static void step(R s[4], Carry& c, Action force, const Params& p, int freq_rate,
                                                R o[4], R& rew, bool& term) {
        for (int k = 0; k < freq_rate; ++k) substep(s, c, force, p);
        Base::obs_of(s, o);
        rew = Base::reward(o, c, p);
        term = Base::terminal(o, c, p);
    }
};

// =============================================================================================
// InvertedPendulum — emei/envs/mujoco/inverted_pendulum.py on emei/envs/mujoco/mujoco_env.py
// VARIANT 0 ReboundBalancing, 1 BoundaryBalancing, 2 ReboundSwingUp, 3 BoundarySwingUp.
// state = (x, theta_unwrapped, v, omega) = (qpos, qvel); obs wraps theta (:45-49).
// Dynamics: MuJoCo's 2-DoF model in closed form + emei's forward-Euler position override
// (mujoco_env.py:91-97,169-195).  Parity with libmujoco is unpinned.
// =============================================================================================
// Model constants from assets/inverted_pendulum.xml (gravity :8; slider range :14; cart capsule :15; hinge :17; pole
// capsule :18; motor :23), capsule mass / inertia by MuJoCo's inertiafromgeom forms — evaluated at COMPILE time
// (constexpr_math.h): as kernel arguments the 25 doubles of PendParams + the sincos constants + the pointers exceeded
// the SGPR file (154 spilled SGPRs = v_readlane / v_writelane VECTOR instructions in the rollout loop; round 1).  Only
// what depends on the run-time dt travels as an argument.  oracle/emei_oracle.c:emei_oracle_ip_model derives the same
// numbers with libm; abi.hip:pend_params is the host twin used by the Body path.
struct IpModel {
    double M11, M22, M11M22, mpr, inv_mpr, gravity, gear, ctrl_lo, ctrl_hi, x_lo, x_hi;
    double phi_off, sin_off, cos_off;  // phi = theta + phi_off is the angle of the pole's com from +z
    double invw, dmin, dmax, inv_width;
    // what the XML's geoms compile to (inertiafromgeom, density 1000): cart mass, pole mass, pole inertia about its com, com
    // distance from the hinge, tilt of the pole's axis at theta = 0 (fromto 0 0 0 0.001 0 0.6); the default solref time constant.
    // THE one hand-typed copy of inverted_pendulum.xml on the kernel side: ipend_model.h and abi.hip:pend_params read these
    // fields, emei_model_constants exports them, tests/test_model_constants.py pins them to the file.
    double mc, mp, Icom, r, phi0, solref_tc, width;
    // hinge range (xml:17, +-90 degrees, `limited` by the joint default) and its dof_invweight0: a limit row of the Balancing
    // variants only (SwingUp's _update_model sets the range to +-inf, inverted_pendulum.py:135-137)
    double th_lo, th_hi, invw_hinge;
};
constexpr IpModel ip_make_model(bool swingup) {
    IpModel m{};
    const double rho = 1000.0, pi = ce::kPi;
    const auto capsule_mass = [&](double r, double half) { return rho * (pi * r * r * 2 * half + 4.0 / 3.0 * pi * r * r * r); };
    const auto capsule_inertia_perp = [&](double r, double half) {
        const double h = 2 * half, mcyl = rho * pi * r * r * h, msph = rho * 4.0 / 3.0 * pi * r * r * r;
        return mcyl * (3 * r * r + h * h) / 12 + msph * (2 * r * r / 5 + h * h / 4 + 3 * h * r / 8);
    };
    const double mc = capsule_mass(0.1, 0.1);
    const double fx = 0.001, fz = 0.6, len = ce::sqrt(fx * fx + fz * fz);
    const double mp = capsule_mass(0.049, len / 2), Icom = capsule_inertia_perp(0.049, len / 2), r = len / 2;
    const double phi0 = ce::atan_small(fx, fz);
    m.M11 = mc + mp, m.M22 = Icom + mp * r * r, m.M11M22 = m.M11 * m.M22;
    m.mpr = mp * r, m.inv_mpr = 1.0 / m.mpr, m.gravity = 9.81;
    m.gear = 100.0, m.ctrl_lo = -3.0, m.ctrl_hi = 3.0, m.x_lo = -2.0, m.x_hi = 2.0;
    m.phi_off = phi0 + (swingup ? pi : 0.0);  // _update_model: pole body turned by pi about y (:135-137)
    m.sin_off = ce::sin(m.phi_off), m.cos_off = ce::cos(m.phi_off);
    const double M12 = m.mpr * ce::cos(phi0);
    m.invw = m.M22 / (m.M11 * m.M22 - M12 * M12);  // dof_invweight0 of the slider at qpos0
    m.invw_hinge = m.M11 / (m.M11 * m.M22 - M12 * M12);
    m.th_lo = -pi / 2, m.th_hi = pi / 2;
    m.dmin = 0.9, m.dmax = 0.95, m.width = 0.001, m.inv_width = 1.0 / 0.001;  // default solimp (.9 .95 .001)
    m.mc = mc, m.mp = mp, m.Icom = Icom, m.r = r, m.phi0 = phi0, m.solref_tc = 0.02;  // default solref (.02 1)
    return m;
}
__device__ constexpr IpModel kIpUpright = ip_make_model(false), kIpHanging = ip_make_model(true);

// emei_model_constants (include/emei_hip.h): [gravity, mc, mp, Icom, r, phi0, gear, ctrl_lo, ctrl_hi, x_lo, x_hi, solref tc,
// solimp dmin, dmax, width, hinge range lo, hi (Balancing variants)]
inline int ip_xml_constants(double* out) {
    constexpr IpModel m = ip_make_model(false);
    const double v[17] = {m.gravity, m.mc, m.mp, m.Icom, m.r, m.phi0, m.gear, m.ctrl_lo, m.ctrl_hi, m.x_lo, m.x_hi, m.solref_tc,
                          m.dmin, m.dmax, m.width, m.th_lo, m.th_hi};
    for (int i = 0; i < 17; ++i) out[i] = v[i];
    return 17;
}

// Joint-limit rows of the InvertedPendulum in their general form — the slider's range and, for the Balancing variants, the hinge's
// +-90 degree range — as oracle/emei_oracle.c:ip_accel states them: MuJoCo's primal problem with at most two rows is the 2 x 2
// complementarity problem f >= 0, (A + R) f - b >= 0, f'((A + R) f - b) = 0 (A = J M^-1 J', b_i = aref_i - J_i a0), solved by
// enumerating its four active sets (A + R is positive definite: exactly one is consistent).  The hinge stop is only reached by a
// POST-terminal state (Rebound terminates at cos theta < 0.9, Boundary at cos theta < 0; the reference keeps stepping after
// `terminal`), so this runs on a cold, wave-uniform path; the hot path keeps the one-row closed form of the slider.
// M12 = mpr cos(phi), idet = 1 / (M11 M22 - M12^2); a0 / a1 enter as the smooth accelerations and leave constrained.
template <typename R>
__device__ __forceinline__ void ip_limit_rows(const IpModel& m, R x, R th, R v0, R v1, R M12, R idet, R limK, R limB, R& a0, R& a1) {
    R J[2], b[2], Rr[2];
    auto row = [&](int i, R qi, R vi, R lo, R hi, R invw, R ai) __attribute__((always_inline)) {
        const bool low = qi < lo, high = qi > hi, on = low | high;
        const R dist = low ? qi - lo : hi - qi;
        const R Ji = low ? R(1) : R(-1);
        const R xx = fabs(dist) * (R)m.inv_width;
        const R u1 = R(1) - xx;
        const R y = xx >= R(1) ? R(1) : (xx <= R(0.5) ? R(2) * xx * xx : fma_r(R(-2) * u1, u1, R(1)));
        const R imp = fma_r(y, (R)(m.dmax - m.dmin), (R)m.dmin);
        const R aref = fma_r(-limK * imp, dist, -limB * (Ji * vi));
        J[i] = on ? Ji : R(0);
        b[i] = on ? aref - Ji * ai : R(0);
        Rr[i] = on ? div_r(R(1) - imp, imp) * invw : R(1);
    };
    row(0, x, v0, (R)m.x_lo, (R)m.x_hi, (R)m.invw, a0);
    row(1, th, v1, (R)m.th_lo, (R)m.th_hi, (R)m.invw_hinge, a1);
    const bool has0 = J[0] != R(0), has1 = J[1] != R(0);
    const R A01 = J[0] * J[1] * (-M12 * idet);
    const R H00 = fma_r((R)m.M22, idet, Rr[0]), H11 = fma_r((R)m.M11, idet, Rr[1]);
    const R s0 = div_r(b[0], H00), s1 = div_r(b[1], H11);  // one row alone
    const R id2 = rcp_r(fma_r(H00, H11, -(A01 * A01)));
    const R t0 = fma_r(H11, b[0], -(A01 * b[1])) * id2, t1 = fma_r(H00, b[1], -(A01 * b[0])) * id2;  // both rows
    const bool both = has0 & has1 & (t0 > R(0)) & (t1 > R(0));
    const bool only0 = !both & has0 & (s0 > R(0)) & !(has1 & (b[1] - A01 * s0 > R(0)));
    const bool only1 = !both & !only0 & has1 & (s1 > R(0)) & !(has0 & (b[0] - A01 * s1 > R(0)));
    const R f0 = both ? t0 : (only0 ? s0 : R(0)), f1 = both ? t1 : (only1 ? s1 : R(0));
    const R g0 = J[0] * f0, g1 = J[1] * f1;  // generalised force J' f, through M^-1
    a0 = fma_r(fma_r((R)m.M22, g0, -(M12 * g1)), idet, a0);
    a1 = fma_r(fma_r((R)m.M11, g1, -(M12 * g0)), idet, a1);
}

// The same solve for a lane that is beyond BOTH stops (the only lanes the staged kernel's cold path keeps a result for:
// InvPend::substep), bit for bit — J = +-1 is a sign, recomputed where it is used instead of kept; no "is this row on" selects;
// the two-row solution first, the single-row ones after it.  Written for its register footprint: with ip_limit_rows the
// Balancing variants needed one float64 more than the 128 registers of a 4-waves-per-SIMD kernel (12 B of scratch; round 5).
template <typename R>
__device__ __forceinline__ void ip_limit_both(const IpModel& m, R x, R th, R v0, R v1, R M12, R idet, R limK, R limB, R& a0, R& a1) {
    auto row = [&](R qi, R vi, R lo, R hi, R invw, R Mdd, R ai, R& bi, R& Hii) __attribute__((always_inline)) {
        const bool low = qi < lo;
        const R dist = low ? qi - lo : hi - qi;
        const R xx = fabs(dist) * (R)m.inv_width;
        const R u1 = R(1) - xx;
        const R y = xx >= R(1) ? R(1) : (xx <= R(0.5) ? R(2) * xx * xx : fma_r(R(-2) * u1, u1, R(1)));
        const R imp = fma_r(y, (R)(m.dmax - m.dmin), (R)m.dmin);
        const R Jv = low ? vi : -vi, Ja = low ? ai : -ai;
        bi = fma_r(-limK * imp, dist, -limB * Jv) - Ja;
        Hii = fma_r(Mdd, idet, div_r(R(1) - imp, imp) * invw);
    };
    R b0, b1, H00, H11;
    row(x, v0, (R)m.x_lo, (R)m.x_hi, (R)m.invw, (R)m.M22, a0, b0, H00);
    row(th, v1, (R)m.th_lo, (R)m.th_hi, (R)m.invw_hinge, (R)m.M11, a1, b1, H11);
    const R mA = M12 * idet;
    const R A01 = ((x < (R)m.x_lo) == (th < (R)m.th_lo)) ? -mA : mA;  // J0 J1 (-M12 idet)
    const R id2 = rcp_r(fma_r(H00, H11, -(A01 * A01)));
    const R t0 = fma_r(H11, b0, -(A01 * b1)) * id2, t1 = fma_r(H00, b1, -(A01 * b0)) * id2;  // both rows
    const bool both = (t0 > R(0)) & (t1 > R(0));
    const R s0 = div_r(b0, H00);  // one row alone
    const bool only0 = !both & (s0 > R(0)) & !(b1 - A01 * s0 > R(0));
    const R s1 = div_r(b1, H11);
    const bool only1 = !both & !only0 & (s1 > R(0)) & !(b0 - A01 * s1 > R(0));
    const R f0 = both ? t0 : (only0 ? s0 : R(0)), f1 = both ? t1 : (only1 ? s1 : R(0));
    const R g0 = x < (R)m.x_lo ? f0 : -f0, g1 = th < (R)m.th_lo ? f1 : -f1;  // J' f
    a0 = fma_r(fma_r((R)m.M22, g0, -(M12 * g1)), idet, a0);
    a1 = fma_r(fma_r((R)m.M11, g1, -(M12 * g0)), idet, a1);
}

#ifndef EMEI_IP_XCD_CONTIGUOUS
#define EMEI_IP_XCD_CONTIGUOUS 0  // the InvertedPendulum kernels are not bound by their write stream: identity map (A/B: nothing)
#endif
#ifndef EMEI_IP_BAL_WAVES
#define EMEI_IP_BAL_WAVES 4  // waves per SIMD the Balancing variants are compiled for (see kMinWavesPerEU; 3 = rounds 3-4, for A/B runs)
#endif
template <int VARIANT, typename R>
struct InvPend {
    using real = R;
    static constexpr bool kDiscrete = false;
    static constexpr int kActDim = 1;
    static constexpr bool kF64 = sizeof(R) == 8;
    // BASELINE configs[2]: 262 144 envs = 4 waves per SIMD, which only fit with <= 128 registers per lane (and <= 40 KiB
    // of LDS per block): the spare initial state of the reset path goes to LDS, the rollout is compiled for 4 waves
    static constexpr bool kSpareInLds = true;
    static constexpr bool kSpareFlagInVgpr = true;  // pendulum_kernels.h:maybe_reset
    static constexpr bool kRotatePriority = true;   // four waves per SIMD, bound by vector issue: pendulum_kernels.h:rotate_priority
    static constexpr bool kSplitLaunch = false;     // ... and wants its four resident waves
    static constexpr unsigned kSplitMaxRounds = 0;
    static constexpr bool kXcdContiguous = EMEI_IP_XCD_CONTIGUOUS != 0;
    static constexpr bool kPeerWrite = false;  // emei_set_obs_peers: built for the CartPole family (configs[4]); UNSUPPORTED here
    static constexpr bool kTileBarrier = false;  // measured: nothing (+-1 %)
    // config 3: some lane of a wave resets in 95 % of its env-steps (random pushes of +-300 N run the cart off the rail in
    // ~20 steps): the reset block is laid out in line, not behind two taken branches
    static constexpr bool kResetLikely = true;
    // Every variant is compiled for 4 waves per SIMD (<= 128 registers, no scratch: tests/test_isa_guards.py).  Rounds 3-4 ran the
    // Balancing variants at 3: their two-row limit solve of the hinge stop (a cold path) needed one float64 more than 128 registers
    // hold; round 5 rewrote it for the lanes that are beyond both stops (ip_limit_both): 0.744 -> 0.696 ms per launch at
    // 262 144 envs (the fourth wave of a SIMD used to queue behind the first three).
    static constexpr int kMinWavesPerEU = VARIANT >= 2 ? 4 : EMEI_IP_BAL_WAVES;
    __device__ static constexpr const IpModel& km() { return VARIANT >= 2 ? kIpHanging : kIpUpright; }
    // The dynamics only use mpr * sin(phi) and mpr * cos(phi), phi = theta + phi_off: the float64 kernels stage the {sin,cos}
    // table rotated by phi_off and pre-multiplied by the pole's mass moment (emei_device.h:stage_trig_table), look theta
    // itself up (trig_angle) and the carry holds the products
    __host__ __device__ static constexpr double trig_rot_c() { return kF64 ? ip_make_model(VARIANT >= 2).mpr * ip_make_model(VARIANT >= 2).cos_off : 1.0; }
    __host__ __device__ static constexpr double trig_rot_s() { return kF64 ? ip_make_model(VARIANT >= 2).mpr * ip_make_model(VARIANT >= 2).sin_off : 0.0; }
    __device__ __forceinline__ static R trig_angle(R theta) { return kF64 ? theta : theta + (R)km().phi_off; }
    struct Carry {
        R sn, cs;  // mpr * sin(phi), mpr * cos(phi), phi = theta + phi_off
        TrigCtx trig;
    };
    using Action = R;  // gear * clipped ctrl
    struct Params {
        double dt, limK, limB;  // time step; solref stiffness / damping of the slider limit (refsafe'd: depend on dt)
        float init_sigma[4];
        int32_t noise_shared;
    };
    static Params make_params(const PendParams& p) {
        Params q;
        q.dt = p.dt, q.limK = p.limK, q.limB = p.limB, q.noise_shared = p.noise_shared;
        for (int i = 0; i < 4; ++i) q.init_sigma[i] = p.init_sigma[i];
        return q;
    }

    using RawAction = float;
    __device__ __forceinline__ static RawAction load_raw(const void* p, int dtype, int64_t idx) {
        return (dtype == EMEI_ACT_F32) ? ((const float*)p)[idx] : (float)load_discrete_action(p, dtype, idx);
    }
    // ctrllimited motor (xml:23): the actuator force is constant over the substeps of a step.  The clamp runs in
    // float32 (the bounds are exact there): one v_med3_f32
    __device__ __forceinline__ static Action decode(RawAction a) {
        // (the builtin: fminf(fmaxf()) compiles to three instructions, the first a canonicalising v_max_f32 a, a; a NaN action
        // gives ctrl_lo either way — v_med3_f32 returns min3 when an input is NaN)
        const float u = __builtin_amdgcn_fmed3f(a, (float)km().ctrl_lo, (float)km().ctrl_hi);
        return (R)km().gear * (R)u;
    }
    template <typename T>
    __device__ __forceinline__ static Action decode_t(T a) {
        return decode((float)a);
    }
    __device__ __forceinline__ static Action load_action(const void* p, int dtype, int64_t idx) {
        return decode(load_raw(p, dtype, idx));
    }

    __device__ __forceinline__ static void prime(const R s[4], Carry& c, const Params&) {
        sincos_ctx(c.trig, trig_angle(s[1]), c.sn, c.cs);
        if (!kF64) c.sn *= (R)km().mpr, c.cs *= (R)km().mpr;
    }
    __device__ __forceinline__ static void after_reset(const R[4], Carry&) {}  // the carry is trigonometry only
    __device__ __forceinline__ static void save_extra(const Carry&, float&, float&) {}
    __device__ __forceinline__ static void load_extra(Carry&, float, float) {}

    // One substep.  P = mpr sin(phi), Q = mpr cos(phi) (= M12) at the OLD angle, gu = gear * ctrl:
    //   f1 = gu + P omega^2,  f2 = g P,  det = M11 M22 - Q^2
    //   a0 = (M22 f1 - Q f2) / det,  a1 = (M11 f2 - Q f1) / det        (+ the soft slider-limit force)
    //   x += dt v_old, theta += dt omega_old (get_euler_pos, mujoco_env.py:189-191);  v += dt a0, omega += dt a1
    __device__ __forceinline__ static void substep(R s[4], Carry& c, R gu, const Params& p) {
        constexpr IpModel m = ip_make_model(VARIANT >= 2);
        const R dt = (R)p.dt;
        // the new angle needs only the OLD angular velocity, so its sin/cos is started first and lands under the dynamics
        const R x_old = s[0], v_old = s[2], om_old = s[3];
        [[maybe_unused]] const R th_old = s[1];
        s[0] = fma_r(dt, v_old, x_old);
        s[1] = fma_r(dt, om_old, s[1]);
        R P = c.sn, Q = c.cs;
        auto pending = sincos_begin_ctx(c.trig, trig_angle(s[1]));
        sincos_pin(pending, P, Q);
        const R f1 = fma_r(P, om_old * om_old, gu);
        const R gP = (R)m.gravity * P;  // f2
        const R idet = rcp1_r(fma_r(-Q, Q, (R)m.M11M22));
        R a0 = fma_r((R)m.M22, f1, -(Q * gP)) * idet;
        R a1 = fma_r((R)m.M11, gP, -(Q * f1)) * idet;
        // Soft slider-limit constraint (MuJoCo joint limit, default solref / solimp).  The rail is symmetric (xml:14,
        // range -2 2): the violated side, if any, is the one x is on, and its distance is x_hi - |x| < 0.
        static_assert(m.x_lo == -m.x_hi, "symmetric slider range");
        const bool beyond = abs_r(x_old) > (R)m.x_hi;
        EMEI_STAT_WAVE(19);  // substeps (waves)
        // One limit row alone, in closed form (the hot path's limit block: config 3 runs the slider's in 55 % of a wave's substeps
        // for 1.7 % of the lanes, so every instruction here is worth half an instruction of the substep itself).  For the slider:
        //   J = -sign(x) (+1 at the lower stop), dist = x_hi - |x|, imp = impedance(|dist| / width),
        //   aref = -K imp dist - B J v,  R = (1 - imp) / imp invw,  A = J M^-1 J' = M22 / det
        //   force = max(0, (aref - J a0) / (A + R)),  a += M^-1 J' force
        // written with aref - J a0 = J (-B v - a0) - K imp dist so that J only ever multiplies (a +-1.0 factor of an fma, not
        // sign-bit surgery on copies), with -K imp dist = fma(K imp, |x|, -K imp x_hi) (exact: x_hi - |x| is, by Sterbenz), and
        // with the force clamped by a maximum instead of a branch.
        // (one closed form for either joint — DOF 0: the slider's rail, DOF 1: the Balancing variants' +-90 degree hinge stop, whose
        // range is symmetric as well; M^-1 = idet [[M22, -Q], [-Q, M11]])
        auto limit_row = [&](auto dof_c) __attribute__((always_inline)) {
            constexpr int DOF = decltype(dof_c)::value;
            constexpr double q_hi = DOF == 0 ? m.x_hi : m.th_hi, invw = DOF == 0 ? m.invw : m.invw_hinge, Mdd = DOF == 0 ? m.M22 : m.M11;
            const R qd = DOF == 0 ? x_old : th_old, vd = DOF == 0 ? v_old : om_old;
            R& ad = DOF == 0 ? a0 : a1;
            R& ao = DOF == 0 ? a1 : a0;
            if (DOF == 0) {
                EMEI_STAT_WAVE(20);  // ... with the limit block
                EMEI_STAT_LANE(21);  // lanes beyond the rail
            }
            const R nJ = copysign_r(R(1), qd);  // -J
            const R n1 = fma_r((R)p.limB, vd, ad);  // aref - J a = nJ (B v + a) - K imp dist
            const R A = (R)Mdd * idet;
            // impedance: xx = |dist| / width; y = 1 beyond the width (1 mm: every violating lane of the wave, almost always).
            // The two terms that depend on it — the position part of aref and the denominator — are formed for the saturated
            // impedance first, from constants; a wave with a lane inside the width overwrites them for that lane in a cold branch
            // (as two arms of a branch that share the rest, hipcc routed the constants through v_mov_b64).  A lane's bits do not
            // depend on the branch its WAVE took: a lane beyond the width keeps the constants' values either way.
            constexpr double kRfull = (1.0 - m.dmax) / m.dmax * invw;
            const R Kfull = (R)p.limK * (R)m.dmax, Kfull_hi = Kfull * (R)q_hi;
            R kpos = fma_r(Kfull, abs_r(qd), -Kfull_hi);  // -K imp dist, dist = q_hi - |q|
            R den = A + (R)kRfull;                         // J M^-1 J' + R
            const bool full = !(abs_r(qd) < (R)(q_hi + m.width));
            if (__builtin_expect(__ballot(!full) != 0ull, 0)) {
                const R xx = (abs_r(qd) - (R)q_hi) * (R)m.inv_width, u1 = R(1) - xx;
                const R y = xx <= R(0.5) ? R(2) * xx * xx : fma_r(R(-2) * u1, u1, R(1));
                const R imp = fma_r(y, (R)(m.dmax - m.dmin), (R)m.dmin);
                const R Kimp = (R)p.limK * imp;
                kpos = full ? kpos : fma_r(Kimp, abs_r(qd), -(Kimp * (R)q_hi));
                den = full ? den : A + (R(1) - imp) * (R)invw * rcp1_r(imp);
            }
            const R force = fmax_r(fma_r(nJ, n1, kpos) * rcp1_r(den), R(0));
            const R g = (nJ * force) * idet;  // M^-1 J' force = -(Mdd, -Q) g
            ad = fma_r(-(R)Mdd, g, ad);
            ao = fma_r(Q, g, ao);
        };
        // Balancing variants: the hinge's +-90 degree stop.  BoundaryBalancing terminates exactly there (cos theta < 0), so with
        // auto-reset every terminating lane spends the last substeps of its last step beyond the stop, and some lane of a wave
        // does in most substeps (config 3's shape with this variant: 1.24 ms per launch when every such lane took the general
        // two-row solve).  A lane at ONE stop takes that joint's closed form; only a lane at both takes the general solve
        // (ip_limit_rows).  Which form a lane takes depends on its own state alone: its bits never depend on its wave-mates.
        bool hinge = false;
        if constexpr (VARIANT < 2) {
            static_assert(m.th_lo == -m.th_hi, "symmetric hinge range");
            hinge = fabs(th_old) > (R)m.th_hi;
        }
        if (__builtin_expect(VARIANT < 2 && __ballot(hinge) != 0ull, 0)) {
            if (__builtin_expect(__ballot(hinge & beyond) != 0ull, 0)) {
                if (hinge & beyond) ip_limit_both(m, x_old, th_old, v_old, om_old, Q, idet, (R)p.limK, (R)p.limB, a0, a1);
            }
            if (hinge & !beyond) limit_row(std::integral_constant<int, 1>{});
            else if (beyond & !hinge) limit_row(std::integral_constant<int, 0>{});
        } else if (beyond) {
            limit_row(std::integral_constant<int, 0>{});
        }
        s[2] = fma_r(dt, a0, v_old);  // MuJoCo Euler on qvel (no joint damping in this model)
        s[3] = fma_r(dt, a1, om_old);
        sincos_end_ctx(pending, a0, a1, c.sn, c.cs);
        if (!kF64) c.sn *= (R)m.mpr, c.cs *= (R)m.mpr;
        sincos_post_ctx(trig_angle(s[1]), c.sn, c.cs);
    }

    __device__ __forceinline__ static void obs_of(const R s[4], R o[4]) {
        o[0] = s[0];
        o[1] = wrap_pi(s[1]);  // inverted_pendulum.py:45-49
        o[2] = s[2], o[3] = s[3];
    }

    // cos(theta) from the carry of phi = theta + off (the carry is scaled by mpr)
    __device__ __forceinline__ static R cos_theta(const Carry& c, const Params&) {
        return c.cs * (R)(km().cos_off * km().inv_mpr) + c.sn * (R)(km().sin_off * km().inv_mpr);
    }
    __device__ __forceinline__ static R reward(const R o[4], const Carry& c, const Params& p) {
        // inverted_pendulum.py:139-142,174-177: (1 - cos theta) / 2 with cos theta = cs k1 + sn k2 (cos_theta below), the
        // halving and the sign folded into the two constants: two fused multiply-adds instead of five operations
        if (VARIANT >= 2)
            return fma_r(c.cs, (R)(-0.5 * km().cos_off * km().inv_mpr), fma_r(c.sn, (R)(-0.5 * km().sin_off * km().inv_mpr), R(0.5)));
        return R(1);  // :73-74,103-104
    }
    // the stateless entry points (emei_reward_io on float64 rows) keep the expression as the reference writes it
    __device__ __forceinline__ static R reward_exact(const R o[4], const Carry& c, const Params& p) {
        return VARIANT >= 2 ? (R(1) - cos_theta(c, p)) / R(2) : R(1);
    }
    __device__ __forceinline__ static bool terminal(const R o[4], const Carry& c, const Params& p) {
        return terminal(o, c, p, finite_r(o[1]));
    }
    // fin1: whether o[1] is finite (step() has it from wrap_pi for nothing in the usual case)
    __device__ __forceinline__ static bool terminal(const R o[4], const Carry& c, const Params& p, bool fin1) {
        bool inx = ((R)km().x_lo < o[0]) & (o[0] < (R)km().x_hi);
        // np.isfinite(obs).all(); where `inx` is part of the test it already fails for a NaN / infinite x
        bool fin = fin1 & finite_r(o[2]) & finite_r(o[3]);
        if (VARIANT == 0 || VARIANT == 2) fin &= finite_r(o[0]);
        R y = cos_theta(c, p);
        bool notdone;
        if (VARIANT == 0) notdone = (y >= R(0.9)) & fin;           // :76-79
        else if (VARIANT == 1) notdone = (y >= R(0)) & inx & fin;  // :106-111
        else if (VARIANT == 2) notdone = fin;                      // :144-146
        else notdone = inx & fin;                                  // :179-183
        return !notdone;
    }

    __device__ __forceinline__ static void step(R s[4], Carry& c, Action gu, const Params& p, int freq_rate,
                                                R o[4], R& rew, bool& term) {
        for (int k = 0; k < freq_rate; ++k) substep(s, c, gu, p);  // mujoco_env.py:88-97
        bool fin1;
        o[0] = s[0], o[1] = wrap_pi(s[1], fin1), o[2] = s[2], o[3] = s[3];  // obs_of
        rew = reward(o, c, p);
        term = terminal(o, c, p, fin1);
    }

    // device reset: init_qpos/qvel (zeros) + sigma * N(0,1) (mujoco_env.py:137-140); same draws as
    // body_kernels.h:gauss_state so both rollout paths of this env reset identically
    __device__ __forceinline__ static void init(R s[4], uint64_t seed, uint64_t env, uint32_t episode,
                                                const Params& p) {
        u32x4 r = philox4x32_10<kF64>(seed, env, episode, 0);  // seed: a kernel argument (wave-uniform)
        float z[4];
        boxmuller(r.v[0], r.v[1], z[0], z[1]);
        boxmuller(r.v[2], r.v[3], z[2], z[3]);
        // selects, not branches: stores to s[] behind a branch end up behind a pointer phi and s[] on the stack (scratch)
        const bool sh = p.noise_shared != 0;
        const float v0 = __fmul_rn(p.init_sigma[0], z[0]);
        const float v1 = sh ? v0 : __fmul_rn(p.init_sigma[1], z[1]);
        const float v2 = __fmul_rn(p.init_sigma[2], sh ? z[1] : z[2]);
        const float v3 = sh ? v2 : __fmul_rn(p.init_sigma[3], z[3]);
        s[0] = (R)v0, s[1] = (R)v1, s[2] = (R)v2, s[3] = (R)v3;
    }
};

}  // namespace emei
