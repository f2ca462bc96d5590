/*
 * emei_oracle.c — CPU restatement of the reference's env-step arithmetic.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; emei_amd/ never does.  It exists to check the
 * HIP kernels and to be timed as the CPU baseline ("port").
 *
 * Parity status
 *   - CartPole (classic control): PINNED.  Bit-compared against tests/golden/cartpole_golden.npz,
 *     which oracle/gen_golden.py produced by running the unmodified reference files
 *     (emei/envs/classic_control/{base_control,cartpole}.py) in the build container.
 *   - InvertedPendulum reward/terminal/wrap/euler-position/noise helpers: PINNED by
 *     tests/golden/mujoco_firstparty_golden.npz (same generator).
 *   - InvertedPendulum *dynamics* (MuJoCo mj_step): PARITY UNPINNED.  The arithmetic lives in the
 *     third-party `mujoco` package (requirements/main.txt:7, "mujoco >= 2.2.0", no lock, not
 *     vendored, absent from the image).  What is restated here is MuJoCo's published algorithm for
 *     this 2-DoF model (inertia-from-geom of the capsules in assets/inverted_pendulum.xml:12-23,
 *     CRBA/RNE closed form, explicit Euler on qvel, the soft joint-limit constraint of the
 *     "Computation" chapter) combined with emei's forward-Euler position override
 *     (emei/envs/mujoco/mujoco_env.py:86-109, 169-195).  The smooth equations of motion themselves
 *     (ip_accel without the limit) ARE pinned to the reference's own SymPy derivation
 *     (classic_control/auxiliary/lagrange_eqs.py) by tests/golden/lagrange_golden.npz.
 *
 * Every function cites the reference file:line it follows.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * CartPole.  Variants: 0 = CartPoleSwingUp, 1 = CartPoleBalancing.
 * Constants: emei/envs/classic_control/cartpole.py:22-31, SwingUp x_threshold :140.
 * ---------------------------------------------------------------------------------------- */
#define CP_GRAVITY 9.8
#define CP_MASS_CART 1.0
#define CP_MASS_POLE 0.1
#define CP_LENGTH 0.5
#define CP_FORCE_MAG 10.0

/* cartpole.py:48-60 — evaluated in double (Python floats, math.sin/cos, ** 2 via pow), then
 * rounded to float32 on return (np.array(..., dtype=np.float32)). */
static void cp_dsdt(const double y[5], float d[5]) {
    const double total_mass = CP_MASS_POLE + CP_MASS_CART; /* cartpole.py:25 */
    double x_dot = y[1], theta = y[2], theta_dot = y[3], force = y[4];
    double pole_mass_length = CP_MASS_POLE * CP_LENGTH;
    double cos_theta = cos(theta);
    double sin_theta = sin(theta);
    double temp = (force + pole_mass_length * pow(theta_dot, 2.0) * sin_theta) / total_mass;
    double theta_acc = (CP_GRAVITY * sin_theta - cos_theta * temp) /
                       (CP_LENGTH * (4.0 / 3.0 - CP_MASS_POLE * pow(cos_theta, 2.0) / total_mass));
    double x_acc = temp - pole_mass_length * theta_acc * cos_theta / total_mass;
    d[0] = (float)x_dot;
    d[1] = (float)x_acc;
    d[2] = (float)theta_dot;
    d[3] = (float)theta_acc;
    d[4] = 0.0f;
}

/* base_control.py:133-173, "euler" branch (:162-164):  y += derivs(y) * dt.
 * derivs(y) is float32, dt a Python float (weak scalar under NumPy 2 promotion), so the product is
 * rounded to float32; the in-place add promotes it back to the float64 accumulator. */
static void cp_ode_euler(double y[5], double dt, int steps) {
    const float dt32 = (float)dt;
    for (int s = 0; s < steps; ++s) {
        float d[5];
        cp_dsdt(y, d);
        for (int i = 0; i < 5; ++i) {
            volatile float inc = d[i] * dt32; /* volatile: forbid fusing/widening the f32 product */
            y[i] += (double)inc;
        }
    }
}

/* base_control.py:133-173, "rk4" branch (:165-170) — unreachable from step() (:73 never passes `method`), callable directly:
 *     k1 = np.asarray(derivs(y));  k2 = np.asarray(derivs(y + dt * k1 / 2));  k3 = ... k2 ...;  k4 = np.asarray(derivs(y + dt * k3))
 *     y += (k1 + 2 * k2 + 2 * k3 + k4) * dt / 6
 * NumPy 2 promotion, checked bit for bit against tests/golden/cartpole_rk4_golden.npz rather than reasoned out: the k's are
 * float32 arrays and dt, 2, 6 weak Python scalars, so `dt * k` and `/ 2` round to float32, the stage state `y + ...` is a
 * float64 sum of y and that float32 value, and the final combination is a chain of float32 roundings (2 k2; k1 + 2 k2;
 * + 2 k3; + k4; * float32(dt); / 6) added to the float64 accumulator. */
static void cp_ode_rk4(double y[5], double dt, int steps) {
    const float dt32 = (float)dt;
    for (int s = 0; s < steps; ++s) {
        float k1[5], k2[5], k3[5], k4[5];
        double ys[5];
        cp_dsdt(y, k1);
        for (int i = 0; i < 5; ++i) {
            volatile float h = dt32 * k1[i];
            volatile float h2 = h / 2.0f;
            ys[i] = y[i] + (double)h2;
        }
        cp_dsdt(ys, k2);
        for (int i = 0; i < 5; ++i) {
            volatile float h = dt32 * k2[i];
            volatile float h2 = h / 2.0f;
            ys[i] = y[i] + (double)h2;
        }
        cp_dsdt(ys, k3);
        for (int i = 0; i < 5; ++i) {
            volatile float h = dt32 * k3[i];
            ys[i] = y[i] + (double)h;
        }
        cp_dsdt(ys, k4);
        for (int i = 0; i < 5; ++i) {
            volatile float a = 2.0f * k2[i];
            volatile float b = k1[i] + a;
            volatile float c = 2.0f * k3[i];
            volatile float d = b + c;
            volatile float e = d + k4[i];
            volatile float f = e * dt32;
            volatile float g = f / 6.0f;
            y[i] += (double)g;
        }
    }
}

/* method: 0 = "euler" (what step() runs), 1 = "rk4" (ODE_approximation's other branch) */
static inline void cp_ode(double y[5], double dt, int steps, int method) {
    if (method == 1) cp_ode_rk4(y, dt, steps);
    else cp_ode_euler(y, dt, steps);
}

static inline double cp_x_threshold(int variant) { return variant == 0 ? 5.0 : 2.4; }
static inline double cp_theta_threshold(void) { return 12 * 2 * M_PI / 360; } /* cartpole.py:30 */

/* cartpole.py:145-147 (SwingUp), :124-126 (Balancing): terminal = not(notdone); NaN => terminal. */
static inline uint8_t cp_terminal(int variant, const double* obs) {
    int notdone;
    if (variant == 0)
        notdone = fabs(obs[0]) < 5.0;
    else
        notdone = (fabs(obs[2]) < cp_theta_threshold()) & (fabs(obs[0]) < 2.4);
    return (uint8_t)!notdone;
}

/* cartpole.py:149-151 (SwingUp): (cos(theta)+1)/2; :128-129 (Balancing): 1.0 */
static inline double cp_reward(int variant, const double* obs) {
    return variant == 0 ? (cos(obs[2]) + 1) / 2 : 1.0;
}

/* base_control.py:61-83 for a batch of independent envs.  state: [n,4] row-major float64, in/out.
 * action: {0,1}; force = +force_mag if action == 1 else -force_mag (cartpole.py:121-122,142-143). */
EXPORT void emei_oracle_cartpole_step_method(int variant, int64_t n, int freq_rate, double dt, int method, double* state,
                                             const int32_t* action, double* reward, uint8_t* terminal) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double y[5];
        memcpy(y, state + 4 * i, 4 * sizeof(double));
        y[4] = action[i] == 1 ? CP_FORCE_MAG : -CP_FORCE_MAG;
        cp_ode(y, dt, freq_rate, method);
        memcpy(state + 4 * i, y, 4 * sizeof(double));
        reward[i] = cp_reward(variant, y);
        terminal[i] = cp_terminal(variant, y);
    }
}
EXPORT void emei_oracle_cartpole_step(int variant, int64_t n, int freq_rate, double dt, double* state,
                                      const int32_t* action, double* reward, uint8_t* terminal) {
    emei_oracle_cartpole_step_method(variant, n, freq_rate, dt, 0, state, action, reward, terminal);
}

EXPORT void emei_oracle_cartpole_reward(int variant, int64_t n, const double* obs, double* reward) {
    for (int64_t i = 0; i < n; ++i) reward[i] = cp_reward(variant, obs + 4 * i);
}

EXPORT void emei_oracle_cartpole_terminal(int variant, int64_t n, const double* obs, uint8_t* terminal) {
    for (int64_t i = 0; i < n; ++i) terminal[i] = cp_terminal(variant, obs + 4 * i);
}

/* ------------------------------------------------------------------------------------------
 * float32 "port" of the same step: the arithmetic the HIP kernel performs (float32 state,
 * float32 derivative, unfused product-then-add), with libm sinf/cosf.  Used (a) to bound the
 * float32-vs-reference error on the CPU before going to the GPU and (b) as the SoA CPU baseline.
 * state: SoA float32 arrays x, xd, th, thd of length n.
 * ---------------------------------------------------------------------------------------- */
EXPORT void emei_oracle_cartpole_step_f32(int variant, int64_t n, int freq_rate, float dt, float* x, float* xd,
                                          float* th, float* thd, const int32_t* action, float* reward,
                                          uint8_t* terminal) {
    const float total_mass = (float)(CP_MASS_POLE + CP_MASS_CART);
    const float pml = (float)(CP_MASS_POLE * CP_LENGTH);
    const float th_thr = (float)cp_theta_threshold();
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        float X = x[i], XD = xd[i], TH = th[i], THD = thd[i];
        float force = action[i] == 1 ? (float)CP_FORCE_MAG : -(float)CP_FORCE_MAG;
        for (int s = 0; s < freq_rate; ++s) {
            float c = cosf(TH), sn = sinf(TH);
            float temp = (force + pml * (THD * THD) * sn) / total_mass;
            float thacc = ((float)CP_GRAVITY * sn - c * temp) /
                          ((float)CP_LENGTH * ((float)(4.0 / 3.0) - (float)CP_MASS_POLE * (c * c) / total_mass));
            float xacc = temp - pml * thacc * c / total_mass;
            volatile float i0 = XD * dt, i1 = xacc * dt, i2 = THD * dt, i3 = thacc * dt;
            X += i0;
            XD += i1;
            TH += i2;
            THD += i3;
        }
        x[i] = X, xd[i] = XD, th[i] = TH, thd[i] = THD;
        if (variant == 0) {
            reward[i] = (cosf(TH) + 1.0f) * 0.5f;
            terminal[i] = (uint8_t)!(fabsf(X) < 5.0f);
        } else {
            reward[i] = 1.0f;
            terminal[i] = (uint8_t)!((fabsf(TH) < th_thr) & (fabsf(X) < 2.4f));
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Counter-based RNG used by the engine's *device-side* reset (perf mode, auto-reset).  The
 * reference seeds PCG64 on the host (base_control.py:38-47; cartpole.py:153-156); that is what
 * parity mode uploads.  The device generator is the build's own specification (Philox4x32-10,
 * Salmon et al. 2011): key = 64-bit seed, counter = (env index lo, env index hi, episode, block).
 * The oracle restates it so auto-reset rollouts can be checked bit-for-bit.
 * ---------------------------------------------------------------------------------------- */
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0, c[1] = n1, c[2] = n2, c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

EXPORT void emei_oracle_philox(uint64_t seed, uint64_t env, uint32_t episode, uint32_t block, uint32_t out[4]) {
    uint32_t c[4] = {(uint32_t)env, (uint32_t)(env >> 32), episode, block};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    memcpy(out, c, sizeof(c));
}

static inline float u01(uint32_t r) { return (float)(r >> 8) * 0x1.0p-24f; } /* [0,1) on a 2^-24 grid */

/* U(-0.05, 0.05)^4, SwingUp adds pi to theta: the distribution of cartpole.py:131-132,153-156. */
EXPORT void emei_oracle_cartpole_init_f32(int variant, uint64_t seed, uint64_t env, uint32_t episode, float s[4]) {
    uint32_t r[4];
    emei_oracle_philox(seed, env, episode, 0, r);
    for (int i = 0; i < 4; ++i) s[i] = fmaf(0.1f, u01(r[i]), -0.05f); /* single rounding */
    if (variant == 0) s[2] += (float)M_PI;
}

/* The CPU twin of the fused device rollout WITH auto-reset (what bench.py times and tests/test_gpu_bench_shape.py checks):
 * T steps of base_control.py:61-83 per env with the float64 reference arithmetic above; TimeLimit (register_env.py:14-23:
 * `truncated` after max_steps steps of an episode, 0 = never); on done (terminal | truncated) the env is re-initialised from
 * the device reset generator's specification, cartpole_init_f32(seed, global env index, episode + 1).
 *   state [n,4] float64 in/out, steps [n] int32 in/out, episode [n] uint32 in/out, env_ids [n] global indices,
 *   actions [T,n] uint8; outputs as the device writes them: obs [T,n,4] float32 (the state after the step, before any reset),
 *   reward [T,n] float32, done [T,n] uint8 (bit 0 terminal, bit 1 truncated); any output may be NULL.
 * Parallel over blocks of 64 envs (no per-step fork / join). */
static void cartpole_rollout_autoreset_impl(int variant, int64_t n, int T, int freq_rate, double dt, int method, int max_steps, uint64_t seed,
                                            const int64_t* env_ids, double* state, int32_t* steps, uint32_t* episode,
                                            const uint8_t* actions, float* obs, float* reward, uint8_t* done) {
    /* blocks of envs (static over the threads), steps outermost inside a block: a block's state (64 envs x 32 B) stays in the
     * core's L1 and every output row is written in contiguous 64-env pieces */
    const int64_t BLK = 64, nblk = (n + BLK - 1) / BLK;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < nblk; ++b) {
        const int64_t lo = b * BLK, hi = lo + BLK < n ? lo + BLK : n;
        for (int t = 0; t < T; ++t) {
            for (int64_t i = lo; i < hi; ++i) {
                double y[5];
                memcpy(y, state + 4 * i, 4 * sizeof(double));
                y[4] = actions[(int64_t)t * n + i] == 1 ? CP_FORCE_MAG : -CP_FORCE_MAG;
                cp_ode(y, dt, freq_rate, method);
                const int32_t st = ++steps[i];
                const uint8_t d = (uint8_t)(cp_terminal(variant, y) | ((max_steps > 0 && st >= max_steps) ? 2 : 0));
                const int64_t row = (int64_t)t * n + i;
                if (obs) for (int k = 0; k < 4; ++k) obs[4 * row + k] = (float)y[k];
                if (reward) reward[row] = (float)cp_reward(variant, y);
                if (done) done[row] = d;
                if (d) {
                    float f[4];
                    steps[i] = 0;
                    emei_oracle_cartpole_init_f32(variant, seed, (uint64_t)(env_ids ? env_ids[i] : i), ++episode[i], f);
                    for (int k = 0; k < 4; ++k) y[k] = (double)f[k];
                }
                memcpy(state + 4 * i, y, 4 * sizeof(double));
            }
        }
    }
}

EXPORT void emei_oracle_cartpole_rollout_autoreset(int variant, int64_t n, int T, int freq_rate, double dt, int max_steps, uint64_t seed,
                                                   const int64_t* env_ids, double* state, int32_t* steps, uint32_t* episode,
                                                   const uint8_t* actions, float* obs, float* reward, uint8_t* done) {
    cartpole_rollout_autoreset_impl(variant, n, T, freq_rate, dt, 0, max_steps, seed, env_ids, state, steps, episode, actions, obs, reward, done);
}
EXPORT void emei_oracle_cartpole_rollout_autoreset_method(int variant, int64_t n, int T, int freq_rate, double dt, int method, int max_steps,
                                                          uint64_t seed, const int64_t* env_ids, double* state, int32_t* steps,
                                                          uint32_t* episode, const uint8_t* actions, float* obs, float* reward, uint8_t* done) {
    cartpole_rollout_autoreset_impl(variant, n, T, freq_rate, dt, method, max_steps, seed, env_ids, state, steps, episode, actions, obs, reward, done);
}

#include "integrators.h"
#define boxmuller oracle_boxmuller /* float32 Box-Muller of the device reset (integrators.h) */

/* ------------------------------------------------------------------------------------------
 * InvertedPendulum (MuJoCo-backed; dynamics parity UNPINNED — see header).
 * Variants: 0 ReboundBalancing, 1 BoundaryBalancing, 2 ReboundSwingUp, 3 BoundarySwingUp
 * (emei/envs/mujoco/inverted_pendulum.py:52-183).
 * Model constants from emei/envs/mujoco/assets/inverted_pendulum.xml:
 *   gravity 9.81 (:8); slider range [-2,2] (:14); cart capsule r=.1 half-length .1 (:15);
 *   hinge about +y, range +-90 deg (:17); pole capsule fromto (0,0,0)-(0.001,0,0.6) r=.049 (:18);
 *   motor gear 100, ctrlrange [-3,3] (:23); default geom density 1000 (MuJoCo default).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    double mc, mp, r, Icom, phi0; /* cart mass, pole mass, |com|, inertia about com (y), com angle */
    double g, gear, ctrl_lo, ctrl_hi, x_lo, x_hi;
    double invweight_slider; /* dof_invweight0 of the slider (at qpos0 of the compiled model) */
    double invweight_hinge;  /* dof_invweight0 of the hinge */
    double th_lo, th_hi;     /* hinge range (:17, -90 90 degrees; the SwingUp variants free it, inverted_pendulum.py:135-137) */
    /* soft joint-limit constraint: default solref (0.02, 1), solimp (0.9, 0.95, 0.001, 0.5, 2) */
    double timeconst, dampratio, dmin, dmax, width;
} ip_model_t;

static double capsule_mass(double rho, double r, double half) {
    return rho * (M_PI * r * r * 2 * half + 4.0 / 3.0 * M_PI * r * r * r);
}
/* inertia of a capsule about an axis through its centre, perpendicular to its long axis
 * (cylinder + two hemispherical caps; the closed form MuJoCo's compiler uses for inertiafromgeom) */
static double capsule_inertia_perp(double rho, double r, double half) {
    double h = 2 * half;
    double mcyl = rho * M_PI * r * r * h;
    double msph = rho * 4.0 / 3.0 * M_PI * r * r * r;
    return mcyl * (3 * r * r + h * h) / 12 + msph * (2 * r * r / 5 + h * h / 4 + 3 * h * r / 8);
}

EXPORT void emei_oracle_ip_model(ip_model_t* m) {
    const double rho = 1000.0;
    m->mc = capsule_mass(rho, 0.1, 0.1);
    double fx = 0.001, fz = 0.6;
    double len = sqrt(fx * fx + fz * fz);
    m->mp = capsule_mass(rho, 0.049, len / 2);
    m->Icom = capsule_inertia_perp(rho, 0.049, len / 2);
    m->r = len / 2;
    m->phi0 = atan2(fx, fz); /* com direction measured from +z towards +x (rotation about +y) */
    m->g = 9.81;
    m->gear = 100.0;
    m->ctrl_lo = -3.0, m->ctrl_hi = 3.0;
    m->x_lo = -2.0, m->x_hi = 2.0;
    m->timeconst = 0.02, m->dampratio = 1.0, m->dmin = 0.9, m->dmax = 0.95, m->width = 0.001;
    /* invweight0: (M^-1)_00 at qpos0 (theta = 0, pole upright in the compiled model) */
    double c = cos(m->phi0);
    double M11 = m->mc + m->mp, M12 = m->mp * m->r * c, M22 = m->Icom + m->mp * m->r * m->r;
    m->invweight_slider = M22 / (M11 * M22 - M12 * M12);
    m->invweight_hinge = M11 / (M11 * M22 - M12 * M12);
    m->th_lo = -M_PI / 2, m->th_hi = M_PI / 2;
}

EXPORT int emei_oracle_ip_model_size(void) { return (int)sizeof(ip_model_t); }

/* the oracle's InvertedPendulum constants in the layout of emei_model_constants (include/emei_hip.h), for
 * tests/test_model_constants.py, which pins them to the reference's XML */
EXPORT int emei_oracle_ip_xml_constants(double* out) {
    ip_model_t m;
    emei_oracle_ip_model(&m);
    const double v[17] = {m.g, m.mc, m.mp, m.Icom, m.r, m.phi0, m.gear, m.ctrl_lo, m.ctrl_hi, m.x_lo, m.x_hi, m.timeconst,
                          m.dmin, m.dmax, m.width, m.th_lo, m.th_hi};
    memcpy(out, v, sizeof(v));
    return 17;
}

static inline int ip_is_swingup(int variant) { return variant >= 2; }
static inline int ip_is_rebound(int variant) { return (variant & 1) == 0; }

/* Forward dynamics of the 2-DoF model: qacc at (q, v) with the actuator and the soft slider limit.
 * The integrators around it (mujoco_env.py:70-79,91-97) are in integrators.h. */
typedef struct { const ip_model_t* m; int variant; } ip_ctx_t;
static void ip_accel(const void* ctx, double dt, double hd, const double* q, const double* v, const double* ctrl_in, double* qacc) {
    const ip_model_t* m = ((const ip_ctx_t*)ctx)->m;
    const int variant = ((const ip_ctx_t*)ctx)->variant;
    const double u = ctrl_in[0];
    (void)hd; /* no joint damping in this model */
    /* SwingUp's _update_model (inverted_pendulum.py:135-137,170-172) turns the pole body by pi about y */
    double phi = q[1] + m->phi0 + (ip_is_swingup(variant) ? M_PI : 0.0);
    double s = sin(phi), c = cos(phi);
    double M11 = m->mc + m->mp, M12 = m->mp * m->r * c, M22 = m->Icom + m->mp * m->r * m->r;
    double ctrl = u < m->ctrl_lo ? m->ctrl_lo : (u > m->ctrl_hi ? m->ctrl_hi : u);
    double f1 = m->gear * ctrl + m->mp * m->r * s * v[1] * v[1];
    double f2 = m->mp * m->g * m->r * s;
    double det = M11 * M22 - M12 * M12;
    double a0 = (M22 * f1 - M12 * f2) / det; /* unconstrained ("smooth") accelerations */
    double a1 = (M11 * f2 - M12 * f1) / det;
    /* Soft joint-limit constraints (mjCNSTR_LIMIT_JOINT, margin 0, default solref / solimp): the slider's range (:14) for every
     * variant — Boundary variants terminate as soon as x leaves (x_lo, x_hi) (inverted_pendulum.py:106-111,179-183) but MuJoCo
     * still applies the limit force while the reference keeps stepping — and the hinge's range of +-90 degrees (:17; every
     * joint is `limited` by the file's default) for the Balancing variants, which only a POST-terminal state reaches (Rebound
     * terminates at cos theta < 0.9, Boundary at cos theta < 0); the SwingUp variants set the hinge range to +-inf.
     * Each row i: cost D_i / 2 min(0, J_i a - aref_i)^2 with D_i = 1 / R_i, R_i = (1 - imp) / imp dof_invweight0.  With at
     * most two rows the minimiser is found by enumerating the active sets of the 2 x 2 linear complementarity problem
     *   f >= 0,  (A + R) f - b >= 0,  f' ((A + R) f - b) = 0,   A = J M^-1 J',  b_i = aref_i - J_i a0
     * (A + R positive definite: exactly one of the four sets is consistent). */
    const double tc = m->timeconst < 2 * dt ? 2 * dt : m->timeconst; /* refsafe */
    const double K = 1.0 / (m->dmax * m->dmax * tc * tc * m->dampratio * m->dampratio), B = 2.0 / (m->dmax * tc);
    double Jr[2] = {0, 0}, b[2] = {0, 0}, Rr[2] = {1, 1}; /* row 0: slider, row 1: hinge; J = 0: the row does not exist */
    for (int i = 0; i < 2; ++i) {
        if (i == 1 && ip_is_swingup(variant)) continue;
        const double lo = i == 0 ? m->x_lo : m->th_lo, hi = i == 0 ? m->x_hi : m->th_hi;
        double dist;
        if (q[i] - lo < 0) dist = q[i] - lo, Jr[i] = 1.0;
        else if (hi - q[i] < 0) dist = hi - q[i], Jr[i] = -1.0;
        else continue;
        const double xx = fabs(dist) / m->width;
        const double y = xx >= 1 ? 1.0 : (xx <= 0.5 ? 2 * xx * xx : 1 - 2 * (1 - xx) * (1 - xx));
        const double imp = m->dmin + y * (m->dmax - m->dmin);
        const double aref = -B * (Jr[i] * v[i]) - K * imp * dist;
        b[i] = aref - Jr[i] * (i == 0 ? a0 : a1);
        Rr[i] = (1 - imp) / imp * (i == 0 ? m->invweight_slider : m->invweight_hinge);
    }
    if (Jr[0] != 0.0 || Jr[1] != 0.0) {
        const double A00 = M22 / det, A11 = M11 / det, A01 = Jr[0] * Jr[1] * (-M12 / det);
        const double H00 = A00 + Rr[0], H11 = A11 + Rr[1];
        double f0 = 0, f1 = 0;
        const int has0 = Jr[0] != 0.0, has1 = Jr[1] != 0.0;
        const double s0 = b[0] / H00, s1 = b[1] / H11; /* single-row solutions */
        const double d2 = H00 * H11 - A01 * A01;
        const double t0 = (H11 * b[0] - A01 * b[1]) / d2, t1 = (H00 * b[1] - A01 * b[0]) / d2; /* both rows */
        if (has0 && has1 && t0 > 0 && t1 > 0) f0 = t0, f1 = t1;
        else if (has0 && s0 > 0 && !(has1 && b[1] - A01 * s0 > 0)) f0 = s0;
        else if (has1 && s1 > 0 && !(has0 && b[0] - A01 * s1 > 0)) f1 = s1;
        /* generalised force J' f through M^-1 */
        const double g0 = Jr[0] * f0, g1 = Jr[1] * f1;
        a0 += (M22 * g0 - M12 * g1) / det;
        a1 += (M11 * g1 - M12 * g0) / det;
    }
    qacc[0] = a0, qacc[1] = a1;
}

/* inverted_pendulum.py:45-49: theta_obs = (theta + pi) % (2 pi) - pi  (Python/NumPy floored mod) */
static inline double ip_wrap(double th) {
    double a = th + M_PI, p = 2 * M_PI;
    double mod = fmod(a, p);
    if (mod != 0 && ((mod < 0) != (p < 0))) mod += p;
    return mod - M_PI;
}

static inline int ip_finite4(const double* o) { return isfinite(o[0]) && isfinite(o[1]) && isfinite(o[2]) && isfinite(o[3]); }

/* inverted_pendulum.py:73-79,103-111,139-146,174-183 */
static inline double ip_reward(int variant, const double* obs) {
    return ip_is_swingup(variant) ? (1 - cos(obs[1])) / 2 : 1.0;
}
static inline uint8_t ip_terminal(const ip_model_t* m, int variant, const double* obs) {
    double y = cos(obs[1]), x = obs[0];
    int fin = ip_finite4(obs);
    int inx = (m->x_lo < x) && (x < m->x_hi);
    int notdone;
    switch (variant) {
        case 0: notdone = (y >= 0.9) & fin; break;
        case 1: notdone = (y >= 0) & inx & fin; break;
        case 2: notdone = fin; break;
        default: notdone = inx & fin; break;
    }
    return (uint8_t)!notdone;
}

EXPORT void emei_oracle_ip_reward(int variant, int64_t n, const double* obs, double* reward) {
    for (int64_t i = 0; i < n; ++i) reward[i] = ip_reward(variant, obs + 4 * i);
}
EXPORT void emei_oracle_ip_terminal(int variant, int64_t n, const double* obs, uint8_t* terminal) {
    ip_model_t m;
    emei_oracle_ip_model(&m);
    for (int64_t i = 0; i < n; ++i) terminal[i] = ip_terminal(&m, variant, obs + 4 * i);
}
EXPORT void emei_oracle_ip_wrap(int64_t n, const double* th, double* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = ip_wrap(th[i]);
}

/* mujoco_env.py:157-167 for a batch.  state [n,4] = (x, theta, v, omega) with theta UNWRAPPED (the
 * internal qpos, inverted_pendulum.py:45-49 wraps only the observation); obs [n,4] is written
 * with the wrapped angle; action [n] float64. */
EXPORT void emei_oracle_ip_step_ex(int variant, int64_t n, int freq_rate, double dt, double* state, const double* action,
                                   double* obs, double* reward, uint8_t* terminal, const oracle_opts_t* opts) {
    ip_model_t m;
    emei_oracle_ip_model(&m);
    ip_ctx_t ctx = {&m, variant};
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double q[2] = {state[4 * i], state[4 * i + 1]}, v[2] = {state[4 * i + 2], state[4 * i + 3]};
        oracle_env_step(ip_accel, &ctx, 2, freq_rate, dt, opts, i, q, v, action + i);
        state[4 * i] = q[0], state[4 * i + 1] = q[1], state[4 * i + 2] = v[0], state[4 * i + 3] = v[1];
        double o[4] = {q[0], ip_wrap(q[1]), v[0], v[1]};
        memcpy(obs + 4 * i, o, sizeof(o));
        reward[i] = ip_reward(variant, o);
        terminal[i] = ip_terminal(&m, variant, o);
    }
}

/* T env-steps in one call (bench.py's cpu_baseline: no Python and no fork / join between steps): blocks of 64 envs over the
 * threads, steps outermost inside a block; float32 actions [T,n] in, the outputs the device writes per env-step out — obs
 * float32 [T,n,4], reward float32 [T,n], terminal uint8 [T,n] (any may be NULL).  No reset: the reference's step() keeps
 * integrating after `terminal` (mujoco_env.py:157-167). */
EXPORT void emei_oracle_ip_rollout(int variant, int64_t n, int T, int freq_rate, double dt, double* state, const float* actions,
                                   float* obs, float* reward, uint8_t* terminal, const oracle_opts_t* opts) {
    ip_model_t m;
    emei_oracle_ip_model(&m);
    ip_ctx_t ctx = {&m, variant};
    const int64_t BLK = 64, nblk = (n + BLK - 1) / BLK;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < nblk; ++b) {
        const int64_t lo = b * BLK, hi = lo + BLK < n ? lo + BLK : n;
        for (int t = 0; t < T; ++t)
            for (int64_t i = lo; i < hi; ++i) {
                double q[2] = {state[4 * i], state[4 * i + 1]}, v[2] = {state[4 * i + 2], state[4 * i + 3]};
                const double a = (double)actions[(int64_t)t * n + i];
                oracle_env_step(ip_accel, &ctx, 2, freq_rate, dt, opts, i, q, v, &a);
                state[4 * i] = q[0], state[4 * i + 1] = q[1], state[4 * i + 2] = v[0], state[4 * i + 3] = v[1];
                const double o[4] = {q[0], ip_wrap(q[1]), v[0], v[1]};
                const int64_t row = (int64_t)t * n + i;
                if (obs) for (int k = 0; k < 4; ++k) obs[4 * row + k] = (float)o[k];
                if (reward) reward[row] = (float)ip_reward(variant, o);
                if (terminal) terminal[row] = ip_terminal(&m, variant, o);
            }
    }
}

/* The same smooth dynamics with caller-supplied parameters (cart mass, pole mass, com distance, inertia about
 * the com, gravity), a plain force instead of gear * clipped ctrl and no slider limit: used by the tests to
 * compare the equations of motion with the reference's own SymPy derivation (auxiliary/lagrange_eqs.py). */
EXPORT void emei_oracle_ip_accel_custom(double mc, double mp, double r, double Icom, double g, const double* q,
                                        const double* v, double force, double* acc) {
    ip_model_t m;
    memset(&m, 0, sizeof(m));
    m.mc = mc, m.mp = mp, m.r = r, m.Icom = Icom, m.phi0 = 0.0, m.g = g;
    m.gear = 1.0, m.ctrl_lo = -INFINITY, m.ctrl_hi = INFINITY, m.x_lo = -INFINITY, m.x_hi = INFINITY;
    m.timeconst = 0.02, m.dampratio = 1.0, m.dmin = 0.9, m.dmax = 0.95, m.width = 0.001, m.invweight_slider = 1.0;
    m.invweight_hinge = 1.0, m.th_lo = -INFINITY, m.th_hi = INFINITY; /* no limit row: the smooth equations only */
    ip_ctx_t ctx = {&m, 0};
    ip_accel(&ctx, 0.02, 0.0, q, v, &force, acc);
}

/* forward dynamics of one state with the model's own constants and limits: (x'', theta'') — for the tests that check the
 * limit rows through their optimality conditions */
EXPORT void emei_oracle_ip_accel(int variant, double dt, const double* q, const double* v, double ctrl, double* acc) {
    ip_model_t m;
    emei_oracle_ip_model(&m);
    ip_ctx_t ctx = {&m, variant};
    ip_accel(&ctx, dt, 0.0, q, v, &ctrl, acc);
}

EXPORT void emei_oracle_ip_step(int variant, int64_t n, int freq_rate, double dt, double* state, const double* action,
                                double* obs, double* reward, uint8_t* terminal) {
    emei_oracle_ip_step_ex(variant, n, freq_rate, dt, state, action, obs, reward, terminal, NULL);
}

/* Device-side init of any MuJoCo-backed body: init_qpos/qvel = 0 + Gaussian noise, layout per `shared`
 * (body_kernels.h:body_init).  s = (q[nv], v[nv]) float64 holding float32-rounded draws. */
EXPORT void emei_oracle_body_init(uint64_t seed, uint64_t env, uint32_t episode, int nv, float sigma_pos, float sigma_vel,
                                  int shared, double* s) {
    oracle_gauss_state(seed, env, episode, 0u, nv, sigma_pos, sigma_vel, shared, 1, s, s + nv);
}

/* Device-side init for InvertedPendulum: zeros + sigma * N(0,1) i.i.d. per coordinate (the
 * distribution SURVEY 8d config 3 prescribes; the reference's row-slicing quirk, mujoco_env.py:243-244,
 * is reproduced on the host path instead). */
EXPORT void emei_oracle_ip_init_f32(uint64_t seed, uint64_t env, uint32_t episode, float sigma, float s[4]) {
    uint32_t r[4];
    emei_oracle_philox(seed, env, episode, 0, r);
    float z[4];
    boxmuller(r[0], r[1], &z[0], &z[1]);
    boxmuller(r[2], r[3], &z[2], &z[3]);
    for (int i = 0; i < 4; ++i) s[i] = sigma * z[i];
}
