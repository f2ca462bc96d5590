/*
 * dpend_oracle.c — CPU restatement of the InvertedDoublePendulum (float64).
 * TEST INFRASTRUCTURE, NOT PRODUCT (same rules as emei_oracle.c).
 *
 * Reference: emei/envs/mujoco/inverted_double_pendulum.py on emei/envs/mujoco/mujoco_env.py, model
 * emei/envs/mujoco/assets/inverted_double_pendulum.xml.  Dynamics parity with libmujoco is
 * UNPINNED (mujoco >= 2.2.0 is absent from the image); the first-party pieces — the quirky
 * observation "wrap" (:59), the four reward/terminal functions (:84-196) — are pinned by
 * tests/golden/dpend_firstparty_golden.npz, and the smooth equations of motion (dp_dynamics) are pinned
 * to the reference's own SymPy derivation (auxiliary/lagrange_eqs.py) by tests/golden/lagrange_golden.npz
 * (tests/test_oracle_lagrange.py); capsule inertias, the soft limit and the integrators stay unpinned.
 *
 * Formulation (deliberately different from the HIP kernel's absolute-angle closed form): joint
 * coordinates q = (x, th1, th2); inertia matrix M = sum_b m_b Jc_b^T Jc_b + I_b Jw_b^T Jw_b from the
 * explicit centre-of-mass Jacobians; bias = sum_b m_b Jc_b^T (d/dt Jc_b) qd + gravity; dense 3x3 solve
 * by Cramer's rule; MuJoCo's soft joint-limit constraint on the slider (range +-3, margin 0.01,
 * default solref/solimp, regulariser from dof_invweight0 at qpos0); emei's forward-Euler position
 * override (mujoco_env.py:94-97,189-191).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "integrators.h"

#define EXPORT __attribute__((visibility("default")))

typedef struct {
    double mc, mp, Ip, lc, L1, gx, gz, gear, x_lo, x_hi, margin, invw, tc, dmin, dmax, width;
    double ctrl_lo, ctrl_hi; /* motor ctrlrange (xml:45): the clamp dp_accel applies */
} dp_model_t;

static double capsule_mass(double rho, double r, double half) { return rho * (M_PI * r * r * 2 * half + 4.0 / 3.0 * M_PI * r * r * r); }
static double capsule_inertia_perp(double rho, double r, double half) {
    double h = 2 * half, mcyl = rho * M_PI * r * r * h, msph = rho * 4.0 / 3.0 * M_PI * r * r * r;
    return mcyl * (3 * r * r + h * h) / 12 + msph * (2 * r * r / 5 + h * h / 4 + 3 * h * r / 8);
}

/* joint-space inertia and bias at (q, qd); phi1 = th1 + off, phi2 = phi1 + th2, angles from +z towards +x */
static void dp_dynamics(const dp_model_t* m, double off, const double* q, const double* qd, double M[3][3], double bias[3]) {
    double p1 = q[1] + off, p2 = p1 + q[2];
    double s1 = sin(p1), c1 = cos(p1), s2 = sin(p2), c2 = cos(p2);
    double w1 = qd[1], w2 = qd[1] + qd[2];
    /* com Jacobians (rows x,z; columns x, th1, th2) */
    double J1[2][3] = {{1, m->lc * c1, 0}, {0, -m->lc * s1, 0}};
    double J2[2][3] = {{1, m->L1 * c1 + m->lc * c2, m->lc * c2}, {0, -m->L1 * s1 - m->lc * s2, -m->lc * s2}};
    double Jw1[3] = {0, 1, 0}, Jw2[3] = {0, 1, 1};
    /* velocity-product accelerations of the coms: -w^2 * (link vectors) */
    double a1[2] = {-w1 * w1 * m->lc * s1, -w1 * w1 * m->lc * c1};
    double a2[2] = {-w1 * w1 * m->L1 * s1 - w2 * w2 * m->lc * s2, -w1 * w1 * m->L1 * c1 - w2 * w2 * m->lc * c2};
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            M[i][j] = m->mp * (J1[0][i] * J1[0][j] + J1[1][i] * J1[1][j]) + m->Ip * Jw1[i] * Jw1[j] +
                      m->mp * (J2[0][i] * J2[0][j] + J2[1][i] * J2[1][j]) + m->Ip * Jw2[i] * Jw2[j];
        }
        /* bias = J^T m (a_bias - g), g = (gx, -gz) */
        bias[i] = m->mp * (J1[0][i] * (a1[0] - m->gx) + J1[1][i] * (a1[1] + m->gz)) +
                  m->mp * (J2[0][i] * (a2[0] - m->gx) + J2[1][i] * (a2[1] + m->gz));
    }
    M[0][0] += m->mc;
    bias[0] += m->mc * (0.0 - m->gx);
}

static void solve3(double M[3][3], const double* r, double* x) {
    double a = M[0][0], b = M[0][1], c = M[0][2], d = M[1][1], e = M[1][2], f = M[2][2];
    double det = a * (d * f - e * e) - b * (b * f - e * c) + c * (b * e - d * c);
    x[0] = (r[0] * (d * f - e * e) - b * (r[1] * f - e * r[2]) + c * (r[1] * e - d * r[2])) / det;
    x[1] = (a * (r[1] * f - e * r[2]) - r[0] * (b * f - e * c) + c * (b * r[2] - r[1] * c)) / det;
    x[2] = (a * (d * r[2] - r[1] * e) - b * (b * r[2] - r[1] * c) + r[0] * (b * e - d * c)) / det;
}

EXPORT void dpend_oracle_model(dp_model_t* m, double dt) {
    const double rho = 1000.0;
    m->mc = capsule_mass(rho, 0.1, 0.1);          /* xml:32 */
    m->mp = capsule_mass(rho, 0.045, 0.3);        /* xml:35,38 */
    m->Ip = capsule_inertia_perp(rho, 0.045, 0.3);
    m->lc = 0.3, m->L1 = 0.6;                     /* xml:35-36 */
    m->gx = 1e-5, m->gz = 9.81;                   /* xml:26 */
    m->gear = 500.0, m->ctrl_lo = -1, m->ctrl_hi = 1; /* xml:45 */
    m->x_lo = -3, m->x_hi = 3, m->margin = 0.01;  /* xml:31 */
    m->tc = 0.02 < 2 * dt ? 2 * dt : 0.02;        /* default solref timeconst, refsafe */
    m->dmin = 0.9, m->dmax = 0.95, m->width = 0.001;
    double M[3][3], bias[3], q0[3] = {0, 0, 0}, z[3] = {0, 0, 0}, ex[3] = {1, 0, 0}, w[3];
    dp_dynamics(m, 0.0, q0, z, M, bias);
    solve3(M, ex, w);
    m->invw = w[0]; /* dof_invweight0 of the slider at qpos0 (compiled model: poles upright) */
}

/* the oracle's constants in the layout of emei_model_constants (include/emei_hip.h); tests/test_model_constants.py */
EXPORT int dpend_oracle_xml_constants(double* out) {
    dp_model_t m;
    dpend_oracle_model(&m, 0.002);
    const double v[17] = {m.gx, m.gz, m.mc, m.mp, m.Ip, m.lc, m.L1, m.gear, m.ctrl_lo, m.ctrl_hi /* the fields dp_accel clamps with */, m.x_lo, m.x_hi,
                          m.margin, m.tc /* at dt = 0.002: the unclamped default */, m.dmin, m.dmax, m.width};
    memcpy(out, v, sizeof(v));
    return 17;
}

/* dof_invweight0 of the slider at qpos0, for the comparison with the kernels' closed form (tests/test_oracle_solver.py) */
EXPORT double dpend_oracle_invweight(void) {
    dp_model_t m;
    dpend_oracle_model(&m, 0.002);
    return m.invw;
}

typedef struct { const dp_model_t* m; double off; } dp_ctx_t;
/* forward dynamics (no joint damping: hd unused); integrators in integrators.h */
static void dp_accel(const void* ctx, double dt, double hd, const double* q, const double* v, const double* ctrl_in, double* acc) {
    const dp_model_t* m = ((const dp_ctx_t*)ctx)->m;
    const double off = ((const dp_ctx_t*)ctx)->off, u = ctrl_in[0];
    (void)dt, (void)hd;
    double M[3][3], bias[3], rhs[3];
    dp_dynamics(m, off, q, v, M, bias);
    double ctrl = u < m->ctrl_lo ? m->ctrl_lo : (u > m->ctrl_hi ? m->ctrl_hi : u);
    rhs[0] = m->gear * ctrl - bias[0], rhs[1] = -bias[1], rhs[2] = -bias[2];
    solve3(M, rhs, acc);
    double dist = 0, J = 0;
    if (q[0] - m->x_lo < m->margin) dist = q[0] - m->x_lo, J = 1;
    else if (m->x_hi - q[0] < m->margin) dist = m->x_hi - q[0], J = -1;
    if (J != 0) {
        double ex[3] = {1, 0, 0}, w[3];
        solve3(M, ex, w);
        double pos = dist - m->margin;
        double xx = fabs(pos) / m->width;
        double y = xx >= 1 ? 1.0 : (xx <= 0.5 ? 2 * xx * xx : 1 - 2 * (1 - xx) * (1 - xx));
        double imp = m->dmin + y * (m->dmax - m->dmin);
        double K = 1 / (m->dmax * m->dmax * m->tc * m->tc), B = 2 / (m->dmax * m->tc);
        double aref = -B * (J * v[0]) - K * imp * pos;
        double R = (1 - imp) / imp * m->invw;
        double force = (aref - J * acc[0]) / (w[0] + R);
        if (force > 0) for (int i = 0; i < 3; ++i) acc[i] += w[i] * J * force;
    }
}

/* inverted_double_pendulum.py:59 (sic): (theta + pi) % 2 * pi - pi, Python floored modulo */
static double quirk_wrap(double th) {
    double a = th + M_PI, mod = fmod(a, 2.0);
    if (mod != 0 && mod < 0) mod += 2.0;
    return mod * M_PI - M_PI;
}

static void dp_reward_terminal(int variant, const double* o, double* rew, uint8_t* term) {
    double y = cos(o[1]) + cos(o[1] + o[2]);
    int fin = 1;
    for (int i = 0; i < 6; ++i) fin &= isfinite(o[i]) != 0;
    int inx = (-3.0 < o[0]) && (o[0] < 3.0);
    switch (variant) {
        case 0: *rew = 1.0, *term = (uint8_t)!((y >= 1.5) & fin); break;       /* :84-90 */
        case 1: *rew = 1.0, *term = (uint8_t)!((y >= 0) & inx & fin); break;   /* :114-123 */
        case 2: *rew = (2 - y) / 4, *term = (uint8_t)!fin; break;              /* :144-153 */
        default: *rew = (2 - y) / 4 - (5e-3 * o[4] * o[4] + 1e-4 * o[5] * o[5]), *term = (uint8_t)!(inx & fin); /* :183-196 */
    }
}

/* mujoco_env.py:157-167 for a batch: state [n,6] = (x, th1, th2, v, w1, w2) in/out; obs [n,6] with the wrap */
EXPORT void dpend_oracle_step_ex(int variant, int64_t n, int freq_rate, double dt, double* state, const double* action,
                                 double* obs, double* reward, uint8_t* terminal, const oracle_opts_t* opts) {
    dp_model_t m;
    dpend_oracle_model(&m, dt);
    dp_ctx_t ctx = {&m, variant >= 2 ? M_PI : 0.0}; /* _update_model: pole body turned by pi (:139-141,170-172) */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double* s = state + 6 * i;
        oracle_env_step(dp_accel, &ctx, 3, freq_rate, dt, opts, i, s, s + 3, action + i);
        double* o = obs + 6 * i;
        o[0] = s[0], o[1] = quirk_wrap(s[1]), o[2] = quirk_wrap(s[2]), o[3] = s[3], o[4] = s[4], o[5] = s[5];
        dp_reward_terminal(variant, o, reward + i, terminal + i);
    }
}

/* T env-steps in one call (bench.py's cpu_baseline; see emei_oracle_ip_rollout): float32 actions [T,n], outputs obs float32
 * [T,n,6], reward float32 [T,n], terminal uint8 [T,n] (any may be NULL); no reset */
EXPORT void dpend_oracle_rollout(int variant, int64_t n, int T, int freq_rate, double dt, double* state, const float* actions,
                                 float* obs, float* reward, uint8_t* terminal, const oracle_opts_t* opts) {
    dp_model_t m;
    dpend_oracle_model(&m, dt);
    dp_ctx_t ctx = {&m, variant >= 2 ? M_PI : 0.0};
    const int64_t BLK = 64, nblk = (n + BLK - 1) / BLK;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < nblk; ++b) {
        const int64_t lo = b * BLK, hi = lo + BLK < n ? lo + BLK : n;
        for (int t = 0; t < T; ++t)
            for (int64_t i = lo; i < hi; ++i) {
                double* s = state + 6 * i;
                const double a = (double)actions[(int64_t)t * n + i];
                oracle_env_step(dp_accel, &ctx, 3, freq_rate, dt, opts, i, s, s + 3, &a);
                const double o[6] = {s[0], quirk_wrap(s[1]), quirk_wrap(s[2]), s[3], s[4], s[5]};
                double r;
                uint8_t d;
                dp_reward_terminal(variant, o, &r, &d);
                const int64_t row = (int64_t)t * n + i;
                if (obs) for (int k = 0; k < 6; ++k) obs[6 * row + k] = (float)o[k];
                if (reward) reward[row] = (float)r;
                if (terminal) terminal[row] = d;
            }
    }
}

EXPORT void dpend_oracle_step(int variant, int64_t n, int freq_rate, double dt, double* state, const double* action,
                              double* obs, double* reward, uint8_t* terminal) {
    dpend_oracle_step_ex(variant, n, freq_rate, dt, state, action, obs, reward, terminal, NULL);
}

/* The same smooth dynamics with caller-supplied parameters (cart mass, pole mass, pole inertia about its
 * com, com distance, pole-1 length, gravity), generalized forces on (x, theta1, theta2) instead of the
 * actuator, no limit: used by the tests to compare
 * the equations of motion with the reference's own SymPy derivation (auxiliary/lagrange_eqs.py). */
EXPORT void dpend_oracle_accel_custom(double mc, double mp, double Ip, double lc, double L1, double g, const double* q,
                                      const double* v, const double* gen_force, double* acc) {
    dp_model_t m;
    memset(&m, 0, sizeof(m));
    m.mc = mc, m.mp = mp, m.Ip = Ip, m.lc = lc, m.L1 = L1, m.gx = 0.0, m.gz = g;
    double M[3][3], bias[3], rhs[3];
    dp_dynamics(&m, 0.0, q, v, M, bias);
    rhs[0] = gen_force[0] - bias[0], rhs[1] = gen_force[1] - bias[1], rhs[2] = gen_force[2] - bias[2];
    solve3(M, rhs, acc);
}

EXPORT void dpend_oracle_reward_terminal(int variant, int64_t n, const double* obs, double* reward, uint8_t* terminal) {
    for (int64_t i = 0; i < n; ++i) dp_reward_terminal(variant, obs + 6 * i, reward + i, terminal + i);
}
EXPORT void dpend_oracle_wrap(int64_t n, const double* th, double* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = quirk_wrap(th[i]);
}
/* total mechanical energy without the cart push (for a conservation check of the restated dynamics) */
EXPORT double dpend_oracle_energy(int variant, const double* s) {
    dp_model_t m;
    dpend_oracle_model(&m, 0.02);
    const double off = variant >= 2 ? M_PI : 0.0;
    double M[3][3], bias[3];
    dp_dynamics(&m, off, s, s + 3, M, bias);
    double T = 0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T += 0.5 * s[3 + i] * M[i][j] * s[3 + j];
    double p1 = s[1] + off, p2 = p1 + s[2];
    double z1 = m.lc * cos(p1), z2 = m.L1 * cos(p1) + m.lc * cos(p2);
    double x1 = s[0] + m.lc * sin(p1), x2 = s[0] + m.L1 * sin(p1) + m.lc * sin(p2);
    double U = m.mp * m.gz * (z1 + z2) - m.gx * (m.mc * s[0] + m.mp * (x1 + x2));
    return T + U;
}
