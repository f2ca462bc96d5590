/*
 * integrators.h — the substep of emei/envs/mujoco/mujoco_env.py:86-104 around a body's forward
 * dynamics, shared by the CPU oracles of the MuJoCo-backed bodies.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (same rules as emei_oracle.c).  PARITY UNPINNED like the body
 * models themselves: the integrators are MuJoCo's (third-party `mujoco >= 2.2.0`, absent from the
 * image), restated from its published pipeline:
 *   euler               mj_Euler velocity update, then emei's position override q += dt * v_old
 *                       (mujoco_env.py:94-97, get_euler_pos :169-195)
 *   semi_implicit_euler mj_Euler as it is: v' = v + dt*qacc, q += dt*v'   (mujoco_env.py:74-75)
 *   rk4                 mj_RungeKutta(N = 4): classic tableau, F(X) = (v, qacc(q, v)) from the full
 *                       forward dynamics at every stage, no implicit damping   (mujoco_env.py:76-77)
 * and the observation noise of mujoco_env.py:98-104 (added to qpos/qvel after EVERY substep) drawn
 * from the device's counter-based stream so that kernel and oracle can be compared draw for draw.
 */
#ifndef EMEI_ORACLE_INTEGRATORS_H
#define EMEI_ORACLE_INTEGRATORS_H
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ORACLE_NV_MAX 16
enum { ORACLE_EULER = 0, ORACLE_SEMI_IMPLICIT = 1, ORACLE_RK4 = 2 };

/* options of the *_step_ex entry points; all zero = the default path */
typedef struct {
    int32_t integrator; /* ORACLE_* */
    int32_t shared;     /* noise layout: 1 = one draw for all of qpos, one for all of qvel (B = 1 quirk) */
    float obs_pos, obs_vel;
    uint64_t seed, env_offset;
    uint32_t episode, step_index; /* counter words of the noise stream (same for every env of the call) */
    int32_t solver;               /* constraint solver of the planar bodies: ORACLE_SOLVER_* (0 = MuJoCo's: converged Newton) */
    int32_t reserved;
} oracle_opts_t;
enum { ORACLE_SOLVER_NEWTON = 0, ORACLE_SOLVER_SWEEP1 = 1 };

/* forward dynamics: qacc at (q, v); hd = dt when joint damping is integrated implicitly, else 0 */
typedef void (*oracle_accel_fn)(const void* ctx, double dt, double hd, const double* q, const double* v, const double* ctrl,
                                double* qacc);

void emei_oracle_philox(uint64_t seed, uint64_t env, uint32_t episode, uint32_t block, uint32_t out[4]);

/* Box-Muller pair from two Philox words (emei_device.h:boxmuller).  The SPECIFICATION is the exact value, rounded once to
 * float32: u1 = ((a >> 8) + 1) / 2^24 in (0, 1], the angle t = (b >> 8) / 2^24 turns, z = sqrt(-2 ln u1) (cos, sin)(2 pi t),
 * evaluated here in float64.  The device uses the hardware float32 transcendentals on the same exact inputs and stays within
 * 1.3e-7 (sin / cos) and 4.9e-7 (radius) of it over all 2^24 inputs (tools/bm_accuracy.hip). */
static inline void oracle_boxmuller(uint32_t a, uint32_t b, float* z0, float* z1) {
    const double u1 = ((double)(a >> 8) + 1.0) * 0x1.0p-24;
    const double ang = 6.283185307179586476925 * ((double)(b >> 8) * 0x1.0p-24);
    const double rad = sqrt(-2.0 * log(u1));
    *z0 = (float)(rad * cos(ang));
    *z1 = (float)(rad * sin(ang));
}

/* s (+)= sigma * N(0,1) over (q[nv], v[nv]); draws as body_kernels.h:gauss_state */
static inline void oracle_gauss_state(uint64_t key, uint64_t env, uint32_t episode, uint32_t blk0, int nv, float sp, float sv,
                                      int shared, int assign, double* q, double* v) {
    uint32_t r[4];
    if (shared) {
        float z0, z1;
        emei_oracle_philox(key, env, episode, blk0, r);
        oracle_boxmuller(r[0], r[1], &z0, &z1);
        for (int i = 0; i < nv; ++i) q[i] = (assign ? 0.0 : q[i]) + (double)(sp * z0), v[i] = (assign ? 0.0 : v[i]) + (double)(sv * z1);
        return;
    }
    float z[2 * ORACLE_NV_MAX + 4];
    int nb = (2 * nv + 3) / 4;
    for (int b = 0; b < nb; ++b) {
        emei_oracle_philox(key, env, episode, blk0 + (uint32_t)b, r);
        oracle_boxmuller(r[0], r[1], &z[4 * b], &z[4 * b + 1]);
        oracle_boxmuller(r[2], r[3], &z[4 * b + 2], &z[4 * b + 3]);
    }
    for (int i = 0; i < nv; ++i) {
        double dq = (double)(sp * z[i]), dv = (double)(sv * z[nv + i]);
        q[i] = assign ? dq : q[i] + dq;
        v[i] = assign ? dv : v[i] + dv;
    }
}
#define ORACLE_OBS_NOISE_KEY 0x6F62736E6F697365ull

static inline void oracle_substep(oracle_accel_fn f, const void* ctx, int nv, int integrator, double dt, double* q, double* v,
                                  const double* ctrl) {
    double a[ORACLE_NV_MAX];
    if (integrator != ORACLE_RK4) {
        f(ctx, dt, dt, q, v, ctrl, a);
        for (int i = 0; i < nv; ++i) {
            double vn = v[i] + dt * a[i];
            q[i] += dt * (integrator == ORACLE_EULER ? v[i] : vn);
            v[i] = vn;
        }
        return;
    }
    /* classic RK4 on X = (q, v) */
    double k_q[4][ORACLE_NV_MAX], k_v[4][ORACLE_NV_MAX], qs[ORACLE_NV_MAX], vs[ORACLE_NV_MAX];
    static const double c[4] = {0.0, 0.5, 0.5, 1.0};
    for (int st = 0; st < 4; ++st) {
        for (int i = 0; i < nv; ++i) {
            qs[i] = st ? q[i] + dt * c[st] * k_q[st - 1][i] : q[i];
            vs[i] = st ? v[i] + dt * c[st] * k_v[st - 1][i] : v[i];
        }
        f(ctx, dt, 0.0, qs, vs, ctrl, a);
        for (int i = 0; i < nv; ++i) k_q[st][i] = vs[i], k_v[st][i] = a[i];
    }
    for (int i = 0; i < nv; ++i) {
        q[i] += dt * (k_q[0][i] + 2 * k_q[1][i] + 2 * k_q[2][i] + k_q[3][i]) / 6.0;
        v[i] += dt * (k_v[0][i] + 2 * k_v[1][i] + 2 * k_v[2][i] + k_v[3][i]) / 6.0;
    }
}

/* freq_rate substeps of one env-step with the optional per-substep observation noise */
static inline void oracle_env_step(oracle_accel_fn f, const void* ctx, int nv, int freq_rate, double dt, const oracle_opts_t* o,
                                   int64_t env, double* q, double* v, const double* ctrl) {
    for (int k = 0; k < freq_rate; ++k) {
        oracle_substep(f, ctx, nv, o ? o->integrator : ORACLE_EULER, dt, q, v, ctrl);
        if (o && (o->obs_pos != 0.f || o->obs_vel != 0.f)) {
            uint32_t nb = (uint32_t)((2 * nv + 3) / 4);
            oracle_gauss_state(o->seed ^ ORACLE_OBS_NOISE_KEY, o->env_offset + (uint64_t)env, o->episode,
                               (o->step_index * (uint32_t)freq_rate + (uint32_t)k) * nb, nv, o->obs_pos, o->obs_vel, o->shared, 0,
                               q, v);
        }
    }
}
#endif
