/*
 * cheetah_oracle.c — CPU restatement of the HalfCheetah-style body (float64).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (same rules as emei_oracle.c).
 *
 * PARITY UNPINNED for the dynamics: the reference steps this body with the third-party `mujoco`
 * package (emei/envs/mujoco/mujoco_env.py:86-109; requirements/main.txt:7, "mujoco >= 2.2.0", not
 * vendored, absent from the image).  This file restates, for the planar 9-DoF tree of
 * emei/envs/mujoco/assets/half_cheetah.xml, MuJoCo's published pipeline with the algorithms MuJoCo
 * itself uses — inertia-from-geom with settotalmass, recursive Newton-Euler for the bias forces,
 * unit-acceleration RNE columns for the joint-space inertia, spring/damper passive forces, armature,
 * Euler with implicit joint damping, soft constraints with solref/solimp impedance — combined with
 * emei's forward-Euler position override (mujoco_env.py:94-97,189-191).  One documented
 * simplification: the constraint forces (joint limits, foot/floor contacts with friction) are
 * obtained by ONE fixed-order Gauss-Seidel sweep over the active constraints (limits first, then
 * contact points in geom order; normal then tangent inside a contact) with regulariser
 * R = (1-d)/d * A_ii, instead of MuJoCo's converged Newton solve with pyramidal cones.  The HIP
 * kernel implements the same model with a different formulation (absolute-angle closed forms,
 * emei_amd/csrc/cheetah_model.h), so kernel-vs-oracle agreement checks both.
 *
 * First-party pieces (pinned by tests/golden/mujoco_firstparty_golden.npz): reward and terminal
 * (emei/envs/mujoco/half_cheetah.py:59-67), the Euler position rule, obs = concat(qpos, qvel).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))
#define NB 7  /* bodies: torso, bthigh, bshin, bfoot, fthigh, fshin, ffoot */
#define NV 9  /* rootx, rootz, rooty, bthigh, bshin, bfoot, fthigh, fshin, ffoot */
#define NG 8  /* capsule geoms: torso, head, 6 leg segments */

typedef struct { double x, z; } v2;
static inline v2 V(double x, double z) { v2 r = {x, z}; return r; }
static inline v2 add(v2 a, v2 b) { return V(a.x + b.x, a.z + b.z); }
static inline v2 sub(v2 a, v2 b) { return V(a.x - b.x, a.z - b.z); }
static inline v2 scl(double s, v2 a) { return V(s * a.x, s * a.z); }
static inline double dot(v2 a, v2 b) { return a.x * b.x + a.z * b.z; }
/* rotation about +y by phi (MuJoCo hinge axis "0 1 0"): x' = x c + z s, z' = -x s + z c */
static inline v2 rot(double phi, v2 a) { double c = cos(phi), s = sin(phi); return V(a.x * c + a.z * s, -a.x * s + a.z * c); }
/* d/dphi of rot(phi, a) for the rotated vector v: (v.z, -v.x) */
static inline v2 perp(v2 v) { return V(v.z, -v.x); }

typedef struct {
    int parent[NB];          /* parent body, -1 = world */
    v2 body_pos[NB];         /* body origin in the parent frame (torso: world, z = 0.7)   xml:62,69,72,75,80,83,86 */
    double mass[NB], inertia[NB]; /* after settotalmass = 14                              xml:35 */
    v2 com[NB];              /* centre of mass in the body frame */
    /* capsule geoms for contact: end-sphere centres in the body frame, radius          xml:66-67,71,74,77,82,85,88 */
    int geom_body[NG];
    v2 geom_end[NG][2];
    double geom_radius;
    double stiffness[NV], damping[NV], armature[NV], range_lo[NV], range_hi[NV];
    int limited[NV];
    double gear[6];
    double gravity, friction;
    /* solref / solimp: contacts (xml:38) and joint limits (xml:37) */
    double c_tc, c_dr, c_dmin, c_dmax, c_width;
    double l_tc, l_dr, l_dmin, l_dmax, l_width;
} cheetah_model_t;

static double capsule_mass(double rho, double r, double half) {
    return rho * (M_PI * r * r * 2 * half + 4.0 / 3.0 * M_PI * r * r * r);
}
static double capsule_inertia_perp(double rho, double r, double half) {
    double h = 2 * half, mcyl = rho * M_PI * r * r * h, msph = rho * 4.0 / 3.0 * M_PI * r * r * r;
    return mcyl * (3 * r * r + h * h) / 12 + msph * (2 * r * r / 5 + h * h / 4 + 3 * h * r / 8);
}

/* a capsule given by centre, rotation `ang` about y of the default +z axis, half-length */
static void capsule_ends(v2 centre, double ang, double half, v2 out[2]) {
    v2 axis = rot(ang, V(0, 1));
    out[0] = sub(centre, scl(half, axis));
    out[1] = add(centre, scl(half, axis));
}

EXPORT void cheetah_oracle_model(cheetah_model_t* m) {
    memset(m, 0, sizeof(*m));
    const double r = 0.046, rho = 1000.0;
    const int parent[NB] = {-1, 0, 1, 2, 0, 4, 5};
    const v2 bpos[NB] = {{0, 0.7}, {-0.5, 0}, {0.16, -0.25}, {-0.28, -0.14}, {0.5, 0}, {-0.14, -0.24}, {0.13, -0.18}};
    memcpy(m->parent, parent, sizeof(parent));
    memcpy(m->body_pos, bpos, sizeof(bpos));
    /* geoms: body, centre, angle about y, half-length */
    struct { int body; v2 c; double ang, half; } g[NG] = {
        {0, {0, 0}, M_PI / 2, 0.5},            /* torso: fromto (-.5,0,0)-(.5,0,0): axis along +x = z rotated by +pi/2 */
        {0, {0.6, 0.1}, 0.87, 0.15},           /* head */
        {1, {0.1, -0.13}, -3.8, 0.145},        /* bthigh */
        {2, {-0.14, -0.07}, -2.03, 0.15},      /* bshin */
        {3, {0.03, -0.097}, -0.27, 0.094},     /* bfoot */
        {4, {-0.07, -0.12}, 0.52, 0.133},      /* fthigh */
        {5, {0.065, -0.09}, -0.6, 0.106},      /* fshin */
        {6, {0.045, -0.07}, -0.6, 0.07},       /* ffoot */
    };
    double gm[NG], gi[NG], total = 0;
    for (int k = 0; k < NG; ++k) {
        gm[k] = capsule_mass(rho, r, g[k].half);
        gi[k] = capsule_inertia_perp(rho, r, g[k].half);
        m->geom_body[k] = g[k].body;
        capsule_ends(g[k].c, g[k].ang, g[k].half, m->geom_end[k]);
        total += gm[k];
    }
    m->geom_radius = r;
    /* per body: mass, com, inertia about the com (parallel axis over its geoms) */
    for (int b = 0; b < NB; ++b) {
        double mb = 0; v2 c = V(0, 0);
        for (int k = 0; k < NG; ++k) if (g[k].body == b) { mb += gm[k]; c = add(c, scl(gm[k], g[k].c)); }
        c = scl(1.0 / mb, c);
        double I = 0;
        for (int k = 0; k < NG; ++k) if (g[k].body == b) { v2 d = sub(g[k].c, c); I += gi[k] + gm[k] * dot(d, d); }
        m->mass[b] = mb, m->com[b] = c, m->inertia[b] = I;
    }
    double s = 14.0 / total; /* settotalmass: masses and inertias scale together */
    for (int b = 0; b < NB; ++b) m->mass[b] *= s, m->inertia[b] *= s;
    const double stiff[6] = {240, 180, 120, 180, 120, 60}, damp[6] = {6, 4.5, 3, 4.5, 3, 1.5};
    const double lo[6] = {-0.52, -0.785, -0.4, -1.0, -1.2, -0.5}, hi[6] = {1.05, 0.785, 0.785, 0.7, 0.87, 0.5};
    const double gear[6] = {120, 90, 60, 120, 60, 30};
    for (int k = 0; k < 6; ++k) {
        m->stiffness[3 + k] = stiff[k], m->damping[3 + k] = damp[k], m->armature[3 + k] = 0.1;
        m->range_lo[3 + k] = lo[k], m->range_hi[3 + k] = hi[k], m->limited[3 + k] = 1;
        m->gear[k] = gear[k];
    }
    m->gravity = 9.81, m->friction = 0.4;
    m->c_tc = 0.02, m->c_dr = 1, m->c_dmin = 0.0, m->c_dmax = 0.8, m->c_width = 0.01;
    m->l_tc = 0.02, m->l_dr = 1, m->l_dmin = 0.0, m->l_dmax = 0.8, m->l_width = 0.03;
}
EXPORT int cheetah_oracle_model_size(void) { return (int)sizeof(cheetah_model_t); }

/* ---------------------------------------------------------------------------------------------
 * kinematics: absolute angle, origin and com of every body */
typedef struct { double phi[NB]; v2 org[NB], com[NB]; } kin_t;

/* dof of the hinge that moves body b relative to its parent (torso: rooty = 2) */
static const int hinge_dof[NB] = {2, 3, 4, 5, 6, 7, 8};

static void kinematics(const cheetah_model_t* m, const double* q, kin_t* k) {
    for (int b = 0; b < NB; ++b) {
        int p = m->parent[b];
        if (p < 0) {
            k->phi[b] = q[2];
            k->org[b] = V(m->body_pos[b].x + q[0], m->body_pos[b].z + q[1]);
        } else {
            k->phi[b] = k->phi[p] + q[hinge_dof[b]];
            k->org[b] = add(k->org[p], rot(k->phi[p], m->body_pos[b]));
        }
        k->com[b] = add(k->org[b], rot(k->phi[b], m->com[b]));
    }
}

/* Recursive Newton-Euler (the algorithm behind mj_rne): generalized forces that produce
 * accelerations qdd at (q, qd); with qdd = 0 this is the bias vector c(q, qd) (+ gravity). */
static void rnea(const cheetah_model_t* m, const kin_t* k, const double* qd, const double* qdd, double grav, double* tau) {
    double w[NB], al[NB];
    v2 vo[NB], ao[NB], f[NB];
    double n[NB]; /* moment about the body origin accumulated from the subtree */
    for (int b = 0; b < NB; ++b) {
        int p = m->parent[b];
        if (p < 0) {
            w[b] = qd[2], al[b] = qdd[2];
            vo[b] = V(qd[0], qd[1]);
            ao[b] = V(qdd[0], qdd[1] + grav); /* gravity as an upward acceleration of the base */
        } else {
            v2 d = sub(k->org[b], k->org[p]);
            w[b] = w[p] + qd[hinge_dof[b]];
            al[b] = al[p] + qdd[hinge_dof[b]];
            vo[b] = add(vo[p], scl(w[p], perp(d)));
            ao[b] = add(ao[p], add(scl(al[p], perp(d)), scl(-w[p] * w[p], d)));
        }
        v2 rc = sub(k->com[b], k->org[b]);
        v2 ac = add(ao[b], add(scl(al[b], perp(rc)), scl(-w[b] * w[b], rc)));
        f[b] = scl(m->mass[b], ac);
        /* moment about the body origin: I*alpha + rc x f  (planar: generalized torque of f at offset rc = f . perp(rc)) */
        n[b] = m->inertia[b] * al[b] + dot(f[b], perp(rc));
    }
    for (int b = NB - 1; b >= 0; --b) {
        int p = m->parent[b];
        tau[hinge_dof[b]] = n[b];
        if (p >= 0) {
            v2 d = sub(k->org[b], k->org[p]);
            n[p] += n[b] + dot(f[b], perp(d));
            f[p] = add(f[p], f[b]);
        } else {
            tau[0] = f[b].x, tau[1] = f[b].z;
        }
    }
}

/* Jacobian (2 x NV) of a world point attached to body b */
static void point_jacobian(const cheetah_model_t* m, const kin_t* k, int b, v2 p, double Jx[NV], double Jz[NV]) {
    memset(Jx, 0, NV * sizeof(double));
    memset(Jz, 0, NV * sizeof(double));
    Jx[0] = 1, Jz[1] = 1;
    for (int a = b; a >= 0; a = m->parent[a]) {
        v2 d = perp(sub(p, k->org[a]));
        Jx[hinge_dof[a]] = d.x, Jz[hinge_dof[a]] = d.z;
    }
}

/* dense LDL^T of a symmetric positive definite NV x NV matrix (in place: L below the diagonal, D on it) */
static void ldl_factor(double A[NV][NV]) {
    for (int j = 0; j < NV; ++j) {
        for (int k = 0; k < j; ++k) A[j][j] -= A[j][k] * A[j][k] * A[k][k];
        for (int i = j + 1; i < NV; ++i) {
            for (int k = 0; k < j; ++k) A[i][j] -= A[i][k] * A[j][k] * A[k][k];
            A[i][j] /= A[j][j];
        }
    }
}
static void ldl_solve(const double A[NV][NV], double* x) {
    for (int i = 0; i < NV; ++i) for (int k = 0; k < i; ++k) x[i] -= A[i][k] * x[k];
    for (int i = 0; i < NV; ++i) x[i] /= A[i][i];
    for (int i = NV - 1; i >= 0; --i) for (int k = i + 1; k < NV; ++k) x[i] -= A[k][i] * x[k];
}

static double impedance(double dist, double dmin, double dmax, double width) {
    double x = fabs(dist) / width;
    double y = x >= 1 ? 1.0 : (x <= 0.5 ? 2 * x * x : 1 - 2 * (1 - x) * (1 - x)); /* midpoint .5, power 2 */
    double d = dmin + y * (dmax - dmin);
    return d < 1e-4 ? 1e-4 : (d > 0.9999 ? 0.9999 : d); /* mjMINIMP, mjMAXIMP */
}

/* one substep: mj_step with Euler (implicit joint damping) + emei's position override */
static void cheetah_substep(const cheetah_model_t* m, double dt, double* q, double* v, const double* ctrl) {
    kin_t k;
    kinematics(m, q, &k);
    double zero[NV] = {0}, bias[NV], M[NV][NV];
    rnea(m, &k, v, zero, m->gravity, bias);
    for (int c = 0; c < NV; ++c) { /* inertia column c = RNE with unit acceleration, no velocity, no gravity */
        double e[NV] = {0}, col[NV];
        e[c] = 1;
        rnea(m, &k, zero, e, 0.0, col);
        for (int r = 0; r < NV; ++r) M[r][c] = col[r];
    }
    double f[NV];
    for (int i = 0; i < NV; ++i) {
        f[i] = -bias[i] - m->stiffness[i] * q[i] - m->damping[i] * v[i]; /* passive: spring to 0, damper */
        M[i][i] += m->armature[i] + dt * m->damping[i];                  /* Euler with implicit damping: M + h D */
    }
    for (int a = 0; a < 6; ++a) {
        double c = ctrl[a] < -1 ? -1 : (ctrl[a] > 1 ? 1 : ctrl[a]); /* ctrllimited, ctrlrange +-1 (xml:39) */
        f[3 + a] += m->gear[a] * c;
    }
    ldl_factor(M);
    double acc[NV];
    memcpy(acc, f, sizeof(f));
    ldl_solve(M, acc); /* unconstrained acceleration */

    /* --- soft constraints, one Gauss-Seidel sweep -------------------------------------------- */
    double l_tc = m->l_tc < 2 * dt ? 2 * dt : m->l_tc, c_tc = m->c_tc < 2 * dt ? 2 * dt : m->c_tc; /* refsafe */
    for (int i = 3; i < NV; ++i) { /* joint limits (mjCNSTR_LIMIT_JOINT) */
        double dist, J;
        if (q[i] - m->range_lo[i] < 0) dist = q[i] - m->range_lo[i], J = 1;
        else if (m->range_hi[i] - q[i] < 0) dist = m->range_hi[i] - q[i], J = -1;
        else continue;
        double w[NV] = {0};
        w[i] = J;
        ldl_solve(M, w);
        double A = J * w[i];
        double imp = impedance(dist, m->l_dmin, m->l_dmax, m->l_width);
        double K = 1 / (m->l_dmax * m->l_dmax * l_tc * l_tc * m->l_dr * m->l_dr), B = 2 / (m->l_dmax * l_tc);
        double aref = -B * (J * v[i]) - K * imp * dist;
        double R = (1 - imp) / imp * A;
        double force = (aref - J * acc[i]) / (A + R);
        if (force > 0) for (int r = 0; r < NV; ++r) acc[r] += w[r] * force;
    }
    double cK = 1 / (m->c_dmax * m->c_dmax * c_tc * c_tc * m->c_dr * m->c_dr), cB = 2 / (m->c_dmax * c_tc);
    for (int g = 0; g < NG; ++g) { /* capsule end spheres against the floor plane z = 0 */
        int b = m->geom_body[g];
        for (int e = 0; e < 2; ++e) {
            v2 s = add(k.org[b], rot(k.phi[b], m->geom_end[g][e]));
            double dist = s.z - m->geom_radius;
            if (!(dist < 0)) continue;
            v2 p = V(s.x, 0.5 * dist); /* MuJoCo places the contact midway between the surfaces */
            double Jx[NV], Jz[NV], wx[NV], wz[NV];
            point_jacobian(m, &k, b, p, Jx, Jz);
            memcpy(wx, Jx, sizeof(wx));
            memcpy(wz, Jz, sizeof(wz));
            ldl_solve(M, wx);
            ldl_solve(M, wz);
            double Ann = 0, Att = 0, Atn = 0, an = 0, at = 0, vn = 0, vt = 0;
            for (int r = 0; r < NV; ++r) {
                Ann += Jz[r] * wz[r], Att += Jx[r] * wx[r], Atn += Jx[r] * wz[r];
                an += Jz[r] * acc[r], at += Jx[r] * acc[r];
                vn += Jz[r] * v[r], vt += Jx[r] * v[r];
            }
            double imp = impedance(dist, m->c_dmin, m->c_dmax, m->c_width);
            double Rn = (1 - imp) / imp * Ann, Rt = (1 - imp) / imp * Att;
            double fn = (-cB * vn - cK * imp * dist - an) / (Ann + Rn);
            if (!(fn > 0)) continue;
            double ft = (-cB * vt - at - Atn * fn) / (Att + Rt);
            double lim = m->friction * fn;
            ft = ft > lim ? lim : (ft < -lim ? -lim : ft);
            for (int r = 0; r < NV; ++r) acc[r] += wz[r] * fn + wx[r] * ft;
        }
    }
    for (int i = 0; i < NV; ++i) {
        q[i] += dt * v[i];   /* emei: position from the OLD velocity (mujoco_env.py:189-191) */
        v[i] += dt * acc[i]; /* MuJoCo Euler on qvel */
    }
}

/* half_cheetah.py:59-63 with step() semantics (one env per call) and :65-67 */
static double cheetah_reward(const double* obs, const double* pre_obs, const double* act, double dt_env) {
    double cost = 0;
    for (int a = 0; a < 6; ++a) cost += act[a] * act[a];
    return 1.0 * (obs[0] - pre_obs[0]) / dt_env - 0.1 * cost;
}
static uint8_t cheetah_terminal(const double* obs) {
    int fin = 1;
    for (int i = 0; i < 18; ++i) fin &= isfinite(obs[i]) != 0;
    return (uint8_t)!fin;
}

/* mujoco_env.py:157-167 for a batch: state [n,18] = (qpos, qvel) in/out, action [n,6] */
EXPORT void cheetah_oracle_step(int64_t n, int freq_rate, double dt, double* state, const double* action,
                                double* reward, uint8_t* terminal) {
    cheetah_model_t m;
    cheetah_oracle_model(&m);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double* s = state + 18 * i;
        double pre[18];
        memcpy(pre, s, sizeof(pre));
        for (int k = 0; k < freq_rate; ++k) cheetah_substep(&m, dt, s, s + 9, action + 6 * i);
        reward[i] = cheetah_reward(s, pre, action + 6 * i, dt * freq_rate);
        terminal[i] = cheetah_terminal(s);
    }
}

EXPORT void cheetah_oracle_reward(int64_t n, const double* obs, const double* pre_obs, const double* act, double dt_env, double* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = cheetah_reward(obs + 18 * i, pre_obs + 18 * i, act + 6 * i, dt_env);
}
EXPORT void cheetah_oracle_terminal(int64_t n, const double* obs, uint8_t* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = cheetah_terminal(obs + 18 * i);
}

/* diagnostics for the tests: mass matrix, bias and total mechanical energy at (q, v) */
EXPORT void cheetah_oracle_inertia(const double* q, const double* v, double* M_out, double* bias_out, double* energy_out) {
    cheetah_model_t m;
    cheetah_oracle_model(&m);
    kin_t k;
    kinematics(&m, q, &k);
    double zero[NV] = {0};
    rnea(&m, &k, v, zero, m.gravity, bias_out);
    for (int c = 0; c < NV; ++c) {
        double e[NV] = {0}, col[NV];
        e[c] = 1;
        rnea(&m, &k, zero, e, 0.0, col);
        for (int r = 0; r < NV; ++r) M_out[r * NV + c] = col[r];
    }
    double T = 0, U = 0;
    for (int r = 0; r < NV; ++r) for (int c = 0; c < NV; ++c) T += 0.5 * v[r] * M_out[r * NV + c] * v[c];
    for (int b = 0; b < NB; ++b) U += m.mass[b] * m.gravity * k.com[b].z;
    for (int i = 0; i < NV; ++i) T += 0.5 * m.armature[i] * v[i] * v[i], U += 0.5 * m.stiffness[i] * q[i] * q[i];
    *energy_out = T + U;
}
