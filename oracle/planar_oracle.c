/*
 * planar_oracle.c — CPU restatement (float64) of the planar articulated bodies the reference steps
 * with MuJoCo: the HalfCheetah-style body (emei/envs/mujoco/half_cheetah.py, assets/half_cheetah.xml)
 * and the Hopper (emei/envs/mujoco/hopper.py, assets/hopper.xml).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (same rules as emei_oracle.c).
 *
 * PARITY UNPINNED for the dynamics: the reference steps these bodies with the third-party `mujoco`
 * package (emei/envs/mujoco/mujoco_env.py:86-109; requirements/main.txt:7, "mujoco >= 2.2.0", not
 * vendored, absent from the image).  This file restates, for a planar kinematic tree (slide-x,
 * slide-z, hinge root + hinge children) described by a table, MuJoCo's published pipeline with the
 * algorithms MuJoCo itself uses — inertia-from-geom (with settotalmass where the xml asks for it),
 * recursive Newton-Euler for the bias forces, unit-acceleration RNE columns for the joint-space
 * inertia, spring/damper passive forces, armature, Euler with implicit joint damping or RK4, soft
 * constraints with solref/solimp impedance — combined with emei's integrator switch
 * (mujoco_env.py:70-79,94-97,189-191; integrators.h).  Constraint forces (joint limits, capsule/floor
 * contacts with friction), two selectable solvers (oracle_opts_t.solver):
 *   NEWTON (default) MuJoCo's own formulation as its documentation states it ("Computation" chapter): the
 *          acceleration minimises  1/2 (a - a0)' M (a - a0) + sum_i s_i(J_i a - aref_i),  s_i(x) = D_i x^2 / 2 for
 *          x < 0, over the active rows — one row per violated joint limit, the 2 (condim - 1) = 4 edges
 *          J_n +- mu J_t of the PYRAMIDAL friction cone (MuJoCo's default cone) per contact — with
 *          R_i = 1 / D_i = (1 - d) / d * diagApprox_i from the qpos0 inverse weights (dof_invweight0 for a limit;
 *          body_invweight0 (1 + mu^2), times 2 mu^2, for a pyramid edge), solved to convergence by Newton's method
 *          with an exact line search (mj_solNewton's scheme; MuJoCo itself stops at tolerance 1e-8), and the Euler
 *          integrator's implicit joint damping applied AFTER the solve, (M + h B) qacc = qfrc_smooth + J' f.
 *   SWEEP1 round 1's simplification, kept for the speed comparison: ONE fixed-order Gauss-Seidel sweep (limits first,
 *          then contact points in geom order; normal then box-clamped tangent inside a contact) with regulariser
 *          R = (1-d)/d * A_ii and the implicit damping folded into the matrix the sweep uses.
 * The HIP kernels implement the same models
 * with a different formulation (absolute-angle closed forms, emei_amd/csrc/cheetah_model.h and
 * hopper_model.h), so kernel-vs-oracle agreement checks both.
 *
 * First-party pieces (pinned by tests/golden/mujoco_firstparty_golden.npz and
 * hopper_firstparty_golden.npz): reward and terminal (half_cheetah.py:59-67; hopper.py:79-106), the
 * Euler position rule, obs = concat(qpos, qvel).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "integrators.h"

#define EXPORT __attribute__((visibility("default")))
/* table capacities; cheetah: 7 bodies / 9 dof / 8 geoms / 6 actuators, hopper: 4 / 6 / 4 / 3 */
#define NB 7
#define NV 9
#define NG 8
#define NU 6

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
    int nb, nv, ng, nu;      /* bodies, dofs (= 2 + nb: rootx, rootz, then one hinge per body in body order), geoms, actuators */
    int parent[NB];          /* parent body, -1 = world */
    v2 body_pos[NB];         /* body origin (= its joint anchor) in the parent frame; root: world */
    double hinge_sign[NB];   /* +1: hinge axis +y, -1: axis -y (hopper.xml:21,25,29) */
    double z_ref;            /* `ref` of the rootz slide (hopper.xml:16): world z = body_pos.z + q[1] - z_ref */
    double mass[NB], inertia[NB];
    v2 com[NB];              /* centre of mass in the body frame */
    /* capsule geoms for contact: end-sphere centres in the body frame, radius, pair friction with the floor */
    int geom_body[NG];
    v2 geom_end[NG][2];
    double geom_radius[NG], geom_friction[NG];
    double contact_margin;
    /* geom-geom contacts (hopper.xml:5: every geom contype = conaffinity = 1, condim 1): capsule pairs of bodies that are
     * neither the same nor parent and child (MuJoCo's default parent-child filter) collide, frictionless (condim = max(1, 1)) */
    int self_collide;
    double stiffness[NV], damping[NV], armature[NV], range_lo[NV], range_hi[NV];
    int limited[NV];
    int act_dof[NU];
    double gear[NU];
    double ctrl_lo, ctrl_hi; /* motors: ctrllimited, ctrlrange (half_cheetah.xml:88-95, hopper.xml:37-39): the clamp of smooth_terms */
    double gravity;
    /* solref / solimp: contacts and joint limits */
    double c_tc, c_dr, c_dmin, c_dmax, c_width;
    double l_tc, l_dr, l_dmin, l_dmax, l_width;
    /* inverse weights at qpos0 (mj_setConst): (M0^-1)_jj per dof; mean translational inverse inertia per body,
     * trace(J_com M0^-1 J_com') / 3 over the three world axes (a planar tree never moves along y) */
    double dof_invweight0[NV], body_invweight0[NB];
    int solver; /* ORACLE_SOLVER_* */
} planar_model_t;
typedef planar_model_t cheetah_model_t;
static void set_invweights(planar_model_t* m);

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

/* mass / com / inertia of every body from its capsule geoms (inertiafromgeom, density rho) */
typedef struct { int body; v2 c; double ang, half, r; } geom_spec_t;
static double bodies_from_geoms(planar_model_t* m, const geom_spec_t* g, double rho) {
    double gm[NG], gi[NG], total = 0;
    for (int k = 0; k < m->ng; ++k) {
        gm[k] = capsule_mass(rho, g[k].r, g[k].half);
        gi[k] = capsule_inertia_perp(rho, g[k].r, g[k].half);
        m->geom_body[k] = g[k].body;
        m->geom_radius[k] = g[k].r;
        capsule_ends(g[k].c, g[k].ang, g[k].half, m->geom_end[k]);
        total += gm[k];
    }
    for (int b = 0; b < m->nb; ++b) { /* parallel axis over the body's geoms */
        double mb = 0; v2 c = V(0, 0);
        for (int k = 0; k < m->ng; ++k) if (g[k].body == b) { mb += gm[k]; c = add(c, scl(gm[k], g[k].c)); }
        c = scl(1.0 / mb, c);
        double I = 0;
        for (int k = 0; k < m->ng; ++k) if (g[k].body == b) { v2 d = sub(g[k].c, c); I += gi[k] + gm[k] * dot(d, d); }
        m->mass[b] = mb, m->com[b] = c, m->inertia[b] = I;
    }
    return total;
}

/* assets/half_cheetah.xml: bodies torso, bthigh, bshin, bfoot, fthigh, fshin, ffoot (:62-88); dofs
 * rootx, rootz, rooty, bthigh, bshin, bfoot, fthigh, fshin, ffoot; capsule geoms torso, head, 6 legs */
EXPORT void cheetah_oracle_model(planar_model_t* m) {
    memset(m, 0, sizeof(*m));
    m->nb = 7, m->nv = 9, m->ng = 8, m->nu = 6;
    const double r = 0.046, rho = 1000.0;
    const int parent[7] = {-1, 0, 1, 2, 0, 4, 5};
    const v2 bpos[7] = {{0, 0.7}, {-0.5, 0}, {0.16, -0.25}, {-0.28, -0.14}, {0.5, 0}, {-0.14, -0.24}, {0.13, -0.18}};
    memcpy(m->parent, parent, sizeof(parent));
    memcpy(m->body_pos, bpos, sizeof(bpos));
    for (int b = 0; b < 7; ++b) m->hinge_sign[b] = 1.0;
    /* geoms: body, centre, angle about y, half-length, radius */
    const geom_spec_t g[8] = {
        {0, {0, 0}, M_PI / 2, 0.5, r},            /* torso: fromto (-.5,0,0)-(.5,0,0): axis along +x = z rotated by +pi/2 */
        {0, {0.6, 0.1}, 0.87, 0.15, r},           /* head */
        {1, {0.1, -0.13}, -3.8, 0.145, r},        /* bthigh */
        {2, {-0.14, -0.07}, -2.03, 0.15, r},      /* bshin */
        {3, {0.03, -0.097}, -0.27, 0.094, r},     /* bfoot */
        {4, {-0.07, -0.12}, 0.52, 0.133, r},      /* fthigh */
        {5, {0.065, -0.09}, -0.6, 0.106, r},      /* fshin */
        {6, {0.045, -0.07}, -0.6, 0.07, r},       /* ffoot */
    };
    double total = bodies_from_geoms(m, g, rho);
    double s = 14.0 / total; /* settotalmass (xml:35): masses and inertias scale together */
    for (int b = 0; b < 7; ++b) m->mass[b] *= s, m->inertia[b] *= s;
    const double stiff[6] = {240, 180, 120, 180, 120, 60}, damp[6] = {6, 4.5, 3, 4.5, 3, 1.5};
    const double lo[6] = {-0.52, -0.785, -0.4, -1.0, -1.2, -0.5}, hi[6] = {1.05, 0.785, 0.785, 0.7, 0.87, 0.5};
    const double gear[6] = {120, 90, 60, 120, 60, 30};
    for (int k = 0; k < 6; ++k) {
        m->stiffness[3 + k] = stiff[k], m->damping[3 + k] = damp[k], m->armature[3 + k] = 0.1;
        m->range_lo[3 + k] = lo[k], m->range_hi[3 + k] = hi[k], m->limited[3 + k] = 1;
        m->gear[k] = gear[k], m->act_dof[k] = 3 + k;
    }
    for (int k = 0; k < 8; ++k) m->geom_friction[k] = 0.4; /* default geom friction .4 (xml:38), floor the same */
    m->gravity = 9.81;
    m->ctrl_lo = -1, m->ctrl_hi = 1;
    m->c_tc = 0.02, m->c_dr = 1, m->c_dmin = 0.0, m->c_dmax = 0.8, m->c_width = 0.01;
    m->l_tc = 0.02, m->l_dr = 1, m->l_dmin = 0.0, m->l_dmax = 0.8, m->l_width = 0.03;
    set_invweights(m);
}
EXPORT int cheetah_oracle_model_size(void) { return (int)sizeof(planar_model_t); }

/* assets/hopper.xml (coordinate="global", angles in degrees): bodies torso, thigh, leg, foot; dofs
 * rootx, rootz (ref 1.25), rooty, thigh, leg, foot; the three leg hinges turn about -y (:21,25,29).
 * Body origins are placed at the joint anchors (world, at qpos0): torso (0,1.25) :17; thigh (0,1.05)
 * :21; leg (0,0.6) :25; foot (0,0.1) :29.  Capsules: torso r .05 (0,1.45)-(0,1.05) :18; thigh r .05
 * (0,1.05)-(0,.6) :22; leg r .04 (0,.6)-(0,.1) :26; foot r .06 (-.13,.1)-(.26,.1) :30.  Default
 * density 1000, no settotalmass.  Joints: armature 1, damping 1, limited (:5), root joints 0/0/free
 * (:15-17); ranges -150..0, -150..0, -45..45 deg; motors gear 200, ctrlrange +-1 (:37-39).
 * Contacts: geom solref (.02 1), solimp (.8 .8 .01), margin .001 (:5); pair friction = max(floor 1
 * (MuJoCo default, :14 sets none), geom): .9 -> 1 for torso/thigh/leg, 2 for the foot.  The geoms also collide with
 * EACH OTHER (contype = conaffinity = 1, :5) where their bodies are not parent and child: torso-leg, torso-foot,
 * thigh-foot, each one frictionless row (geom condim 1): capsule_pair / build_rows below. */
EXPORT void hopper_oracle_model(planar_model_t* m) {
    memset(m, 0, sizeof(*m));
    m->nb = 4, m->nv = 6, m->ng = 4, m->nu = 3;
    const double rho = 1000.0, deg = M_PI / 180.0;
    const int parent[4] = {-1, 0, 1, 2};
    const v2 bpos[4] = {{0, 1.25}, {0, -0.2}, {0, -0.45}, {0, -0.5}};
    memcpy(m->parent, parent, sizeof(parent));
    memcpy(m->body_pos, bpos, sizeof(bpos));
    m->hinge_sign[0] = 1.0, m->hinge_sign[1] = m->hinge_sign[2] = m->hinge_sign[3] = -1.0;
    m->z_ref = 1.25;
    const geom_spec_t g[4] = {
        {0, {0, 0}, 0.0, 0.2, 0.05},             /* torso: centre (0,1.25) = the body origin, axis z */
        {1, {0, -0.225}, 0.0, 0.225, 0.05},      /* thigh: centre (0,.825) */
        {2, {0, -0.25}, 0.0, 0.25, 0.04},        /* leg: centre (0,.35) */
        {3, {0.065, 0}, M_PI / 2, 0.195, 0.06},  /* foot: centre (.065,.1), axis x */
    };
    bodies_from_geoms(m, g, rho);
    const double lo[3] = {-150 * deg, -150 * deg, -45 * deg}, hi[3] = {0, 0, 45 * deg};
    for (int k = 0; k < 3; ++k) {
        m->damping[3 + k] = 1.0, m->armature[3 + k] = 1.0;
        m->range_lo[3 + k] = lo[k], m->range_hi[3 + k] = hi[k], m->limited[3 + k] = 1;
        m->gear[k] = 200.0, m->act_dof[k] = 3 + k;
    }
    m->geom_friction[0] = m->geom_friction[1] = m->geom_friction[2] = 1.0, m->geom_friction[3] = 2.0;
    m->contact_margin = 0.001;
    m->self_collide = 1; /* :5; the cheetah's geoms have conaffinity 0 (half_cheetah.xml:39): floor only */
    m->gravity = 9.81;
    m->ctrl_lo = -1, m->ctrl_hi = 1;
    m->c_tc = 0.02, m->c_dr = 1, m->c_dmin = 0.8, m->c_dmax = 0.8, m->c_width = 0.01;
    m->l_tc = 0.02, m->l_dr = 1, m->l_dmin = 0.9, m->l_dmax = 0.95, m->l_width = 0.001; /* MuJoCo joint defaults */
    set_invweights(m);
}

/* the oracle's tables in the layout of emei_model_constants (include/emei_hip.h): gravity; per body {mass, com, inertia,
 * origin in the parent frame}; per capsule {body, two end-sphere centres, radius, pair friction}; per actuated joint
 * {stiffness, damping, armature, range, gear}; margins / solref / solimp; ctrlrange; rootz ref; hinge sign.
 * tests/test_model_constants.py pins them to the reference's XML. */
EXPORT int planar_oracle_xml_constants(int body, double* out) {
    planar_model_t m;
    if (body == 0) cheetah_oracle_model(&m); else hopper_oracle_model(&m);
    int n = 0;
    out[n++] = m.gravity;
    for (int b = 0; b < m.nb; ++b) {
        out[n++] = m.mass[b], out[n++] = m.com[b].x, out[n++] = m.com[b].z, out[n++] = m.inertia[b];
        out[n++] = m.body_pos[b].x, out[n++] = m.body_pos[b].z;
    }
    for (int g = 0; g < m.ng; ++g) {
        out[n++] = m.geom_body[g];
        out[n++] = m.geom_end[g][0].x, out[n++] = m.geom_end[g][0].z, out[n++] = m.geom_end[g][1].x, out[n++] = m.geom_end[g][1].z;
        out[n++] = m.geom_radius[g], out[n++] = m.geom_friction[g];
    }
    for (int k = 0; k < m.nu; ++k) {
        const int d = m.act_dof[k];
        out[n++] = m.stiffness[d], out[n++] = m.damping[d], out[n++] = m.armature[d], out[n++] = m.range_lo[d], out[n++] = m.range_hi[d];
        out[n++] = m.gear[k];
    }
    out[n++] = m.contact_margin;
    out[n++] = m.c_tc, out[n++] = m.c_dmin, out[n++] = m.c_dmax, out[n++] = m.c_width;
    out[n++] = m.l_tc, out[n++] = m.l_dmin, out[n++] = m.l_dmax, out[n++] = m.l_width;
    out[n++] = m.ctrl_lo, out[n++] = m.ctrl_hi; /* the fields smooth_terms / planar_accel_sweep1 clamp with */
    out[n++] = m.z_ref, out[n++] = m.hinge_sign[m.nb - 1];
    return n;
}

/* ---------------------------------------------------------------------------------------------
 * kinematics: absolute angle, origin and com of every body */
typedef struct { double phi[NB]; v2 org[NB], com[NB]; } kin_t;

/* dof of the hinge that moves body b relative to its parent (root body: rooty = 2) */
static inline int hinge_dof(int b) { return 2 + b; }

static void kinematics(const planar_model_t* m, const double* q, kin_t* k) {
    for (int b = 0; b < m->nb; ++b) {
        int p = m->parent[b];
        if (p < 0) {
            k->phi[b] = m->hinge_sign[b] * q[2];
            k->org[b] = V(m->body_pos[b].x + q[0], m->body_pos[b].z + q[1] - m->z_ref);
        } else {
            k->phi[b] = k->phi[p] + m->hinge_sign[b] * q[hinge_dof(b)];
            k->org[b] = add(k->org[p], rot(k->phi[p], m->body_pos[b]));
        }
        k->com[b] = add(k->org[b], rot(k->phi[b], m->com[b]));
    }
}

/* Recursive Newton-Euler (the algorithm behind mj_rne): generalized forces that produce
 * accelerations qdd at (q, qd); with qdd = 0 this is the bias vector c(q, qd) (+ gravity). */
static void rnea(const planar_model_t* m, const kin_t* k, const double* qd, const double* qdd, double grav, double* tau) {
    double w[NB], al[NB];
    v2 vo[NB], ao[NB], f[NB];
    double n[NB]; /* moment about the body origin accumulated from the subtree */
    for (int b = 0; b < m->nb; ++b) {
        int p = m->parent[b];
        double sg = m->hinge_sign[b];
        if (p < 0) {
            w[b] = sg * qd[2], al[b] = sg * qdd[2];
            vo[b] = V(qd[0], qd[1]);
            ao[b] = V(qdd[0], qdd[1] + grav); /* gravity as an upward acceleration of the base */
        } else {
            v2 d = sub(k->org[b], k->org[p]);
            w[b] = w[p] + sg * qd[hinge_dof(b)];
            al[b] = al[p] + sg * qdd[hinge_dof(b)];
            vo[b] = add(vo[p], scl(w[p], perp(d)));
            ao[b] = add(ao[p], add(scl(al[p], perp(d)), scl(-w[p] * w[p], d)));
        }
        v2 rc = sub(k->com[b], k->org[b]);
        v2 ac = add(ao[b], add(scl(al[b], perp(rc)), scl(-w[b] * w[b], rc)));
        f[b] = scl(m->mass[b], ac);
        /* moment about the body origin: I*alpha + rc x f  (planar: generalized torque of f at offset rc = f . perp(rc)) */
        n[b] = m->inertia[b] * al[b] + dot(f[b], perp(rc));
    }
    for (int b = m->nb - 1; b >= 0; --b) {
        int p = m->parent[b];
        tau[hinge_dof(b)] = m->hinge_sign[b] * n[b];
        if (p >= 0) {
            v2 d = sub(k->org[b], k->org[p]);
            n[p] += n[b] + dot(f[b], perp(d));
            f[p] = add(f[p], f[b]);
        } else {
            tau[0] = f[b].x, tau[1] = f[b].z;
        }
    }
}

/* Jacobian (2 x nv) of a world point attached to body b */
static void point_jacobian(const planar_model_t* m, const kin_t* k, int b, v2 p, double Jx[NV], double Jz[NV]) {
    memset(Jx, 0, NV * sizeof(double));
    memset(Jz, 0, NV * sizeof(double));
    Jx[0] = 1, Jz[1] = 1;
    for (int a = b; a >= 0; a = m->parent[a]) {
        v2 d = scl(m->hinge_sign[a], perp(sub(p, k->org[a])));
        Jx[hinge_dof(a)] = d.x, Jz[hinge_dof(a)] = d.z;
    }
}

/* dense LDL^T of a symmetric positive definite n x n matrix (in place: L below the diagonal, D on it) */
static void ldl_factor(int n, double A[NV][NV]) {
    for (int j = 0; j < n; ++j) {
        for (int k = 0; k < j; ++k) A[j][j] -= A[j][k] * A[j][k] * A[k][k];
        for (int i = j + 1; i < n; ++i) {
            for (int k = 0; k < j; ++k) A[i][j] -= A[i][k] * A[j][k] * A[k][k];
            A[i][j] /= A[j][j];
        }
    }
}
static void ldl_solve(int n, const double A[NV][NV], double* x) {
    for (int i = 0; i < n; ++i) for (int k = 0; k < i; ++k) x[i] -= A[i][k] * x[k];
    for (int i = 0; i < n; ++i) x[i] /= A[i][i];
    for (int i = n - 1; i >= 0; --i) for (int k = i + 1; k < n; ++k) x[i] -= A[k][i] * x[k];
}

static double impedance(double dist, double dmin, double dmax, double width) {
    double x = fabs(dist) / width;
    double y = x >= 1 ? 1.0 : (x <= 0.5 ? 2 * x * x : 1 - 2 * (1 - x) * (1 - x)); /* midpoint .5, power 2 */
    double d = dmin + y * (dmax - dmin);
    return d < 1e-4 ? 1e-4 : (d > 0.9999 ? 0.9999 : d); /* mjMINIMP, mjMAXIMP */
}

/* joint-space inertia (armature included) and smooth forces at (q, v) */
static void smooth_terms(const planar_model_t* m, const kin_t* k, const double* q, const double* v, const double* ctrl,
                         double M[NV][NV], double* f) {
    const int nv = m->nv;
    double zero[NV] = {0}, bias[NV];
    rnea(m, k, v, zero, m->gravity, bias);
    for (int c = 0; c < nv; ++c) { /* inertia column c = RNE with unit acceleration, no velocity, no gravity */
        double e[NV] = {0}, col[NV];
        e[c] = 1;
        rnea(m, k, zero, e, 0.0, col);
        for (int r = 0; r < nv; ++r) M[r][c] = col[r];
    }
    for (int i = 0; i < nv; ++i) {
        f[i] = -bias[i] - m->stiffness[i] * q[i] - m->damping[i] * v[i]; /* passive: spring to 0, damper */
        M[i][i] += m->armature[i];
    }
    for (int a = 0; a < m->nu; ++a) {
        double c = ctrl[a] < m->ctrl_lo ? m->ctrl_lo : (ctrl[a] > m->ctrl_hi ? m->ctrl_hi : ctrl[a]); /* ctrllimited */
        f[m->act_dof[a]] += m->gear[a] * c;
    }
}

/* mj_setConst: inverse weights at qpos0 */
static void set_invweights(planar_model_t* m) {
    const int nv = m->nv;
    double q0[NV] = {0}, M[NV][NV], f[NV], ctrl0[NU] = {0};
    q0[1] = m->z_ref; /* qpos0 of a joint with `ref` is its ref */
    kin_t k;
    kinematics(m, q0, &k);
    smooth_terms(m, &k, q0, q0 + 0 * nv, ctrl0, M, f); /* v = 0: only M is used */
    ldl_factor(nv, M);
    for (int j = 0; j < nv; ++j) {
        double e[NV] = {0};
        e[j] = 1;
        ldl_solve(nv, M, e);
        m->dof_invweight0[j] = e[j];
    }
    for (int b = 0; b < m->nb; ++b) {
        double Jx[NV], Jz[NV], wx[NV], wz[NV], A = 0;
        point_jacobian(m, &k, b, k.com[b], Jx, Jz);
        memcpy(wx, Jx, sizeof(wx)), memcpy(wz, Jz, sizeof(wz));
        ldl_solve(nv, M, wx), ldl_solve(nv, M, wz);
        for (int r = 0; r < nv; ++r) A += Jx[r] * wx[r] + Jz[r] * wz[r];
        m->body_invweight0[b] = A / 3.0;
    }
}

/* one scalar constraint row of the primal problem: cost D/2 (J.a - aref)^2 where negative */
typedef struct { double J[NV], aref, D; } crow_t;
#define MAXPAIRS (NG * (NG - 1) / 2)
#define MAXROWS (NV + 4 * 2 * NG + MAXPAIRS)
#define MJ_MINVAL 1e-15
/* closest points nearer than this have no usable direction.  MuJoCo's own threshold is mjMINVAL, but between 1e-15 and 1e-10 the
 * difference of two O(1) positions is rounding noise, and two implementations would not agree on it: both sides use 1e-10 */
#define PAIR_MINLEN 1e-10

/* Do capsules g1 < g2 collide at all?  MuJoCo's filters: contype / conaffinity (the model's self_collide), not the same body,
 * not parent and child. */
static int pair_collides(const planar_model_t* m, int g1, int g2) {
    const int b1 = m->geom_body[g1], b2 = m->geom_body[g2];
    return m->self_collide && b1 != b2 && m->parent[b1] != b2 && m->parent[b2] != b1;
}

/* Capsule against capsule, both axes in the x-z plane.  MuJoCo's capsule-capsule collider: the closest points of the two axis
 * SEGMENTS, then sphere against sphere there — dist = |c2 - c1| - r1 - r2, normal from geom 1 to geom 2, contact position midway
 * between the surfaces; a contact exists while dist < margin.  The closest points are found the way MuJoCo's source and Ericson
 * (Real-Time Collision Detection, 5.1.9) do it: minimise over the two infinite lines (x1, x2 = signed distances from the
 * capsule centres along the unit axes), clamp x1 to its segment, re-solve x2 for it and clamp, re-solve x1 if x2 was clamped.
 * Coincident closest points (the axes cross: a penetration deeper than r1 + r2, which no trajectory reaches through the soft
 * contact) leave no direction: normal (1, 0), as MuJoCo's sphere-sphere falls back to the x axis (PAIR_MINLEN above).  Exactly parallel axes: MuJoCo emits up to TWO contacts (at the overlapping ends);
 * this restatement keeps the closer end pair only — a measure-zero configuration, and outside the joint ranges for the three
 * Hopper pairs (a parallel pair within r1 + r2 would need a hinge folded by 180 degrees).
 * Returns 1 and (*n, *pos, *dist) if dist < margin. */
static int capsule_pair(const planar_model_t* m, const kin_t* k, int g1, int g2, v2* n, v2* pos, double* dist_out) {
    const int b1 = m->geom_body[g1], b2 = m->geom_body[g2];
    v2 e10 = add(k->org[b1], rot(k->phi[b1], m->geom_end[g1][0])), e11 = add(k->org[b1], rot(k->phi[b1], m->geom_end[g1][1]));
    v2 e20 = add(k->org[b2], rot(k->phi[b2], m->geom_end[g2][0])), e21 = add(k->org[b2], rot(k->phi[b2], m->geom_end[g2][1]));
    v2 p1 = scl(0.5, add(e10, e11)), p2 = scl(0.5, add(e20, e21));
    v2 h1 = scl(0.5, sub(e11, e10)), h2 = scl(0.5, sub(e21, e20));
    const double l1 = sqrt(dot(h1, h1)), l2 = sqrt(dot(h2, h2)); /* half lengths */
    v2 a1 = scl(1 / l1, h1), a2 = scl(1 / l2, h2);
    v2 dif = sub(p1, p2);
    const double mb = -dot(a1, a2), u = -dot(a1, dif), w = dot(a2, dif), det = 1 - mb * mb;
    double x1, x2;
    if (fabs(det) >= MJ_MINVAL) {
        x1 = (u - mb * w) / det, x2 = (w - mb * u) / det;
        if (x1 > l1) x1 = l1, x2 = w - mb * x1;
        else if (x1 < -l1) x1 = -l1, x2 = w - mb * x1;
        if (x2 > l2) { x2 = l2, x1 = u - mb * x2; x1 = x1 > l1 ? l1 : (x1 < -l1 ? -l1 : x1); }
        else if (x2 < -l2) { x2 = -l2, x1 = u - mb * x2; x1 = x1 > l1 ? l1 : (x1 < -l1 ? -l1 : x1); }
    } else { /* parallel: the closest of the four end-to-segment pairs */
        double best = INFINITY;
        x1 = x2 = 0;
        for (int c = 0; c < 4; ++c) {
            double t1, t2;
            if (c < 2) { t1 = c ? l1 : -l1; t2 = w - mb * t1; t2 = t2 > l2 ? l2 : (t2 < -l2 ? -l2 : t2); }
            else { t2 = c == 3 ? l2 : -l2; t1 = u - mb * t2; t1 = t1 > l1 ? l1 : (t1 < -l1 ? -l1 : t1); }
            v2 d = sub(add(p2, scl(t2, a2)), add(p1, scl(t1, a1)));
            if (dot(d, d) < best) best = dot(d, d), x1 = t1, x2 = t2;
        }
    }
    v2 c1 = add(p1, scl(x1, a1)), c2 = add(p2, scl(x2, a2)), d = sub(c2, c1);
    const double len = sqrt(dot(d, d)), r1 = m->geom_radius[g1], r2 = m->geom_radius[g2];
    const double dist = len - r1 - r2;
    if (!(dist < m->contact_margin)) return 0;
    *n = len < PAIR_MINLEN ? V(1, 0) : scl(1 / len, d);
    *pos = add(c1, scl(r1 + 0.5 * dist, *n));
    *dist_out = dist;
    return 1;
}
/* The frictionless row of a capsule pair: J = n . (velocity of the contact point as part of body 2 - as part of body 1) */
static void pair_jacobian(const planar_model_t* m, const kin_t* k, int g1, int g2, v2 n, v2 pos, double J[NV]) {
    double J1x[NV], J1z[NV], J2x[NV], J2z[NV];
    point_jacobian(m, k, m->geom_body[g1], pos, J1x, J1z);
    point_jacobian(m, k, m->geom_body[g2], pos, J2x, J2z);
    for (int c = 0; c < NV; ++c) J[c] = n.x * (J2x[c] - J1x[c]) + n.z * (J2z[c] - J1z[c]);
}

static int build_rows(const planar_model_t* m, const kin_t* k, double dt, const double* q, const double* v, crow_t* rows) {
    const int nv = m->nv;
    int nr = 0;
    double l_tc = m->l_tc < 2 * dt ? 2 * dt : m->l_tc, c_tc = m->c_tc < 2 * dt ? 2 * dt : m->c_tc; /* refsafe */
    for (int i = 3; i < nv; ++i) { /* joint limits (mjCNSTR_LIMIT_JOINT), margin 0 */
        double dist, J;
        if (!m->limited[i]) continue;
        if (q[i] - m->range_lo[i] < 0) dist = q[i] - m->range_lo[i], J = 1;
        else if (m->range_hi[i] - q[i] < 0) dist = m->range_hi[i] - q[i], J = -1;
        else continue;
        crow_t* r = &rows[nr++];
        memset(r, 0, sizeof(*r));
        r->J[i] = J;
        double imp = impedance(dist, m->l_dmin, m->l_dmax, m->l_width);
        double K = 1 / (m->l_dmax * m->l_dmax * l_tc * l_tc * m->l_dr * m->l_dr), B = 2 / (m->l_dmax * l_tc);
        r->aref = -B * (J * v[i]) - K * imp * dist;
        double R = (1 - imp) / imp * m->dof_invweight0[i];
        r->D = 1.0 / (R > MJ_MINVAL ? R : MJ_MINVAL);
    }
    double cK = 1 / (m->c_dmax * m->c_dmax * c_tc * c_tc * m->c_dr * m->c_dr), cB = 2 / (m->c_dmax * c_tc);
    for (int g = 0; g < m->ng; ++g) { /* capsule end spheres against the floor plane z = 0 (mjCNSTR_CONTACT_PYRAMIDAL) */
        int b = m->geom_body[g];
        for (int e = 0; e < 2; ++e) {
            v2 s = add(k->org[b], rot(k->phi[b], m->geom_end[g][e]));
            double dist = s.z - m->geom_radius[g];
            if (!(dist < m->contact_margin)) continue;
            v2 p = V(s.x, 0.5 * dist); /* MuJoCo places the contact midway between the surfaces */
            double Jx[NV], Jz[NV];
            point_jacobian(m, k, b, p, Jx, Jz);
            double pos = dist - m->contact_margin, mu = m->geom_friction[g];
            double imp = impedance(pos, m->c_dmin, m->c_dmax, m->c_width);
            /* diagApprox of a pyramid edge n +- mu t in the isotropic approximation, then the edges' common regulariser as the
             * friction-curvature match of the elliptic cone at impratio 1 (the pair n +- mu t costs D (x_n^2 + mu^2 x_t^2)):
             * R_py = 2 mu^2 R.  Derivation and check on these rows: tests/test_oracle_solver.py::
             * test_pyramid_edge_regulariser_is_the_friction_match_of_the_elliptic_cone */
            double R = (1 - imp) / imp * (m->body_invweight0[b] * (1 + mu * mu));
            double Rpy = 2 * mu * mu * R;
            double D = 1.0 / (Rpy > MJ_MINVAL ? Rpy : MJ_MINVAL);
            /* edges n + mu t, n - mu t in the plane; the two edges n +- mu y of condim 3 have no y-motion to act on: J_n */
            const double sgn[4] = {1, -1, 0, 0};
            for (int ed = 0; ed < 4; ++ed) {
                crow_t* r = &rows[nr++];
                double Jv = 0;
                for (int c = 0; c < NV; ++c) r->J[c] = c < nv ? Jz[c] + sgn[ed] * mu * Jx[c] : 0.0;
                for (int c = 0; c < nv; ++c) Jv += r->J[c] * v[c];
                r->aref = -cB * Jv - cK * imp * pos;
                r->D = D;
            }
        }
    }
    for (int g1 = 0; g1 < m->ng; ++g1) /* capsule against capsule (mjCNSTR_CONTACT_FRICTIONLESS: both geoms condim 1) */
        for (int g2 = g1 + 1; g2 < m->ng; ++g2) {
            v2 n, p;
            double dist;
            if (!pair_collides(m, g1, g2) || !capsule_pair(m, k, g1, g2, &n, &p, &dist)) continue;
            crow_t* r = &rows[nr++];
            memset(r, 0, sizeof(*r));
            pair_jacobian(m, k, g1, g2, n, p, r->J);
            double Jv = 0;
            for (int c = 0; c < nv; ++c) Jv += r->J[c] * v[c];
            double pos = dist - m->contact_margin; /* margin = max of the two geoms' (equal here) */
            double imp = impedance(pos, m->c_dmin, m->c_dmax, m->c_width);
            /* diagApprox of a contact's normal row: the translational inverse weights of BOTH bodies (the floor's is 0) */
            double R = (1 - imp) / imp * (m->body_invweight0[m->geom_body[g1]] + m->body_invweight0[m->geom_body[g2]]);
            r->aref = -cB * Jv - cK * imp * pos;
            r->D = 1.0 / (R > MJ_MINVAL ? R : MJ_MINVAL);
        }
    return nr;
}

static void matvec(int n, const double A[NV][NV], const double* x, double* y) {
    for (int r = 0; r < n; ++r) {
        y[r] = 0;
        for (int c = 0; c < n; ++c) y[r] += A[r][c] * x[c];
    }
}

/* MuJoCo's constraint solve restated: Newton on the primal cost with an exact line search, to convergence.
 * *iters_out / *resid_out (optional): iterations used and the final scaled gradient norm. */
/* unit_steps != 0 (diagnostics only, planar_oracle_solve_unit): no line search, alpha = 1 — the iteration the HIP kernels run
 * (cheetah_model.h:accel_newton), started from `start` when given; stops on the kernels' criterion |g|_inf <= 1e-11 |f|_inf
 * with fmax passed in `unit_fmax`. */
static void newton_solve_ex(int nv, const double M[NV][NV], const double* a0, int nr, const crow_t* rows, double* a, int* iters_out,
                            double* resid_out, int unit_steps, const double* start, double unit_fmax, int max_iter);
static void newton_solve(int nv, const double M[NV][NV], const double* a0, int nr, const crow_t* rows, double* a, int* iters_out,
                         double* resid_out) {
    newton_solve_ex(nv, M, a0, nr, rows, a, iters_out, resid_out, 0, NULL, 0.0, 200);
}
static void newton_solve_ex(int nv, const double M[NV][NV], const double* a0, int nr, const crow_t* rows, double* a, int* iters_out,
                            double* resid_out, int unit_steps, const double* start, double unit_fmax, int max_iter) {
    double scale = 0; /* 1 / (mean inertia * nv), mj_solNewton's scaling of the gradient */
    for (int i = 0; i < nv; ++i) scale += M[i][i];
    scale = 1.0 / scale;
    memcpy(a, start ? start : a0, nv * sizeof(double));
    int it = 0;
    double gn = 0;
    for (; it < max_iter; ++it) {
        double H[NV][NV], g[NV], d[NV], da[NV], Mda[NV];
        for (int i = 0; i < nv; ++i) da[i] = a[i] - a0[i];
        matvec(nv, M, da, g);
        memcpy(H, M, sizeof(H));
        for (int r = 0; r < nr; ++r) {
            double x = -rows[r].aref;
            for (int c = 0; c < nv; ++c) x += rows[r].J[c] * a[c];
            if (!(x < 0)) continue;
            for (int i = 0; i < nv; ++i) {
                g[i] += rows[r].D * x * rows[r].J[i];
                for (int j = 0; j < nv; ++j) H[i][j] += rows[r].D * rows[r].J[i] * rows[r].J[j];
            }
        }
        gn = 0;
        for (int i = 0; i < nv; ++i) gn += g[i] * g[i];
        gn = sqrt(gn) * scale;
        if (unit_steps) {
            double gmax = 0;
            for (int i = 0; i < nv; ++i) gmax = fmax(gmax, fabs(g[i]));
            gn = gmax / unit_fmax;
            if (gmax <= 1e-11 * unit_fmax) break;
        } else if (gn < 1e-11) break; /* MuJoCo's own tolerance is 1e-8; the rounding floor of this gradient is ~1e-13 */
        ldl_factor(nv, H);
        for (int i = 0; i < nv; ++i) d[i] = -g[i];
        ldl_solve(nv, H, d);
        if (unit_steps) {
            for (int i = 0; i < nv; ++i) a[i] += d[i];
            continue;
        }
        /* exact line search: phi'(alpha) = d.M (a + alpha d - a0) + sum_active D (x_r + alpha jd_r) jd_r is piecewise linear
         * and increasing; Newton on it from alpha = 1 (the minimiser when the active set does not change), bisection guard */
        matvec(nv, M, d, Mda);
        double dMd = 0, dMa = 0;
        for (int i = 0; i < nv; ++i) dMd += d[i] * Mda[i], dMa += Mda[i] * da[i];
        double xr[MAXROWS], jd[MAXROWS];
        for (int r = 0; r < nr; ++r) {
            xr[r] = -rows[r].aref, jd[r] = 0;
            for (int c = 0; c < nv; ++c) xr[r] += rows[r].J[c] * a[c], jd[r] += rows[r].J[c] * d[c];
        }
        double lo = 0, hi = -1, alpha = 1;
        for (int ls = 0; ls < 100; ++ls) {
            double p1 = dMa + alpha * dMd, p2 = dMd;
            for (int r = 0; r < nr; ++r) {
                double x = xr[r] + alpha * jd[r];
                if (x < 0) p1 += rows[r].D * x * jd[r], p2 += rows[r].D * jd[r] * jd[r];
            }
            if (fabs(p1) <= 1e-15 * (fabs(dMa) + fabs(dMd) + 1e-300)) break;
            if (p1 > 0) hi = alpha; else lo = alpha;
            double next = alpha - p1 / p2;
            if (hi > 0 && !(next > lo && next < hi)) next = 0.5 * (lo + hi);
            if (hi < 0 && !(next > lo)) next = 2 * alpha;
            if (fabs(next - alpha) <= 1e-16 * fabs(alpha)) { alpha = next; break; }
            alpha = next;
        }
        for (int i = 0; i < nv; ++i) a[i] += alpha * d[i];
    }
    if (iters_out) *iters_out = it;
    if (resid_out) *resid_out = gn;
}

/* forward dynamics with MuJoCo's constraint formulation (see the header): qacc at (q, v); hd = dt for the Euler
 * integrator (implicit joint damping after the solve), 0 for RK4 */
static void planar_accel_newton(const planar_model_t* m, double dt, double hd, const double* q, const double* v, const double* ctrl,
                                double* acc, int* iters_out, double* resid_out, int* nrows_out) {
    const int nv = m->nv;
    kin_t k;
    kinematics(m, q, &k);
    double M[NV][NV], L[NV][NV], f[NV], a0[NV], a[NV];
    smooth_terms(m, &k, q, v, ctrl, M, f);
    memcpy(L, M, sizeof(L));
    ldl_factor(nv, L);
    memcpy(a0, f, nv * sizeof(double));
    ldl_solve(nv, L, a0); /* qacc_smooth */
    crow_t rows[MAXROWS];
    int nr = build_rows(m, &k, dt, q, v, rows);
    if (nrows_out) *nrows_out = nr;
    if (nr == 0) {
        memcpy(a, a0, nv * sizeof(double));
        if (iters_out) *iters_out = 0;
        if (resid_out) *resid_out = 0;
    } else {
        newton_solve(nv, M, a0, nr, rows, a, iters_out, resid_out);
    }
    if (hd > 0) { /* mj_EulerSkip: (M + h B) qacc = qfrc_smooth + qfrc_constraint = M a at the optimum */
        double rhs[NV];
        matvec(nv, M, a, rhs);
        for (int i = 0; i < nv; ++i) M[i][i] += hd * m->damping[i];
        ldl_factor(nv, M);
        ldl_solve(nv, M, rhs);
        memcpy(acc, rhs, nv * sizeof(double));
    } else {
        memcpy(acc, a, nv * sizeof(double));
    }
}

static void planar_accel_sweep1(const void* ctx, double dt, double hd, const double* q, const double* v, const double* ctrl, double* acc);
/* forward dynamics (mj_forward): qacc at (q, v) with the solver the model selects */
static void planar_accel(const void* ctx, double dt, double hd, const double* q, const double* v, const double* ctrl, double* acc) {
    const planar_model_t* m = (const planar_model_t*)ctx;
    if (m->solver == ORACLE_SOLVER_SWEEP1) planar_accel_sweep1(ctx, dt, hd, q, v, ctrl, acc);
    else planar_accel_newton(m, dt, hd, q, v, ctrl, acc, NULL, NULL, NULL);
}

/* SWEEP1 (round 1).  hd = dt for MuJoCo's Euler (implicit joint damping folded into the matrix:
 * (M + h D) qacc = f), 0 for RK4. */
static void planar_accel_sweep1(const void* ctx, double dt, double hd, const double* q, const double* v, const double* ctrl, double* acc) {
    const planar_model_t* m = (const planar_model_t*)ctx;
    const int nv = m->nv;
    kin_t k;
    kinematics(m, q, &k);
    double zero[NV] = {0}, bias[NV], M[NV][NV];
    rnea(m, &k, v, zero, m->gravity, bias);
    for (int c = 0; c < nv; ++c) { /* inertia column c = RNE with unit acceleration, no velocity, no gravity */
        double e[NV] = {0}, col[NV];
        e[c] = 1;
        rnea(m, &k, zero, e, 0.0, col);
        for (int r = 0; r < nv; ++r) M[r][c] = col[r];
    }
    double f[NV];
    for (int i = 0; i < nv; ++i) {
        f[i] = -bias[i] - m->stiffness[i] * q[i] - m->damping[i] * v[i]; /* passive: spring to 0, damper */
        M[i][i] += m->armature[i] + hd * m->damping[i];                  /* Euler with implicit damping: M + h D */
    }
    for (int a = 0; a < m->nu; ++a) {
        double c = ctrl[a] < m->ctrl_lo ? m->ctrl_lo : (ctrl[a] > m->ctrl_hi ? m->ctrl_hi : ctrl[a]); /* ctrllimited */
        f[m->act_dof[a]] += m->gear[a] * c;
    }
    ldl_factor(nv, M);
    memcpy(acc, f, nv * sizeof(double));
    ldl_solve(nv, M, acc); /* unconstrained acceleration */

    /* --- soft constraints, one Gauss-Seidel sweep -------------------------------------------- */
    double l_tc = m->l_tc < 2 * dt ? 2 * dt : m->l_tc, c_tc = m->c_tc < 2 * dt ? 2 * dt : m->c_tc; /* refsafe */
    for (int i = 3; i < nv; ++i) { /* joint limits (mjCNSTR_LIMIT_JOINT) */
        double dist, J;
        if (!m->limited[i]) continue;
        if (q[i] - m->range_lo[i] < 0) dist = q[i] - m->range_lo[i], J = 1;
        else if (m->range_hi[i] - q[i] < 0) dist = m->range_hi[i] - q[i], J = -1;
        else continue;
        double w[NV] = {0};
        w[i] = J;
        ldl_solve(nv, M, w);
        double A = J * w[i];
        double imp = impedance(dist, m->l_dmin, m->l_dmax, m->l_width);
        double K = 1 / (m->l_dmax * m->l_dmax * l_tc * l_tc * m->l_dr * m->l_dr), B = 2 / (m->l_dmax * l_tc);
        double aref = -B * (J * v[i]) - K * imp * dist;
        double R = (1 - imp) / imp * A;
        double force = (aref - J * acc[i]) / (A + R);
        if (force > 0) for (int r = 0; r < nv; ++r) acc[r] += w[r] * force;
    }
    double cK = 1 / (m->c_dmax * m->c_dmax * c_tc * c_tc * m->c_dr * m->c_dr), cB = 2 / (m->c_dmax * c_tc);
    for (int g = 0; g < m->ng; ++g) { /* capsule end spheres against the floor plane z = 0 */
        int b = m->geom_body[g];
        for (int e = 0; e < 2; ++e) {
            v2 s = add(k.org[b], rot(k.phi[b], m->geom_end[g][e]));
            double dist = s.z - m->geom_radius[g];
            if (!(dist < m->contact_margin)) continue;
            v2 p = V(s.x, 0.5 * dist); /* MuJoCo places the contact midway between the surfaces */
            double Jx[NV], Jz[NV], wx[NV], wz[NV];
            point_jacobian(m, &k, b, p, Jx, Jz);
            memcpy(wx, Jx, sizeof(wx));
            memcpy(wz, Jz, sizeof(wz));
            ldl_solve(nv, M, wx);
            ldl_solve(nv, M, wz);
            double Ann = 0, Att = 0, Atn = 0, an = 0, at = 0, vn = 0, vt = 0;
            for (int r = 0; r < nv; ++r) {
                Ann += Jz[r] * wz[r], Att += Jx[r] * wx[r], Atn += Jx[r] * wz[r];
                an += Jz[r] * acc[r], at += Jx[r] * acc[r];
                vn += Jz[r] * v[r], vt += Jx[r] * v[r];
            }
            double pos = dist - m->contact_margin;
            double imp = impedance(pos, m->c_dmin, m->c_dmax, m->c_width);
            double Rn = (1 - imp) / imp * Ann, Rt = (1 - imp) / imp * Att;
            double fn = (-cB * vn - cK * imp * pos - an) / (Ann + Rn);
            if (!(fn > 0)) continue;
            double ft = (-cB * vt - at - Atn * fn) / (Att + Rt);
            double lim = m->geom_friction[g] * fn;
            ft = ft > lim ? lim : (ft < -lim ? -lim : ft);
            for (int r = 0; r < nv; ++r) acc[r] += wz[r] * fn + wx[r] * ft;
        }
    }
    for (int g1 = 0; g1 < m->ng; ++g1) /* capsule against capsule: one frictionless row each, after the floor points */
        for (int g2 = g1 + 1; g2 < m->ng; ++g2) {
            v2 n, p;
            double dist, J[NV], w[NV], A = 0, an = 0, vn = 0;
            if (!pair_collides(m, g1, g2) || !capsule_pair(m, &k, g1, g2, &n, &p, &dist)) continue;
            pair_jacobian(m, &k, g1, g2, n, p, J);
            memcpy(w, J, sizeof(w));
            ldl_solve(nv, M, w);
            for (int r = 0; r < nv; ++r) A += J[r] * w[r], an += J[r] * acc[r], vn += J[r] * v[r];
            double pos = dist - m->contact_margin;
            double imp = impedance(pos, m->c_dmin, m->c_dmax, m->c_width);
            double fn = (-cB * vn - cK * imp * pos - an) / (A + (1 - imp) / imp * A);
            if (fn > 0) for (int r = 0; r < nv; ++r) acc[r] += w[r] * fn;
        }
}

/* Constructor parameters of the reward / terminal functions in the order of emei_hip.h's enum emei_env_param:
 * forward_reward_weight, ctrl_cost_weight, healthy_reward, terminate_when_unhealthy, healthy_state lo / hi,
 * healthy_z lo / hi.  NULL = the reference's defaults. */
static const double kCheetahDefaults[8] = {1.0, 0.1, 0, 0, 0, 0, 0, 0};                 /* half_cheetah.py:23-24 */
static const double kHopperDefaults[8] = {1.0, 1e-3, 1.0, 1.0, -100.0, 100.0, 0.7, INFINITY}; /* hopper.py:25-30 */

/* half_cheetah.py:59-63 with step() semantics (one env per call) and :65-67 */
static double cheetah_reward(const double* obs, const double* pre_obs, const double* act, double dt_env, const double* P) {
    double cost = 0;
    for (int a = 0; a < 6; ++a) cost += act[a] * act[a];
    return P[0] * (obs[0] - pre_obs[0]) / dt_env - P[1] * cost;
}
static uint8_t cheetah_terminal(const double* obs) {
    int fin = 1;
    for (int i = 0; i < 18; ++i) fin &= isfinite(obs[i]) != 0;
    return (uint8_t)!fin;
}

/* mujoco_env.py:157-167 for a batch: state [n,18] = (qpos, qvel) in/out, action [n,6] */
EXPORT void cheetah_oracle_step_p(int64_t n, int freq_rate, double dt, double* state, const double* action, double* reward,
                                  uint8_t* terminal, const oracle_opts_t* opts, const double* params) {
    const double* P = params ? params : kCheetahDefaults;
    planar_model_t m;
    cheetah_oracle_model(&m);
    m.solver = opts ? opts->solver : ORACLE_SOLVER_NEWTON;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double* s = state + 18 * i;
        double pre[18];
        memcpy(pre, s, sizeof(pre));
        oracle_env_step(planar_accel, &m, 9, freq_rate, dt, opts, i, s, s + 9, action + 6 * i);
        reward[i] = cheetah_reward(s, pre, action + 6 * i, dt * freq_rate, P);
        terminal[i] = cheetah_terminal(s);
    }
}
EXPORT void cheetah_oracle_step_ex(int64_t n, int freq_rate, double dt, double* state, const double* action, double* reward,
                                   uint8_t* terminal, const oracle_opts_t* opts) {
    cheetah_oracle_step_p(n, freq_rate, dt, state, action, reward, terminal, opts, NULL);
}
EXPORT void cheetah_oracle_step(int64_t n, int freq_rate, double dt, double* state, const double* action,
                                double* reward, uint8_t* terminal) {
    cheetah_oracle_step_p(n, freq_rate, dt, state, action, reward, terminal, NULL, NULL);
}

EXPORT void cheetah_oracle_reward_p(int64_t n, const double* obs, const double* pre_obs, const double* act, double dt_env,
                                    const double* params, double* out) {
    const double* P = params ? params : kCheetahDefaults;
    for (int64_t i = 0; i < n; ++i) out[i] = cheetah_reward(obs + 18 * i, pre_obs + 18 * i, act + 6 * i, dt_env, P);
}
EXPORT void cheetah_oracle_reward(int64_t n, const double* obs, const double* pre_obs, const double* act, double dt_env, double* out) {
    cheetah_oracle_reward_p(n, obs, pre_obs, act, dt_env, NULL, out);
}
EXPORT void cheetah_oracle_terminal(int64_t n, const double* obs, uint8_t* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = cheetah_terminal(obs + 18 * i);
}

/* ---------------------------------------------------------------------------------------------
 * Hopper (hopper.py).  is_healthy (:79-93) as the reference EXECUTES it: np.logical_and(healthy_state,
 * healthy_z, healthy_angle) passes healthy_angle as `out=`, so the result is healthy_state & healthy_z
 * and the angle range never matters.  healthy reward = (is_healthy | terminate_when_unhealthy) *
 * healthy_reward (:99); terminal = ~(is_healthy | terminate_when_unhealthy) (:104-106): with the default
 * flag (True) the reward term is constant and the env never terminates.  Per-env control cost (step()). */
static int hopper_is_healthy(const double* obs, const double* P) {
    int st = 1;
    for (int i = 2; i < 12; ++i) st &= (P[4] < obs[i]) && (obs[i] < P[5]);
    return st && (P[6] < obs[1]) && (obs[1] < P[7]);
}
static double hopper_reward(const double* obs, const double* pre_obs, const double* act, double dt_env, const double* P) {
    double cost = 0;
    for (int a = 0; a < 3; ++a) cost += act[a] * act[a];
    int ok = hopper_is_healthy(obs, P) || (P[3] != 0.0);
    return (ok ? P[2] : 0.0) + P[0] * (obs[0] - pre_obs[0]) / dt_env - P[1] * cost;
}
static uint8_t hopper_terminal(const double* obs, const double* P) { return (uint8_t)!(hopper_is_healthy(obs, P) || (P[3] != 0.0)); }

EXPORT void hopper_oracle_step_p(int64_t n, int freq_rate, double dt, double* state, const double* action, double* reward,
                                 uint8_t* terminal, const oracle_opts_t* opts, const double* params) {
    const double* P = params ? params : kHopperDefaults;
    planar_model_t m;
    hopper_oracle_model(&m);
    m.solver = opts ? opts->solver : ORACLE_SOLVER_NEWTON;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double* s = state + 12 * i;
        double pre[12];
        memcpy(pre, s, sizeof(pre));
        oracle_env_step(planar_accel, &m, 6, freq_rate, dt, opts, i, s, s + 6, action + 3 * i);
        reward[i] = hopper_reward(s, pre, action + 3 * i, dt * freq_rate, P);
        terminal[i] = hopper_terminal(s, P);
    }
}
EXPORT void hopper_oracle_step_ex(int64_t n, int freq_rate, double dt, double* state, const double* action, double* reward,
                                  uint8_t* terminal, const oracle_opts_t* opts) {
    hopper_oracle_step_p(n, freq_rate, dt, state, action, reward, terminal, opts, NULL);
}
EXPORT void hopper_oracle_reward_p(int64_t n, const double* obs, const double* pre_obs, const double* act, double dt_env,
                                   const double* params, double* out) {
    const double* P = params ? params : kHopperDefaults;
    for (int64_t i = 0; i < n; ++i) out[i] = hopper_reward(obs + 12 * i, pre_obs + 12 * i, act + 3 * i, dt_env, P);
}
EXPORT void hopper_oracle_reward(int64_t n, const double* obs, const double* pre_obs, const double* act, double dt_env, double* out) {
    hopper_oracle_reward_p(n, obs, pre_obs, act, dt_env, NULL, out);
}
EXPORT void hopper_oracle_is_healthy_p(int64_t n, const double* obs, const double* params, uint8_t* healthy, uint8_t* terminal) {
    const double* P = params ? params : kHopperDefaults;
    for (int64_t i = 0; i < n; ++i) {
        if (healthy) healthy[i] = (uint8_t)hopper_is_healthy(obs + 12 * i, P);
        if (terminal) terminal[i] = hopper_terminal(obs + 12 * i, P);
    }
}
EXPORT void hopper_oracle_is_healthy(int64_t n, const double* obs, uint8_t* out) {
    hopper_oracle_is_healthy_p(n, obs, NULL, out, NULL);
}

/* T env-steps of a planar body in one call (bench.py's cpu_baseline; see emei_oracle_ip_rollout): body 0 = cheetah, 1 = hopper;
 * state [n, 2 nv] in/out, float32 actions [T,n,nu], outputs obs float32 [T,n,2 nv], reward float32 [T,n], terminal uint8 [T,n]
 * (any may be NULL); no reset */
EXPORT void planar_oracle_rollout(int body, int64_t n, int T, int freq_rate, double dt, double* state, const float* actions,
                                  float* obs, float* reward, uint8_t* terminal, const oracle_opts_t* opts, const double* params) {
    planar_model_t m;
    if (body == 0) cheetah_oracle_model(&m); else hopper_oracle_model(&m);
    m.solver = opts ? opts->solver : ORACLE_SOLVER_NEWTON;
    const double* P = params ? params : (body == 0 ? kCheetahDefaults : kHopperDefaults);
    const int nv = m.nv, nu = m.nu, ns = 2 * nv;
    const int64_t BLK = 16, nblk = (n + BLK - 1) / BLK;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t b = 0; b < nblk; ++b) {
        const int64_t lo = b * BLK, hi = lo + BLK < n ? lo + BLK : n;
        for (int t = 0; t < T; ++t)
            for (int64_t i = lo; i < hi; ++i) {
                double* s = state + ns * i;
                double pre[2 * NV], a[NU];
                memcpy(pre, s, ns * sizeof(double));
                for (int k = 0; k < nu; ++k) a[k] = (double)actions[((int64_t)t * n + i) * nu + k];
                oracle_env_step(planar_accel, &m, nv, freq_rate, dt, opts, i, s, s + nv, a);
                const int64_t row = (int64_t)t * n + i;
                if (obs) for (int k = 0; k < ns; ++k) obs[ns * row + k] = (float)s[k];
                if (reward) reward[row] = (float)(body == 0 ? cheetah_reward(s, pre, a, dt * freq_rate, P) : hopper_reward(s, pre, a, dt * freq_rate, P));
                if (terminal) terminal[row] = body == 0 ? cheetah_terminal(s) : hopper_terminal(s, P);
            }
    }
}

/* diagnostics for the tests: one forward-dynamics evaluation with the Newton solver and what it took — active rows,
 * iterations, final scaled gradient norm — plus, for comparison, the one-sweep acceleration (either output may be NULL) */
EXPORT void planar_oracle_solve(int body, double dt, double hd, const double* q, const double* v, const double* ctrl, double* acc_newton,
                                double* acc_sweep1, int* nrows, int* iters, double* resid) {
    planar_model_t m;
    if (body == 0) cheetah_oracle_model(&m); else hopper_oracle_model(&m);
    double a[NV];
    planar_accel_newton(&m, dt, hd, q, v, ctrl, a, iters, resid, nrows);
    if (acc_newton) memcpy(acc_newton, a, m.nv * sizeof(double));
    if (acc_sweep1) planar_accel_sweep1(&m, dt, hd, q, v, ctrl, acc_sweep1);
}
/* diagnostics: the kernels' iteration (unit Newton steps, their stopping rule, optional warm start = the minimiser of a
 * previous evaluation) on one state -> the minimiser `a` (before the Euler damping step), *passes = gradient evaluations used
 * (the kernels' pass count; max_iter + 1 if the cap was hit), *nrows. */
EXPORT void planar_oracle_solve_unit(int body, double dt, const double* q, const double* v, const double* ctrl, const double* warm,
                                     int max_iter, double* a_out, int* passes, int* nrows) {
    planar_model_t m;
    if (body == 0) cheetah_oracle_model(&m); else hopper_oracle_model(&m);
    const int nv = m.nv;
    kin_t k;
    kinematics(&m, q, &k);
    double M[NV][NV], L[NV][NV], f[NV], a0[NV];
    smooth_terms(&m, &k, q, v, ctrl, M, f);
    memcpy(L, M, sizeof(L));
    ldl_factor(nv, L);
    memcpy(a0, f, nv * sizeof(double));
    ldl_solve(nv, L, a0);
    crow_t rows[MAXROWS];
    const int nr = build_rows(&m, &k, dt, q, v, rows);
    *nrows = nr;
    if (nr == 0) {
        memcpy(a_out, a0, nv * sizeof(double));
        *passes = 0;
        return;
    }
    double fm = 1.0;
    for (int i = 0; i < nv; ++i) fm = fmax(fm, fabs(f[i]));
    int it = 0;
    newton_solve_ex(nv, M, a0, nr, rows, a_out, &it, NULL, 1, warm, fm, max_iter);
    *passes = it + 1;
}
/* diagnostics for the long-horizon tests: number of scalar constraint rows (limit rows + 4 pyramid edges per contact point)
 * the solver sees at each of n states [n, 2 nv] = (qpos, qvel); -1 for a non-finite state */
EXPORT void planar_oracle_count_rows(int body, int64_t n, double dt, const double* state, int32_t* nrows_out) {
    planar_model_t m;
    if (body == 0) cheetah_oracle_model(&m); else hopper_oracle_model(&m);
    const int nv = m.nv;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const double* q = state + 2 * nv * i;
        int finite = 1;
        for (int c = 0; c < 2 * nv; ++c) finite &= isfinite(q[c]) != 0;
        if (!finite) { nrows_out[i] = -1; continue; }
        kin_t k;
        kinematics(&m, q, &k);
        crow_t rows[MAXROWS];
        nrows_out[i] = build_rows(&m, &k, dt, q, q + nv, rows);
    }
}
/* diagnostics: which rows exist at each state, as a bit mask — bit (i - 3) for a violated limit of joint dof i, bit
 * (nv - 3 + 2 g + e) for end sphere e of capsule g inside the contact margin, then one bit per colliding capsule pair (the kernels'
 * `rows` word has the same meaning) */
EXPORT void planar_oracle_row_mask(int body, int64_t n, const double* state, uint32_t* mask_out) {
    planar_model_t m;
    if (body == 0) cheetah_oracle_model(&m); else hopper_oracle_model(&m);
    const int nv = m.nv;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const double* q = state + 2 * nv * i;
        kin_t k;
        kinematics(&m, q, &k);
        uint32_t mask = 0;
        for (int d = 3; d < nv; ++d)
            if (m.limited[d] && (q[d] < m.range_lo[d] || q[d] > m.range_hi[d])) mask |= 1u << (d - 3);
        for (int g = 0; g < m.ng; ++g)
            for (int e = 0; e < 2; ++e) {
                v2 sp = add(k.org[m.geom_body[g]], rot(k.phi[m.geom_body[g]], m.geom_end[g][e]));
                if (sp.z - m.geom_radius[g] < m.contact_margin) mask |= 1u << (nv - 3 + 2 * g + e);
            }
        int pi = 0; /* colliding capsule pairs in (g1, g2) order: hopper torso-leg, torso-foot, thigh-foot */
        for (int g1 = 0; g1 < m.ng; ++g1)
            for (int g2 = g1 + 1; g2 < m.ng; ++g2) {
                v2 n, p;
                double dist;
                if (!pair_collides(&m, g1, g2)) continue;
                if (capsule_pair(&m, &k, g1, g2, &n, &p, &dist)) mask |= 1u << (nv - 3 + 2 * m.ng + pi);
                ++pi;
            }
        mask_out[i] = mask;
    }
}
/* diagnostics for the tests: the scalar constraint rows the Newton solver sees at one state — Jacobians [nr][nv], reference
 * accelerations, weights D = 1 / R — in build_rows' order (violated limits, then 4 pyramid edges per touching end sphere:
 * n + mu t, n - mu t, n, n; then one row per touching capsule pair).  Returns the row count; at most `cap` rows are written. */
EXPORT int planar_oracle_rows(int body, double dt, const double* q, const double* v, int cap, double* J_out, double* aref_out, double* D_out) {
    planar_model_t m;
    if (body == 0) cheetah_oracle_model(&m); else hopper_oracle_model(&m);
    kin_t k;
    kinematics(&m, q, &k);
    crow_t rows[MAXROWS];
    const int nr = build_rows(&m, &k, dt, q, v, rows);
    for (int r = 0; r < nr && r < cap; ++r) {
        memcpy(J_out + (size_t)r * m.nv, rows[r].J, m.nv * sizeof(double));
        aref_out[r] = rows[r].aref, D_out[r] = rows[r].D;
    }
    return nr;
}
/* diagnostics for the tests: every colliding capsule pair at q in (g1, g2) order -> out[pair][7] = {g1, g2, touching (dist <
 * margin), dist, normal x, normal z, 0}; dist / normal are reported for NON-touching pairs too (margin lifted for the query).
 * Returns the number of pairs. */
EXPORT int planar_oracle_pairs(int body, const double* q, double* out) {
    planar_model_t m;
    if (body == 0) cheetah_oracle_model(&m); else hopper_oracle_model(&m);
    kin_t k;
    kinematics(&m, q, &k);
    const double margin = m.contact_margin;
    int np = 0;
    for (int g1 = 0; g1 < m.ng; ++g1)
        for (int g2 = g1 + 1; g2 < m.ng; ++g2) {
            v2 n = V(0, 0), p = V(0, 0);
            double dist = 0;
            if (!pair_collides(&m, g1, g2)) continue;
            m.contact_margin = INFINITY;
            capsule_pair(&m, &k, g1, g2, &n, &p, &dist);
            m.contact_margin = margin;
            double* o = out + 7 * np++;
            o[0] = g1, o[1] = g2, o[2] = dist < margin, o[3] = dist, o[4] = n.x, o[5] = n.z, o[6] = 0;
        }
    return np;
}
EXPORT void planar_oracle_invweights(int body, double* dof_out, double* body_out) {
    planar_model_t m;
    if (body == 0) cheetah_oracle_model(&m); else hopper_oracle_model(&m);
    memcpy(dof_out, m.dof_invweight0, m.nv * sizeof(double));
    memcpy(body_out, m.body_invweight0, m.nb * sizeof(double));
}

/* diagnostics for the tests: mass matrix [nv,nv], bias and total mechanical energy at (q, v); body 0 = cheetah, 1 = hopper */
EXPORT void planar_oracle_inertia(int body, const double* q, const double* v, double* M_out, double* bias_out, double* energy_out) {
    planar_model_t m;
    if (body == 0) cheetah_oracle_model(&m); else hopper_oracle_model(&m);
    const int nv = m.nv;
    kin_t k;
    kinematics(&m, q, &k);
    double zero[NV] = {0};
    rnea(&m, &k, v, zero, m.gravity, bias_out);
    for (int c = 0; c < nv; ++c) {
        double e[NV] = {0}, col[NV];
        e[c] = 1;
        rnea(&m, &k, zero, e, 0.0, col);
        for (int r = 0; r < nv; ++r) M_out[r * nv + c] = col[r];
    }
    double T = 0, U = 0;
    for (int r = 0; r < nv; ++r) for (int c = 0; c < nv; ++c) T += 0.5 * v[r] * M_out[r * nv + c] * v[c];
    for (int b = 0; b < m.nb; ++b) U += m.mass[b] * m.gravity * k.com[b].z;
    for (int i = 0; i < nv; ++i) T += 0.5 * m.armature[i] * v[i] * v[i], U += 0.5 * m.stiffness[i] * q[i] * q[i];
    *energy_out = T + U;
}
EXPORT void cheetah_oracle_inertia(const double* q, const double* v, double* M_out, double* bias_out, double* energy_out) {
    planar_oracle_inertia(0, q, v, M_out, bias_out, energy_out);
}
/* body masses / world contact-sphere centres at q, for geometry checks in the tests; optionally (non-NULL)
 * each body's centre of mass [nb][2], absolute angle [nb] and inertia about the com [nb] */
EXPORT void planar_oracle_geometry(int body, const double* q, double* mass_out, double* ends_out /* [ng][2][2] */,
                                   double* com_out, double* phi_out, double* inertia_out) {
    planar_model_t m;
    if (body == 0) cheetah_oracle_model(&m); else hopper_oracle_model(&m);
    kin_t k;
    kinematics(&m, q, &k);
    for (int b = 0; b < m.nb; ++b) {
        mass_out[b] = m.mass[b];
        if (com_out) com_out[2 * b] = k.com[b].x, com_out[2 * b + 1] = k.com[b].z;
        if (phi_out) phi_out[b] = k.phi[b];
        if (inertia_out) inertia_out[b] = m.inertia[b];
    }
    for (int g = 0; g < m.ng; ++g)
        for (int e = 0; e < 2; ++e) {
            v2 s = add(k.org[m.geom_body[g]], rot(k.phi[m.geom_body[g]], m.geom_end[g][e]));
            ends_out[(g * 2 + e) * 2] = s.x, ends_out[(g * 2 + e) * 2 + 1] = s.z;
        }
}
