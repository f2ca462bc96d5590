// cheetah_model.h — device arithmetic of the HalfCheetah-style planar body (9 DoF, 7 links).
//
// Model: emei/envs/mujoco/assets/half_cheetah.xml stepped the way emei steps it
// (emei/envs/mujoco/mujoco_env.py:86-109,169-195): MuJoCo's Euler substep (implicit joint damping)
// with emei's forward-Euler position override.  Parity with libmujoco is UNPINNED (no MuJoCo in
// the image): the CPU oracle (oracle/planar_oracle.c) restates the same model with MuJoCo's own
// algorithms (recursive Newton-Euler in joint coordinates, dense factorisation); this file is an
// independent formulation of the same equations, chosen for the GPU:
//
//   * ABSOLUTE link angles phi_b as velocity coordinates u = (xdot, zdot, Omega_0..Omega_6).  In
//     these coordinates the inertia matrix of a planar tree is
//         M[phi_i, phi_i] = const,   M[phi_i, phi_j] = (R_i d_ij) . S_j   (i ancestor of j),
//         M[x|z, phi_j]   = perp(S_j),   S_j = R(phi_j) s_j,
//     with one constant "mass-moment" vector s_j per link, and the velocity-product forces are
//         -sum_j Omega_j^2 (...) of the same rotated vectors: no recursion, 7 sincos per substep.
//   * joint springs/dampers/armature/actuators/limits act on theta_k = phi_k - phi_parent(k): a
//     torque tau_k enters as +tau_k on phi_k and -tau_k on phi_parent(k).
//   * elimination order (leaves first: bfoot, bshin, bthigh, ffoot, fshin, fthigh, torso, x, z)
//     makes the LDL^T factorisation fill-free; the zero blocks between the two legs are
//     compile-time zeros of the fully unrolled loops.
//   * constraints (6 joint limits, 16 capsule-end/floor contact points with friction), two solvers like the oracle
//     (emei_config.solver): NEWTON (default) = MuJoCo's primal formulation — rows of the pyramidal friction cone,
//     regularisers from the qpos0 inverse weights, Newton's method to convergence, implicit joint damping after
//     the solve (accel_newton below) — and SWEEP1 = round 1's single fixed-order Gauss-Seidel sweep (accel).
#pragma once
#include <cmath>
#include <cstring>
#include <type_traits>

#include "constexpr_math.h"
#include "emei_device.h"

namespace emei {
namespace cheetah {

constexpr int NV = 9;
// permuted index of each velocity coordinate
enum { P_BFOOT = 0, P_BSHIN = 1, P_BTHIGH = 2, P_FFOOT = 3, P_FSHIN = 4, P_FTHIGH = 5, P_TORSO = 6, P_X = 7, P_Z = 8 };

// structural non-zero of the (filled) lower triangle: same leg chain, or a row of torso/x/z
__host__ __device__ constexpr bool nz(int i, int j) { return i >= 6 || (i / 3) == (j / 3); }

// Model constants (computed on the host in double by cheetah_make_model below, passed by value).
struct Model {
    // mass-moment vector s_j (body frame) and constant diagonal inertia per link, permuted link order 0..6
    double sx[7], sz[7], diag[7];
    // link vectors to the child joint (body frame): torso->bthigh, torso->fthigh, bthigh->bshin,
    // bshin->bfoot, fthigh->fshin, fshin->ffoot
    double d_tb[2], d_tf[2], d_bt_bs[2], d_bs_bf[2], d_ft_fs[2], d_fs_ff[2];
    double mtot, gravity, z0;                     // total mass, g, torso height at qpos0 (0.7)
    double stiff[6], damp[6], arm[6], lo[6], hi[6], gear[6];  // joint order bthigh,bshin,bfoot,fthigh,fshin,ffoot
    double geom_end[16][2];                       // capsule end-sphere centres, body frame; geom order torso,head,6 legs
    double radius, friction;
    double cK, cB, c_dmin, c_dmax, c_width;       // contact solref (refsafe'd for dt) / solimp
    double lK, lB, l_dmin, l_dmax, l_width;       // joint-limit solref / solimp
    double dt;
    double w_forward, w_ctrl;  // reward weights (half_cheetah.py:23-24), run-time
};

template <typename R>
struct V2 {
    R x, z;
};
template <typename R>
__device__ __forceinline__ V2<R> rot(R c, R s, R ax, R az) {  // rotation about +y: x' = x c + z s, z' = -x s + z c
    return V2<R>{fma_r(az, s, ax * c), fma_r(az, c, -(ax * s))};
}
template <typename R>
__device__ __forceinline__ R dot(V2<R> a, V2<R> b) { return fma_r(a.x, b.x, a.z * b.z); }
// a . perp(b), perp(b) = (b.z, -b.x)
template <typename R>
__device__ __forceinline__ R dotperp(V2<R> a, V2<R> b) { return fma_r(a.x, b.z, -(a.z * b.x)); }

template <typename R>
__device__ __forceinline__ R impedance(R dist, R dmin, R dmax, R inv_width) {  // inv_width: a compile-time 1 / solimp width
    R x = fabs(dist) * inv_width;
    R y = x >= R(1) ? R(1) : (x <= R(0.5) ? R(2) * x * x : R(1) - R(2) * (R(1) - x) * (R(1) - x));
    R d = dmin + y * (dmax - dmin);
    return d < R(1e-4) ? R(1e-4) : (d > R(0.9999) ? R(0.9999) : d);
}

// ---------------------------------------------------------------------------------------------
// Model constants from assets/half_cheetah.xml (inertiafromgeom, settotalmass = 14, xml:35), evaluated at
// COMPILE time: the kernels read them from the constexpr object kGeom below, so they are immediates
// (s_mov) instead of ~220 scalar registers of kernel arguments that would not fit the SGPR file and came
// back through v_readlane spills (28 % of the executed vector instructions).  Only what depends on the
// run-time dt (solref stiffness / damping, dt itself) travels as a kernel argument.
namespace cheetah_host {
struct H2 {
    double x, z;
};
constexpr H2 hrot(double a, H2 v) {  // compile-time rotation (constexpr_math.h)
    double sn = 0, cs = 0;
    ce::sincos(a, sn, cs);
    return {v.x * cs + v.z * sn, -v.x * sn + v.z * cs};
}
constexpr double capsule_mass(double rho, double r, double half) { return rho * (M_PI * r * r * 2 * half + 4.0 / 3.0 * M_PI * r * r * r); }
constexpr double capsule_inertia_perp(double rho, double r, double half) {
    double h = 2 * half, mcyl = rho * M_PI * r * r * h, msph = rho * 4.0 / 3.0 * M_PI * r * r * r;
    return mcyl * (3 * r * r + h * h) / 12 + msph * (2 * r * r / 5 + h * h / 4 + 3 * h * r / 8);
}
}  // namespace cheetah_host

// the XML-level tables (hand-typed from half_cheetah.xml:62-95; pinned to the file by tests/test_model_constants.py through
// emei_model_constants) and what MuJoCo's compiler derives from them: mass, centre of mass and inertia of every body
constexpr double kCtrlLo = -1.0, kCtrlHi = 1.0;  // <motor ctrllimited ctrlrange="-1 1"> (xml:40)
constexpr double kSolrefTc = 0.02;               // solref / solreflimit ".02 1" (xml:37-38): time constant, damping ratio 1
struct CheetahGeom {
    int body;
    cheetah_host::H2 c;
    double ang, half;
};
struct CheetahBodies {
    int parent[7];
    cheetah_host::H2 bpos[7];  // body origin = joint anchor, parent frame (torso: world)
    CheetahGeom g[8];          // capsules torso, head, bthigh, bshin, bfoot, fthigh, fshin, ffoot: body, centre, angle about y, half-length
    double r, total_mass;
    double mass[7], inertia[7];  // after settotalmass
    cheetah_host::H2 com[7];
};
constexpr CheetahBodies cheetah_bodies() {
    using namespace cheetah_host;
    const double r = 0.046, rho = 1000.0;
    // bodies in the xml's order: torso, bthigh, bshin, bfoot, fthigh, fshin, ffoot
    CheetahBodies B{{-1, 0, 1, 2, 0, 4, 5},
                    {{0, 0.7}, {-0.5, 0}, {0.16, -0.25}, {-0.28, -0.14}, {0.5, 0}, {-0.14, -0.24}, {0.13, -0.18}},
                    {{0, {0, 0}, M_PI / 2, 0.5},       {0, {0.6, 0.1}, 0.87, 0.15},       {1, {0.1, -0.13}, -3.8, 0.145},
                     {2, {-0.14, -0.07}, -2.03, 0.15}, {3, {0.03, -0.097}, -0.27, 0.094}, {4, {-0.07, -0.12}, 0.52, 0.133},
                     {5, {0.065, -0.09}, -0.6, 0.106}, {6, {0.045, -0.07}, -0.6, 0.07}},
                    r, 14.0, {}, {}, {}};
    const CheetahGeom(&g)[8] = B.g;
    double gm[8] = {}, gi[8] = {}, total = 0;
    for (int k = 0; k < 8; ++k) gm[k] = capsule_mass(rho, r, g[k].half), gi[k] = capsule_inertia_perp(rho, r, g[k].half), total += gm[k];
    for (int b = 0; b < 7; ++b) {
        double mb = 0;
        H2 c = {0, 0};
        for (int k = 0; k < 8; ++k)
            if (g[k].body == b) mb += gm[k], c.x += gm[k] * g[k].c.x, c.z += gm[k] * g[k].c.z;
        c.x /= mb, c.z /= mb;
        double I = 0;
        for (int k = 0; k < 8; ++k)
            if (g[k].body == b) {
                double dx = g[k].c.x - c.x, dz = g[k].c.z - c.z;
                I += gi[k] + gm[k] * (dx * dx + dz * dz);
            }
        B.mass[b] = mb, B.com[b] = c, B.inertia[b] = I;
    }
    const double s = B.total_mass / total;  // settotalmass (xml:35): masses and inertias scale together
    for (int b = 0; b < 7; ++b) B.mass[b] *= s, B.inertia[b] *= s;
    return B;
}

constexpr Model cheetah_make_model(double dt) {
    using namespace cheetah_host;
    Model m{};
    const CheetahBodies B = cheetah_bodies();
    const double r = B.r;
    const int(&parent)[7] = B.parent;
    const H2(&bpos)[7] = B.bpos;
    const CheetahGeom(&g)[8] = B.g;
    const double(&mass)[7] = B.mass;
    const double(&inertia)[7] = B.inertia;
    const H2(&com)[7] = B.com;
    // subtree masses
    double sub[7] = {};
    for (int b = 0; b < 7; ++b) sub[b] = mass[b];
    for (int b = 6; b > 0; --b) sub[parent[b]] += sub[b];
    // permuted link order: 0 bfoot 1 bshin 2 bthigh 3 ffoot 4 fshin 5 fthigh 6 torso  <- xml body index
    const int perm_body[7] = {3, 2, 1, 6, 5, 4, 0};
    for (int p = 0; p < 7; ++p) {
        const int b = perm_body[p];
        double sx = mass[b] * com[b].x, sz = mass[b] * com[b].z;
        double dg = inertia[b] + mass[b] * (com[b].x * com[b].x + com[b].z * com[b].z);
        for (int c = 1; c < 7; ++c)
            if (parent[c] == b) {
                sx += sub[c] * bpos[c].x, sz += sub[c] * bpos[c].z;
                dg += sub[c] * (bpos[c].x * bpos[c].x + bpos[c].z * bpos[c].z);
            }
        m.sx[p] = sx, m.sz[p] = sz, m.diag[p] = dg;
    }
    m.d_tb[0] = bpos[1].x, m.d_tb[1] = bpos[1].z;
    m.d_tf[0] = bpos[4].x, m.d_tf[1] = bpos[4].z;
    m.d_bt_bs[0] = bpos[2].x, m.d_bt_bs[1] = bpos[2].z;
    m.d_bs_bf[0] = bpos[3].x, m.d_bs_bf[1] = bpos[3].z;
    m.d_ft_fs[0] = bpos[5].x, m.d_ft_fs[1] = bpos[5].z;
    m.d_fs_ff[0] = bpos[6].x, m.d_fs_ff[1] = bpos[6].z;
    m.mtot = 14.0, m.gravity = 9.81, m.z0 = bpos[0].z;
    const double stiff[6] = {240, 180, 120, 180, 120, 60}, damp[6] = {6, 4.5, 3, 4.5, 3, 1.5};
    const double lo[6] = {-0.52, -0.785, -0.4, -1.0, -1.2, -0.5}, hi[6] = {1.05, 0.785, 0.785, 0.7, 0.87, 0.5};
    const double gear[6] = {120, 90, 60, 120, 60, 30};
    for (int k = 0; k < 6; ++k)
        m.stiff[k] = stiff[k], m.damp[k] = damp[k], m.arm[k] = 0.1, m.lo[k] = lo[k], m.hi[k] = hi[k], m.gear[k] = gear[k];
    for (int k = 0; k < 8; ++k) {  // capsule end spheres: centre -/+ half * axis, axis = +z rotated by ang about y
        H2 ax = hrot(g[k].ang, {0, 1});
        m.geom_end[2 * k][0] = g[k].c.x - g[k].half * ax.x, m.geom_end[2 * k][1] = g[k].c.z - g[k].half * ax.z;
        m.geom_end[2 * k + 1][0] = g[k].c.x + g[k].half * ax.x, m.geom_end[2 * k + 1][1] = g[k].c.z + g[k].half * ax.z;
    }
    m.radius = r, m.friction = 0.4;
    // solref (.02, 1) with MuJoCo's refsafe clamp timeconst >= 2 dt; solimp contacts (0,.8,.01), limits (0,.8,.03)
    const double tc = kSolrefTc < 2 * dt ? 2 * dt : kSolrefTc, dmax = 0.8;
    m.cK = m.lK = 1.0 / (dmax * dmax * tc * tc), m.cB = m.lB = 2.0 / (dmax * tc);
    m.c_dmin = 0.0, m.c_dmax = dmax, m.c_width = 0.01;
    m.l_dmin = 0.0, m.l_dmax = dmax, m.l_width = 0.03;
    m.dt = dt;
    return m;
}

// every dt-independent constant of the model, as compile-time immediates for the device code
__device__ constexpr Model kGeom = cheetah_make_model(0.002);

// emei_model_constants (include/emei_hip.h): the tables above in the layout documented there, for the test that pins them
// to the reference's XML.  Host only; reads the SAME constexpr objects the kernels are compiled from.
inline int xml_constants(double* out) {
    constexpr CheetahBodies B = cheetah_bodies();
    constexpr Model m = cheetah_make_model(0.002);
    int n = 0;
    out[n++] = m.gravity;
    for (int b = 0; b < 7; ++b) {
        const double row[6] = {B.mass[b], B.com[b].x, B.com[b].z, B.inertia[b], B.bpos[b].x, B.bpos[b].z};
        for (double v : row) out[n++] = v;
    }
    for (int k = 0; k < 8; ++k) {
        const double row[7] = {(double)B.g[k].body, m.geom_end[2 * k][0], m.geom_end[2 * k][1], m.geom_end[2 * k + 1][0],
                               m.geom_end[2 * k + 1][1], m.radius, m.friction};
        for (double v : row) out[n++] = v;
    }
    for (int k = 0; k < 6; ++k) {
        const double row[6] = {m.stiff[k], m.damp[k], m.arm[k], m.lo[k], m.hi[k], m.gear[k]};
        for (double v : row) out[n++] = v;
    }
    const double tail[13] = {0.0 /* contact margin */, kSolrefTc,
                             m.c_dmin, m.c_dmax, m.c_width, kSolrefTc, m.l_dmin, m.l_dmax, m.l_width, kCtrlLo, kCtrlHi,
                             0.0 /* rootz ref */, 1.0 /* hinges about +y */};
    for (double v : tail) out[n++] = v;
    return n;
}

// Inverse weights at qpos0 (MuJoCo's mj_setConst; the diagonal approximation of J M^-1 J' that scales the constraint
// regularisers): dof_invweight0 = (M0^-1)_jj of the six leg joints (bthigh .. ffoot) and body_invweight0 = the mean
// translational inverse inertia trace(J_com M0^-1 J_com') / 3 of every link, in the PERMUTED link order (bfoot, bshin,
// bthigh, ffoot, fshin, fthigh, torso).  Derived HERE, at compile time, from this file's own model — the absolute-angle inertia
// M_u(qpos0) assembled from the mass-moment vectors exactly as accel() does it (every link angle is 0 at qpos0: R = 1) — and not
// from the oracle: a joint velocity theta_k = Omega_child - Omega_parent is the row J = e_child - e_parent of the absolute
// coordinates, so (M_q^-1)_kk = J M_u^-1 J'; a link's com Jacobian is perp() of the link vectors on its root path.  The oracle
// inverts the joint-space RNE inertia instead (planar_oracle.c:set_invweights); tests/test_oracle_solver.py compares the two
// derivations (and a third, energy-based one in NumPy from the XML constants) through emei_model_invweights.
struct InvWeights {
    double dof[6], link[7];
};
constexpr InvWeights cheetah_invweights() {
    using namespace cheetah_host;
    const Model m = cheetah_make_model(0.002);
    const CheetahBodies B = cheetah_bodies();
    double A[NV][NV] = {};
    for (int b = 0; b < 7; ++b) A[b][b] = m.diag[b], A[P_X][b] = A[b][P_X] = m.sz[b], A[P_Z][b] = A[b][P_Z] = -m.sx[b];
    A[P_X][P_X] = A[P_Z][P_Z] = m.mtot;
    // permuted link -> xml body, and the chain of link vectors from an ancestor towards a descendant: the ancestor's vector to
    // the child ON that path = the child's body origin in the ancestor's frame (bpos)
    const int perm_body[7] = {3, 2, 1, 6, 5, 4, 0};
    int body_perm[7] = {};
    for (int p = 0; p < 7; ++p) body_perm[perm_body[p]] = p;
    for (int j = 0; j < 7; ++j) {  // descendant j, every proper ancestor a: M[a][j] = d_(a -> towards j) . s_j
        int child = perm_body[j];
        for (int a = B.parent[child]; a >= 0; child = a, a = B.parent[a]) {
            const int pa = body_perm[a];
            A[pa][j] = A[j][pa] = B.bpos[child].x * m.sx[j] + B.bpos[child].z * m.sz[j];
        }
    }
    const int jc[6] = {P_BTHIGH, P_BSHIN, P_BFOOT, P_FTHIGH, P_FSHIN, P_FFOOT};
    const int jp[6] = {P_TORSO, P_BTHIGH, P_BSHIN, P_TORSO, P_FTHIGH, P_FSHIN};
    for (int k = 0; k < 6; ++k) {  // armature acts on theta_k = phi_child - phi_parent
        A[jc[k]][jc[k]] += m.arm[k], A[jp[k]][jp[k]] += m.arm[k];
        A[jc[k]][jp[k]] -= m.arm[k], A[jp[k]][jc[k]] -= m.arm[k];
    }
    InvWeights w{};
    for (int k = 0; k < 6; ++k) {
        double J[NV] = {};
        J[jc[k]] = 1, J[jp[k]] = -1;
        w.dof[k] = ce::spd_quad<NV>(A, J);
    }
    for (int p = 0; p < 7; ++p) {
        double Jx[NV] = {}, Jz[NV] = {};
        Jx[P_X] = 1, Jz[P_Z] = 1;
        const int b = perm_body[p];
        Jx[p] = B.com[b].z, Jz[p] = -B.com[b].x;  // d/dphi of R(phi) c at phi = 0: perp(c)
        int child = b;
        for (int a = B.parent[b]; a >= 0; child = a, a = B.parent[a]) Jx[body_perm[a]] = B.bpos[child].z, Jz[body_perm[a]] = -B.bpos[child].x;
        w.link[p] = (ce::spd_quad<NV>(A, Jx) + ce::spd_quad<NV>(A, Jz)) / 3.0;  // three world axes; nothing moves along y
    }
    return w;
}
__device__ constexpr InvWeights kInvW = cheetah_invweights();
// emei_model_invweights (include/emei_hip.h): dof_invweight0 of the six leg joints, then body_invweight0 in XML body order
inline int xml_invweights(double* out) {
    constexpr InvWeights w = cheetah_invweights();
    const int perm_body[7] = {3, 2, 1, 6, 5, 4, 0};
    int n = 0;
    for (int k = 0; k < 6; ++k) out[n++] = w.dof[k];
    for (int p = 0; p < 7; ++p) out[6 + perm_body[p]] = w.link[p];
    return n + 7;
}
#ifndef EMEI_MAX_NEWTON
#define EMEI_MAX_NEWTON 24
#endif
// Constraint-space ("dual") form of the same active-set iteration for lanes with at most kDualSlots row blocks (contact
// points, violated joint limits; accel_newton): slots of kSlotFields values per lane in the block's LDS scratch
#ifndef EMEI_CHEETAH_DUAL_SLOTS
#define EMEI_CHEETAH_DUAL_SLOTS 2  // 0: every lane iterates in the primal loop (A/B: 10.1 vs 8.6 ms per 100 steps of config 4)
#endif
constexpr int kDualSlots = EMEI_CHEETAH_DUAL_SLOTS;
constexpr int kSlotFields = 26;  // Yn[9], Yt[9], u0n, u0t, bn, bt, Dw, start un, start ut, mu
static_assert(kDualSlots == 0 || kDualSlots == 2, "the elimination below is written for two slots");
constexpr int kMaxNewton = EMEI_MAX_NEWTON;  // iteration cap of accel_newton (the oracle's statistics: <= 9 over 40 000 random states)

// Sparse LDL^T in the permuted order.  L is stored in the strict lower triangle of A, 1/D in invd.
template <typename R>
__device__ __forceinline__ void ldl_factor(R (&A)[NV][NV], R (&invd)[NV]) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        invd[j] = rcp_r(A[j][j]);
#pragma unroll
        for (int i = j + 1; i < NV; ++i) {
            if (!nz(i, j)) continue;
            const R l = A[i][j] * invd[j];
            // A[i][k] -= l * A[k][j] for j < k <= i (right-looking update with the unscaled column)
#pragma unroll
            for (int k = j + 1; k <= i; ++k)
                if (nz(k, j) && nz(i, k)) A[i][k] = fma_r(-l, A[k][j], A[i][k]);
        }
#pragma unroll
        for (int i = j + 1; i < NV; ++i)
            if (nz(i, j)) A[i][j] *= invd[j];
    }
}
// Structural non-zeros of L^-1 J for a constraint whose Jacobian starts at coordinate FIRST (a link: its own
// angle, its ancestors on the same leg, then torso, x, z): everything before FIRST and the other leg stay 0.
__host__ __device__ constexpr bool in_pat(int first, int i) { return i >= 6 || (first < 6 && (i / 3) == (first / 3) && i >= first); }

// y <- L^-1 y for a right-hand side with the pattern of FIRST (FIRST = 0 with every entry set: the dense case)
template <int FIRST, bool DENSE, typename R>
__device__ __forceinline__ void ldl_forward(const R (&A)[NV][NV], R (&y)[NV]) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        if (!(DENSE || in_pat(FIRST, j))) continue;
#pragma unroll
        for (int i = j + 1; i < NV; ++i)
            if (nz(i, j)) y[i] = fma_r(-A[i][j], y[j], y[i]);
    }
}
// x <- L^-T x
template <typename R>
__device__ __forceinline__ void ldl_backward(const R (&A)[NV][NV], R (&x)[NV]) {
#pragma unroll
    for (int j = NV - 1; j >= 0; --j)
#pragma unroll
        for (int i = j + 1; i < NV; ++i)
            if (nz(i, j)) x[j] = fma_r(-A[i][j], x[i], x[j]);
}

// Forward dynamics: qacc at (q, v) with actuator, passive and soft-constraint forces.  `hd` is the
// time step when joint damping is treated implicitly (MuJoCo's Euler: (M + h D) qacc = f), 0 for RK4.
// q, v in the reference's order (rootx, rootz, rooty, bthigh, bshin, bfoot, fthigh, fshin, ffoot).
template <typename R>
__device__ __forceinline__ void accel(const R (&q)[NV], const R (&v)[NV], const R (&ctrl)[6], const Model& m, R hd,
                                      R (&qacc)[NV], const TrigCtx& trig) {
    // ---- absolute angles / rates, permuted link order: 0 bfoot 1 bshin 2 bthigh 3 ffoot 4 fshin 5 fthigh 6 torso
    R phi[7], om[7];
    phi[6] = q[2], om[6] = v[2];
    phi[2] = phi[6] + q[3], om[2] = om[6] + v[3];
    phi[1] = phi[2] + q[4], om[1] = om[2] + v[4];
    phi[0] = phi[1] + q[5], om[0] = om[1] + v[5];
    phi[5] = phi[6] + q[6], om[5] = om[6] + v[6];
    phi[4] = phi[5] + q[7], om[4] = om[5] + v[7];
    phi[3] = phi[4] + q[8], om[3] = om[4] + v[8];
    R cs[7], sn[7];
#pragma unroll
    for (int b = 0; b < 7; ++b) sincos_ctx(trig, phi[b], sn[b], cs[b]);
    V2<R> S[7];
#pragma unroll
    for (int b = 0; b < 7; ++b) S[b] = rot(cs[b], sn[b], (R)kGeom.sx[b], (R)kGeom.sz[b]);
    // rotated link vectors
    const V2<R> Dtb = rot(cs[6], sn[6], (R)kGeom.d_tb[0], (R)kGeom.d_tb[1]);          // torso -> bthigh joint
    const V2<R> Dtf = rot(cs[6], sn[6], (R)kGeom.d_tf[0], (R)kGeom.d_tf[1]);          // torso -> fthigh joint
    const V2<R> Dbt = rot(cs[2], sn[2], (R)kGeom.d_bt_bs[0], (R)kGeom.d_bt_bs[1]);    // bthigh -> bshin joint
    const V2<R> Dbs = rot(cs[1], sn[1], (R)kGeom.d_bs_bf[0], (R)kGeom.d_bs_bf[1]);    // bshin -> bfoot joint
    const V2<R> Dft = rot(cs[5], sn[5], (R)kGeom.d_ft_fs[0], (R)kGeom.d_ft_fs[1]);    // fthigh -> fshin joint
    const V2<R> Dfs = rot(cs[4], sn[4], (R)kGeom.d_fs_ff[0], (R)kGeom.d_fs_ff[1]);    // fshin -> ffoot joint

    // ---- inertia matrix (lower triangle, permuted) and generalized forces in absolute coordinates
    R A[NV][NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < NV; ++j) A[i][j] = R(0);
#pragma unroll
    for (int b = 0; b < 7; ++b) A[b][b] = (R)kGeom.diag[b];
    A[P_X][P_X] = (R)kGeom.mtot, A[P_Z][P_Z] = (R)kGeom.mtot;
    // angle-angle couplings along each chain: (ancestor link vector towards the descendant) . S_descendant
    A[P_BSHIN][P_BFOOT] = dot(Dbs, S[0]);
    A[P_BTHIGH][P_BFOOT] = dot(Dbt, S[0]);
    A[P_BTHIGH][P_BSHIN] = dot(Dbt, S[1]);
    A[P_TORSO][P_BFOOT] = dot(Dtb, S[0]);
    A[P_TORSO][P_BSHIN] = dot(Dtb, S[1]);
    A[P_TORSO][P_BTHIGH] = dot(Dtb, S[2]);
    A[P_FSHIN][P_FFOOT] = dot(Dfs, S[3]);
    A[P_FTHIGH][P_FFOOT] = dot(Dft, S[3]);
    A[P_FTHIGH][P_FSHIN] = dot(Dft, S[4]);
    A[P_TORSO][P_FFOOT] = dot(Dtf, S[3]);
    A[P_TORSO][P_FSHIN] = dot(Dtf, S[4]);
    A[P_TORSO][P_FTHIGH] = dot(Dtf, S[5]);
#pragma unroll
    for (int b = 0; b < 7; ++b) {  // translation-angle: perp(S_b) = (S.z, -S.x)
        A[P_X][b] = S[b].z;
        A[P_Z][b] = -S[b].x;
    }
    // velocity-product (centripetal) and gravity forces, moved to the right-hand side
    R f[NV];
    R w2[7];
#pragma unroll
    for (int b = 0; b < 7; ++b) w2[b] = om[b] * om[b];
    {
        R fx = R(0), fz = R(0);
#pragma unroll
        for (int b = 0; b < 7; ++b) fx = fma_r(w2[b], S[b].x, fx), fz = fma_r(w2[b], S[b].z, fz);
        f[P_X] = fx;
        f[P_Z] = fz - (R)kGeom.mtot * (R)kGeom.gravity;
    }
    // Q_phi_i = g S_i.x + sum_{j != i, same chain} Omega_j^2 * (l_bj . perp(l_bi) summed over bodies)
    //   i ancestor of j:  S_j . perp(D_i->j);   j ancestor of i:  D_j->i . perp(S_i)
    const R g = (R)kGeom.gravity;
    f[P_BFOOT] = fma_r(g, S[0].x, w2[1] * dotperp(Dbs, S[0]) + w2[2] * dotperp(Dbt, S[0]) + w2[6] * dotperp(Dtb, S[0]));
    f[P_BSHIN] = fma_r(g, S[1].x, w2[0] * dotperp(S[0], Dbs) + w2[2] * dotperp(Dbt, S[1]) + w2[6] * dotperp(Dtb, S[1]));
    f[P_BTHIGH] = fma_r(g, S[2].x, w2[0] * dotperp(S[0], Dbt) + w2[1] * dotperp(S[1], Dbt) + w2[6] * dotperp(Dtb, S[2]));
    f[P_FFOOT] = fma_r(g, S[3].x, w2[4] * dotperp(Dfs, S[3]) + w2[5] * dotperp(Dft, S[3]) + w2[6] * dotperp(Dtf, S[3]));
    f[P_FSHIN] = fma_r(g, S[4].x, w2[3] * dotperp(S[3], Dfs) + w2[5] * dotperp(Dft, S[4]) + w2[6] * dotperp(Dtf, S[4]));
    f[P_FTHIGH] = fma_r(g, S[5].x, w2[3] * dotperp(S[3], Dft) + w2[4] * dotperp(S[4], Dft) + w2[6] * dotperp(Dtf, S[5]));
    f[P_TORSO] = fma_r(g, S[6].x,
                       w2[0] * dotperp(S[0], Dtb) + w2[1] * dotperp(S[1], Dtb) + w2[2] * dotperp(S[2], Dtb) +
                           w2[3] * dotperp(S[3], Dtf) + w2[4] * dotperp(S[4], Dtf) + w2[5] * dotperp(S[5], Dtf));

    // ---- joints: child link / parent link (permuted) for joints bthigh,bshin,bfoot,fthigh,fshin,ffoot
    constexpr int jc[6] = {P_BTHIGH, P_BSHIN, P_BFOOT, P_FTHIGH, P_FSHIN, P_FFOOT};
    constexpr int jp[6] = {P_TORSO, P_BTHIGH, P_BSHIN, P_TORSO, P_FTHIGH, P_FSHIN};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const R c = ctrl[k] < R(kCtrlLo) ? R(kCtrlLo) : (ctrl[k] > R(kCtrlHi) ? R(kCtrlHi) : ctrl[k]);   // ctrlrange +-1
        const R tau = (R)kGeom.gear[k] * c - (R)kGeom.stiff[k] * q[3 + k] - (R)kGeom.damp[k] * v[3 + k];
        f[jc[k]] += tau;
        f[jp[k]] -= tau;
        const R e = (R)kGeom.arm[k] + hd * (R)kGeom.damp[k];  // armature + implicit damping: M + h D on theta_k
        A[jc[k]][jc[k]] += e;
        A[jp[k]][jp[k]] += e;
        // the (child,parent) entry lives in the lower triangle at [max][min]
        const int hi_ = jc[k] > jp[k] ? jc[k] : jp[k], lo_ = jc[k] > jp[k] ? jp[k] : jc[k];
        A[hi_][lo_] -= e;
    }

    // M = L D L^T.  The acceleration is carried in "half-solved" form z = D^-1 L^-1 (f + sum J^T lambda):
    // for a constraint row J with y = L^-1 J^T, J M^-1 J^T = y . D^-1 y and J acc = y . z, so a constraint costs
    // one sparse forward substitution per row and no backward one; acc = L^-T z once at the end.
    R invd[NV];
    ldl_factor(A, invd);
    R z[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) z[i] = f[i];
    ldl_forward<0, true>(A, z);
#pragma unroll
    for (int i = 0; i < NV; ++i) z[i] *= invd[i];

    // ---- soft constraints, one Gauss-Seidel sweep (oracle/planar_oracle.c order) -------------------
    auto limit = [&](auto kc) __attribute__((always_inline)) {  // joint limit on theta_k
        constexpr int k = decltype(kc)::value, C = jc[k], P = jp[k];  // child index < parent index
        const R th = q[3 + k];
        // lo < hi: at most one side is violated, the smaller of the two distances is it (branch-free pick)
        const R dlo = th - (R)kGeom.lo[k], dhi = (R)kGeom.hi[k] - th;
        const bool lower = dlo < dhi;
        const R dist = lower ? dlo : dhi, J = lower ? R(1) : R(-1);
        if (dist < R(0)) {
            R y[NV], yd[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) y[i] = R(0);
            y[C] = J, y[P] = -J;
            ldl_forward<C, false>(A, y);
            R Aii = R(0), acur = R(0);
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (in_pat(C, i)) yd[i] = y[i] * invd[i], Aii = fma_r(y[i], yd[i], Aii), acur = fma_r(y[i], z[i], acur);
            const R imp = impedance(dist, (R)kGeom.l_dmin, (R)kGeom.l_dmax, (R)(1.0 / kGeom.l_width));
            const R aref = -(R)m.lB * (J * v[3 + k]) - (R)m.lK * imp * dist;
            const R Rr = div_r(R(1) - imp, imp) * Aii;
            const R force = div_r(aref - acur, Aii + Rr);
            if (force > R(0)) {
#pragma unroll
                for (int i = 0; i < NV; ++i)
                    if (in_pat(C, i)) z[i] = fma_r(yd[i], force, z[i]);
            }
        }
    };
    limit(std::integral_constant<int, 0>{}), limit(std::integral_constant<int, 1>{}), limit(std::integral_constant<int, 2>{});
    limit(std::integral_constant<int, 3>{}), limit(std::integral_constant<int, 4>{}), limit(std::integral_constant<int, 5>{});
    // body origins (world) for the contact points
    const V2<R> o_t = {q[0], (R)kGeom.z0 + q[1]};
    const V2<R> o_bt = {o_t.x + Dtb.x, o_t.z + Dtb.z}, o_bs = {o_bt.x + Dbt.x, o_bt.z + Dbt.z},
                o_bf = {o_bs.x + Dbs.x, o_bs.z + Dbs.z};
    const V2<R> o_ft = {o_t.x + Dtf.x, o_t.z + Dtf.z}, o_fs = {o_ft.x + Dft.x, o_ft.z + Dft.z},
                o_ff = {o_fs.x + Dfs.x, o_fs.z + Dfs.z};
    // velocities in absolute coordinates (permuted): u = (Omega_links..., xdot, zdot)
    R u[NV];
#pragma unroll
    for (int b = 0; b < 7; ++b) u[b] = om[b];
    u[P_X] = v[0], u[P_Z] = v[1];

    // one capsule end sphere of a body against the floor: LNK = permuted link of the body, `org` its origin,
    // (a1,v1)..(a3,v3) = (link, rotated link vector) of the ancestors on its chain
    auto contact = [&](int pt, auto lnk_c, V2<R> org, int a1, V2<R> v1, int a2, V2<R> v2, int a3, V2<R> v3)
                       __attribute__((always_inline)) {
        constexpr int LNK = decltype(lnk_c)::value;
        const V2<R> e = rot(cs[LNK], sn[LNK], (R)kGeom.geom_end[pt][0], (R)kGeom.geom_end[pt][1]);
        const R sz_ = org.z + e.z;
        const R dist = sz_ - (R)kGeom.radius;
        if (dist < R(0)) {
            // contact point midway between the surfaces: p = (s.x, dist/2); offset from the body origin
            const V2<R> r = {e.x, R(0.5) * dist - org.z};
            R Jx[NV], Jz[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) Jx[i] = R(0), Jz[i] = R(0);
            Jx[P_X] = R(1), Jz[P_Z] = R(1);
            // d p / d phi_j = perp(vector from link j's contribution): own link uses r, ancestors their link vector
            Jx[LNK] = r.z, Jz[LNK] = -r.x;
            if (a1 >= 0) Jx[a1] = v1.z, Jz[a1] = -v1.x;
            if (a2 >= 0) Jx[a2] = v2.z, Jz[a2] = -v2.x;
            if (a3 >= 0) Jx[a3] = v3.z, Jz[a3] = -v3.x;
            R vn = R(0), vt = R(0);
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (in_pat(LNK, i)) vn = fma_r(Jz[i], u[i], vn), vt = fma_r(Jx[i], u[i], vt);
            ldl_forward<LNK, false>(A, Jx);  // Jx, Jz now hold L^-1 J^T
            ldl_forward<LNK, false>(A, Jz);
            R dx[NV], dz[NV];
            R Ann = R(0), Att = R(0), Atn = R(0), an = R(0), at = R(0);
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (in_pat(LNK, i)) {
                    dx[i] = Jx[i] * invd[i], dz[i] = Jz[i] * invd[i];
                    Ann = fma_r(Jz[i], dz[i], Ann), Att = fma_r(Jx[i], dx[i], Att), Atn = fma_r(Jx[i], dz[i], Atn);
                    an = fma_r(Jz[i], z[i], an), at = fma_r(Jx[i], z[i], at);
                }
            const R imp = impedance(dist, (R)kGeom.c_dmin, (R)kGeom.c_dmax, (R)(1.0 / kGeom.c_width));
            const R k1 = div_r(R(1) - imp, imp);
            const R fn = div_r(-(R)m.cB * vn - (R)m.cK * imp * dist - an, Ann + k1 * Ann);
            if (fn > R(0)) {
                R ft = div_r(-(R)m.cB * vt - at - Atn * fn, Att + k1 * Att);
                const R lim = (R)kGeom.friction * fn;
                ft = ft > lim ? lim : (ft < -lim ? -lim : ft);
#pragma unroll
                for (int i = 0; i < NV; ++i)
                    if (in_pat(LNK, i)) z[i] = fma_r(dz[i], fn, fma_r(dx[i], ft, z[i]));
            }
        }
    };
    const V2<R> none = {R(0), R(0)};
    using std::integral_constant;
    // geom order: torso(0,1) head(2,3) bthigh(4,5) bshin(6,7) bfoot(8,9) fthigh(10,11) fshin(12,13) ffoot(14,15)
    contact(0, integral_constant<int, P_TORSO>{}, o_t, -1, none, -1, none, -1, none);
    contact(1, integral_constant<int, P_TORSO>{}, o_t, -1, none, -1, none, -1, none);
    contact(2, integral_constant<int, P_TORSO>{}, o_t, -1, none, -1, none, -1, none);
    contact(3, integral_constant<int, P_TORSO>{}, o_t, -1, none, -1, none, -1, none);
    contact(4, integral_constant<int, P_BTHIGH>{}, o_bt, P_TORSO, Dtb, -1, none, -1, none);
    contact(5, integral_constant<int, P_BTHIGH>{}, o_bt, P_TORSO, Dtb, -1, none, -1, none);
    contact(6, integral_constant<int, P_BSHIN>{}, o_bs, P_TORSO, Dtb, P_BTHIGH, Dbt, -1, none);
    contact(7, integral_constant<int, P_BSHIN>{}, o_bs, P_TORSO, Dtb, P_BTHIGH, Dbt, -1, none);
    contact(8, integral_constant<int, P_BFOOT>{}, o_bf, P_TORSO, Dtb, P_BTHIGH, Dbt, P_BSHIN, Dbs);
    contact(9, integral_constant<int, P_BFOOT>{}, o_bf, P_TORSO, Dtb, P_BTHIGH, Dbt, P_BSHIN, Dbs);
    contact(10, integral_constant<int, P_FTHIGH>{}, o_ft, P_TORSO, Dtf, -1, none, -1, none);
    contact(11, integral_constant<int, P_FTHIGH>{}, o_ft, P_TORSO, Dtf, -1, none, -1, none);
    contact(12, integral_constant<int, P_FSHIN>{}, o_fs, P_TORSO, Dtf, P_FTHIGH, Dft, -1, none);
    contact(13, integral_constant<int, P_FSHIN>{}, o_fs, P_TORSO, Dtf, P_FTHIGH, Dft, -1, none);
    contact(14, integral_constant<int, P_FFOOT>{}, o_ff, P_TORSO, Dtf, P_FTHIGH, Dft, P_FSHIN, Dfs);
    contact(15, integral_constant<int, P_FFOOT>{}, o_ff, P_TORSO, Dtf, P_FTHIGH, Dft, P_FSHIN, Dfs);

    R acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = z[i];
    ldl_backward(A, acc);

    // ---- back to joint coordinates
    qacc[0] = acc[P_X], qacc[1] = acc[P_Z], qacc[2] = acc[P_TORSO];
    qacc[3] = acc[P_BTHIGH] - acc[P_TORSO], qacc[4] = acc[P_BSHIN] - acc[P_BTHIGH], qacc[5] = acc[P_BFOOT] - acc[P_BSHIN];
    qacc[6] = acc[P_FTHIGH] - acc[P_TORSO], qacc[7] = acc[P_FSHIN] - acc[P_FTHIGH], qacc[8] = acc[P_FFOOT] - acc[P_FSHIN];
}

// y = A x for the symmetric matrix stored in the lower triangle with the pattern nz()
template <typename R>
__device__ __forceinline__ void sym_matvec(const R (&A)[NV][NV], const R (&x)[NV], R (&y)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        R acc = A[i][i] * x[i];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (j < i && nz(i, j)) acc = fma_r(A[i][j], x[j], acc);
            if (j > i && nz(j, i)) acc = fma_r(A[j][i], x[j], acc);
        }
        y[i] = acc;
    }
}

// Forward dynamics with MuJoCo's constraint formulation (see oracle/planar_oracle.c, header): the acceleration minimises
//     1/2 (a - a0)' M (a - a0) + sum_rows D/2 min(0, J a - aref)^2
// over one row per violated joint limit and the four pyramid edges J_n +- mu J_t, J_n, J_n of every contact point
// (the two edges along y have no motion to act on in a planar tree).  Newton's method with unit steps on the active
// set (piecewise-linear gradient with positive definite pieces: converges in <= 9 iterations over the oracle's random
// states; the oracle itself adds MuJoCo's exact line search — both end at the unique minimiser); H = M + sum D J J' has
// the fill pattern of M (a contact row lives on a root path), so the sparse LDL^T serves it.  Everything is carried in
// the absolute-angle coordinates of accel(); `hd` > 0: MuJoCo's Euler applies the joint damping implicitly AFTER the
// solve, (M + h B) qacc = qfrc_smooth + J' f = M a.
// `warm`: the minimiser of the previous RK4 stage evaluation of the same env-step (body_kernels.h:body_substep; Euler
// evaluations start cold — measured there): stage states share their active set almost always.
template <typename R>
struct NewtonWarm {
    R a[NV];
    bool valid;
};
template <typename R>
__device__ __forceinline__ void accel_newton(const R (&q)[NV], const R (&v)[NV], const R (&ctrl)[6], const Model& m, R hd,
                                             R (&qacc)[NV], const TrigCtx& trig, NewtonWarm<R>& warm) {
    EMEI_MARK(nw_trig);
    R phi[7], om[7];
    phi[6] = q[2], om[6] = v[2];
    phi[2] = phi[6] + q[3], om[2] = om[6] + v[3];
    phi[1] = phi[2] + q[4], om[1] = om[2] + v[4];
    phi[0] = phi[1] + q[5], om[0] = om[1] + v[5];
    phi[5] = phi[6] + q[6], om[5] = om[6] + v[6];
    phi[4] = phi[5] + q[7], om[4] = om[5] + v[7];
    phi[3] = phi[4] + q[8], om[3] = om[4] + v[8];
    R cs[7], sn[7];
#pragma unroll
    for (int b = 0; b < 7; ++b) sincos_ctx(trig, phi[b], sn[b], cs[b]);
    V2<R> S[7];
#pragma unroll
    for (int b = 0; b < 7; ++b) S[b] = rot(cs[b], sn[b], (R)kGeom.sx[b], (R)kGeom.sz[b]);
    const V2<R> Dtb = rot(cs[6], sn[6], (R)kGeom.d_tb[0], (R)kGeom.d_tb[1]);
    const V2<R> Dtf = rot(cs[6], sn[6], (R)kGeom.d_tf[0], (R)kGeom.d_tf[1]);
    const V2<R> Dbt = rot(cs[2], sn[2], (R)kGeom.d_bt_bs[0], (R)kGeom.d_bt_bs[1]);
    const V2<R> Dbs = rot(cs[1], sn[1], (R)kGeom.d_bs_bf[0], (R)kGeom.d_bs_bf[1]);
    const V2<R> Dft = rot(cs[5], sn[5], (R)kGeom.d_ft_fs[0], (R)kGeom.d_ft_fs[1]);
    const V2<R> Dfs = rot(cs[4], sn[4], (R)kGeom.d_fs_ff[0], (R)kGeom.d_fs_ff[1]);
    constexpr int jc[6] = {P_BTHIGH, P_BSHIN, P_BFOOT, P_FTHIGH, P_FSHIN, P_FFOOT};
    constexpr int jp[6] = {P_TORSO, P_BTHIGH, P_BSHIN, P_TORSO, P_FTHIGH, P_FSHIN};

    // inertia (lower triangle, permuted) with `e_k` added on the joint coordinate theta_k = phi_child - phi_parent
    auto build_inertia = [&](R (&A)[NV][NV], R hdamp) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < NV; ++j) A[i][j] = R(0);
#pragma unroll
        for (int b = 0; b < 7; ++b) A[b][b] = (R)kGeom.diag[b];
        A[P_X][P_X] = (R)kGeom.mtot, A[P_Z][P_Z] = (R)kGeom.mtot;
        A[P_BSHIN][P_BFOOT] = dot(Dbs, S[0]);
        A[P_BTHIGH][P_BFOOT] = dot(Dbt, S[0]);
        A[P_BTHIGH][P_BSHIN] = dot(Dbt, S[1]);
        A[P_TORSO][P_BFOOT] = dot(Dtb, S[0]);
        A[P_TORSO][P_BSHIN] = dot(Dtb, S[1]);
        A[P_TORSO][P_BTHIGH] = dot(Dtb, S[2]);
        A[P_FSHIN][P_FFOOT] = dot(Dfs, S[3]);
        A[P_FTHIGH][P_FFOOT] = dot(Dft, S[3]);
        A[P_FTHIGH][P_FSHIN] = dot(Dft, S[4]);
        A[P_TORSO][P_FFOOT] = dot(Dtf, S[3]);
        A[P_TORSO][P_FSHIN] = dot(Dtf, S[4]);
        A[P_TORSO][P_FTHIGH] = dot(Dtf, S[5]);
#pragma unroll
        for (int b = 0; b < 7; ++b) A[P_X][b] = S[b].z, A[P_Z][b] = -S[b].x;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const R e = (R)kGeom.arm[k] + hdamp * (R)kGeom.damp[k];
            A[jc[k]][jc[k]] += e;
            A[jp[k]][jp[k]] += e;
            const int hi_ = jc[k] > jp[k] ? jc[k] : jp[k], lo_ = jc[k] > jp[k] ? jp[k] : jc[k];
            A[hi_][lo_] -= e;
        }
    };

    EMEI_MARK(nw_forces);
    // smooth generalised forces in absolute coordinates (as accel())
    R f[NV], w2[7];
#pragma unroll
    for (int b = 0; b < 7; ++b) w2[b] = om[b] * om[b];
    {
        R fx = R(0), fz = R(0);
#pragma unroll
        for (int b = 0; b < 7; ++b) fx = fma_r(w2[b], S[b].x, fx), fz = fma_r(w2[b], S[b].z, fz);
        f[P_X] = fx;
        f[P_Z] = fz - (R)kGeom.mtot * (R)kGeom.gravity;
    }
    const R g = (R)kGeom.gravity;
    f[P_BFOOT] = fma_r(g, S[0].x, w2[1] * dotperp(Dbs, S[0]) + w2[2] * dotperp(Dbt, S[0]) + w2[6] * dotperp(Dtb, S[0]));
    f[P_BSHIN] = fma_r(g, S[1].x, w2[0] * dotperp(S[0], Dbs) + w2[2] * dotperp(Dbt, S[1]) + w2[6] * dotperp(Dtb, S[1]));
    f[P_BTHIGH] = fma_r(g, S[2].x, w2[0] * dotperp(S[0], Dbt) + w2[1] * dotperp(S[1], Dbt) + w2[6] * dotperp(Dtb, S[2]));
    f[P_FFOOT] = fma_r(g, S[3].x, w2[4] * dotperp(Dfs, S[3]) + w2[5] * dotperp(Dft, S[3]) + w2[6] * dotperp(Dtf, S[3]));
    f[P_FSHIN] = fma_r(g, S[4].x, w2[3] * dotperp(S[3], Dfs) + w2[5] * dotperp(Dft, S[4]) + w2[6] * dotperp(Dtf, S[4]));
    f[P_FTHIGH] = fma_r(g, S[5].x, w2[3] * dotperp(S[3], Dft) + w2[4] * dotperp(S[4], Dft) + w2[6] * dotperp(Dtf, S[5]));
    f[P_TORSO] = fma_r(g, S[6].x,
                       w2[0] * dotperp(S[0], Dtb) + w2[1] * dotperp(S[1], Dtb) + w2[2] * dotperp(S[2], Dtb) +
                           w2[3] * dotperp(S[3], Dtf) + w2[4] * dotperp(S[4], Dtf) + w2[5] * dotperp(S[5], Dtf));
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const R c = ctrl[k] < R(kCtrlLo) ? R(kCtrlLo) : (ctrl[k] > R(kCtrlHi) ? R(kCtrlHi) : ctrl[k]);
        const R tau = (R)kGeom.gear[k] * c - (R)kGeom.stiff[k] * q[3 + k] - (R)kGeom.damp[k] * v[3 + k];
        f[jc[k]] += tau;
        f[jp[k]] -= tau;
    }

    EMEI_MARK(nw_rows);
    // ---- which rows exist (geometry only: fixed during the solve): bits 0-5 joint limits, 6-21 contact points
    const V2<R> o_t = {q[0], (R)kGeom.z0 + q[1]};
    const V2<R> o_bt = {o_t.x + Dtb.x, o_t.z + Dtb.z}, o_bs = {o_bt.x + Dbt.x, o_bt.z + Dbt.z},
                o_bf = {o_bs.x + Dbs.x, o_bs.z + Dbs.z};
    const V2<R> o_ft = {o_t.x + Dtf.x, o_t.z + Dtf.z}, o_fs = {o_ft.x + Dft.x, o_ft.z + Dft.z},
                o_ff = {o_fs.x + Dfs.x, o_fs.z + Dfs.z};
    constexpr int pt_link[16] = {P_TORSO, P_TORSO, P_TORSO, P_TORSO, P_BTHIGH, P_BTHIGH, P_BSHIN, P_BSHIN,
                                 P_BFOOT, P_BFOOT, P_FTHIGH, P_FTHIGH, P_FSHIN, P_FSHIN, P_FFOOT, P_FFOOT};
    const V2<R> org_of[7] = {o_bf, o_bs, o_bt, o_ff, o_fs, o_ft, o_t};  // by permuted link
    uint32_t rows = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) rows |= ((q[3 + k] < (R)kGeom.lo[k]) | (q[3 + k] > (R)kGeom.hi[k])) ? (1u << k) : 0u;
#pragma unroll
    for (int pt = 0; pt < 16; ++pt) {
        const int L = pt_link[pt];
        const R ez = fma_r((R)kGeom.geom_end[pt][1], cs[L], -((R)kGeom.geom_end[pt][0] * sn[L]));
        rows |= (org_of[L].z + ez - (R)kGeom.radius < R(0)) ? (1u << (6 + pt)) : 0u;
    }

    // A lane with THREE row blocks borrows, for its third block, the second slot of a lane of its wave that leaves it unused (a
    // lane with at most one block, or none: 74 % of the lanes): same LDS, same layout, another column.  Wave-uniform scalar loop
    // over the few such lanes (3 per 1000); a lane without a donor keeps the primal loop.  (Round 4: the primal loop ran for 0.3 %
    // of the lanes but in 15.6 % of the wave evaluations, and cost 27 % of config 4: 7.75 -> 5.65 ms with the loop cut out.)
    int donor = -1;
    if (kDualSlots > 0 && trig.scratch != nullptr) {
        const int nb = __popc(rows);
        unsigned long long tri_m = __ballot(nb == 3);
        if (__builtin_expect(tri_m != 0ull, 0)) {  // 16 % of the wave evaluations; one or two iterations
            unsigned long long don_m = __ballot(nb <= 1);
            const int ln = (int)(threadIdx.x & (kWave - 1));
            while (tri_m != 0ull && don_m != 0ull) {
                if (ln == __ffsll(tri_m) - 1) donor = __ffsll(don_m) - 1;
                tri_m &= tri_m - 1ull, don_m &= don_m - 1ull;
            }
        }
    }
    EMEI_MARK(nw_direct);
    EMEI_STAT_LANE(0);
    EMEI_STAT_WAVE(7);
    R A[NV][NV], invd[NV], a[NV];
    if (rows == 0u) {  // free flight: qacc = (M + h B)^-1 qfrc_smooth
        if (hd > R(0)) {
            // Euler: solved together with the constrained lanes' damping step at the end (one factorisation of M + h B per
            // evaluation for the whole wave instead of one per branch: nearly every wave has lanes of both kinds)
#pragma unroll
            for (int i = 0; i < NV; ++i) a[i] = R(0);
        } else {
            build_inertia(A, R(0));
            ldl_factor(A, invd);
#pragma unroll
            for (int i = 0; i < NV; ++i) a[i] = f[i];
            ldl_forward<0, true>(A, a);
#pragma unroll
            for (int i = 0; i < NV; ++i) a[i] *= invd[i];
            ldl_backward(A, a);
        }
    } else {
        EMEI_MARK(nw_smooth0);
        EMEI_STAT_LANE(1);
        EMEI_STAT_LANE(25 + (__popc(rows) < 6 ? __popc(rows) : 6));  // 26..31: lanes with 1, 2, 3, 4, 5, >= 6 row blocks
        // the start of the iteration: the previous minimiser, else qacc_smooth = M^-1 qfrc_smooth (the dual path always
        // needs M's factor and qacc_smooth; a previous minimiser then only provides its first active set)
        const bool dual = kDualSlots > 0 && trig.scratch != nullptr && (__popc(rows) <= kDualSlots || donor >= 0);
        if (warm.valid && !dual) {
#pragma unroll
            for (int i = 0; i < NV; ++i) a[i] = warm.a[i];
        } else {
            build_inertia(A, R(0));
            ldl_factor(A, invd);
#pragma unroll
            for (int i = 0; i < NV; ++i) a[i] = f[i];
            ldl_forward<0, true>(A, a);
#pragma unroll
            for (int i = 0; i < NV; ++i) a[i] *= invd[i];
            ldl_backward(A, a);
        }
        R u[NV];  // velocities in absolute coordinates
#pragma unroll
        for (int b = 0; b < 7; ++b) u[b] = om[b];
        u[P_X] = v[0], u[P_Z] = v[1];
        R fmax = R(1);
#pragma unroll
        for (int i = 0; i < NV; ++i) fmax = fmax > fabs(f[i]) ? fmax : fabs(f[i]);

        [[maybe_unused]] int n_pass = 0;
        bool converged = false;  // the stopping rule was met (otherwise the loop ran into kMaxNewton: report_cap_hit)
        if (dual) {
            // ---- The same iteration in constraint space.  With M = L D L' and, per contact point p, Y_p = L^-1 (J_n, J_t)':
            //   a = a0 - L^-T D^-1 sum_p Y_p g_p,   g_p = W_p (u_p + b_p),   u_q = J_q a = u0_q - sum_p G_qp g_p,   G_qp = Y_q' D^-1 Y_p
            // where W_p (2x2, from the active pyramid edges) and b_p are what the row blocks of the primal loop below
            // accumulate into H and the gradient.  For a fixed active set this is the LINEAR system (I + W G) g = W (u0 + b)
            // of size 2 x slots: a pass costs ~160 instructions instead of ~740 (refactoring the 9x9 H), the iterates are
            // those of the unit-step Newton iteration (piecewise-quadratic cost: a Newton step IS the minimiser of the
            // current active set), and it ends when the set reproduces itself.  Y, u0, b live in LDS slots because which
            // points a lane has is only known at run time.
            EMEI_STAT_LANE(22);
            EMEI_STAT_WAVE(23);
            EMEI_MARK(dual_fill);
            R* const sl = (R*)trig.scratch + threadIdx.x;
            // slot 2 = slot 1 of the donor's column (this lane writes and reads it itself; the donor never touches its slot 1)
            R* const sl1 = sl + kSlotFields * trig.scratch_stride;
            R* const sl2 = (R*)trig.scratch + ((int)(threadIdx.x & ~(kWave - 1)) + (donor >= 0 ? donor : (int)(threadIdx.x & (kWave - 1)))) +
                           kSlotFields * trig.scratch_stride;
            // the block's slot: chosen ONCE per block (as a select inside every store it cost 5 800 cycles per evaluation)
            auto slot_base = [&](int sidx) __attribute__((always_inline)) { return sidx == 0 ? sl : (sidx == 1 ? sl1 : sl2); };
            auto put = [&](R* sp, int fld, R val) __attribute__((always_inline)) { sp[fld * trig.scratch_stride] = val; };
            auto get = [&](int sidx, int fld) __attribute__((always_inline)) { return sl[(sidx * kSlotFields + fld) * trig.scratch_stride]; };
            auto get2 = [&](int fld) __attribute__((always_inline)) { return sl2[fld * trig.scratch_stride]; };
            const R mu = (R)kGeom.friction;
            int slot = 0;
            // a violated joint limit is a slot with one direction: J = +-(e_C - e_P), no tangent, mu = 0 and D / 4 (with
            // mu = 0 the three edge tests below coincide and their weights add up to 4)
            auto dlimit = [&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value, C = jc[k], P = jp[k];
                if (rows & (1u << k)) {
                    R* const sp = slot_base(slot);
                    const R th = q[3 + k];
                    const bool lower = th < (R)kGeom.lo[k];
                    const R dist = lower ? th - (R)kGeom.lo[k] : (R)kGeom.hi[k] - th, J = lower ? R(1) : R(-1);
                    const R imp = impedance(dist, (R)kGeom.l_dmin, (R)kGeom.l_dmax, (R)(1.0 / kGeom.l_width));
                    const R aref = -(R)m.lB * (J * v[3 + k]) - (R)m.lK * imp * dist;
                    R Jn[NV];
#pragma unroll
                    for (int i = 0; i < NV; ++i) Jn[i] = R(0);
                    Jn[C] = J, Jn[P] = -J;
                    const R an = J * (a[C] - a[P]), wn = J * (warm.a[C] - warm.a[P]);
                    ldl_forward<C, false>(A, Jn);
#pragma unroll
                    for (int i = 0; i < NV; ++i) put(sp, i, in_pat(C, i) ? Jn[i] : R(0)), put(sp, NV + i, R(0));
                    put(sp, 18, an), put(sp, 19, R(0)), put(sp, 20, -aref), put(sp, 21, R(0));
                    put(sp, 22, R(0.25) * div_r(imp, (R(1) - imp) * (R)kInvW.dof[k]));
                    put(sp, 23, warm.valid ? wn : an), put(sp, 24, R(0)), put(sp, 25, R(0));
                    ++slot;
                }
            };
            dlimit(std::integral_constant<int, 0>{}), dlimit(std::integral_constant<int, 1>{}), dlimit(std::integral_constant<int, 2>{});
            dlimit(std::integral_constant<int, 3>{}), dlimit(std::integral_constant<int, 4>{}), dlimit(std::integral_constant<int, 5>{});
            auto dcontact = [&](int pt, auto lnk_c, V2<R> org, int a1, V2<R> v1, int a2, V2<R> v2, int a3, V2<R> v3)
                                __attribute__((always_inline)) {
                constexpr int LNK = decltype(lnk_c)::value;
                if (rows & (1u << (6 + pt))) {
                    R* const sp = slot_base(slot);
                    const V2<R> e = rot(cs[LNK], sn[LNK], (R)kGeom.geom_end[pt][0], (R)kGeom.geom_end[pt][1]);
                    const R dist = org.z + e.z - (R)kGeom.radius;
                    const V2<R> r = {e.x, R(0.5) * dist - org.z};
                    R Jx[NV], Jz[NV];
#pragma unroll
                    for (int i = 0; i < NV; ++i) Jx[i] = R(0), Jz[i] = R(0);
                    Jx[P_X] = R(1), Jz[P_Z] = R(1);
                    Jx[LNK] = r.z, Jz[LNK] = -r.x;
                    if (a1 >= 0) Jx[a1] = v1.z, Jz[a1] = -v1.x;
                    if (a2 >= 0) Jx[a2] = v2.z, Jz[a2] = -v2.x;
                    if (a3 >= 0) Jx[a3] = v3.z, Jz[a3] = -v3.x;
                    R vn = R(0), vt = R(0), an = R(0), at = R(0), wn = R(0), wt = R(0);
#pragma unroll
                    for (int i = 0; i < NV; ++i)
                        if (in_pat(LNK, i)) {
                            vn = fma_r(Jz[i], u[i], vn), vt = fma_r(Jx[i], u[i], vt);
                            an = fma_r(Jz[i], a[i], an), at = fma_r(Jx[i], a[i], at);
                            wn = fma_r(Jz[i], warm.a[i], wn), wt = fma_r(Jx[i], warm.a[i], wt);
                        }
                    const R imp = impedance(dist, (R)kGeom.c_dmin, (R)kGeom.c_dmax, (R)(1.0 / kGeom.c_width));
                    const R Dw = div_r(imp, (R(1) - imp) * (R)(2.0 * kGeom.friction * kGeom.friction * (1.0 + kGeom.friction * kGeom.friction)) *
                                                (R)kInvW.link[LNK]);
                    ldl_forward<LNK, false>(A, Jz);  // Y_n, Y_t
                    ldl_forward<LNK, false>(A, Jx);
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        put(sp, i, in_pat(LNK, i) ? Jz[i] : R(0));
                        put(sp, NV + i, in_pat(LNK, i) ? Jx[i] : R(0));
                    }
                    put(sp, 18, an), put(sp, 19, at);
                    put(sp, 20, (R)m.cB * vn + (R)m.cK * imp * dist), put(sp, 21, (R)m.cB * vt), put(sp, 22, Dw);
                    put(sp, 23, warm.valid ? wn : an), put(sp, 24, warm.valid ? wt : at), put(sp, 25, mu);
                    ++slot;
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            {
                const V2<R> none = {R(0), R(0)};
                using std::integral_constant;
                dcontact(0, integral_constant<int, P_TORSO>{}, o_t, -1, none, -1, none, -1, none);
                dcontact(1, integral_constant<int, P_TORSO>{}, o_t, -1, none, -1, none, -1, none);
                dcontact(2, integral_constant<int, P_TORSO>{}, o_t, -1, none, -1, none, -1, none);
                dcontact(3, integral_constant<int, P_TORSO>{}, o_t, -1, none, -1, none, -1, none);
                dcontact(4, integral_constant<int, P_BTHIGH>{}, o_bt, P_TORSO, Dtb, -1, none, -1, none);
                dcontact(5, integral_constant<int, P_BTHIGH>{}, o_bt, P_TORSO, Dtb, -1, none, -1, none);
                dcontact(6, integral_constant<int, P_BSHIN>{}, o_bs, P_TORSO, Dtb, P_BTHIGH, Dbt, -1, none);
                dcontact(7, integral_constant<int, P_BSHIN>{}, o_bs, P_TORSO, Dtb, P_BTHIGH, Dbt, -1, none);
                dcontact(8, integral_constant<int, P_BFOOT>{}, o_bf, P_TORSO, Dtb, P_BTHIGH, Dbt, P_BSHIN, Dbs);
                dcontact(9, integral_constant<int, P_BFOOT>{}, o_bf, P_TORSO, Dtb, P_BTHIGH, Dbt, P_BSHIN, Dbs);
                dcontact(10, integral_constant<int, P_FTHIGH>{}, o_ft, P_TORSO, Dtf, -1, none, -1, none);
                dcontact(11, integral_constant<int, P_FTHIGH>{}, o_ft, P_TORSO, Dtf, -1, none, -1, none);
                dcontact(12, integral_constant<int, P_FSHIN>{}, o_fs, P_TORSO, Dtf, P_FTHIGH, Dft, -1, none);
                dcontact(13, integral_constant<int, P_FSHIN>{}, o_fs, P_TORSO, Dtf, P_FTHIGH, Dft, -1, none);
                dcontact(14, integral_constant<int, P_FFOOT>{}, o_ff, P_TORSO, Dtf, P_FTHIGH, Dft, P_FSHIN, Dfs);
                dcontact(15, integral_constant<int, P_FFOOT>{}, o_ff, P_TORSO, Dtf, P_FTHIGH, Dft, P_FSHIN, Dfs);
            }
            if (slot <= 2) {
            EMEI_MARK(dual_gram);
            // slot data back (static slot index now); an absent second slot stays all zero: W_1 = 0, g_1 = 0
            R G00nn = R(0), G00nt = R(0), G00tt = R(0), G11nn = R(0), G11nt = R(0), G11tt = R(0);
            R G01nn = R(0), G01nt = R(0), G01tn = R(0), G01tt = R(0);
            R Y0n[NV], Y0t[NV], Y1n[NV], Y1t[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) Y0n[i] = get(0, i), Y0t[i] = get(0, NV + i), Y1n[i] = R(0), Y1t[i] = R(0);
            const R u0n0 = get(0, 18), u0t0 = get(0, 19), bn0 = get(0, 20), bt0 = get(0, 21), Dw0 = get(0, 22);
            const R mu0 = get(0, 25);
            R un0 = get(0, 23), ut0 = get(0, 24);
            R u0n1 = R(0), u0t1 = R(0), bn1 = R(0), bt1 = R(0), Dw1 = R(0), un1 = R(0), ut1 = R(0), mu1 = R(0);
            if (slot > 1) {
#pragma unroll
                for (int i = 0; i < NV; ++i) Y1n[i] = get(1, i), Y1t[i] = get(1, NV + i);
                u0n1 = get(1, 18), u0t1 = get(1, 19), bn1 = get(1, 20), bt1 = get(1, 21), Dw1 = get(1, 22);
                un1 = get(1, 23), ut1 = get(1, 24), mu1 = get(1, 25);
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const R d0n = invd[i] * Y0n[i], d0t = invd[i] * Y0t[i], d1n = invd[i] * Y1n[i], d1t = invd[i] * Y1t[i];
                G00nn = fma_r(d0n, Y0n[i], G00nn), G00nt = fma_r(d0n, Y0t[i], G00nt), G00tt = fma_r(d0t, Y0t[i], G00tt);
                G11nn = fma_r(d1n, Y1n[i], G11nn), G11nt = fma_r(d1n, Y1t[i], G11nt), G11tt = fma_r(d1t, Y1t[i], G11tt);
                G01nn = fma_r(d0n, Y1n[i], G01nn), G01nt = fma_r(d0n, Y1t[i], G01nt);
                G01tn = fma_r(d0t, Y1n[i], G01tn), G01tt = fma_r(d0t, Y1t[i], G01tt);
            }
            R g0n = R(0), g0t = R(0), g1n = R(0), g1t = R(0);
            // cold start (u = u0): g = 0 is the solution of the empty set, so if no edge is active there, a0 is the minimiser;
            // from a warm start the first pass always solves
            uint32_t used = warm.valid ? 0xffffffffu : 0u;
            EMEI_MARK(dual_loop);
#pragma unroll 1
            for (int it = 0; it < kMaxNewton; ++it) {
                ++n_pass;
                EMEI_STAT_LANE(2);
                EMEI_STAT_WAVE(3);
                // active pyramid edges at the current iterate (cheetah_model.h, primal row block: x1, x2, xn)
                const R xn0 = un0 + bn0, xt0 = mu0 * (ut0 + bt0), xn1 = un1 + bn1, xt1 = mu1 * (ut1 + bt1);
                const bool p0 = xn0 + xt0 < R(0), m0 = xn0 - xt0 < R(0), y0 = xn0 < R(0);
                const bool p1 = xn1 + xt1 < R(0), m1 = xn1 - xt1 < R(0), y1 = xn1 < R(0);
                const uint32_t flags = (p0 ? 1u : 0u) | (m0 ? 2u : 0u) | (y0 ? 4u : 0u) | (p1 ? 8u : 0u) | (m1 ? 16u : 0u) | (y1 ? 32u : 0u);
                if (flags == used) {  // the set the iterate was computed with reproduces itself: the minimiser
                    converged = true;
                    break;
                }
                used = flags;
                const R c10 = p0 ? R(1) : R(0), c20 = m0 ? R(1) : R(0), cy0 = y0 ? R(2) : R(0);
                const R c11 = p1 ? R(1) : R(0), c21 = m1 ? R(1) : R(0), cy1 = y1 ? R(2) : R(0);
                const R w0nn = Dw0 * (c10 + c20 + cy0), w0nt = Dw0 * mu0 * (c10 - c20), w0tt = Dw0 * mu0 * mu0 * (c10 + c20);
                const R w1nn = Dw1 * (c11 + c21 + cy1), w1nt = Dw1 * mu1 * (c11 - c21), w1tt = Dw1 * mu1 * mu1 * (c11 + c21);
                const R s0n = u0n0 + bn0, s0t = u0t0 + bt0, s1n = u0n1 + bn1, s1t = u0t1 + bt1;
                const R r0n = fma_r(w0nn, s0n, w0nt * s0t), r0t = fma_r(w0nt, s0n, w0tt * s0t);
                const R r1n = fma_r(w1nn, s1n, w1nt * s1t), r1t = fma_r(w1nt, s1n, w1tt * s1t);
                // P0 = I + W0 G00 (det >= 1: W0, G00 positive semidefinite), T = P0^-1 W0, y0v = P0^-1 r0
                const R p00 = R(1) + fma_r(w0nn, G00nn, w0nt * G00nt), p01 = fma_r(w0nn, G00nt, w0nt * G00tt);
                const R p10 = fma_r(w0nt, G00nn, w0tt * G00nt), p11 = R(1) + fma_r(w0nt, G00nt, w0tt * G00tt);
                const R id0 = rcp_r(fma_r(p00, p11, -(p01 * p10)));
                const R t00 = fma_r(p11, w0nn, -(p01 * w0nt)) * id0, t01 = fma_r(p11, w0nt, -(p01 * w0tt)) * id0;
                const R t10 = fma_r(p00, w0nt, -(p10 * w0nn)) * id0, t11 = fma_r(p00, w0tt, -(p10 * w0nt)) * id0;
                const R y0n = fma_r(p11, r0n, -(p01 * r0t)) * id0, y0t = fma_r(p00, r0t, -(p10 * r0n)) * id0;
                // E = T G01, C = G10 E, Q = I + W1 (G11 - C) (det >= 1: G11 - C is a Schur complement)
                const R e00 = fma_r(t00, G01nn, t01 * G01tn), e01 = fma_r(t00, G01nt, t01 * G01tt);
                const R e10 = fma_r(t10, G01nn, t11 * G01tn), e11 = fma_r(t10, G01nt, t11 * G01tt);
                const R h00 = G11nn - fma_r(G01nn, e00, G01tn * e10), h01 = G11nt - fma_r(G01nn, e01, G01tn * e11);
                const R h10 = G11nt - fma_r(G01nt, e00, G01tt * e10), h11 = G11tt - fma_r(G01nt, e01, G01tt * e11);
                const R q00 = R(1) + fma_r(w1nn, h00, w1nt * h10), q01 = fma_r(w1nn, h01, w1nt * h11);
                const R q10 = fma_r(w1nt, h00, w1tt * h10), q11 = R(1) + fma_r(w1nt, h01, w1tt * h11);
                const R d0 = fma_r(G01nn, y0n, G01tn * y0t), d1 = fma_r(G01nt, y0n, G01tt * y0t);
                const R k1n = r1n - fma_r(w1nn, d0, w1nt * d1), k1t = r1t - fma_r(w1nt, d0, w1tt * d1);
                const R id1 = rcp_r(fma_r(q00, q11, -(q01 * q10)));
                g1n = fma_r(q11, k1n, -(q01 * k1t)) * id1, g1t = fma_r(q00, k1t, -(q10 * k1n)) * id1;
                g0n = y0n - fma_r(e00, g1n, e01 * g1t), g0t = y0t - fma_r(e10, g1n, e11 * g1t);
                // u = u0 - G g
                un0 = u0n0 - (fma_r(G00nn, g0n, G00nt * g0t) + fma_r(G01nn, g1n, G01nt * g1t));
                ut0 = u0t0 - (fma_r(G00nt, g0n, G00tt * g0t) + fma_r(G01tn, g1n, G01tt * g1t));
                un1 = u0n1 - (fma_r(G01nn, g0n, G01tn * g0t) + fma_r(G11nn, g1n, G11nt * g1t));
                ut1 = u0t1 - (fma_r(G01nt, g0n, G01tt * g0t) + fma_r(G11nt, g1n, G11tt * g1t));
            }
            EMEI_MARK(dual_final);
            // a = a0 - L^-T D^-1 sum_p Y_p g_p (Y read again from the slots: 36 values are not worth holding through the loop)
            R z[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) z[i] = fma_r(get(0, i), g0n, get(0, NV + i) * g0t);
            if (slot > 1) {
#pragma unroll
                for (int i = 0; i < NV; ++i) z[i] += fma_r(get(1, i), g1n, get(1, NV + i) * g1t);
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) z[i] *= invd[i];
            ldl_backward(A, z);
#pragma unroll
            for (int i = 0; i < NV; ++i) a[i] -= z[i];
            } else {
                // ---- three row blocks: the same linear system (I + W G) g = W (u0 + b), now 6 x 6.  Block 0 is eliminated first,
                //   g0 = y0 - E01 g1 - E02 g2,  y0 = P0^-1 W0 s0,  E0j = P0^-1 W0 G0j,  P0 = I + W0 G00   (det P0 >= 1)
                // which leaves a system of the SAME form for blocks 1, 2 with the Schur complements G'ij = Gij - Gi0 E0j (symmetric:
                // P0^-1 W0 is) and right-hand sides W_i (s_i - Gi0 y0): solved by the two-block elimination above.
                EMEI_MARK(tri_gram);
                struct S2 { R nn, nt, tt; };      // symmetric 2 x 2
                struct G2 { R nn, nt, tn, tt; };  // general 2 x 2: rows = components (n, t) of the first block
                S2 Ga = {R(0), R(0), R(0)}, Gb = Ga, Gc = Ga;                     // G00, G11, G22
                G2 Gab = {R(0), R(0), R(0), R(0)}, Gac = Gab, Gbc = Gab;            // G01, G02, G12
#pragma unroll
                for (int i = 0; i < NV; ++i) {  // one row of the three Y pairs at a time: 21 accumulators, 6 values in flight
                    const R an = get(0, i), at = get(0, NV + i), bn = get(1, i), bt = get(1, NV + i), cn = get2(i), ct = get2(NV + i);
                    const R dan = invd[i] * an, dat = invd[i] * at, dbn = invd[i] * bn, dbt = invd[i] * bt, dcn = invd[i] * cn, dct = invd[i] * ct;
                    Ga.nn = fma_r(dan, an, Ga.nn), Ga.nt = fma_r(dan, at, Ga.nt), Ga.tt = fma_r(dat, at, Ga.tt);
                    Gb.nn = fma_r(dbn, bn, Gb.nn), Gb.nt = fma_r(dbn, bt, Gb.nt), Gb.tt = fma_r(dbt, bt, Gb.tt);
                    Gc.nn = fma_r(dcn, cn, Gc.nn), Gc.nt = fma_r(dcn, ct, Gc.nt), Gc.tt = fma_r(dct, ct, Gc.tt);
                    Gab.nn = fma_r(dan, bn, Gab.nn), Gab.nt = fma_r(dan, bt, Gab.nt), Gab.tn = fma_r(dat, bn, Gab.tn), Gab.tt = fma_r(dat, bt, Gab.tt);
                    Gac.nn = fma_r(dan, cn, Gac.nn), Gac.nt = fma_r(dan, ct, Gac.nt), Gac.tn = fma_r(dat, cn, Gac.tn), Gac.tt = fma_r(dat, ct, Gac.tt);
                    Gbc.nn = fma_r(dbn, cn, Gbc.nn), Gbc.nt = fma_r(dbn, ct, Gbc.nt), Gbc.tn = fma_r(dbt, cn, Gbc.tn), Gbc.tt = fma_r(dbt, ct, Gbc.tt);
                    // hipcc would hoist all 54 slot reads of the unrolled loop in front of it (108 registers the kernel does not
                    // have: scratch); a scheduling barrier per two rows keeps 12 in flight
                    if (i & 1) __builtin_amdgcn_sched_barrier(0);
                }
                // per block: the current u (n, t) in registers; u0, b, weight, friction stay in the slots and are read per pass (18
                // values that would otherwise be live across the whole loop: the kernel is at the register file's limit)
                auto fld = [&](int k, int f) __attribute__((always_inline)) { return k < 2 ? get(k, f) : get2(f); };
                R un[3], ut[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) un[k] = fld(k, 23), ut[k] = fld(k, 24);
                R gn[3] = {R(0), R(0), R(0)}, gt[3] = {R(0), R(0), R(0)};
                uint32_t used = warm.valid ? 0xffffffffu : 0u;
                EMEI_MARK(tri_loop);
#pragma unroll 1
                for (int it = 0; it < kMaxNewton; ++it) {
                    ++n_pass;
                    EMEI_STAT_LANE(2);
                    EMEI_STAT_WAVE(3);
                    uint32_t flags = 0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {  // active pyramid edges of block k at the current iterate
                        const R bn_k = fld(k, 20), bt_k = fld(k, 21), mu_k = fld(k, 25);
                        const R xn = un[k] + bn_k, xt = mu_k * (ut[k] + bt_k);
                        flags |= ((xn + xt < R(0) ? 1u : 0u) | (xn - xt < R(0) ? 2u : 0u) | (xn < R(0) ? 4u : 0u)) << (3 * k);
                    }
                    if (flags == used) {
                        converged = true;
                        break;
                    }
                    used = flags;
                    // weight matrix W_k of block k's active edges and r_k = W_k (u0_k + b_k): formed where they are consumed (every
                    // value kept live across the elimination is one the register file does not have)
                    auto weights = [&](int k, S2& W, R& rn, R& rt) __attribute__((always_inline)) {
                        const R Dw_k = fld(k, 22), mu_k = fld(k, 25);
                        const uint32_t fk = flags >> (3 * k);
                        const R c1 = (fk & 1u) ? R(1) : R(0), c2 = (fk & 2u) ? R(1) : R(0), cy = (fk & 4u) ? R(2) : R(0);
                        W.nn = Dw_k * (c1 + c2 + cy), W.nt = Dw_k * mu_k * (c1 - c2), W.tt = Dw_k * mu_k * mu_k * (c1 + c2);
                        const R sn_ = fld(k, 18) + fld(k, 20), st_ = fld(k, 19) + fld(k, 21);
                        rn = fma_r(W.nn, sn_, W.nt * st_), rt = fma_r(W.nt, sn_, W.tt * st_);
                    };
                    // block 0 out: P0 = I + W0 G00, T = P0^-1 W0, y0 = P0^-1 r0
                    R y0n, y0t;
                    G2 Eb, Ec;
                    {
                        S2 W0;
                        R r0n, r0t;
                        weights(0, W0, r0n, r0t);
                        const R p00 = R(1) + fma_r(W0.nn, Ga.nn, W0.nt * Ga.nt), p01 = fma_r(W0.nn, Ga.nt, W0.nt * Ga.tt);
                        const R p10 = fma_r(W0.nt, Ga.nn, W0.tt * Ga.nt), p11 = R(1) + fma_r(W0.nt, Ga.nt, W0.tt * Ga.tt);
                        const R id0 = rcp_r(fma_r(p00, p11, -(p01 * p10)));
                        const R t00 = fma_r(p11, W0.nn, -(p01 * W0.nt)) * id0, t01 = fma_r(p11, W0.nt, -(p01 * W0.tt)) * id0;
                        const R t10 = fma_r(p00, W0.nt, -(p10 * W0.nn)) * id0, t11 = fma_r(p00, W0.tt, -(p10 * W0.nt)) * id0;
                        y0n = fma_r(p11, r0n, -(p01 * r0t)) * id0, y0t = fma_r(p00, r0t, -(p10 * r0n)) * id0;
                        // E0j = T G0j (j = b, c)
                        Eb = G2{fma_r(t00, Gab.nn, t01 * Gab.tn), fma_r(t00, Gab.nt, t01 * Gab.tt), fma_r(t10, Gab.nn, t11 * Gab.tn), fma_r(t10, Gab.nt, t11 * Gab.tt)};
                        Ec = G2{fma_r(t00, Gac.nn, t01 * Gac.tn), fma_r(t00, Gac.nt, t01 * Gac.tt), fma_r(t10, Gac.nn, t11 * Gac.tn), fma_r(t10, Gac.nt, t11 * Gac.tt)};
                    }
                    // Schur complements: G'bb = Gbb - Gba Eb, G'cc = Gcc - Gca Ec, G'bc = Gbc - Gba Ec   (Gba = Gab')
                    const S2 Hb = {Gb.nn - fma_r(Gab.nn, Eb.nn, Gab.tn * Eb.tn), Gb.nt - fma_r(Gab.nn, Eb.nt, Gab.tn * Eb.tt),
                                   Gb.tt - fma_r(Gab.nt, Eb.nt, Gab.tt * Eb.tt)};
                    const S2 Hc = {Gc.nn - fma_r(Gac.nn, Ec.nn, Gac.tn * Ec.tn), Gc.nt - fma_r(Gac.nn, Ec.nt, Gac.tn * Ec.tt),
                                   Gc.tt - fma_r(Gac.nt, Ec.nt, Gac.tt * Ec.tt)};
                    const G2 Hbc = {Gbc.nn - fma_r(Gab.nn, Ec.nn, Gab.tn * Ec.tn), Gbc.nt - fma_r(Gab.nn, Ec.nt, Gab.tn * Ec.tt),
                                    Gbc.tn - fma_r(Gab.nt, Ec.nn, Gab.tt * Ec.tn), Gbc.tt - fma_r(Gab.nt, Ec.nt, Gab.tt * Ec.tt)};
                    // Gi0 y0 for the right-hand sides r'_i = r_i - W_i (Gi0 y0)
                    const R db_n = fma_r(Gab.nn, y0n, Gab.tn * y0t), db_t = fma_r(Gab.nt, y0n, Gab.tt * y0t);
                    const R dc_n = fma_r(Gac.nn, y0n, Gac.tn * y0t), dc_t = fma_r(Gac.nt, y0n, Gac.tt * y0t);
                    // the two-block elimination on (Hb, Hc, Hbc), as in the two-slot path: block b out, then block c
                    R ybn, ybt;
                    G2 F;
                    {
                        S2 W1;
                        R r1n, r1t;
                        weights(1, W1, r1n, r1t);
                        const R rbn = r1n - fma_r(W1.nn, db_n, W1.nt * db_t), rbt = r1t - fma_r(W1.nt, db_n, W1.tt * db_t);
                        const R q00 = R(1) + fma_r(W1.nn, Hb.nn, W1.nt * Hb.nt), q01 = fma_r(W1.nn, Hb.nt, W1.nt * Hb.tt);
                        const R q10 = fma_r(W1.nt, Hb.nn, W1.tt * Hb.nt), q11 = R(1) + fma_r(W1.nt, Hb.nt, W1.tt * Hb.tt);
                        const R id1 = rcp_r(fma_r(q00, q11, -(q01 * q10)));
                        const R v00 = fma_r(q11, W1.nn, -(q01 * W1.nt)) * id1, v01 = fma_r(q11, W1.nt, -(q01 * W1.tt)) * id1;
                        const R v10 = fma_r(q00, W1.nt, -(q10 * W1.nn)) * id1, v11 = fma_r(q00, W1.tt, -(q10 * W1.nt)) * id1;
                        ybn = fma_r(q11, rbn, -(q01 * rbt)) * id1, ybt = fma_r(q00, rbt, -(q10 * rbn)) * id1;
                        F = G2{fma_r(v00, Hbc.nn, v01 * Hbc.tn), fma_r(v00, Hbc.nt, v01 * Hbc.tt), fma_r(v10, Hbc.nn, v11 * Hbc.tn), fma_r(v10, Hbc.nt, v11 * Hbc.tt)};
                    }
                    {
                        S2 W2;
                        R r2n, r2t;
                        weights(2, W2, r2n, r2t);
                        const R rcn = r2n - fma_r(W2.nn, dc_n, W2.nt * dc_t), rct = r2t - fma_r(W2.nt, dc_n, W2.tt * dc_t);
                        const R k00 = Hc.nn - fma_r(Hbc.nn, F.nn, Hbc.tn * F.tn), k01 = Hc.nt - fma_r(Hbc.nn, F.nt, Hbc.tn * F.tt);
                        const R k10 = Hc.nt - fma_r(Hbc.nt, F.nn, Hbc.tt * F.tn), k11 = Hc.tt - fma_r(Hbc.nt, F.nt, Hbc.tt * F.tt);
                        const R m00 = R(1) + fma_r(W2.nn, k00, W2.nt * k10), m01 = fma_r(W2.nn, k01, W2.nt * k11);
                        const R m10 = fma_r(W2.nt, k00, W2.tt * k10), m11 = R(1) + fma_r(W2.nt, k01, W2.tt * k11);
                        const R e0 = fma_r(Hbc.nn, ybn, Hbc.tn * ybt), e1 = fma_r(Hbc.nt, ybn, Hbc.tt * ybt);
                        const R kcn = rcn - fma_r(W2.nn, e0, W2.nt * e1), kct = rct - fma_r(W2.nt, e0, W2.tt * e1);
                        const R id2 = rcp_r(fma_r(m00, m11, -(m01 * m10)));
                        gn[2] = fma_r(m11, kcn, -(m01 * kct)) * id2, gt[2] = fma_r(m00, kct, -(m10 * kcn)) * id2;
                    }
                    gn[1] = ybn - fma_r(F.nn, gn[2], F.nt * gt[2]), gt[1] = ybt - fma_r(F.tn, gn[2], F.tt * gt[2]);
                    gn[0] = y0n - (fma_r(Eb.nn, gn[1], Eb.nt * gt[1]) + fma_r(Ec.nn, gn[2], Ec.nt * gt[2]));
                    gt[0] = y0t - (fma_r(Eb.tn, gn[1], Eb.tt * gt[1]) + fma_r(Ec.tn, gn[2], Ec.tt * gt[2]));
                    // u = u0 - G g
                    un[0] = fld(0, 18) - (fma_r(Ga.nn, gn[0], Ga.nt * gt[0]) + fma_r(Gab.nn, gn[1], Gab.nt * gt[1]) + fma_r(Gac.nn, gn[2], Gac.nt * gt[2]));
                    ut[0] = fld(0, 19) - (fma_r(Ga.nt, gn[0], Ga.tt * gt[0]) + fma_r(Gab.tn, gn[1], Gab.tt * gt[1]) + fma_r(Gac.tn, gn[2], Gac.tt * gt[2]));
                    un[1] = fld(1, 18) - (fma_r(Gab.nn, gn[0], Gab.tn * gt[0]) + fma_r(Gb.nn, gn[1], Gb.nt * gt[1]) + fma_r(Gbc.nn, gn[2], Gbc.nt * gt[2]));
                    ut[1] = fld(1, 19) - (fma_r(Gab.nt, gn[0], Gab.tt * gt[0]) + fma_r(Gb.nt, gn[1], Gb.tt * gt[1]) + fma_r(Gbc.tn, gn[2], Gbc.tt * gt[2]));
                    un[2] = fld(2, 18) - (fma_r(Gac.nn, gn[0], Gac.tn * gt[0]) + fma_r(Gbc.nn, gn[1], Gbc.tn * gt[1]) + fma_r(Gc.nn, gn[2], Gc.nt * gt[2]));
                    ut[2] = fld(2, 19) - (fma_r(Gac.nt, gn[0], Gac.tt * gt[0]) + fma_r(Gbc.nt, gn[1], Gbc.tt * gt[1]) + fma_r(Gc.nt, gn[2], Gc.tt * gt[2]));
                }
                EMEI_MARK(tri_final);
                // a = a0 - L^-T D^-1 sum_p Y_p g_p
                R z[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    z[i] = fma_r(get(0, i), gn[0], get(0, NV + i) * gt[0]) + fma_r(get(1, i), gn[1], get(1, NV + i) * gt[1]) +
                           fma_r(get2(i), gn[2], get2(NV + i) * gt[2]);
                    if (i & 1) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < NV; ++i) z[i] *= invd[i];
                ldl_backward(A, z);
#pragma unroll
                for (int i = 0; i < NV; ++i) a[i] -= z[i];
            }
        } else {
        EMEI_STAT_LANE(24);
        EMEI_STAT_WAVE(25);
#pragma unroll 1
        for (int it = 0; it < kMaxNewton; ++it) {
            EMEI_MARK(nw_pass_base);
            R gr[NV];
            ++n_pass;
            EMEI_STAT_LANE(2);
            EMEI_STAT_WAVE(3);
            build_inertia(A, R(0));
            sym_matvec(A, a, gr);
#pragma unroll
            for (int i = 0; i < NV; ++i) gr[i] -= f[i];
            EMEI_MARK(nw_limits);
            // joint-limit rows
            auto limit = [&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value, C = jc[k], P = jp[k];
                if (rows & (1u << k)) {
                    EMEI_STAT_WAVE(6);
                    // opaque, as the contact rows below: a violated limit is rare (0.15 rows per pass of a wave in config 4),
                    // but its impedance / reference terms do not depend on the iterate, so hipcc would compute them for all
                    // limits of every evaluation ahead of the loop (~30 instructions each, two divisions; A/B on one box:
                    // 10.9 vs 11.25 ms per 100 steps of config 4)
                    R th = q[3 + k], vk = v[3 + k];
                    asm volatile("" : "+v"(th), "+v"(vk));
                    const bool lower = th < (R)kGeom.lo[k];
                    const R dist = lower ? th - (R)kGeom.lo[k] : (R)kGeom.hi[k] - th, J = lower ? R(1) : R(-1);
                    const R imp = impedance(dist, (R)kGeom.l_dmin, (R)kGeom.l_dmax, (R)(1.0 / kGeom.l_width));
                    const R aref = -(R)m.lB * (J * vk) - (R)m.lK * imp * dist;
                    const R x = J * (a[C] - a[P]) - aref;
                    if (x < R(0)) {
                        const R Dw = div_r(imp, (R(1) - imp) * (R)kInvW.dof[k]);  // 1 / R
                        const R t = Dw * x * J;
                        gr[C] += t, gr[P] -= t;
                        A[C][C] += Dw, A[P][P] += Dw;
                        A[P > C ? P : C][P > C ? C : P] -= Dw;
                    }
                }
            };
            limit(std::integral_constant<int, 0>{}), limit(std::integral_constant<int, 1>{}), limit(std::integral_constant<int, 2>{});
            limit(std::integral_constant<int, 3>{}), limit(std::integral_constant<int, 4>{}), limit(std::integral_constant<int, 5>{});
            EMEI_MARK(nw_contacts);
            // contact rows: the four edges of the pyramid of one capsule end sphere
            auto contact = [&](int pt, auto lnk_c, V2<R> org, int a1, V2<R> v1, int a2, V2<R> v2, int a3, V2<R> v3)
                               __attribute__((always_inline)) {
                constexpr int LNK = decltype(lnk_c)::value;
                if (rows & (1u << (6 + pt))) {
                    EMEI_STAT_WAVE(4);
                    EMEI_STAT_LANE(5);
                    // The geometry of a point does not change during the iteration, so hipcc would hoist all of it (16
                    // points x ~15 doubles) out of the Newton loop and spill ~1 KB per lane to scratch around every
                    // evaluation — measured: 3x the time of the whole solve.  An opaque copy of the link's sin / cos
                    // makes the block recompute instead (~40 instructions per active point and pass).
                    R csl = cs[LNK], snl = sn[LNK];
                    asm volatile("" : "+v"(csl), "+v"(snl));
                    const V2<R> e = rot(csl, snl, (R)kGeom.geom_end[pt][0], (R)kGeom.geom_end[pt][1]);
                    const R dist = org.z + e.z - (R)kGeom.radius;
                    const V2<R> r = {e.x, R(0.5) * dist - org.z};
                    R Jx[NV], Jz[NV];
#pragma unroll
                    for (int i = 0; i < NV; ++i) Jx[i] = R(0), Jz[i] = R(0);
                    Jx[P_X] = R(1), Jz[P_Z] = R(1);
                    Jx[LNK] = r.z, Jz[LNK] = -r.x;
                    if (a1 >= 0) Jx[a1] = v1.z, Jz[a1] = -v1.x;
                    if (a2 >= 0) Jx[a2] = v2.z, Jz[a2] = -v2.x;
                    if (a3 >= 0) Jx[a3] = v3.z, Jz[a3] = -v3.x;
                    R vn = R(0), vt = R(0), an = R(0), at = R(0);
#pragma unroll
                    for (int i = 0; i < NV; ++i)
                        if (in_pat(LNK, i)) {
                            vn = fma_r(Jz[i], u[i], vn), vt = fma_r(Jx[i], u[i], vt);
                            an = fma_r(Jz[i], a[i], an), at = fma_r(Jx[i], a[i], at);
                        }
                    const R mu = (R)kGeom.friction;
                    const R imp = impedance(dist, (R)kGeom.c_dmin, (R)kGeom.c_dmax, (R)(1.0 / kGeom.c_width));
                    // x_edge = J_edge a - aref_edge, aref_edge = -B (J_edge v) - K imp pos
                    const R xn = an + (R)m.cB * vn + (R)m.cK * imp * dist, xt = mu * (at + (R)m.cB * vt);
                    const R x1 = xn + xt, x2 = xn - xt;
                    const bool s1 = x1 < R(0), s2 = x2 < R(0), sy = xn < R(0);
                    if (s1 | s2 | sy) {
                        // R_edge = 2 mu^2 (1 - imp) / imp * invweight (1 + mu^2)
                        const R Dw = div_r(imp, (R(1) - imp) * (R)(2.0 * kGeom.friction * kGeom.friction * (1.0 + kGeom.friction * kGeom.friction)) *
                                                    (R)kInvW.link[LNK]);
                        const R c1 = s1 ? R(1) : R(0), c2 = s2 ? R(1) : R(0), cy = sy ? R(2) : R(0);
                        const R gn = Dw * (c1 * x1 + c2 * x2 + cy * xn), gt = Dw * mu * (c1 * x1 - c2 * x2);
                        const R wnn = Dw * (c1 + c2 + cy), wtt = Dw * mu * mu * (c1 + c2), wnt = Dw * mu * (c1 - c2);
#pragma unroll
                        for (int i = 0; i < NV; ++i)
                            if (in_pat(LNK, i)) {
                                gr[i] = fma_r(Jz[i], gn, fma_r(Jx[i], gt, gr[i]));
                                const R ux = fma_r(wtt, Jx[i], wnt * Jz[i]), uz = fma_r(wnt, Jx[i], wnn * Jz[i]);
#pragma unroll
                                for (int j = 0; j <= i; ++j)
                                    if (in_pat(LNK, j)) A[i][j] = fma_r(ux, Jx[j], fma_r(uz, Jz[j], A[i][j]));
                            }
                    }
                }
                // one row block at a time: without the fence hipcc interleaves the 16 blocks (and their geometry) for ILP
                // and the live ranges spill ~1 KB per lane to scratch
                __builtin_amdgcn_sched_barrier(0);
            };
            const V2<R> none = {R(0), R(0)};
            using std::integral_constant;
            contact(0, integral_constant<int, P_TORSO>{}, o_t, -1, none, -1, none, -1, none);
            contact(1, integral_constant<int, P_TORSO>{}, o_t, -1, none, -1, none, -1, none);
            contact(2, integral_constant<int, P_TORSO>{}, o_t, -1, none, -1, none, -1, none);
            contact(3, integral_constant<int, P_TORSO>{}, o_t, -1, none, -1, none, -1, none);
            contact(4, integral_constant<int, P_BTHIGH>{}, o_bt, P_TORSO, Dtb, -1, none, -1, none);
            contact(5, integral_constant<int, P_BTHIGH>{}, o_bt, P_TORSO, Dtb, -1, none, -1, none);
            contact(6, integral_constant<int, P_BSHIN>{}, o_bs, P_TORSO, Dtb, P_BTHIGH, Dbt, -1, none);
            contact(7, integral_constant<int, P_BSHIN>{}, o_bs, P_TORSO, Dtb, P_BTHIGH, Dbt, -1, none);
            contact(8, integral_constant<int, P_BFOOT>{}, o_bf, P_TORSO, Dtb, P_BTHIGH, Dbt, P_BSHIN, Dbs);
            contact(9, integral_constant<int, P_BFOOT>{}, o_bf, P_TORSO, Dtb, P_BTHIGH, Dbt, P_BSHIN, Dbs);
            contact(10, integral_constant<int, P_FTHIGH>{}, o_ft, P_TORSO, Dtf, -1, none, -1, none);
            contact(11, integral_constant<int, P_FTHIGH>{}, o_ft, P_TORSO, Dtf, -1, none, -1, none);
            contact(12, integral_constant<int, P_FSHIN>{}, o_fs, P_TORSO, Dtf, P_FTHIGH, Dft, -1, none);
            contact(13, integral_constant<int, P_FSHIN>{}, o_fs, P_TORSO, Dtf, P_FTHIGH, Dft, -1, none);
            contact(14, integral_constant<int, P_FFOOT>{}, o_ff, P_TORSO, Dtf, P_FTHIGH, Dft, P_FSHIN, Dfs);
            contact(15, integral_constant<int, P_FFOOT>{}, o_ff, P_TORSO, Dtf, P_FTHIGH, Dft, P_FSHIN, Dfs);

            EMEI_MARK(nw_conv);
            // a lane leaves when ITS gradient is down (its result does not depend on its wave-mates); the passes the
            // slower lanes still need skip every row block none of them has
            R gmax = R(0);
#pragma unroll
            for (int i = 0; i < NV; ++i) gmax = gmax > fabs(gr[i]) ? gmax : fabs(gr[i]);
            if (gmax <= R(sizeof(R) == 8 ? 1e-11 : 1e-5) * fmax) {
                converged = true;
                break;
            }
            EMEI_MARK(nw_step);
            ldl_factor(A, invd);
            ldl_forward<0, true>(A, gr);
#pragma unroll
            for (int i = 0; i < NV; ++i) gr[i] *= invd[i];
            ldl_backward(A, gr);
#pragma unroll
            for (int i = 0; i < NV; ++i) a[i] -= gr[i];
        }
        }  // primal loop / dual path
        EMEI_MARK(nw_final);
        EMEI_STAT_LANE(8 + (n_pass < 13 ? n_pass : 13));
        report_cap_hit(trig, !converged);
#pragma unroll
        for (int i = 0; i < NV; ++i) warm.a[i] = a[i];
        warm.valid = true;
    }
    EMEI_MARK(nw_euler);
    if (hd > R(0)) {
        // mj_EulerSkip for every lane: (M + h B) qacc = M a, as qacc = a - (M + h B)^-1 (h B a), for the lanes with rows;
        // free flight (a = 0 above): qacc = (M + h B)^-1 f = 0 - (M + h B)^-1 (-f)
        R rhs[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) rhs[i] = rows == 0u ? -f[i] : R(0);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const R t = hd * (R)kGeom.damp[k] * (a[jc[k]] - a[jp[k]]);
            rhs[jc[k]] += t, rhs[jp[k]] -= t;
        }
        build_inertia(A, hd);
        ldl_factor(A, invd);
        ldl_forward<0, true>(A, rhs);
#pragma unroll
        for (int i = 0; i < NV; ++i) rhs[i] *= invd[i];
        ldl_backward(A, rhs);
#pragma unroll
        for (int i = 0; i < NV; ++i) a[i] -= rhs[i];
    }
    EMEI_MARK(nw_out);
    qacc[0] = a[P_X], qacc[1] = a[P_Z], qacc[2] = a[P_TORSO];
    qacc[3] = a[P_BTHIGH] - a[P_TORSO], qacc[4] = a[P_BSHIN] - a[P_BTHIGH], qacc[5] = a[P_BFOOT] - a[P_BSHIN];
    qacc[6] = a[P_FTHIGH] - a[P_TORSO], qacc[7] = a[P_FSHIN] - a[P_FTHIGH], qacc[8] = a[P_FFOOT] - a[P_FSHIN];
}

}  // namespace cheetah

// ---------------------------------------------------------------------------------------------
// Body traits for body_kernels.h
// SOLVER: EMEI_SOLVER_NEWTON (MuJoCo's constraint formulation, converged) or EMEI_SOLVER_SWEEP1 (round 1's single sweep)
template <typename R, int SOLVER = EMEI_SOLVER_NEWTON>
struct CheetahBody {
    using real = R;
    using Model = cheetah::Model;
    static constexpr int kMinWavesPerEU = 1;
#ifndef EMEI_CHEETAH_UNROLL_RK4
#define EMEI_CHEETAH_UNROLL_RK4 1
#endif
    static constexpr bool kUnrollRK4 = EMEI_CHEETAH_UNROLL_RK4 != 0;
    static constexpr int kScratchPerLane = SOLVER == EMEI_SOLVER_NEWTON ? cheetah::kDualSlots * cheetah::kSlotFields : 0;
    static constexpr bool kHasCtrlCost = true;
    static constexpr bool kObsIsState = true;
    static constexpr bool kSpareReset = false;
    static constexpr bool kStreamOutputs = false;  // emei_device.h:store_body_out
    static constexpr int NS = 18, NO = 18, NA = 6;
    static Model make_model(double dt, const EnvParams& ep) {
        Model m = cheetah::cheetah_make_model(dt);
        m.w_forward = ep.get(EMEI_PARAM_FORWARD_REWARD_WEIGHT, 1.0), m.w_ctrl = ep.get(EMEI_PARAM_CTRL_COST_WEIGHT, 0.1);
        return m;
    }

    struct WarmNone {};
    using Warm = std::conditional_t<SOLVER == EMEI_SOLVER_SWEEP1, WarmNone, cheetah::NewtonWarm<R>>;
    __device__ __forceinline__ static void begin_stages(Warm&) {}
    __device__ __forceinline__ static void accel(const R (&q)[cheetah::NV], const R (&v)[cheetah::NV], const R (&ctrl)[NA],
                                                 const Model& m, R hd, R (&qacc)[cheetah::NV], const TrigCtx& trig, Warm& warm) {
        if constexpr (SOLVER == EMEI_SOLVER_SWEEP1) cheetah::accel(q, v, ctrl, m, hd, qacc, trig);
        else cheetah::accel_newton(q, v, ctrl, m, hd, qacc, trig, warm);
    }
    // obs = concat(qpos, qvel) (mujoco_env.py:153-155); reward half_cheetah.py:59-63 with step() semantics
    // (per env: w_f (x' - x)/dt_env - w_c sum a^2, dt_env = dt*freq_rate); terminal :65-67 (non-finite)
    __device__ __forceinline__ static void outputs(const R (&s)[NS], const R (&pre)[NS], const R (&ctrl)[NA], const Model& m,
                                                   int freq_rate, float (&o)[NO], R& rew, bool& term, const TrigCtx&) {
        R cost = R(0);
#pragma unroll
        for (int k = 0; k < NA; ++k) cost = fma_r(ctrl[k], ctrl[k], cost);
        rew = (R)m.w_forward * (s[0] - pre[0]) / ((R)m.dt * (R)freq_rate) - (R)m.w_ctrl * cost;
        bool fin = true;
#pragma unroll
        for (int k = 0; k < NS; ++k) fin &= finite_r(s[k]), o[k] = (float)s[k];
        term = !fin;
    }
    __device__ __forceinline__ static void init_base(R (&)[NS]) {}  // init_qpos = init_qvel = 0
    // the state of the padding lanes of a ragged last wave (body_kernels.h): far above the floor, at rest — no row ever, the
    // cheapest path through accel(), and a lane that never needs its second constraint slot
    __device__ __forceinline__ static void park(R (&s)[NS]) {
#pragma unroll
        for (int k = 0; k < NS; ++k) s[k] = R(0);
        s[1] = R(1e6);
    }
    __device__ __forceinline__ static void obs_of(const R (&s)[NS], double (&o)[NO], const Model&) {
#pragma unroll
        for (int k = 0; k < NS; ++k) o[k] = (double)s[k];
    }
    // half_cheetah.py:59-63 for one row (the reference's batch form sums np.square(action) over the WHOLE
    // batch, a quirk documented in DESIGN.md); float32 in, float64 arithmetic
    template <typename T>
    __device__ __forceinline__ static double ctrl_cost(const T* act) {  // this row's sum a^2, float64 (half_cheetah.py:61)
        double cost = 0.0;
#pragma unroll
        for (int k = 0; k < NA; ++k) cost += (double)act[k] * (double)act[k];
        return cost;
    }
    template <typename T>
    __device__ __forceinline__ static double batch_reward(const T* obs, const T* pre_obs, const T* act,
                                                          const Model& m, int freq_rate) {
        double cost = 0.0;
#pragma unroll
        for (int k = 0; k < NA; ++k) cost += (double)act[k] * (double)act[k];
        return m.w_forward * ((double)obs[0] - (double)pre_obs[0]) / (m.dt * freq_rate) - m.w_ctrl * cost;
    }
    template <typename T>
    __device__ __forceinline__ static bool batch_terminal(const T* obs, const Model&) {
        bool fin = true;
#pragma unroll
        for (int k = 0; k < NO; ++k) fin &= finite_r(obs[k]);
        return !fin;
    }
};

}  // namespace emei
