// hopper_model.h — device arithmetic of the Hopper (emei/envs/mujoco/hopper.py on mujoco_env.py;
// model emei/envs/mujoco/assets/hopper.xml): planar 4-link chain torso-thigh-leg-foot, 6 DoF
// (rootx, rootz, rooty, thigh, leg, foot), 3 motors, capsule/floor contacts.
//
// Parity with libmujoco is UNPINNED (no MuJoCo in the image).  The CPU oracle
// (oracle/planar_oracle.c) restates the model with MuJoCo's own algorithms on a generic tree table
// (recursive Newton-Euler, dense factorisation); this file is an independent formulation for the
// GPU, the single-chain case of cheetah_model.h:
//   * ABSOLUTE link angles phi_L as velocity coordinates u = (Omega_foot, Omega_leg, Omega_thigh,
//     Omega_torso, xdot, zdot) — leaves first, so the LDL^T runs down the chain.  Inertia:
//     M[phi_i, phi_i] = const, M[phi_i, phi_j] = D_i . S_j (i ancestor of j), M[x|z, phi_j] =
//     perp(S_j), with S_j = R(phi_j) s_j the rotated mass-moment vector of link j (+ everything it
//     carries) and D_i = R(phi_i) d_i link i's vector to its child joint; velocity-product forces
//     are Omega_j^2 times the same rotated vectors: 4 sincos per forward-dynamics evaluation.
//   * the three leg hinges turn about -y (hopper.xml:21,25,29): theta_k = -(phi_child - phi_parent),
//     so a joint torque tau_k enters as -tau_k on phi_child and +tau_k on phi_parent.
//   * constraints (3 joint limits, 8 capsule-end/floor contact points with margin and friction, 3 capsule-capsule pairs):
//     MuJoCo's primal formulation solved by Newton's method (accel_newton, the default) or round 1's single Gauss-Seidel
//     sweep (accel), exactly as in cheetah_model.h.
//   * the geoms collide with EACH OTHER (hopper.xml:5: contype = conaffinity = 1 for every geom) wherever their bodies are
//     not parent and child: torso-leg, torso-foot, thigh-foot.  Both geoms have condim 1, so each pair is ONE frictionless
//     row along the line between the closest points of the two capsule axes (capsule_pair below).
// The reference steps this env with RK4 by default (hopper.py:22): body_kernels.h:body_substep.
#pragma once
#include <cmath>
#include <cstring>
#include <type_traits>

#include "cheetah_model.h"  // V2, rot, dot, dotperp, impedance
#include "emei_device.h"

namespace emei {
namespace hopper {

constexpr int NV = 6, NL = 4;
enum { L_FOOT = 0, L_LEG = 1, L_THIGH = 2, L_TORSO = 3, P_X = 4, P_Z = 5 };
// geom g (XML order torso, thigh, leg, foot: xml:18,22,26,30) lies on chain link kGeomLink[g]: make_model, the contact rows and the
// constants export all read this table
constexpr int kGeomLink[4] = {L_TORSO, L_THIGH, L_LEG, L_FOOT};

struct Model {
    double sx[NL], sz[NL], diag[NL];  // mass-moment vector (link frame) and constant diagonal inertia per link
    double d[NL][2];                  // link L's vector to the joint of link L-1 (link frame); d[0] unused
    double mtot, gravity, z0;         // z0: world z of the torso origin at qpos[1] = 0 (body z 1.25 - ref 1.25)
    double damp[3], arm[3], lo[3], hi[3], gear[3];  // joints thigh, leg, foot
    double geom_end[8][2];            // capsule end-sphere centres, link frame; geom order torso, thigh, leg, foot
    double radius[4], friction[4], margin;
    double cK, cB, c_dmin, c_dmax, c_width;  // contact solref (refsafe'd for dt) / solimp
    double lK, lB, l_dmin, l_dmax, l_width;  // joint-limit solref / solimp
    double dt;
    // reward / health parameters (hopper.py:25-31), run-time
    double w_forward, w_ctrl, healthy_reward, st_lo, st_hi, z_lo, z_hi;
    int32_t terminate_when_unhealthy;
};

using cheetah::dot;
using cheetah::dotperp;
using cheetah::impedance;
using cheetah::rot;
using cheetah::V2;

// rot() of a LITERAL link-frame vector: a zero component costs nothing and a unit axis is (sin, cos) itself — hipcc does not
// fold `0 * c` (NaN / inf semantics), and the hopper's capsules all lie along their link's z or x axis.  Same values as rot().
template <typename R>
__device__ __forceinline__ V2<R> rot_lit(R c, R s, double ax, double az) {
    if (ax == 0.0 && az == 1.0) return V2<R>{s, c};
    if (ax == 1.0 && az == 0.0) return V2<R>{c, -s};
    if (ax == 0.0 && az == 0.0) return V2<R>{R(0), R(0)};
    if (ax == 0.0) return V2<R>{(R)az * s, (R)az * c};
    if (az == 0.0) return V2<R>{(R)ax * c, -((R)ax * s)};
    return rot(c, s, (R)ax, (R)az);
}

// model constants from assets/hopper.xml (coordinate="global", degrees, inertiafromgeom, density 1000),
// evaluated at compile time like the cheetah's (cheetah_model.h:kGeom)
// the XML-level tables (hand-typed from hopper.xml; pinned to the file by tests/test_model_constants.py through
// emei_model_constants) and the capsule masses / inertias MuJoCo's compiler derives (inertiafromgeom, density 1000)
constexpr double kCtrlLo = -1.0, kCtrlHi = 1.0;  // motors: ctrlrange="-1.0 1.0" (xml:37-39)
// The three leg hinges turn about -y (axis="0 -1 0", xml:21,25,29): theta_k = kHingeSign * (phi_child - phi_parent).  Every place
// the dynamics use that relation goes through hs() (exact: a sign flip the compiler folds into its neighbour), and
// emei_model_constants exports THIS constant, so the XML pin (tests/test_model_constants.py) sees the sign the code runs with.
constexpr double kHingeSign = -1.0;
template <typename T>
__host__ __device__ constexpr T hs(T x) {
    return kHingeSign < 0 ? -x : x;
}
constexpr double kJointStiffness[3] = {0.0, 0.0, 0.0};  // no `stiffness` on the leg joints (xml:5,21,25,29); a non-zero entry adds -k theta
constexpr double kSolrefTc = 0.02;               // geom solref ".02 1" (xml:6); joint limits: MuJoCo's default (.02 1)
struct HopperLinks {
    // links in chain order foot, leg, thigh, torso.  Link frames sit at the joint anchors (world at qpos0):
    // foot (0,.1) xml:29, leg (0,.6) :25, thigh (0,1.05) :21, torso (0,1.25) :17
    double half[NL], rad[NL];                // capsules :30,26,22,18
    cheetah::cheetah_host::H2 gc[NL];        // capsule centres, link frame
    double gang[NL];                         // capsule axis: +z rotated by this angle about y (the foot lies along x)
    cheetah::cheetah_host::H2 dvec[NL];      // link L's vector to the joint of link L-1
    double torso_z, z_ref;                   // body pos z and rootz ref (:14,16)
    double mass[NL], inertia[NL];
};
constexpr HopperLinks hopper_links() {
    using namespace cheetah::cheetah_host;
    HopperLinks K{{0.195, 0.25, 0.225, 0.2}, {0.06, 0.04, 0.05, 0.05}, {{0.065, 0}, {0, -0.25}, {0, -0.225}, {0, 0}},
                  {M_PI / 2, 0, 0, 0}, {{0, 0}, {0, -0.5}, {0, -0.45}, {0, -0.2}}, 1.25, 1.25, {}, {}};
    const double rho = 1000.0;
    for (int b = 0; b < NL; ++b) {
        K.mass[b] = capsule_mass(rho, K.rad[b], K.half[b]);
        K.inertia[b] = capsule_inertia_perp(rho, K.rad[b], K.half[b]);
    }
    return K;
}

// model constants from assets/hopper.xml (coordinate="global", degrees, inertiafromgeom, density 1000),
// evaluated at compile time like the cheetah's (cheetah_model.h:kGeom)
constexpr Model make_model(double dt) {
    using namespace cheetah::cheetah_host;
    Model m{};
    const double deg = M_PI / 180.0;
    const HopperLinks K = hopper_links();
    const double(&half)[NL] = K.half;
    const double(&rad)[NL] = K.rad;
    const H2(&gc)[NL] = K.gc;
    const double(&gang)[NL] = K.gang;
    const H2(&dvec)[NL] = K.dvec;
    const double(&mass)[NL] = K.mass;
    const double(&inertia)[NL] = K.inertia;
    double sub[NL] = {};
    for (int b = 0; b < NL; ++b) sub[b] = mass[b] + (b ? sub[b - 1] : 0.0);  // link b carries links 0..b-1
    for (int b = 0; b < NL; ++b) {
        const double carried = b ? sub[b - 1] : 0.0;
        m.sx[b] = mass[b] * gc[b].x + carried * dvec[b].x, m.sz[b] = mass[b] * gc[b].z + carried * dvec[b].z;
        m.diag[b] = inertia[b] + mass[b] * (gc[b].x * gc[b].x + gc[b].z * gc[b].z) +
                    carried * (dvec[b].x * dvec[b].x + dvec[b].z * dvec[b].z);
        m.d[b][0] = dvec[b].x, m.d[b][1] = dvec[b].z;
    }
    m.mtot = sub[NL - 1], m.gravity = 9.81, m.z0 = K.torso_z - K.z_ref;  // body pos z 1.25, rootz ref 1.25 (:16)
    const double lo[3] = {-150 * deg, -150 * deg, -45 * deg}, hi[3] = {0, 0, 45 * deg};  // :21,25,29
    for (int k = 0; k < 3; ++k) m.damp[k] = 1.0, m.arm[k] = 1.0, m.lo[k] = lo[k], m.hi[k] = hi[k], m.gear[k] = 200.0;  // :5,37-39
    for (int g = 0; g < 4; ++g) {
        const int b = kGeomLink[g];
        const H2 ax = hrot(gang[b], {0, 1});
        m.geom_end[2 * g][0] = gc[b].x - half[b] * ax.x, m.geom_end[2 * g][1] = gc[b].z - half[b] * ax.z;
        m.geom_end[2 * g + 1][0] = gc[b].x + half[b] * ax.x, m.geom_end[2 * g + 1][1] = gc[b].z + half[b] * ax.z;
        m.radius[g] = rad[b];
        m.friction[g] = g == 3 ? 2.0 : 1.0;  // max(floor 1.0, geom .9 | 2.0) (:18,22,26,30)
    }
    m.margin = 0.001;  // :6
    const double tc = kSolrefTc < 2 * dt ? 2 * dt : kSolrefTc;  // solref (.02 1), refsafe
    m.c_dmin = 0.8, m.c_dmax = 0.8, m.c_width = 0.01;  // geom solimp (.8 .8 .01) :6
    m.l_dmin = 0.9, m.l_dmax = 0.95, m.l_width = 0.001;  // MuJoCo's joint-limit defaults
    m.cK = 1.0 / (m.c_dmax * m.c_dmax * tc * tc), m.cB = 2.0 / (m.c_dmax * tc);
    m.lK = 1.0 / (m.l_dmax * m.l_dmax * tc * tc), m.lB = 2.0 / (m.l_dmax * tc);
    m.dt = dt;
    return m;
}

// every dt-independent constant of the model, as compile-time immediates for the device code
__device__ constexpr Model kGeom = make_model(0.002);

// Impedance of a contact row.  hopper.xml:5 gives solimp (.8 .8 .01): dmin = dmax, so d(r) = 0.8 whatever the penetration — a
// compile-time constant (the oracle's impedance() returns dmin + y * 0 = 0.8 for every finite r), and with it every contact
// row's weight D = d / ((1 - d) diagApprox).  A non-finite r still poisons aref through K d r, as before.
template <typename R>
__device__ __forceinline__ R contact_impedance(R pos) {
    if constexpr (kGeom.c_dmin == kGeom.c_dmax && kGeom.c_dmin > 1e-4 && kGeom.c_dmin < 0.9999) return (R)kGeom.c_dmin;
    else return impedance(pos, (R)kGeom.c_dmin, (R)kGeom.c_dmax, (R)(1.0 / kGeom.c_width));
}
// imp / ((1 - imp) * diag): a literal when the impedance is one
template <typename R>
__device__ __forceinline__ R contact_weight(R imp, double diag) {
    if constexpr (kGeom.c_dmin == kGeom.c_dmax && kGeom.c_dmin > 1e-4 && kGeom.c_dmin < 0.9999)
        return (R)(kGeom.c_dmin / ((1.0 - kGeom.c_dmin) * diag));
    else return div_r(imp, (R(1) - imp) * (R)diag);
}

// emei_model_constants (include/emei_hip.h), XML body order torso, thigh, leg, foot = chain order reversed
inline int xml_constants(double* out) {
    constexpr HopperLinks K = hopper_links();
    constexpr Model m = make_model(0.002);
    int n = 0;
    out[n++] = m.gravity;
    for (int b = NL - 1; b >= 0; --b) {  // body pos: the torso in the world, a child at its parent's vector to the child joint
        const double px = b == NL - 1 ? 0.0 : K.dvec[b + 1].x, pz = b == NL - 1 ? K.torso_z : K.dvec[b + 1].z;
        const double row[6] = {K.mass[b], K.gc[b].x, K.gc[b].z, K.inertia[b], px, pz};
        for (double v : row) out[n++] = v;
    }
    for (int g = 0; g < 4; ++g) {  // geom order torso, thigh, leg, foot; body index in XML order
        const double row[7] = {(double)(L_TORSO - kGeomLink[g]) /* body index in XML order */, m.geom_end[2 * g][0], m.geom_end[2 * g][1], m.geom_end[2 * g + 1][0], m.geom_end[2 * g + 1][1],
                               m.radius[g], m.friction[g]};
        for (double v : row) out[n++] = v;
    }
    for (int k = 0; k < 3; ++k) {
        const double row[6] = {kJointStiffness[k], m.damp[k], m.arm[k], m.lo[k], m.hi[k], m.gear[k]};
        for (double v : row) out[n++] = v;
    }
    const double tail[13] = {m.margin, kSolrefTc, m.c_dmin, m.c_dmax, m.c_width, kSolrefTc, m.l_dmin, m.l_dmax, m.l_width, kCtrlLo, kCtrlHi,
                             K.z_ref, kHingeSign};
    for (double v : tail) out[n++] = v;
    return n;
}

// inverse weights at qpos0 (see cheetah_model.h:cheetah_invweights): joints thigh, leg, foot; links in chain order foot, leg,
// thigh, torso.  Derived at compile time from THIS file's absolute-angle inertia at qpos0 (all link angles 0), not from the
// oracle; compared with the oracle's joint-space derivation by tests/test_oracle_solver.py through emei_model_invweights.
struct InvWeights {
    double dof[3], link[NL];
};
constexpr InvWeights hopper_invweights() {
    const Model m = make_model(0.002);
    const HopperLinks K = hopper_links();
    double A[NV][NV] = {};
    for (int i = 0; i < NL; ++i) {
        A[i][i] = m.diag[i], A[P_X][i] = A[i][P_X] = m.sz[i], A[P_Z][i] = A[i][P_Z] = -m.sx[i];
        for (int j = 0; j < i; ++j) A[i][j] = A[j][i] = m.d[i][0] * m.sx[j] + m.d[i][1] * m.sz[j];  // j is carried by i
    }
    A[P_X][P_X] = A[P_Z][P_Z] = m.mtot;
    const int jc[3] = {L_THIGH, L_LEG, L_FOOT}, jp[3] = {L_TORSO, L_THIGH, L_LEG};
    for (int k = 0; k < 3; ++k) {
        A[jc[k]][jc[k]] += m.arm[k], A[jp[k]][jp[k]] += m.arm[k];
        A[jc[k]][jp[k]] -= m.arm[k], A[jp[k]][jc[k]] -= m.arm[k];
    }
    InvWeights w{};
    for (int k = 0; k < 3; ++k) {
        double J[NV] = {};
        J[jc[k]] = kHingeSign, J[jp[k]] = -kHingeSign;  // theta_k = kHingeSign (phi_child - phi_parent)
        w.dof[k] = ce::spd_quad<NV>(A, J);
    }
    for (int b = 0; b < NL; ++b) {
        double Jx[NV] = {}, Jz[NV] = {};
        Jx[P_X] = 1, Jz[P_Z] = 1;
        Jx[b] = K.gc[b].z, Jz[b] = -K.gc[b].x;                             // the link's own com: its capsule centre
        for (int a = b + 1; a < NL; ++a) Jx[a] = m.d[a][1], Jz[a] = -m.d[a][0];  // ancestors: perp(their link vector)
        w.link[b] = (ce::spd_quad<NV>(A, Jx) + ce::spd_quad<NV>(A, Jz)) / 3.0;
    }
    return w;
}
__device__ constexpr InvWeights kInvW = hopper_invweights();
// emei_model_invweights: dof_invweight0 of thigh, leg, foot, then body_invweight0 in XML body order torso, thigh, leg, foot
inline int xml_invweights(double* out) {
    constexpr InvWeights w = hopper_invweights();
    int n = 0;
    for (int k = 0; k < 3; ++k) out[n++] = w.dof[k];
    for (int b = NL - 1; b >= 0; --b) out[n++] = w.link[b];
    return n;
}

// ---- capsule against capsule.  Pair index: 0 torso-leg, 1 torso-foot, 2 thigh-foot (geom order torso, thigh, leg, foot:
// the pairs whose bodies are not parent and child, in MuJoCo's (geom1 < geom2) order).
constexpr int kNumPairs = 3;
constexpr int kPairGeom[kNumPairs][2] = {{0, 2}, {0, 3}, {1, 3}};
// a capsule's axis segment in its link frame: centre, unit direction, half length
struct CapsuleAxis {
    double cx, cz, ax, az, half;
};
constexpr CapsuleAxis capsule_axis(int g) {
    const Model m = make_model(0.002);
    const double hx = 0.5 * (m.geom_end[2 * g + 1][0] - m.geom_end[2 * g][0]), hz = 0.5 * (m.geom_end[2 * g + 1][1] - m.geom_end[2 * g][1]);
    const double len = ce::sqrt(hx * hx + hz * hz);
    return {0.5 * (m.geom_end[2 * g + 1][0] + m.geom_end[2 * g][0]), 0.5 * (m.geom_end[2 * g + 1][1] + m.geom_end[2 * g][1]), hx / len,
            hz / len, len};
}
// Closest points of the two axis segments the way MuJoCo's capsule-capsule collider finds them (x1, x2 = signed distances
// from the centres along the unit axes: minimise over the two lines, clamp x1, re-solve x2 and clamp, re-solve x1 if x2
// was clamped), then sphere against sphere: dist = |c2 - c1| - r1 - r2, normal n from geom 1 to geom 2, contact point p midway
// between the surfaces.  True while dist < margin.  Coincident closest points (crossing axes) -> n = (1, 0) like MuJoCo's
// sphere-sphere fallback; (nearly) parallel axes: the determinant is floored, which sends x1 to an end of its segment
// (MuJoCo emits up to two contacts for exactly parallel capsules: outside the joint ranges for these pairs, not reproduced).
template <int PAIR, typename R>
__device__ __forceinline__ bool capsule_pair(const R (&cs)[NL], const R (&sn)[NL], const V2<R> (&org)[NL], R& dist, V2<R>& n, V2<R>& p) {
    constexpr int G1 = kPairGeom[PAIR][0], G2 = kPairGeom[PAIR][1], LA = kGeomLink[G1], LB = kGeomLink[G2];
    constexpr CapsuleAxis c1 = capsule_axis(G1), c2 = capsule_axis(G2);
    constexpr double r1 = kGeom.radius[G1], r2 = kGeom.radius[G2], reach = r1 + r2 + kGeom.margin;
    const V2<R> o1 = rot_lit(cs[LA], sn[LA], c1.cx, c1.cz), a1 = rot_lit(cs[LA], sn[LA], c1.ax, c1.az);
    const V2<R> o2 = rot_lit(cs[LB], sn[LB], c2.cx, c2.cz), a2 = rot_lit(cs[LB], sn[LB], c2.ax, c2.az);
    const V2<R> p1 = {org[LA].x + o1.x, org[LA].z + o1.z}, p2 = {org[LB].x + o2.x, org[LB].z + o2.z};
    const V2<R> dif = {p1.x - p2.x, p1.z - p2.z};
    const R mb = -dot(a1, a2), u = -dot(a1, dif), w = dot(a2, dif);
    R det = fma_r(-mb, mb, R(1));
    det = det > R(sizeof(R) == 8 ? 1e-15 : 1e-6) ? det : R(sizeof(R) == 8 ? 1e-15 : 1e-6);
    const R l1 = (R)c1.half, l2 = (R)c2.half;
    R x1 = fma_r(-mb, w, u) * rcp_r(det);
    x1 = x1 > l1 ? l1 : (x1 < -l1 ? -l1 : x1);
    R x2 = fma_r(-mb, x1, w);
    if (x2 > l2 || x2 < -l2) {
        x2 = x2 > l2 ? l2 : -l2;
        x1 = fma_r(-mb, x2, u);
        x1 = x1 > l1 ? l1 : (x1 < -l1 ? -l1 : x1);
    }
    const V2<R> q1 = {fma_r(x1, a1.x, p1.x), fma_r(x1, a1.z, p1.z)};
    const V2<R> d = {fma_r(x2, a2.x, p2.x) - q1.x, fma_r(x2, a2.z, p2.z) - q1.z};
    const R len2 = dot(d, d);
    if (!(len2 < (R)(reach * reach))) return false;
    const bool degenerate = len2 < R(sizeof(R) == 8 ? 1e-20 : 1e-9);  // |c2 - c1| < 1e-10: oracle PAIR_MINLEN (float32: 3e-5, its noise floor)
    const R rl = rsqrt_r(degenerate ? R(1) : len2), len = len2 * rl;
    n = degenerate ? V2<R>{R(1), R(0)} : V2<R>{d.x * rl, d.z * rl};
    dist = len - (R)(r1 + r2);
    const R t = fma_r(R(0.5), dist, (R)r1);
    p = V2<R>{fma_r(t, n.x, q1.x), fma_r(t, n.z, q1.z)};
    return true;
}
// The pair's row on the absolute-angle coordinates: J . u = n . (velocity of p as part of link LB - as part of link LA).  The
// root translation and every link above LA cancel: entries LB .. LA only (LB < LA), J_i = n x (lever of link i) with levers
// p - org_LB, the link vectors D_a in between, org_(LA-1) - p.
template <int PAIR, typename R>
__device__ __forceinline__ void pair_row(const V2<R> (&org)[NL], const V2<R> (&D)[NL], V2<R> n, V2<R> p, R (&J)[NV]) {
    constexpr int LA = kGeomLink[kPairGeom[PAIR][0]], LB = kGeomLink[kPairGeom[PAIR][1]];
#pragma unroll
    for (int i = 0; i < NV; ++i) J[i] = R(0);
    J[LB] = dotperp(n, V2<R>{p.x - org[LB].x, p.z - org[LB].z});
#pragma unroll
    for (int a = LB + 1; a < LA; ++a) J[a] = dotperp(n, D[a]);
    J[LA] = dotperp(n, V2<R>{org[LA - 1].x - p.x, org[LA - 1].z - p.z});
}

// dense LDL^T of the symmetric 6x6 (lower triangle of A); L in the strict lower triangle, 1/D in invd
template <typename R>
__device__ __forceinline__ void ldl_factor(R (&A)[NV][NV], R (&invd)[NV]) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        invd[j] = rcp_r(A[j][j]);
#pragma unroll
        for (int i = j + 1; i < NV; ++i) {
            const R l = A[i][j] * invd[j];
#pragma unroll
            for (int k = j + 1; k <= i; ++k) A[i][k] = fma_r(-l, A[k][j], A[i][k]);
        }
#pragma unroll
        for (int i = j + 1; i < NV; ++i) A[i][j] *= invd[j];
    }
}
// y <- L^-1 y for a right-hand side whose entries before FIRST are structurally zero (a constraint on
// link FIRST touches its own angle, its ancestors' - the higher indices - and x, z)
template <int FIRST, typename R>
__device__ __forceinline__ void ldl_forward(const R (&A)[NV][NV], R (&y)[NV]) {
#pragma unroll
    for (int j = FIRST; j < NV; ++j)
#pragma unroll
        for (int i = j + 1; i < NV; ++i) y[i] = fma_r(-A[i][j], y[j], y[i]);
}
// x <- L^-T x
template <typename R>
__device__ __forceinline__ void ldl_backward(const R (&A)[NV][NV], R (&x)[NV]) {
#pragma unroll
    for (int j = NV - 1; j >= 0; --j)
#pragma unroll
        for (int i = j + 1; i < NV; ++i) x[j] = fma_r(-A[i][j], x[i], x[j]);
}

// Forward dynamics: qacc at (q, v); hd = dt for MuJoCo's Euler (implicit joint damping), 0 for RK4.
template <typename R>
__device__ __forceinline__ void accel(const R (&q)[NV], const R (&v)[NV], const R (&ctrl)[3], const Model& m, R hd,
                                      R (&qacc)[NV], const TrigCtx& trig) {
    // absolute angles / rates down the chain (hinges about -y)
    R phi[NL], om[NL];
    phi[L_TORSO] = q[2], om[L_TORSO] = v[2];
    phi[L_THIGH] = phi[L_TORSO] + hs(q[3]), om[L_THIGH] = om[L_TORSO] + hs(v[3]);
    phi[L_LEG] = phi[L_THIGH] + hs(q[4]), om[L_LEG] = om[L_THIGH] + hs(v[4]);
    phi[L_FOOT] = phi[L_LEG] + hs(q[5]), om[L_FOOT] = om[L_LEG] + hs(v[5]);
    R cs[NL], sn[NL], w2[NL];
    V2<R> S[NL], D[NL];
#pragma unroll
    for (int b = 0; b < NL; ++b) {
        sincos_ctx(trig, phi[b], sn[b], cs[b]);
        S[b] = rot(cs[b], sn[b], (R)kGeom.sx[b], (R)kGeom.sz[b]);
        D[b] = rot(cs[b], sn[b], (R)kGeom.d[b][0], (R)kGeom.d[b][1]);
        w2[b] = om[b] * om[b];
    }
    // ---- inertia (lower triangle) and right-hand side in absolute coordinates
    R A[NV][NV], f[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < NV; ++j) A[i][j] = R(0);
    const R g = (R)kGeom.gravity;
    R fx = R(0), fz = R(0);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        A[i][i] = (R)kGeom.diag[i];
        A[P_X][i] = S[i].z, A[P_Z][i] = -S[i].x;  // perp(S_i)
        R fi = g * S[i].x;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            if (j < i) {  // j is carried by i
                A[i][j] = dot(D[i], S[j]);
                fi = fma_r(w2[j], dotperp(S[j], D[i]), fi);
            } else if (j > i) {
                fi = fma_r(w2[j], dotperp(D[j], S[i]), fi);
            }
        }
        f[i] = fi;
        fx = fma_r(w2[i], S[i].x, fx), fz = fma_r(w2[i], S[i].z, fz);
    }
    A[P_X][P_X] = (R)kGeom.mtot, A[P_Z][P_Z] = (R)kGeom.mtot;
    f[P_X] = fx, f[P_Z] = fz - (R)kGeom.mtot * g;
    // ---- joints thigh, leg, foot: child link / parent link; theta_k = -(phi_c - phi_p)
    constexpr int jc[3] = {L_THIGH, L_LEG, L_FOOT}, jp[3] = {L_TORSO, L_THIGH, L_LEG};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const R c = ctrl[k] < R(kCtrlLo) ? R(kCtrlLo) : (ctrl[k] > R(kCtrlHi) ? R(kCtrlHi) : ctrl[k]);  // ctrlrange +-1 (xml:37-39)
        R tau = (R)kGeom.gear[k] * c - (R)kGeom.damp[k] * v[3 + k];
        if (kJointStiffness[k] != 0.0) tau -= (R)kJointStiffness[k] * q[3 + k];  // compile-time false for this model
        f[jc[k]] += hs(tau);  // generalised force of a joint torque on the absolute angles: d theta / d phi_child = kHingeSign
        f[jp[k]] -= hs(tau);
        const R e = (R)kGeom.arm[k] + hd * (R)kGeom.damp[k];  // armature + implicit damping on theta_k
        A[jc[k]][jc[k]] += e;
        A[jp[k]][jp[k]] += e;
        A[jp[k]][jc[k]] -= e;  // parent index > child index: lower triangle
    }
    // M = L D L^T; the acceleration is carried as z = D^-1 L^-1 (f + sum J^T lambda) (see cheetah_model.h):
    // a constraint row J costs one forward substitution y = L^-1 J^T (J M^-1 J^T = y . D^-1 y, J acc = y . z)
    R invd[NV], z[NV];
    ldl_factor(A, invd);
#pragma unroll
    for (int i = 0; i < NV; ++i) z[i] = f[i];
    ldl_forward<0>(A, z);
#pragma unroll
    for (int i = 0; i < NV; ++i) z[i] *= invd[i];

    // ---- soft constraints, one Gauss-Seidel sweep (oracle/planar_oracle.c order)
    auto limit = [&](auto kc) __attribute__((always_inline)) {  // joint limit on theta_k = phi_P - phi_C
        constexpr int k = decltype(kc)::value, C = jc[k], P = jp[k];  // C < P
        const R th = q[3 + k];
        // lo < hi: at most one side is violated, the smaller of the two distances is it (branch-free pick)
        const R dlo = th - (R)kGeom.lo[k], dhi = (R)kGeom.hi[k] - th;
        const bool lower = dlo < dhi;
        const R dist = lower ? dlo : dhi, J = lower ? R(1) : R(-1);
        if (dist < R(0)) {
            R y[NV], yd[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) y[i] = R(0);
            y[C] = hs(J), y[P] = -hs(J);
            ldl_forward<C>(A, y);
            R Aii = R(0), acur = R(0);
#pragma unroll
            for (int i = C; i < NV; ++i) yd[i] = y[i] * invd[i], Aii = fma_r(y[i], yd[i], Aii), acur = fma_r(y[i], z[i], acur);
            const R imp = impedance(dist, (R)kGeom.l_dmin, (R)kGeom.l_dmax, (R)(1.0 / kGeom.l_width));
            const R aref = -(R)m.lB * (J * v[3 + k]) - (R)m.lK * imp * dist;
            const R Rr = div_r(R(1) - imp, imp) * Aii;
            const R force = div_r(aref - acur, Aii + Rr);
            if (force > R(0)) {
#pragma unroll
                for (int i = C; i < NV; ++i) z[i] = fma_r(yd[i], force, z[i]);
            }
        }
    };
    limit(std::integral_constant<int, 0>{}), limit(std::integral_constant<int, 1>{}), limit(std::integral_constant<int, 2>{});
    // link origins (world): torso, then down the chain
    V2<R> org[NL];
    org[L_TORSO] = V2<R>{q[0], (R)kGeom.z0 + q[1]};
    org[L_THIGH] = V2<R>{org[L_TORSO].x + D[L_TORSO].x, org[L_TORSO].z + D[L_TORSO].z};
    org[L_LEG] = V2<R>{org[L_THIGH].x + D[L_THIGH].x, org[L_THIGH].z + D[L_THIGH].z};
    org[L_FOOT] = V2<R>{org[L_LEG].x + D[L_LEG].x, org[L_LEG].z + D[L_LEG].z};
    R u[NV];
#pragma unroll
    for (int b = 0; b < NL; ++b) u[b] = om[b];
    u[P_X] = v[0], u[P_Z] = v[1];
    // one capsule end sphere against the floor; geom order torso, thigh, leg, foot (two ends each)
    auto contact = [&](auto pt_c) __attribute__((always_inline)) {
        constexpr int pt = decltype(pt_c)::value, gi = pt / 2, LNK = kGeomLink[gi];
        const V2<R> e = rot(cs[LNK], sn[LNK], (R)kGeom.geom_end[pt][0], (R)kGeom.geom_end[pt][1]);
        const R dist = org[LNK].z + e.z - (R)kGeom.radius[gi];
        if (dist < (R)kGeom.margin) {
            // contact point midway between the surfaces: p = (s.x, dist/2); r = p - link origin
            const V2<R> r = {e.x, R(0.5) * dist - org[LNK].z};
            R Jx[NV], Jz[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) Jx[i] = R(0), Jz[i] = R(0);
            Jx[P_X] = R(1), Jz[P_Z] = R(1);
            Jx[LNK] = r.z, Jz[LNK] = -r.x;
#pragma unroll
            for (int a = LNK + 1; a < NL; ++a) Jx[a] = D[a].z, Jz[a] = -D[a].x;  // ancestors: perp(their link vector)
            R vn = R(0), vt = R(0);
#pragma unroll
            for (int i = LNK; i < NV; ++i) vn = fma_r(Jz[i], u[i], vn), vt = fma_r(Jx[i], u[i], vt);
            ldl_forward<LNK>(A, Jx);  // Jx, Jz now hold L^-1 J^T
            ldl_forward<LNK>(A, Jz);
            R dx[NV], dz[NV];
            R Ann = R(0), Att = R(0), Atn = R(0), an = R(0), at = R(0);
#pragma unroll
            for (int i = LNK; i < NV; ++i) {
                dx[i] = Jx[i] * invd[i], dz[i] = Jz[i] * invd[i];
                Ann = fma_r(Jz[i], dz[i], Ann), Att = fma_r(Jx[i], dx[i], Att), Atn = fma_r(Jx[i], dz[i], Atn);
                an = fma_r(Jz[i], z[i], an), at = fma_r(Jx[i], z[i], at);
            }
            const R pos = dist - (R)kGeom.margin;
            const R imp = contact_impedance(pos);
            const R k1 = div_r(R(1) - imp, imp);
            const R fn = div_r(-(R)m.cB * vn - (R)m.cK * imp * pos - an, Ann + k1 * Ann);
            if (fn > R(0)) {
                R ft = div_r(-(R)m.cB * vt - at - Atn * fn, Att + k1 * Att);
                const R lim = (R)kGeom.friction[gi] * fn;
                ft = ft > lim ? lim : (ft < -lim ? -lim : ft);
#pragma unroll
                for (int i = LNK; i < NV; ++i) z[i] = fma_r(dz[i], fn, fma_r(dx[i], ft, z[i]));
            }
        }
    };
    using std::integral_constant;
    contact(integral_constant<int, 0>{}), contact(integral_constant<int, 1>{}), contact(integral_constant<int, 2>{});
    contact(integral_constant<int, 3>{}), contact(integral_constant<int, 4>{}), contact(integral_constant<int, 5>{});
    contact(integral_constant<int, 6>{}), contact(integral_constant<int, 7>{});
    // capsule against capsule: one frictionless row per pair, after the floor points (oracle order)
    auto pair = [&](auto pc) __attribute__((always_inline)) {
        constexpr int P = decltype(pc)::value, LA = kGeomLink[kPairGeom[P][0]], LB = kGeomLink[kPairGeom[P][1]];
        R dist;
        V2<R> n, p;
        if (capsule_pair<P>(cs, sn, org, dist, n, p)) {
            R y[NV], yd[NV];
            pair_row<P>(org, D, n, p, y);
            R vn = R(0);
#pragma unroll
            for (int i = LB; i <= LA; ++i) vn = fma_r(y[i], u[i], vn);
            ldl_forward<LB>(A, y);
            R Aii = R(0), acur = R(0);
#pragma unroll
            for (int i = LB; i < NV; ++i) yd[i] = y[i] * invd[i], Aii = fma_r(y[i], yd[i], Aii), acur = fma_r(y[i], z[i], acur);
            const R pos = dist - (R)kGeom.margin;
            const R imp = contact_impedance(pos);
            const R fn = div_r(-(R)m.cB * vn - (R)m.cK * imp * pos - acur, Aii + div_r(R(1) - imp, imp) * Aii);
            if (fn > R(0)) {
#pragma unroll
                for (int i = LB; i < NV; ++i) z[i] = fma_r(yd[i], fn, z[i]);
            }
        }
    };
    pair(integral_constant<int, 0>{}), pair(integral_constant<int, 1>{}), pair(integral_constant<int, 2>{});
    R acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = z[i];
    ldl_backward(A, acc);

    // ---- back to joint coordinates
    qacc[0] = acc[P_X], qacc[1] = acc[P_Z], qacc[2] = acc[L_TORSO];
    qacc[3] = hs(acc[L_THIGH] - acc[L_TORSO]), qacc[4] = hs(acc[L_LEG] - acc[L_THIGH]), qacc[5] = hs(acc[L_FOOT] - acc[L_LEG]);
}

// y = A x, A symmetric in its lower triangle
template <typename R>
__device__ __forceinline__ void sym_matvec(const R (&A)[NV][NV], const R (&x)[NV], R (&y)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        R acc = A[i][i] * x[i];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (j < i) acc = fma_r(A[i][j], x[j], acc);
            if (j > i) acc = fma_r(A[j][i], x[j], acc);
        }
        y[i] = acc;
    }
}

// Forward dynamics with MuJoCo's constraint formulation: see cheetah_model.h:accel_newton (same scheme, single chain).
template <typename R>
struct NewtonWarm {
    R a[NV];
    bool valid;
    bool stages;  // RK4: the four stage evaluations of a substep share this object (body_kernels.h: begin_stages)
};
template <typename R>
__device__ __forceinline__ void accel_newton(const R (&q)[NV], const R (&v)[NV], const R (&ctrl)[3], const Model& m, R hd,
                                             R (&qacc)[NV], const TrigCtx& trig, NewtonWarm<R>& warm) {
    EMEI_MARK(nw_trig);
    R phi[NL], om[NL];
    phi[L_TORSO] = q[2], om[L_TORSO] = v[2];
    phi[L_THIGH] = phi[L_TORSO] + hs(q[3]), om[L_THIGH] = om[L_TORSO] + hs(v[3]);
    phi[L_LEG] = phi[L_THIGH] + hs(q[4]), om[L_LEG] = om[L_THIGH] + hs(v[4]);
    phi[L_FOOT] = phi[L_LEG] + hs(q[5]), om[L_FOOT] = om[L_LEG] + hs(v[5]);
    R cs[NL], sn[NL], w2[NL];
    V2<R> S[NL], D[NL];
#pragma unroll
    for (int b = 0; b < NL; ++b) {
        sincos_ctx(trig, phi[b], sn[b], cs[b]);
        S[b] = rot(cs[b], sn[b], (R)kGeom.sx[b], (R)kGeom.sz[b]);
        D[b] = rot(cs[b], sn[b], (R)kGeom.d[b][0], (R)kGeom.d[b][1]);
        w2[b] = om[b] * om[b];
    }
    constexpr int jc[3] = {L_THIGH, L_LEG, L_FOOT}, jp[3] = {L_TORSO, L_THIGH, L_LEG};
    auto build_inertia = [&](R (&A)[NV][NV], R hdamp) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < NV; ++j) A[i][j] = R(0);
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            A[i][i] = (R)kGeom.diag[i];
            A[P_X][i] = S[i].z, A[P_Z][i] = -S[i].x;
#pragma unroll
            for (int j = 0; j < i; ++j) A[i][j] = dot(D[i], S[j]);
        }
        A[P_X][P_X] = (R)kGeom.mtot, A[P_Z][P_Z] = (R)kGeom.mtot;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const R e = (R)kGeom.arm[k] + hdamp * (R)kGeom.damp[k];
            A[jc[k]][jc[k]] += e, A[jp[k]][jp[k]] += e, A[jp[k]][jc[k]] -= e;
        }
    };
    R f[NV];
    const R g = (R)kGeom.gravity;
    {
        R fx = R(0), fz = R(0);
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            R fi = g * S[i].x;
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                if (j < i) fi = fma_r(w2[j], dotperp(S[j], D[i]), fi);
                else if (j > i) fi = fma_r(w2[j], dotperp(D[j], S[i]), fi);
            }
            f[i] = fi;
            fx = fma_r(w2[i], S[i].x, fx), fz = fma_r(w2[i], S[i].z, fz);
        }
        f[P_X] = fx, f[P_Z] = fz - (R)kGeom.mtot * g;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const R c = ctrl[k] < R(kCtrlLo) ? R(kCtrlLo) : (ctrl[k] > R(kCtrlHi) ? R(kCtrlHi) : ctrl[k]);
        R tau = (R)kGeom.gear[k] * c - (R)kGeom.damp[k] * v[3 + k];
        if (kJointStiffness[k] != 0.0) tau -= (R)kJointStiffness[k] * q[3 + k];  // compile-time false for this model
        f[jc[k]] += hs(tau);  // generalised force of a joint torque on the absolute angles: d theta / d phi_child = kHingeSign
        f[jp[k]] -= hs(tau);
    }
    V2<R> org[NL];
    org[L_TORSO] = V2<R>{q[0], (R)kGeom.z0 + q[1]};
    org[L_THIGH] = V2<R>{org[L_TORSO].x + D[L_TORSO].x, org[L_TORSO].z + D[L_TORSO].z};
    org[L_LEG] = V2<R>{org[L_THIGH].x + D[L_THIGH].x, org[L_THIGH].z + D[L_THIGH].z};
    org[L_FOOT] = V2<R>{org[L_LEG].x + D[L_LEG].x, org[L_LEG].z + D[L_LEG].z};
    EMEI_MARK(nw_rows);
    // rows that exist (geometry only): bits 0-2 joint limits, 3-10 contact points, 11-13 capsule pairs
    uint32_t rows = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) rows |= ((q[3 + k] < (R)kGeom.lo[k]) | (q[3 + k] > (R)kGeom.hi[k])) ? (1u << k) : 0u;
#pragma unroll
    for (int pt = 0; pt < 8; ++pt) {
        const int gi = pt / 2, L = kGeomLink[gi];
        const R ez = rot_lit(cs[L], sn[L], kGeom.geom_end[pt][0], kGeom.geom_end[pt][1]).z;
        rows |= (org[L].z + ez - (R)kGeom.radius[gi] < (R)kGeom.margin) ? (1u << (3 + pt)) : 0u;
    }
    {  // bits 11-13: capsule pairs
        R dist;
        V2<R> n, p;
        rows |= capsule_pair<0>(cs, sn, org, dist, n, p) ? (1u << 11) : 0u;
        rows |= capsule_pair<1>(cs, sn, org, dist, n, p) ? (1u << 12) : 0u;
        rows |= capsule_pair<2>(cs, sn, org, dist, n, p) ? (1u << 13) : 0u;
    }
    R A[NV][NV], invd[NV], a[NV];
    EMEI_STAT_LANE(0);
    EMEI_STAT_WAVE(7);
    EMEI_MARK(nw_smooth0);
    // qacc_smooth = M^-1 qfrc_smooth (cheetah_model.h): free flight without implicit damping, and the cold start of the
    // iteration.  Only COLD evaluations need it — the first of an env-step; the other 15 of an RK4 step start from the previous
    // minimiser in every lane of the wave (body_kernels.h: `warm` lives for one env-step), and a free-flight lane then takes the
    // loop below with no row: its first step from the previous minimiser is a - M^-1 (M a - f) = M^-1 f, which the verify
    // sweep confirms (no row before, no row after) inside passes the wave runs for its other lanes anyway.  Wave-uniform
    // skip of ~150 instructions in 15 of 16 evaluations.
    const bool cold = !warm.valid;
#pragma unroll
    for (int i = 0; i < NV; ++i) a[i] = R(0);
    if (__ballot(cold) != 0ull) {
        build_inertia(A, R(0));
        ldl_factor(A, invd);
#pragma unroll
        for (int i = 0; i < NV; ++i) a[i] = f[i];
        ldl_forward<0>(A, a);
#pragma unroll
        for (int i = 0; i < NV; ++i) a[i] *= invd[i];
        ldl_backward(A, a);
    }
    if (rows == 0u && cold) {  // free flight: qacc = (M + h B)^-1 qfrc_smooth
        if (hd > R(0)) {  // Euler: solved with the constrained lanes' damping step at the end (cheetah_model.h)
#pragma unroll
            for (int i = 0; i < NV; ++i) a[i] = R(0);
        } else {  // the next stage evaluations start from it (or the lane would stay cold for as long as it is airborne)
#pragma unroll
            for (int i = 0; i < NV; ++i) warm.a[i] = a[i];
            warm.valid = true;
        }
    } else {
        EMEI_STAT_LANE(1);
#ifdef EMEI_NEWTON_STATS
        EMEI_STAT_LANE(25 + (__popc(rows) < 6 ? __popc(rows) : 6));  // 26..31: lanes with 1, 2, 3, 4, 5, >= 6 row blocks
        if (__popc(rows) <= 2) {  // how many lanes a two-slot constraint-space path (cheetah_model.h) would serve
            EMEI_STAT_LANE(22);
        } else {
            EMEI_STAT_LANE(24);
        }
#endif
        // the start of the iteration: the previous minimiser if there is one (RK4 stages), else qacc_smooth
        if (warm.valid) {
#pragma unroll
            for (int i = 0; i < NV; ++i) a[i] = warm.a[i];
        }
        R u[NV];
#pragma unroll
        for (int b = 0; b < NL; ++b) u[b] = om[b];
        u[P_X] = v[0], u[P_Z] = v[1];
        R fmax = R(1);
#pragma unroll
        for (int i = 0; i < NV; ++i) fmax = fmax > fabs(f[i]) ? fmax : fabs(f[i]);
        [[maybe_unused]] int n_pass = 0;
        bool converged = false;  // see cheetah_model.h
#pragma unroll 1
        for (int it = 0; it < cheetah::kMaxNewton; ++it) {
            R gr[NV];
            ++n_pass;
            EMEI_STAT_LANE(2);
            EMEI_STAT_WAVE(3);
            EMEI_MARK(nw_pass_base);
            build_inertia(A, R(0));
            sym_matvec(A, a, gr);
#pragma unroll
            for (int i = 0; i < NV; ++i) gr[i] -= f[i];
            EMEI_MARK(nw_limits);
            uint32_t flags = 0;  // the active set this pass assembles: bit k = limit row k, bits 3 + 3 pt .. = the point's edges (s1, s2, sy), bit 27 + pair
            auto limit = [&](auto kc) __attribute__((always_inline)) {  // theta_k = phi_P - phi_C: J = +-(e_P - e_C)
                constexpr int k = decltype(kc)::value, C = jc[k], P = jp[k];
                if (rows & (1u << k)) {
                    EMEI_STAT_WAVE(6);
                    // NOT opaque (cheetah_model.h makes its limit rows so): the hopper sits at a joint limit most of the time
                    // (2.5 limit rows per pass of a wave, tools/newton_stats.py), and the impedance / reference terms hipcc
                    // computes once ahead of the loop are then cheaper than per pass (A/B on one box: 30.1 vs 32.7 ms)
                    const R th = q[3 + k], vk = v[3 + k];
                    const bool lower = th < (R)kGeom.lo[k];
                    const R dist = lower ? th - (R)kGeom.lo[k] : (R)kGeom.hi[k] - th, J = lower ? R(1) : R(-1);
                    const R imp = impedance(dist, (R)kGeom.l_dmin, (R)kGeom.l_dmax, (R)(1.0 / kGeom.l_width));
                    const R aref = -(R)m.lB * (J * vk) - (R)m.lK * imp * dist;
                    const R x = J * hs(a[C] - a[P]) - aref;
                    if (x < R(0)) {
                        flags |= 1u << k;
                        const R Dw = div_r(imp, (R(1) - imp) * (R)kInvW.dof[k]);
                        const R t = Dw * x * J;
                        gr[C] += hs(t), gr[P] -= hs(t);
                        A[C][C] += Dw, A[P][P] += Dw, A[P][C] -= Dw;  // P > C
                    }
                }
            };
            limit(std::integral_constant<int, 0>{}), limit(std::integral_constant<int, 1>{}), limit(std::integral_constant<int, 2>{});
            EMEI_MARK(nw_contacts);
            auto contact = [&](auto pt_c) __attribute__((always_inline)) {
                constexpr int pt = decltype(pt_c)::value, gi = pt / 2, LNK = kGeomLink[gi];
                if (rows & (1u << (3 + pt))) {
                    EMEI_STAT_WAVE(4);
                    EMEI_STAT_LANE(5);
                    // The point's geometry, impedance and reference terms do not depend on the iterate: hipcc computes them
                    // once ahead of the loop.  Right for this body — most of its 8 points are in contact (5.4 row blocks per
                    // pass of a wave) and their ~40 doubles fit the register file (386 VGPRs, no scratch): 25.7 vs 30.3 ms per
                    // 100 RK4 steps against recomputing per pass.  The cheetah (16 points, 2 in contact) does the opposite.
                    const R csl = cs[LNK], snl = sn[LNK];
                    const V2<R> e = rot_lit(csl, snl, kGeom.geom_end[pt][0], kGeom.geom_end[pt][1]);
                    const R dist = org[LNK].z + e.z - (R)kGeom.radius[gi];
                    const V2<R> r = {e.x, R(0.5) * dist - org[LNK].z};
                    // Jacobian rows J_t = Jx, J_n = Jz: the link entries LNK .. torso; the root translation's are the literals
                    // Jx[P_X] = Jz[P_Z] = 1, Jx[P_Z] = Jz[P_X] = 0, written out below as plain additions (hipcc keeps `0 * u` and
                    // `1 * u` as multiplications: ~20 of a block's ~105 instructions per pass)
                    R Jx[NL], Jz[NL];
#pragma unroll
                    for (int i = 0; i < NL; ++i) Jx[i] = R(0), Jz[i] = R(0);
                    Jx[LNK] = r.z, Jz[LNK] = -r.x;
#pragma unroll
                    for (int b = LNK + 1; b < NL; ++b) Jx[b] = D[b].z, Jz[b] = -D[b].x;
                    R vn = u[P_Z], vt = u[P_X], an = a[P_Z], at = a[P_X];
#pragma unroll
                    for (int i = LNK; i < NL; ++i) {
                        vn = fma_r(Jz[i], u[i], vn), vt = fma_r(Jx[i], u[i], vt);
                        an = fma_r(Jz[i], a[i], an), at = fma_r(Jx[i], a[i], at);
                    }
                    constexpr double mu_c = kGeom.friction[gi];
                    const R mu = (R)mu_c;
                    const R pos = dist - (R)kGeom.margin;
                    const R imp = contact_impedance(pos);
                    const R xn = an + (R)m.cB * vn + (R)m.cK * imp * pos, xt = mu * (at + (R)m.cB * vt);
                    const R x1 = xn + xt, x2 = xn - xt;
                    const bool s1 = x1 < R(0), s2 = x2 < R(0), sy = xn < R(0);
                    flags |= ((s1 ? 1u : 0u) | (s2 ? 2u : 0u) | (sy ? 4u : 0u)) << (3 + 3 * pt);
                    if (s1 | s2 | sy) {
                        const R Dw = contact_weight(imp, 2.0 * mu_c * mu_c * (1.0 + mu_c * mu_c) * kInvW.link[LNK]);
                        const R c1 = s1 ? R(1) : R(0), c2 = s2 ? R(1) : R(0), cy = sy ? R(2) : R(0);
                        const R gn = Dw * (c1 * x1 + c2 * x2 + cy * xn), gt = Dw * mu * (c1 * x1 - c2 * x2);
                        const R wnn = Dw * (c1 + c2 + cy), wtt = Dw * mu * mu * (c1 + c2), wnt = Dw * mu * (c1 - c2);
#pragma unroll
                        for (int i = LNK; i < NL; ++i) {
                            gr[i] = fma_r(Jz[i], gn, fma_r(Jx[i], gt, gr[i]));
                            const R ux = fma_r(wtt, Jx[i], wnt * Jz[i]), uz = fma_r(wnt, Jx[i], wnn * Jz[i]);
#pragma unroll
                            for (int j = LNK; j <= i; ++j) A[i][j] = fma_r(ux, Jx[j], fma_r(uz, Jz[j], A[i][j]));
                            A[P_X][i] += ux, A[P_Z][i] += uz;  // rows of the root translation: J' W J against the unit columns
                        }
                        gr[P_X] += gt, gr[P_Z] += gn;
                        A[P_X][P_X] += wtt, A[P_Z][P_X] += wnt, A[P_Z][P_Z] += wnn;
                    }
                }
            };
            using std::integral_constant;
            contact(integral_constant<int, 0>{}), contact(integral_constant<int, 1>{}), contact(integral_constant<int, 2>{});
            contact(integral_constant<int, 3>{}), contact(integral_constant<int, 4>{}), contact(integral_constant<int, 5>{});
            contact(integral_constant<int, 6>{}), contact(integral_constant<int, 7>{});
            EMEI_MARK(hp_pairs);
            auto pair = [&](auto pc) __attribute__((always_inline)) {
                constexpr int P = decltype(pc)::value, G1 = kPairGeom[P][0], G2 = kPairGeom[P][1], LA = kGeomLink[G1], LB = kGeomLink[G2];
                if (rows & (1u << (11 + P))) {
                    EMEI_STAT_WAVE(4);
                    R dist, J[NV];
                    V2<R> n, p;
                    capsule_pair<P>(cs, sn, org, dist, n, p);
                    pair_row<P>(org, D, n, p, J);
                    R vn = R(0), an = R(0);
#pragma unroll
                    for (int i = LB; i <= LA; ++i) vn = fma_r(J[i], u[i], vn), an = fma_r(J[i], a[i], an);
                    const R pos = dist - (R)kGeom.margin;
                    const R imp = contact_impedance(pos);
                    const R x = an + (R)m.cB * vn + (R)m.cK * imp * pos;
                    if (x < R(0)) {
                        flags |= 1u << (27 + P);
                        // frictionless row: diagApprox = the translational inverse weights of both bodies
                        const R Dw = contact_weight(imp, kInvW.link[LA] + kInvW.link[LB]);
                        const R t = Dw * x;
#pragma unroll
                        for (int i = LB; i <= LA; ++i) {
                            gr[i] = fma_r(J[i], t, gr[i]);
                            const R ui = Dw * J[i];
#pragma unroll
                            for (int j = LB; j <= i; ++j) A[i][j] = fma_r(ui, J[j], A[i][j]);
                        }
                    }
                }
            };
            pair(integral_constant<int, 0>{}), pair(integral_constant<int, 1>{}), pair(integral_constant<int, 2>{});
            EMEI_MARK(nw_conv);
            R gmax = R(0);
#pragma unroll
            for (int i = 0; i < NV; ++i) gmax = gmax > fabs(gr[i]) ? gmax : fabs(gr[i]);
            if (gmax <= R(sizeof(R) == 8 ? 1e-11 : 1e-5) * fmax) {  // per lane, as in cheetah_model.h
                converged = true;
                break;
            }
            EMEI_MARK(nw_step);
            ldl_factor(A, invd);
            ldl_forward<0>(A, gr);
#pragma unroll
            for (int i = 0; i < NV; ++i) gr[i] *= invd[i];
            ldl_backward(A, gr);
#pragma unroll
            for (int i = 0; i < NV; ++i) a[i] -= gr[i];
            // The cost is piecewise quadratic: the step just taken IS the minimiser of the piece it was assembled on, so if the
            // active set at the new iterate is the same one, the iteration is over — found with the rows' residuals alone
            // (~20 instructions per row block) instead of a whole further pass whose gradient test would say the same (the
            // iterate is not touched: results are bit-identical; a set that flips on a rounding boundary goes on to that pass).
            // In the RK4 kernels only (`stages` is a compile-time constant after inlining): A/B on one box 25.85 -> 24.80 ms per 100
            // RK4 steps (86 % of the warm-started lanes take one step and then only verify); in the Euler kernel, where every
            // evaluation starts cold, the sweep costs what it saves (8.035 -> 8.08 ms).
            if (!warm.stages) continue;
            EMEI_MARK(hp_verify);
            uint32_t again = 0;
            auto limit2 = [&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value, C = jc[k], P = jp[k];
                if (rows & (1u << k)) {
                    const R th = q[3 + k], vk = v[3 + k];
                    const bool lower = th < (R)kGeom.lo[k];
                    const R dist = lower ? th - (R)kGeom.lo[k] : (R)kGeom.hi[k] - th, J = lower ? R(1) : R(-1);
                    const R imp = impedance(dist, (R)kGeom.l_dmin, (R)kGeom.l_dmax, (R)(1.0 / kGeom.l_width));
                    const R aref = -(R)m.lB * (J * vk) - (R)m.lK * imp * dist;
                    again |= (J * hs(a[C] - a[P]) - aref < R(0)) ? (1u << k) : 0u;
                }
            };
            limit2(std::integral_constant<int, 0>{}), limit2(std::integral_constant<int, 1>{}), limit2(std::integral_constant<int, 2>{});
            auto contact2 = [&](auto pt_c) __attribute__((always_inline)) {
                constexpr int pt = decltype(pt_c)::value, gi = pt / 2, LNK = kGeomLink[gi];
                if (rows & (1u << (3 + pt))) {
                    const V2<R> e = rot_lit(cs[LNK], sn[LNK], kGeom.geom_end[pt][0], kGeom.geom_end[pt][1]);
                    const R dist = org[LNK].z + e.z - (R)kGeom.radius[gi];
                    const V2<R> r = {e.x, R(0.5) * dist - org[LNK].z};
                    R Jx[NL], Jz[NL];  // as in the pass above: link entries; the root translation's are literals
#pragma unroll
                    for (int i = 0; i < NL; ++i) Jx[i] = R(0), Jz[i] = R(0);
                    Jx[LNK] = r.z, Jz[LNK] = -r.x;
#pragma unroll
                    for (int b = LNK + 1; b < NL; ++b) Jx[b] = D[b].z, Jz[b] = -D[b].x;
                    R vn = u[P_Z], vt = u[P_X], an = a[P_Z], at = a[P_X];
#pragma unroll
                    for (int i = LNK; i < NL; ++i) {
                        vn = fma_r(Jz[i], u[i], vn), vt = fma_r(Jx[i], u[i], vt);
                        an = fma_r(Jz[i], a[i], an), at = fma_r(Jx[i], a[i], at);
                    }
                    const R mu = (R)kGeom.friction[gi];
                    const R pos = dist - (R)kGeom.margin;
                    const R imp = contact_impedance(pos);
                    const R xn = an + (R)m.cB * vn + (R)m.cK * imp * pos, xt = mu * (at + (R)m.cB * vt);
                    again |= ((xn + xt < R(0) ? 1u : 0u) | (xn - xt < R(0) ? 2u : 0u) | (xn < R(0) ? 4u : 0u)) << (3 + 3 * pt);
                }
            };
            contact2(integral_constant<int, 0>{}), contact2(integral_constant<int, 1>{}), contact2(integral_constant<int, 2>{});
            contact2(integral_constant<int, 3>{}), contact2(integral_constant<int, 4>{}), contact2(integral_constant<int, 5>{});
            contact2(integral_constant<int, 6>{}), contact2(integral_constant<int, 7>{});
            auto pair2 = [&](auto pc) __attribute__((always_inline)) {
                constexpr int P = decltype(pc)::value, LA = kGeomLink[kPairGeom[P][0]], LB = kGeomLink[kPairGeom[P][1]];
                if (rows & (1u << (11 + P))) {
                    R dist, J[NV];
                    V2<R> n, p;
                    capsule_pair<P>(cs, sn, org, dist, n, p);
                    pair_row<P>(org, D, n, p, J);
                    R vn = R(0), an = R(0);
#pragma unroll
                    for (int i = LB; i <= LA; ++i) vn = fma_r(J[i], u[i], vn), an = fma_r(J[i], a[i], an);
                    const R pos = dist - (R)kGeom.margin;
                    const R imp = contact_impedance(pos);
                    again |= (an + (R)m.cB * vn + (R)m.cK * imp * pos < R(0)) ? (1u << (27 + P)) : 0u;
                }
            };
            pair2(integral_constant<int, 0>{}), pair2(integral_constant<int, 1>{}), pair2(integral_constant<int, 2>{});
            if (again == flags) {
                converged = true;
                break;
            }
        }
        EMEI_MARK(nw_final);
        EMEI_STAT_LANE(8 + (n_pass < 13 ? n_pass : 13));
        report_cap_hit(trig, !converged);
#pragma unroll
        for (int i = 0; i < NV; ++i) warm.a[i] = a[i];
        warm.valid = true;
    }
    EMEI_MARK(nw_euler);
    if (hd > R(0)) {  // mj_EulerSkip for every lane: qacc = a - (M + h B)^-1 (h B a); free flight (a = 0): rhs = -f
        R rhs[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) rhs[i] = rows == 0u ? -f[i] : R(0);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const R t = hd * (R)kGeom.damp[k] * (a[jc[k]] - a[jp[k]]);
            rhs[jc[k]] += t, rhs[jp[k]] -= t;
        }
        build_inertia(A, hd);
        ldl_factor(A, invd);
        ldl_forward<0>(A, rhs);
#pragma unroll
        for (int i = 0; i < NV; ++i) rhs[i] *= invd[i];
        ldl_backward(A, rhs);
#pragma unroll
        for (int i = 0; i < NV; ++i) a[i] -= rhs[i];
    }
    EMEI_MARK(nw_out);
    qacc[0] = a[P_X], qacc[1] = a[P_Z], qacc[2] = a[L_TORSO];
    qacc[3] = hs(a[L_THIGH] - a[L_TORSO]), qacc[4] = hs(a[L_LEG] - a[L_THIGH]), qacc[5] = hs(a[L_FOOT] - a[L_LEG]);
}

}  // namespace hopper

// Body traits for body_kernels.h
template <typename R, int SOLVER = EMEI_SOLVER_NEWTON>
struct HopperBody {
    using real = R;
    using Model = hopper::Model;
    // single sweep: the 256-register cap buys a second resident wave; the Newton solve needs the whole file
    static constexpr int kMinWavesPerEU = SOLVER == EMEI_SOLVER_SWEEP1 ? 2 : 1;
    static constexpr bool kUnrollRK4 = false;  // unrolled: the same time with the Newton solver (25.76 vs 25.73 ms), 4x the code
    static constexpr int kScratchPerLane = 0;
    static constexpr bool kHasCtrlCost = true;
    static constexpr bool kObsIsState = true;
    static constexpr bool kSpareReset = false;
    static constexpr bool kStreamOutputs = false;  // emei_device.h:store_body_out
    static constexpr int NS = 12, NO = 12, NA = 3;
    static Model make_model(double dt, const EnvParams& ep) {
        Model m = hopper::make_model(dt);
        m.w_forward = ep.get(EMEI_PARAM_FORWARD_REWARD_WEIGHT, 1.0), m.w_ctrl = ep.get(EMEI_PARAM_CTRL_COST_WEIGHT, 1e-3);
        m.healthy_reward = ep.get(EMEI_PARAM_HEALTHY_REWARD, 1.0);
        m.terminate_when_unhealthy = ep.get(EMEI_PARAM_TERMINATE_WHEN_UNHEALTHY, 1.0) != 0.0;
        m.st_lo = ep.get(EMEI_PARAM_HEALTHY_STATE_LO, -100.0), m.st_hi = ep.get(EMEI_PARAM_HEALTHY_STATE_HI, 100.0);
        m.z_lo = ep.get(EMEI_PARAM_HEALTHY_Z_LO, 0.7), m.z_hi = ep.get(EMEI_PARAM_HEALTHY_Z_HI, (double)INFINITY);
        return m;
    }

    struct WarmNone {};
    using Warm = std::conditional_t<SOLVER == EMEI_SOLVER_SWEEP1, WarmNone, hopper::NewtonWarm<R>>;
    __device__ __forceinline__ static void begin_stages(Warm& w) {
        if constexpr (SOLVER != EMEI_SOLVER_SWEEP1) w.stages = true;
    }
    __device__ __forceinline__ static void accel(const R (&q)[6], const R (&v)[6], const R (&ctrl)[NA], const Model& m, R hd,
                                                 R (&qacc)[6], const TrigCtx& trig, Warm& warm) {
        if constexpr (SOLVER == EMEI_SOLVER_SWEEP1) hopper::accel(q, v, ctrl, m, hd, qacc, trig);
        else hopper::accel_newton(q, v, ctrl, m, hd, qacc, trig, warm);
    }
    // hopper.py:79-93 as executed: np.logical_and(healthy_state, healthy_z, healthy_angle) takes the
    // third argument as `out=`, so the angle range is never applied
    template <typename T>
    __device__ __forceinline__ static bool is_healthy(const T* o, const Model& m) {
        bool st = true;
#pragma unroll
        for (int k = 2; k < NO; ++k) st &= ((T)m.st_lo < o[k]) & (o[k] < (T)m.st_hi);
        return st & ((T)m.z_lo < o[1]) & (o[1] < (T)m.z_hi);
    }
    // reward (hopper.py:95-102) = (is_healthy | terminate_when_unhealthy) * healthy_reward + w_f (x' - x)/dt_env
    // - w_c sum a^2 (per env, step() semantics); terminal (:104-106) = ~(is_healthy | terminate_when_unhealthy):
    // with the default flag (True) the reward term is constant and the env never terminates
    template <typename T>
    __device__ __forceinline__ static void reward_terminal(const T* o, T x_pre, T cost, const Model& m, int freq_rate, T& rew,
                                                           bool& term) {
        const bool ok = is_healthy(o, m) | (m.terminate_when_unhealthy != 0);
        rew = (ok ? (T)m.healthy_reward : T(0)) + (T)m.w_forward * (o[0] - x_pre) / ((T)m.dt * (T)freq_rate) - (T)m.w_ctrl * cost;
        term = !ok;
    }
    // obs = concat(qpos, qvel) (mujoco_env.py:153-155)
    __device__ __forceinline__ static void outputs(const R (&s)[NS], const R (&pre)[NS], const R (&ctrl)[NA], const Model& m,
                                                   int freq_rate, float (&o)[NO], R& rew, bool& term, const TrigCtx&) {
        R cost = R(0);
#pragma unroll
        for (int k = 0; k < NA; ++k) cost = fma_r(ctrl[k], ctrl[k], cost);
        reward_terminal(s, pre[0], cost, m, freq_rate, rew, term);
#pragma unroll
        for (int k = 0; k < NS; ++k) o[k] = (float)s[k];
    }
    __device__ __forceinline__ static void init_base(R (&s)[NS]) { s[1] += R(1.25); }  // init_qpos[rootz] = ref (xml:16)
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
    template <typename T>
    __device__ __forceinline__ static double ctrl_cost(const T* act) {  // this row's sum a^2, float64 (hopper.py:98)
        double cost = 0.0;
#pragma unroll
        for (int k = 0; k < NA; ++k) cost += (double)act[k] * (double)act[k];
        return cost;
    }
    template <typename T>
    __device__ __forceinline__ static double batch_reward(const T* obs, const T* pre_obs, const T* act,
                                                          const Model& m, int freq_rate) {
        double o[NO], cost = 0.0, rew;
        bool term;
#pragma unroll
        for (int k = 0; k < NO; ++k) o[k] = (double)obs[k];
#pragma unroll
        for (int k = 0; k < NA; ++k) cost += (double)act[k] * (double)act[k];
        reward_terminal(o, (double)pre_obs[0], cost, m, freq_rate, rew, term);
        return rew;
    }
    template <typename T>
    __device__ __forceinline__ static bool batch_terminal(const T* obs, const Model& m) {
        double o[NO], rew;
        bool term;
#pragma unroll
        for (int k = 0; k < NO; ++k) o[k] = (double)obs[k];
        reward_terminal(o, 0.0, 0.0, m, 1, rew, term);
        return term;
    }
};

}  // namespace emei
