// dpend_model.h — InvertedDoublePendulum (emei/envs/mujoco/inverted_double_pendulum.py on
// mujoco_env.py; model emei/envs/mujoco/assets/inverted_double_pendulum.xml): cart on a rail + two
// hinged poles, 3 DoF.  VARIANT 0 ReboundBalancing (:63-90), 1 BoundaryBalancing (:93-123),
// 2 ReboundSwingUp (:126-153), 3 BoundarySwingUp (:156-196).
//
// Same absolute-angle formulation as cheetah_model.h (M[x,phi_j] = S_j.z, M[phi_1,phi_2] = D.S_2,
// constant diagonals, centripetal terms Omega_j^2 * rotated vectors), restated independently by the
// oracle (oracle/dpend_oracle.c: joint-coordinate Jacobians).  Parity with libmujoco is unpinned.
// Reference quirks reproduced on purpose: the observation "wrap" is ((theta+pi) % 2) * pi - pi
// (operator precedence, :59) and reward/terminal are evaluated on those wrapped values, as step()
// does (mujoco_env.py:160-163); gravity has an x component of 1e-5 (xml:26).
#pragma once
#include <cmath>
#include <cstring>

#include "emei_device.h"

namespace emei {
namespace dpend {

struct Model {
    double s1z, s2z, diag1, diag2, L1, mtot;  // mass-moment z of each pole, constant inertias, pole-1 length
    double gx, gz, gear, x_lo, x_hi, margin, invw;
    double K, B, dmin, dmax, width;  // slider-limit solref (refsafe'd) / solimp
    double dt, phi_off;
};

namespace host {
inline double capsule_mass(double rho, double r, double half) { return rho * (M_PI * r * r * 2 * half + 4.0 / 3.0 * M_PI * r * r * r); }
inline double capsule_inertia_perp(double rho, double r, double half) {
    double h = 2 * half, mcyl = rho * M_PI * r * r * h, msph = rho * 4.0 / 3.0 * M_PI * r * r * r;
    return mcyl * (3 * r * r + h * h) / 12 + msph * (2 * r * r / 5 + h * h / 4 + 3 * h * r / 8);
}
}  // namespace host

// the XML-level numbers (hand-typed from inverted_double_pendulum.xml; pinned to the file by tests/test_model_constants.py
// through emei_model_constants)
struct Xml {
    double rho, cart_r, cart_half, pole_r, pole_half, L1, gx, gz, gear, ctrl_lo, ctrl_hi, x_lo, x_hi, margin, solref_tc, dmin, dmax, width;
};
constexpr Xml kXml = {1000.0, 0.1, 0.1,      // density (MuJoCo default); cart capsule size="0.1 0.1" (xml:32)
                      0.045, 0.3, 0.6,       // each pole: fromto 0 0 0 0 0 0.6, r .045 (xml:35,38); pole 2 hangs at z = .6 (xml:36)
                      1e-5, 9.81,            // gravity "1e-5 0 -9.81" (xml:26)
                      500.0, -1.0, 1.0,      // motor gear, ctrlrange (xml:45)
                      -3.0, 3.0, 0.01,       // slider range, margin (xml:31)
                      0.02, 0.9, 0.95, 0.001};  // default solref (.02 1), solimp (.9 .95 .001)

inline Model make_model(bool swingup, double dt) {
    Model m;
    memset(&m, 0, sizeof(m));
    const double rho = kXml.rho;
    const double mc = host::capsule_mass(rho, kXml.cart_r, kXml.cart_half);
    const double mp = host::capsule_mass(rho, kXml.pole_r, kXml.pole_half);
    const double Ip = host::capsule_inertia_perp(rho, kXml.pole_r, kXml.pole_half);
    const double lc = kXml.pole_half, L1 = kXml.L1;                           // pole com, pole-2 attachment
    m.s1z = mp * lc + mp * L1, m.s2z = mp * lc;
    m.diag1 = Ip + mp * lc * lc + mp * L1 * L1, m.diag2 = Ip + mp * lc * lc;
    m.L1 = L1, m.mtot = mc + 2 * mp;
    m.gx = kXml.gx, m.gz = kXml.gz;
    m.gear = kXml.gear;
    m.x_lo = kXml.x_lo, m.x_hi = kXml.x_hi, m.margin = kXml.margin;
    // dof_invweight0 of the slider: (M^-1)_xx at qpos0 (both poles upright)
    const double a = m.mtot, b = m.s1z, c = m.s2z, d = m.diag1, e = L1 * m.s2z, f = m.diag2;
    const double det = a * (d * f - e * e) - b * (b * f - e * c) + c * (b * e - d * c);
    m.invw = (d * f - e * e) / det;
    const double tc = kXml.solref_tc < 2 * dt ? 2 * dt : kXml.solref_tc, dmax = kXml.dmax;
    m.K = 1.0 / (dmax * dmax * tc * tc), m.B = 2.0 / (dmax * tc);
    m.dmin = kXml.dmin, m.dmax = dmax, m.width = kXml.width;
    m.dt = dt;
    m.phi_off = swingup ? M_PI : 0.0;                                         // _update_model: body_quat[2] = (0,0,1,0)
    return m;
}

// emei_model_constants (include/emei_hip.h): [gx, gz, mc, mp, Ip, lc, L1, gear, ctrl_lo, ctrl_hi, x_lo, x_hi, margin, solref tc,
// solimp dmin, dmax, width]
inline int xml_constants(double* out) {
    const double mc = host::capsule_mass(kXml.rho, kXml.cart_r, kXml.cart_half), mp = host::capsule_mass(kXml.rho, kXml.pole_r, kXml.pole_half);
    const double v[17] = {kXml.gx, kXml.gz, mc, mp, host::capsule_inertia_perp(kXml.rho, kXml.pole_r, kXml.pole_half), kXml.pole_half,
                          kXml.L1, kXml.gear, kXml.ctrl_lo, kXml.ctrl_hi, kXml.x_lo, kXml.x_hi, kXml.margin, kXml.solref_tc, kXml.dmin,
                          kXml.dmax, kXml.width};
    for (int i = 0; i < 17; ++i) out[i] = v[i];
    return 17;
}

}  // namespace dpend

template <int VARIANT, typename R>
struct DPendBody {
    using real = R;
    using Model = dpend::Model;
    static constexpr int kMinWavesPerEU = 1;
    static constexpr bool kUnrollRK4 = false;
    static constexpr int kScratchPerLane = 0;
    static constexpr bool kHasCtrlCost = false;
    static constexpr bool kObsIsState = false;
    static constexpr bool kSpareReset = true;
    static constexpr bool kStreamOutputs = true;  // emei_device.h:store_body_out
    static constexpr int NS = 6, NO = 6, NA = 1;
    static Model make_model(double dt, const EnvParams&) { return dpend::make_model(VARIANT >= 2, dt); }

    // q = (x, theta1, theta2), v = (v, omega1, omega2); no joint damping in this model, so `hd` is unused
    struct Warm {};  // a single constraint row: solved exactly in one step
    __device__ __forceinline__ static void begin_stages(Warm&) {}
    __device__ __forceinline__ static void accel(const R (&q)[3], const R (&v)[3], const R (&ctrl)[NA], const Model& m, R,
                                                 R (&qacc)[3], const TrigCtx& trig, Warm&) {
        const R phi1 = q[1] + (R)m.phi_off, phi2 = phi1 + q[2];
        const R w1 = v[1], w2 = v[1] + v[2];
        R s1, c1, s2, c2;
        sincos_ctx(trig, phi1, s1, c1);
        sincos_ctx(trig, phi2, s2, c2);
        // S_j = R(phi_j)(0, s_jz) = (s_jz sin, s_jz cos);  D = R(phi_1)(0, L1)
        const R S1x = (R)m.s1z * s1, S1z = (R)m.s1z * c1, S2x = (R)m.s2z * s2, S2z = (R)m.s2z * c2;
        const R Dx = (R)m.L1 * s1, Dz = (R)m.L1 * c1;
        const R M12 = fma_r(Dx, S2x, Dz * S2z);  // D . S2
        const R u = ctrl[0] < R(dpend::kXml.ctrl_lo) ? R(dpend::kXml.ctrl_lo) : (ctrl[0] > R(dpend::kXml.ctrl_hi) ? R(dpend::kXml.ctrl_hi) : ctrl[0]);
        const R w1s = w1 * w1, w2s = w2 * w2;
        R fx = (R)m.gear * u + (R)m.mtot * (R)m.gx + w1s * S1x + w2s * S2x;
        R f1 = (R)m.gx * S1z + (R)m.gz * S1x + w2s * fma_r(S2x, Dz, -(S2z * Dx));  // S2 . perp(D)
        R f2 = (R)m.gx * S2z + (R)m.gz * S2x + w1s * fma_r(Dx, S2z, -(Dz * S2x));  // D . perp(S2)
        // symmetric 3x3 [[a b c],[b d e],[c e f]] in the order (x, phi1, phi2): LDL^T eliminating phi2, phi1, x
        const R a = (R)m.mtot, b = S1z, c = S2z, d = (R)m.diag1, e = M12, f = (R)m.diag2;
        // (one Newton step on the hardware reciprocal, <= 2.2e-15 relative: emei_device.h:rcp1_r — as the InvertedPendulum)
        const R if_ = rcp1_r(f);
        const R le = e * if_, lc = c * if_;          // column of phi2
        const R d1 = fma_r(-le, e, d), b1 = fma_r(-le, c, b), a1 = fma_r(-lc, c, a);
        const R id1 = rcp1_r(d1);
        const R lb = b1 * id1;
        const R a2 = fma_r(-lb, b1, a1);
        const R ia2 = rcp1_r(a2);
        // solve M acc = rhs for rhs = (fx, f1, f2)
        R ax, a1_, a2_;
        {
            const R y1 = fma_r(-le, f2, f1);                  // forward: phi2 -> phi1, x
            const R yx = fma_r(-lb, y1, fma_r(-lc, f2, fx));
            ax = yx * ia2;                                    // back substitution
            a1_ = fma_r(-lb, ax, y1 * id1);
            a2_ = fma_r(-le, a1_, fma_r(-lc, ax, f2 * if_));
        }
        // soft slider limit with margin 0.01 (mjCNSTR_LIMIT_JOINT: active when dist < margin)
        // x_lo + margin < x_hi - margin: at most one side is active, the smaller distance is it.  The rail is symmetric, so
        // dist = x_hi - |x| and J = -sign(x).  Some lane of a wave is at the rail in about half of the evaluations (random
        // pushes), a few per cent of the lanes: the block is written as the InvertedPendulum's (pendulum_envs.h:slider_row) —
        // M^-1 e_x = (1, -lb, lb le - lc) / a2 from the factor instead of a second substitution, J as a +-1.0 factor of an
        // fma, -K imp pos = fma(K imp, |x|, -K imp (x_hi - margin)), the impedance polynomial and the division by imp only in
        // waves with a lane inside the 1 mm width, the force clamped by a maximum.
        static_assert(dpend::kXml.x_lo == -dpend::kXml.x_hi, "symmetric slider range");
        const R ax_abs = abs_r(q[0]);
        if (ax_abs > (R)(m.x_hi - m.margin)) {
            const R nJ = copysign_r(R(1), q[0]);                  // -J
            const R n1 = fma_r((R)m.B, v[0], ax);                 // aref - J ax = nJ (B v + ax) - K imp pos
            const R wx = ia2, w1_ = -lb * ia2, w2_ = fma_r(lb, le, -lc) * ia2;
            auto finish = [&](R Kimp, R Kimp_edge, R Rr) __attribute__((always_inline)) {
                const R num = fma_r(nJ, n1, fma_r(Kimp, ax_abs, -Kimp_edge));
                const R g = nJ * fmax_r(num * rcp1_r(wx + Rr), R(0));  // -J force
                ax = fma_r(-wx, g, ax);
                a1_ = fma_r(-w1_, g, a1_);
                a2_ = fma_r(-w2_, g, a2_);
            };
            const R edge = (R)(m.x_hi - m.margin);
            const R Kfull = (R)m.K * (R)m.dmax;
            const bool full = !(ax_abs < (R)(m.x_hi - m.margin + m.width));
            if (__builtin_expect(__ballot(!full) != 0ull, 0)) {
                const R xx = (ax_abs - edge) * (R)(1.0 / m.width), u1 = R(1) - xx;
                const R y = xx <= R(0.5) ? R(2) * xx * xx : fma_r(R(-2) * u1, u1, R(1));
                const R imp = fma_r(y, (R)(m.dmax - m.dmin), (R)m.dmin);
                const R Kimp = full ? Kfull : (R)m.K * imp;
                finish(Kimp, Kimp * edge, full ? (R)((1.0 - m.dmax) / m.dmax) * (R)m.invw : (R(1) - imp) * (R)m.invw * rcp1_r(imp));
            } else {
                finish(Kfull, Kfull * edge, (R)((1.0 - m.dmax) / m.dmax) * (R)m.invw);
            }
        }
        qacc[0] = ax, qacc[1] = a1_, qacc[2] = a2_ - a1_;  // theta2'' = Omega2' - Omega1'
    }

    // inverted_double_pendulum.py:57-60: state[1:3] = (state[1:3] + pi) % 2 * pi - pi   (sic)
    template <typename T>
    __device__ __forceinline__ static T quirk_wrap(T th) {
        const T pi = T(3.141592653589793);
        return pymod_pos(th + pi, T(2), T(0.5)) * pi - pi;
    }
    // `sc(x, s, c)`: the caller's sincos (LDS table inside the rollout, polynomial in the stateless kernels)
    template <typename T, typename SC>
    __device__ __forceinline__ static void reward_terminal(const T (&o)[NO], const Model& m, T& rew, bool& term, SC sc) {
        T sa, ca, sb, cb;
        sc(o[1], sa, ca);
        sc(o[1] + o[2], sb, cb);
        const T y = ca + cb;
        bool fin = true;
#pragma unroll
        for (int k = 0; k < NO; ++k) fin &= finite_r(o[k]);
        const bool inx = ((T)m.x_lo < o[0]) & (o[0] < (T)m.x_hi);
        if (VARIANT == 0) rew = T(1), term = !((y >= T(1.5)) & fin);            // :84-90
        else if (VARIANT == 1) rew = T(1), term = !((y >= T(0)) & inx & fin);   // :114-123
        else if (VARIANT == 2) rew = (T(2) - y) / T(4), term = !fin;            // :144-153
        else rew = (T(2) - y) / T(4) - (T(5e-3) * o[4] * o[4] + T(1e-4) * o[5] * o[5]), term = !(inx & fin);  // :183-196
    }
    __device__ __forceinline__ static void outputs(const R (&s)[NS], const R (&pre)[NS], const R (&ctrl)[NA], const Model& m,
                                                   int freq_rate, float (&o)[NO], R& rew, bool& term, const TrigCtx& trig) {
        R ob[NO] = {s[0], quirk_wrap(s[1]), quirk_wrap(s[2]), s[3], s[4], s[5]};
        reward_terminal(ob, m, rew, term, [&](R x, R& sn, R& cs) { sincos_ctx(trig, x, sn, cs); });
#pragma unroll
        for (int k = 0; k < NO; ++k) o[k] = (float)ob[k];
    }
    __device__ __forceinline__ static void init_base(R (&)[NS]) {}  // init_qpos = init_qvel = 0
    // the state of the padding lanes of a ragged last wave (body_kernels.h): at rest in the middle of the rail
    __device__ __forceinline__ static void park(R (&s)[NS]) {
#pragma unroll
        for (int k = 0; k < NS; ++k) s[k] = R(0);
    }
    __device__ __forceinline__ static void obs_of(const R (&s)[NS], double (&o)[NO], const Model&) {
        o[0] = (double)s[0], o[1] = (double)quirk_wrap(s[1]), o[2] = (double)quirk_wrap(s[2]);
        o[3] = (double)s[3], o[4] = (double)s[4], o[5] = (double)s[5];
    }
    template <typename T>
    __device__ __forceinline__ static double batch_reward(const T* obs, const T*, const T*, const Model& m, int) {
        double o[NO], rew;
        bool term;
#pragma unroll
        for (int k = 0; k < NO; ++k) o[k] = (double)obs[k];
        reward_terminal(o, m, rew, term, [](double x, double& sn, double& cs) { sincos_r(x, sn, cs); });
        return rew;
    }
    template <typename T>
    __device__ __forceinline__ static bool batch_terminal(const T* obs, const Model& m) {
        double o[NO], rew;
        bool term;
#pragma unroll
        for (int k = 0; k < NO; ++k) o[k] = (double)obs[k];
        reward_terminal(o, m, rew, term, [](double x, double& sn, double& cs) { sincos_r(x, sn, cs); });
        return term;
    }
};

}  // namespace emei
