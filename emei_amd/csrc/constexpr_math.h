// constexpr_math.h — compile-time sin / cos / sqrt / atan for the model constants of the MuJoCo-backed bodies.
//
// The dt-independent constants of a body (masses, inertias, link vectors, joint tables: functions of the MJCF
// asset only) are evaluated at COMPILE time and reach the kernels as immediates: as kernel arguments they occupy
// scalar registers for the whole kernel, overflow the SGPR file and come back through v_readlane spills, which are
// vector instructions (cheetah_model.h; InvPend in pendulum_envs.h).  Accuracy ~1e-16, the same as libm's.
#pragma once

namespace emei {
namespace ce {

constexpr double kPi = 3.14159265358979323846;

// sin / cos for |a| of a few radians: reduction to [-pi/4, pi/4] by multiples of pi/2 (two-constant Cody-Waite),
// Taylor series to the 21st / 20th power
constexpr void sincos(double a, double& sn, double& cs) {
    const double pio2_hi = 1.5707963267948966, pio2_lo = 6.123233995736766e-17;
    const double kq = a / pio2_hi;
    const long k = (long)(kq + (kq >= 0 ? 0.5 : -0.5));
    const double r = (a - k * pio2_hi) - k * pio2_lo, z = r * r;
    double ts = 0, tc = 0;
    for (int n = 10; n >= 0; --n) {  // Horner over sum_n (-1)^n z^n / (2n+1)!  and  / (2n)!
        double fs = 1, fc = 1;
        for (int q = 1; q <= 2 * n + 1; ++q) fs *= q;
        for (int q = 1; q <= 2 * n; ++q) fc *= q;
        ts = ts * z + ((n & 1) ? -1.0 : 1.0) / fs;
        tc = tc * z + ((n & 1) ? -1.0 : 1.0) / fc;
    }
    const double s0 = r * ts, c0 = tc;
    switch (((k % 4) + 4) % 4) {
        case 0: sn = s0, cs = c0; break;
        case 1: sn = c0, cs = -s0; break;
        case 2: sn = -s0, cs = -c0; break;
        default: sn = -c0, cs = s0; break;
    }
}
constexpr double sin(double a) {
    double s = 0, c = 0;
    sincos(a, s, c);
    return s;
}
constexpr double cos(double a) {
    double s = 0, c = 0;
    sincos(a, s, c);
    return c;
}

// Newton on x^2 = v from a power-of-two seed (quadratic convergence: 8 steps from within a factor 2)
constexpr double sqrt(double v) {
    if (!(v > 0)) return 0;
    double x = 1;
    while (x * x < v) x *= 2;
    while (x * x > v) x *= 0.5;
    for (int i = 0; i < 12; ++i) x = 0.5 * (x + v / x);
    return x;
}

// atan(y / x) for x > 0 and |y / x| <= 0.5 (the pole's fromto offsets are ~1e-3 of its length): alternating series
constexpr double atan_small(double y, double x) {
    const double t = y / x, z = t * t;
    double s = 0;
    for (int n = 40; n >= 0; --n) s = s * z + ((n & 1) ? -1.0 : 1.0) / (2 * n + 1);
    return t * s;
}

// x <- A^-1 x for a symmetric positive definite N x N matrix given in full (both triangles): dense LDL^T without pivoting.
// Used at compile time only (the qpos0 inverse weights of the constraint regularisers).
template <int N>
constexpr void spd_solve(const double (&A)[N][N], double (&x)[N]) {
    double L[N][N] = {}, D[N] = {};
    for (int j = 0; j < N; ++j) {
        double d = A[j][j];
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k] * D[k];
        D[j] = d;
        for (int i = j + 1; i < N; ++i) {
            double v = A[i][j];
            for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k] * D[k];
            L[i][j] = v / d;
        }
    }
    for (int i = 0; i < N; ++i)
        for (int k = 0; k < i; ++k) x[i] -= L[i][k] * x[k];
    for (int i = 0; i < N; ++i) x[i] /= D[i];
    for (int i = N - 1; i >= 0; --i)
        for (int k = i + 1; k < N; ++k) x[i] -= L[k][i] * x[k];
}
// J . A^-1 J
template <int N>
constexpr double spd_quad(const double (&A)[N][N], const double (&J)[N]) {
    double w[N] = {};
    for (int i = 0; i < N; ++i) w[i] = J[i];
    spd_solve<N>(A, w);
    double s = 0;
    for (int i = 0; i < N; ++i) s += J[i] * w[i];
    return s;
}

}  // namespace ce
}  // namespace emei
