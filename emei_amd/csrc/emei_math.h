// emei_math.h — fast full-precision sin/cos and reciprocal for the env kernels.
//
// The env-step arithmetic is a serial chain per env (one sincos + ~20 flops per substep), and at the
// benchmark's 65 536 envs there is exactly one wave per SIMD, so instruction count per step IS the
// speed.  The device library's sincos(double) carries a Payne-Hanek path and double-double
// arithmetic (hundreds of instructions); the angles of this workload are bounded (a swinging pole,
// |theta| < 1e6 rad), which allows a 2-constant Cody-Waite reduction with FMA and the classic
// minimax kernels on [-pi/4, pi/4] (coefficients: Sun fdlibm k_sin.c/k_cos.c, public domain;
// Cephes sinf/cosf for float).  Absolute error <= ~1.5e-16 (double) / ~6e-8 (float), i.e. the
// same last-bit uncertainty as between any two libm implementations.
//
// The header is plain C++ (no HIP types) so tests/host/math_accuracy.cpp can compile the very
// same functions with g++ and compare them with long-double libm on the CPU.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define EMEI_HD __host__ __device__ __forceinline__
#else
#define EMEI_HD inline
#endif

namespace emei {

// Horner step with a CONSTANT addend.  hipcc selects the two-address v_fmac_f64 for
// fma(p, z, C) and then has to copy C into the accumulator first (one v_mov_b64 per coefficient,
// 11 per sincos); the three-address v_fma_f64 keeps the constant where it is.  Plain (non-volatile)
// asm: the scheduler may still move and interleave it.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double horner_step(double p, double z, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(z), "v"(c));
    return d;
}
#else
inline double horner_step(double p, double z, double c) { return __builtin_fma(p, z, c); }
#endif

EMEI_HD double bits_to_f64(uint64_t u) {
    double d;
    memcpy(&d, &u, sizeof(d));
    return d;
}
EMEI_HD uint64_t f64_to_bits(double d) {
    uint64_t u;
    memcpy(&u, &d, sizeof(u));
    return u;
}
EMEI_HD float bits_to_f32(uint32_t u) {
    float f;
    memcpy(&f, &u, sizeof(f));
    return f;
}
EMEI_HD uint32_t f32_to_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, sizeof(u));
    return u;
}

// |x| above which the fast reduction is not used (n*P2 tail error and int32 quadrant stay tiny below)
constexpr double kFastTrigLimitF64 = 1.0e6;
constexpr float kFastTrigLimitF32 = 3.0e4f;

// sin and cos of x for |x| <= kFastTrigLimitF64 (NaN/Inf propagate as NaN).
EMEI_HD void fast_sincos(double x, double& s, double& c) {
    const double two_over_pi = 0.6366197723675814;
    const double P1 = 1.5707963267948966, P2 = 6.123233995736766e-17;  // pi/2 = P1 + P2 (+1.5e-33)
    double n = __builtin_rint(x * two_over_pi);
    double r = __builtin_fma(-n, P1, x);
    r = __builtin_fma(-n, P2, r);
    double z = r * r;
    // sin(r) = r + r^3 * Ps(z)
    double ps = 1.58969099521155010221e-10;
    ps = horner_step(ps, z, -2.50507602534068634195e-08);
    ps = horner_step(ps, z, 2.75573137070700676789e-06);
    ps = horner_step(ps, z, -1.98412698298579493134e-04);
    ps = horner_step(ps, z, 8.33333333332248946124e-03);
    ps = horner_step(ps, z, -1.66666666666666324348e-01);
    // cos(r) = 1 - z/2 + z^2 * Pc(z)
    double pc = -1.13596475577881948265e-11;
    pc = horner_step(pc, z, 2.08757232129817482790e-09);
    pc = horner_step(pc, z, -2.75573143513906633035e-07);
    pc = horner_step(pc, z, 2.48015872894767294178e-05);
    pc = horner_step(pc, z, -1.38888888888741095749e-03);
    pc = horner_step(pc, z, 4.16666666666666019037e-02);
    double sr = __builtin_fma(z * r, ps, r);
    double cr = __builtin_fma(z, __builtin_fma(z, pc, -0.5), 1.0);
    // quadrant: q = n mod 4
    int q = (int)n;
    double s0 = (q & 1) ? cr : sr;
    double c0 = (q & 1) ? sr : cr;
    uint64_t sflip = (uint64_t)(q & 2) << 62;        // bit 63 if q in {2,3}
    uint64_t cflip = (uint64_t)((q + 1) & 2) << 62;  // bit 63 if q in {1,2}
    s = bits_to_f64(f64_to_bits(s0) ^ sflip);
    c = bits_to_f64(f64_to_bits(c0) ^ cflip);
}

// Table-assisted variant: 256 entries {sin, cos}(k * 2pi/256), correctly rounded on the host
// (emei_trig_table, abi.hip) and staged in LDS by every kernel.  x = k*2pi/256 + r with |r| <= pi/256,
// so sin r / cos r need 3 / 4 terms and no quadrant logic:
//     sin x = S_k cos r + C_k sin r,    cos x = C_k cos r - S_k sin r.
// 15 float64 operations + 3 integer ones instead of 37; absolute error <= ~2.3e-16.
struct SinCosEntry {
    double s, c;
};
constexpr int kTrigTableSize = 256;

// Split in two so that a caller can put independent work between the table read and its use:
// begin() issues the LDS read and evaluates sin r / cos r, end() applies the rotation.
struct SinCosPending {
    SinCosEntry e;
    double sr, cr;
};
EMEI_HD SinCosPending fast_sincos_tab_begin(double x, const SinCosEntry* tab) {
    const double inv_step = 40.74366543152521;  // 256 / (2 pi)
    const double H1 = 1.5707963267948966 / 64, H2 = 6.123233995736766e-17 / 64;  // 2pi/256 = H1 + H2 (exact scalings)
    SinCosPending p;
    const double n = __builtin_rint(x * inv_step);
    p.e = tab[(int)n & (kTrigTableSize - 1)];
    double r = __builtin_fma(-n, H1, x);
    r = __builtin_fma(-n, H2, r);
    const double z = r * r;
    p.sr = __builtin_fma(r * z, __builtin_fma(z, 1.0 / 120, -1.0 / 6), r);                      // r - r^3/6 + r^5/120
    p.cr = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, -1.0 / 720, 1.0 / 24), -0.5), 1.0);  // 1 - z/2 + z^2/24 - z^3/720
    return p;
}
EMEI_HD void fast_sincos_tab_end(const SinCosPending& p, double& s, double& c) {
    s = __builtin_fma(p.e.s, p.cr, p.e.c * p.sr);
    c = __builtin_fma(p.e.c, p.cr, -(p.e.s * p.sr));
}
EMEI_HD void fast_sincos_tab(double x, const SinCosEntry* tab, double& s, double& c) {
    fast_sincos_tab_end(fast_sincos_tab_begin(x, tab), s, c);
}

// float version, |x| <= kFastTrigLimitF32
EMEI_HD void fast_sincosf(float x, float& s, float& c) {
    const float two_over_pi = 0.6366197466850281f;
    const float P1 = 1.5707963705062866f, P2 = -4.371138828673793e-08f, P3 = -1.7151245100058819e-15f;
    float n = __builtin_rintf(x * two_over_pi);
    float r = __builtin_fmaf(-n, P1, x);
    r = __builtin_fmaf(-n, P2, r);
    r = __builtin_fmaf(-n, P3, r);
    float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = __builtin_fmaf(ps, z, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
    float pc = 2.443315711809948e-5f;
    pc = __builtin_fmaf(pc, z, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
    float sr = __builtin_fmaf(z * r, ps, r);
    float cr = __builtin_fmaf(z, __builtin_fmaf(z, pc, -0.5f), 1.0f);
    int q = (int)n;
    float s0 = (q & 1) ? cr : sr;
    float c0 = (q & 1) ? sr : cr;
    uint32_t sflip = (uint32_t)(q & 2) << 30;
    uint32_t cflip = (uint32_t)((q + 1) & 2) << 30;
    s = bits_to_f32(f32_to_bits(s0) ^ sflip);
    c = bits_to_f32(f32_to_bits(c0) ^ cflip);
}

// 1/d to ~1 ulp from a hardware seed `r0` (>= 20 good bits) by two Newton steps.
EMEI_HD double refine_rcp(double d, double r0) {
    double e = __builtin_fma(-d, r0, 1.0);
    double r = __builtin_fma(r0, e, r0);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
}

// n/d with a final residual correction (error <= ~1 ulp for normal operands, no scaling:
// the denominators of this workload are O(1) by construction).
EMEI_HD double div_via_rcp(double n, double d, double rcp) {
    double q = n * rcp;
    double res = __builtin_fma(-d, q, n);
    return __builtin_fma(res, rcp, q);
}

}  // namespace emei
