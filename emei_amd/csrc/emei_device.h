// emei_device.h — device-side helpers shared by the env kernels (gfx950 / wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/emei_hip.h"

namespace emei {

constexpr int kBlock = 256;  // 4 waves: one per SIMD of a CU
constexpr int kWave = 64;

// ---------------------------------------------------------------------------------------------
// trigonometry in the precision of the env
__device__ __forceinline__ void sincos_r(double x, double& s, double& c) { ::sincos(x, &s, &c); }
__device__ __forceinline__ void sincos_r(float x, float& s, float& c) { ::sincosf(x, &s, &c); }

// Python / NumPy floored modulo (np.remainder): sign follows the divisor.
__device__ __forceinline__ double pymod(double a, double p) {
    double m = ::fmod(a, p);
    if (m != 0.0 && ((m < 0.0) != (p < 0.0))) m += p;
    return m;
}
__device__ __forceinline__ float pymod(float a, float p) {
    float m = ::fmodf(a, p);
    if (m != 0.0f && ((m < 0.0f) != (p < 0.0f))) m += p;
    return m;
}

template <typename T>
__device__ __forceinline__ bool finite_r(T v) {
    return ::isfinite(v);
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11): the device-side reset generator.
// key = 64-bit seed; counter = (env lo, env hi, episode, block).  Restated in oracle/emei_oracle.c.
struct u32x4 {
    uint32_t v[4];
};

__device__ __forceinline__ u32x4 philox4x32_10(uint64_t seed, uint64_t env, uint32_t episode, uint32_t block) {
    uint32_t c0 = (uint32_t)env, c1 = (uint32_t)(env >> 32), c2 = episode, c3 = block;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0, c1 = lo1, c2 = n2, c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return u32x4{{c0, c1, c2, c3}};
}

__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * 0x1.0p-24f; }  // [0,1)

__device__ __forceinline__ void boxmuller(uint32_t a, uint32_t b, float& z0, float& z1) {
    float u1 = ((float)(a >> 8) + 1.0f) * 0x1.0p-24f;  // (0,1]
    float u2 = u01(b);
    float rad = ::sqrtf(-2.0f * ::logf(u1));
    float ang = 6.283185307179586f * u2;
    float s, c;
    ::sincosf(ang, &s, &c);
    z0 = rad * c;
    z1 = rad * s;
}

// ---------------------------------------------------------------------------------------------
// action loads: dtype is wave-uniform, so the switch is a scalar branch
__device__ __forceinline__ int load_discrete_action(const void* p, int dtype, int64_t idx) {
    switch (dtype) {
        case EMEI_ACT_U8: return (int)((const uint8_t*)p)[idx];
        case EMEI_ACT_I32: return ((const int32_t*)p)[idx];
        case EMEI_ACT_I64: return (int)((const int64_t*)p)[idx];
        default: return (int)((const float*)p)[idx];
    }
}

}  // namespace emei
