// Accuracy of the hardware transcendentals a cheap float32 Box-Muller would use (v_log_f32, v_sqrt_f32, v_sin_f32 /
// v_cos_f32 with the angle in turns), against float64 libm, over ALL 2^24 inputs the generator can produce; and of a
// polynomial sincos on the exact turn fraction.  build: hipcc -O3 --offload-arch=gfx950 tools/bm_accuracy.hip -o tools/bm_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
__device__ inline void sincos_turns_poly(float t, float& s, float& c) {  // t in [0,1)
    const float k = __builtin_rintf(4.0f * t);
    const float x = (t - 0.25f * k) * 6.283185307179586f;  // |x| <= pi/4, the subtraction is exact
    const float z = x * x;
    float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
    float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
    const float sr = __builtin_fmaf(z * x, ps, x), cr = __builtin_fmaf(z, __builtin_fmaf(z, pc, -0.5f), 1.0f);
    const int q = (int)k;
    const float s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}
// err[0..]: max abs errors, reduced with atomicMax on the float bits (non-negative floats order like ints)
__global__ void k(unsigned* err) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;  // 0 .. 2^24-1
    const float t = (float)i * 0x1.0p-24f;
    const double ang = 6.283185307179586476925 * (double)i * 0x1.0p-24;
    const double sd = sin(ang), cd = cos(ang);
    const float hs = __builtin_amdgcn_sinf(t), hc = __builtin_amdgcn_cosf(t);
    float ps, pc;
    sincos_turns_poly(t, ps, pc);
    const float u1 = ((float)i + 1.0f) * 0x1.0p-24f;
    const double radd = sqrt(-2.0 * log((double)u1));
    const float radh = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    const float radl = ::sqrtf(-2.0f * ::logf(u1));
    auto upd = [&](int j, double e) { atomicMax(&err[j], __float_as_uint((float)fabs(e))); };
    upd(0, hs - sd), upd(1, hc - cd), upd(2, ps - sd), upd(3, pc - cd);
    upd(4, (radh - radd) / radd), upd(5, (radl - radd) / radd), upd(6, radh - radd), upd(7, radl - radd);
    float ls, lc;
    ::sincosf(6.283185307179586f * t, &ls, &lc);
    upd(8, ls - sd), upd(9, lc - cd);
}
int main() {
    unsigned* d;
    hipMalloc(&d, 64);
    hipMemset(d, 0, 64);
    k<<<(1 << 24) / 256, 256>>>(d);
    unsigned h[16];
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    auto f = [&](int j) { union { unsigned u; float x; } v; v.u = h[j]; return v.x; };
    printf("angle = 2 pi i / 2^24, all i: max abs err  v_sin_f32 %.3e  v_cos_f32 %.3e | polynomial on the turn fraction sin %.3e cos %.3e | libm sincosf(2 pi t rounded) sin %.3e cos %.3e\n",
           f(0), f(1), f(2), f(3), f(8), f(9));
    printf("radius sqrt(-2 ln u), u = (i+1) / 2^24, all i: max rel err  hardware log2/sqrt %.3e  libm logf/sqrtf %.3e ; max abs err %.3e / %.3e\n",
           f(4), f(5), f(6), f(7));
    return 0;
}
