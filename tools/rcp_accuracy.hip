// Accuracy of the v_rcp_f64 seed and of 1 / 2 Newton steps on it, over the ranges the env kernels divide by.
// build: hipcc -O3 --offload-arch=gfx950 tools/rcp_accuracy.hip -o tools/rcp_accuracy ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* d, double* r0, double* r1, double* r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = d[i], a = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, a, 1.0), b = __builtin_fma(a, e, a);
    e = __builtin_fma(-x, b, 1.0);
    double c = __builtin_fma(b, e, b);
    r0[i] = a, r1[i] = b, r2[i] = c;
}
int main() {
    const int n = 1 << 22;
    std::vector<double> h(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        double u = (double)(s >> 11) * (1.0 / 9007199254740992.0);
        h[i] = (i & 1) ? 0.05 + u * 20.0 : std::ldexp(1.0 + u, (int)(s % 40) - 20);  // (0.05, 20) and 2^-20 .. 2^20
    }
    double *d, *r0, *r1, *r2;
    hipMalloc(&d, n * 8), hipMalloc(&r0, n * 8), hipMalloc(&r1, n * 8), hipMalloc(&r2, n * 8);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(d, r0, r1, r2, n);
    std::vector<double> a(n), b(n), c(n);
    hipMemcpy(a.data(), r0, n * 8, hipMemcpyDeviceToHost), hipMemcpy(b.data(), r1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), r2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / (long double)h[i];
        m0 = std::fmax(m0, (double)fabsl(((long double)a[i] - t) / t));
        m1 = std::fmax(m1, (double)fabsl(((long double)b[i] - t) / t));
        m2 = std::fmax(m2, (double)fabsl(((long double)c[i] - t) / t));
    }
    printf("v_rcp_f64 max rel err: seed %.3e (2^%.1f)  +1 Newton %.3e  +2 Newton %.3e  (ulp/2 = 1.11e-16)\n", m0, std::log2(m0), m1, m2);
    return 0;
}
