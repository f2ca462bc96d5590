// Micro-benchmark: what one global_store_dwordx4 per iteration costs a wave that otherwise runs a
// serial f64 FMA chain (1 wave per SIMD), as a function of where the stored registers come from.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>  // 0: no store, 1: store float4 every iter, 2: store every iter from cvt of chain values, 3: nontemporal store
__global__ void __launch_bounds__(256) k(float4* out, int iters, int n, unsigned long long* clk) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double x = threadIdx.x * 1e-3, y = x + 1, z = x + 2, w = x + 3;
    const double a = 0.999999, b = 1e-7;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < iters; ++t) {
#pragma unroll
        for (int u = 0; u < 25; ++u) {
            x = __builtin_fma(x, a, b);
            y = __builtin_fma(y, a, x);
            z = __builtin_fma(z, a, y);
            w = __builtin_fma(w, a, z);
        }
        if (MODE == 1 || MODE == 2) out[(size_t)t * n + i] = make_float4((float)x, (float)y, (float)z, (float)w);
        if (MODE == 3) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 v = {(float)x, (float)y, (float)z, (float)w};
            __builtin_nontemporal_store(v, (f4*)&out[(size_t)t * n + i]);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) out[i] = make_float4((float)x, (float)y, (float)z, (float)w);
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int MODE>
void run(const char* name, float4* out, int n, int iters) {
    unsigned long long* clk; unsigned long long h;
    hipMalloc(&clk, 8);
    k<MODE><<<n / 256, 256>>>(out, 10, n, clk);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<n / 256, 256>>>(out, iters, n, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("%-34s n=%7d: %.1f cycles/iter (100 dep-chain FMAs), wall %.3f ms, %.0f GB/s\n", name, n, (double)h / iters, ms,
           MODE ? (double)n * 16 * iters / ms / 1e6 : 0.0);
    hipFree(clk);
}
int main() {
    const int iters = 1000;
    float4* out; hipMalloc(&out, (size_t)262144 * 16 * iters);
    for (int n : {65536, 131072, 262144}) {
        run<0>("no store", out, n, iters);
        run<1>("dwordx4 store per iter", out, n, iters);
        run<3>("nontemporal dwordx4 store per iter", out, n, iters);
    }
    return 0;
}
