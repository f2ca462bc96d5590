// Micro-benchmark: per-instruction cost of dependent / independent f64 and f32 FMA chains for one
// wave per SIMD (the occupancy of the 65 536-env benchmark) and two waves per SIMD, plus the
// in-kernel clock (s_memtime / s_memrealtime).  Build: hipcc --offload-arch=gfx950 -O3 issue_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T, int CHAINS>
__global__ void __launch_bounds__(256) chain_kernel(T* out, int iters, unsigned long long* clk) {
    T x[CHAINS];
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) x[k] = (T)(threadIdx.x * 1e-3 + k);
    const T a = (T)0.999999, b = (T)1e-7;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int k = 0; k < CHAINS; ++k) x[k] = __builtin_fma(x[k], a, b);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    T s = 0;
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <typename T, int CHAINS>
void run(const char* name, int blocks) {
    T* out; unsigned long long* clk; unsigned long long h[2];
    hipMalloc(&out, blocks * 256 * sizeof(T)); hipMalloc(&clk, 16);
    int iters = 4000;
    chain_kernel<T, CHAINS><<<blocks, 256>>>(out, 10, clk);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    chain_kernel<T, CHAINS><<<blocks, 256>>>(out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double ninstr = (double)iters * 16 * CHAINS;
    double clock_ghz = (double)h[0] / (double)h[1] * 0.1;
    printf("%-28s blocks=%4d: %.2f shader-cycles/instr/wave, wall %.3f ms, in-kernel clock %.2f GHz\n", name, blocks,
           (double)h[0] / ninstr, ms, clock_ghz);
    hipFree(out); hipFree(clk);
}
int main() {
    for (int blocks : {256, 512, 1024}) {
        run<double, 1>("f64 fma dependent x1", blocks);
        run<double, 2>("f64 fma 2 chains", blocks);
        run<double, 4>("f64 fma 4 chains", blocks);
        run<float, 1>("f32 fma dependent x1", blocks);
        run<float, 2>("f32 fma 2 chains", blocks);
        run<float, 4>("f32 fma 4 chains", blocks);
    }
    return 0;
}
