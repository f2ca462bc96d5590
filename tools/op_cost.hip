// Micro-benchmark: SIMD cycles per wave-instruction for the VALU ops of the env-step kernels, at 4 waves
// per SIMD (throughput) and 1 wave per SIMD, each op in 8 independent chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ void __launch_bounds__(256) k(double* out, int iters, unsigned long long* clk) {
    double d[8]; float f[8]; int n[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { d[j] = 1.0 + threadIdx.x * 1e-3 + j; f[j] = (float)d[j]; n[j] = j + threadIdx.x; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[j]) : "v"(d[(j + 1) & 7]), "v"(d[(j + 2) & 7]));
                if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[j]) : "v"(d[(j + 1) & 7]));
                if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[j]) : "v"(d[(j + 1) & 7]));
                if (OP == 3) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[j]) : "v"(d[j]));
                if (OP == 4) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[j]) : "v"(f[j]));
                if (OP == 5) asm volatile("v_rcp_f64 %0, %1" : "=v"(d[j]) : "v"(d[(j + 1) & 7]));
                if (OP == 6) asm volatile("v_rndne_f64 %0, %1" : "=v"(d[j]) : "v"(d[(j + 1) & 7]));
                if (OP == 7) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[j]) : "v"(d[j]));
                if (OP == 8) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j]) : "v"(f[(j + 1) & 7]), "v"(f[(j + 2) & 7]));
                if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[j]) : "v"(n[(j + 1) & 7]));
                if (OP == 10) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[j]) : "v"(f[(j + 1) & 7]));
                if (OP == 11) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(n[j]) : "v"(n[(j + 1) & 7]));
                if (OP == 12) asm volatile("v_mov_b64 %0, %1" : "=v"(d[j]) : "v"(d[(j + 1) & 7]));
                if (OP == 13) asm volatile("v_cmp_lt_f64 vcc, %0, %1" ::"v"(d[j]), "v"(d[(j + 1) & 7]) : "vcc");
                if (OP == 14) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(n[j]) : "v"(n[(j + 1) & 7]));
                if (OP == 15) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[j]) : "v"(d[(j + 1) & 7]));
                if (OP == 17) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(n[j]) : "v"(n[(j + 1) & 7]) : "s20", "s21");
                if (OP == 18) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(n[j]) : "v"(n[(j + 1) & 7]), "v"(n[(j + 2) & 7]));
                if (OP == 19) asm volatile("v_bfe_i32 %0, %1, 0, 1" : "=v"(n[j]) : "v"(n[(j + 1) & 7]));
                if (OP == 20) asm volatile("v_cmp_eq_u32 vcc, 0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(n[j]) : "v"(n[(j + 1) & 7]), "v"(n[(j + 2) & 7]) : "vcc");
                if (OP == 21) asm volatile("v_and_b32 %0, %0, %1" : "+v"(n[j]) : "v"(n[(j + 1) & 7]));
                if (OP == 22) asm volatile("v_lshlrev_b32 %0, 30, %1" : "=v"(n[j]) : "v"(n[(j + 1) & 7]));
                if (OP == 16) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[j]) : "v"(n[j]), "v"(n[(j + 1) & 7]) : "vcc");
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int j = 0; j < 8; ++j) s += d[j] + f[j] + n[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int OP> void run(const char* name) {
    double* out; unsigned long long* clk; unsigned long long h;
    hipMalloc(&out, 1024 * 256 * 8); hipMalloc(&clk, 8);
    const int iters = 2000;
    double res[2], wall[2];
    int bl[2] = {256, 1024};
    for (int b = 0; b < 2; ++b) {
        k<OP><<<bl[b], 256>>>(out, 10, clk); hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); k<OP><<<bl[b], 256>>>(out, iters, clk); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
        double ninstr = (double)iters * 32;
        double waves_per_simd = bl[b] / 256.0;
        res[b] = (double)h / ninstr;  // wave cycles (s_memtime) per instruction
        wall[b] = ms * 1e6 / (ninstr * waves_per_simd);  // wall ns per wave-instruction per SIMD
    }
    printf("%-16s  1 wave/SIMD: %5.2f cyc/instr/wave, %5.2f ns/instr/SIMD   4 waves/SIMD: %5.2f cyc/instr/wave, %5.2f ns/instr/SIMD\n", name, res[0], wall[0], res[1], wall[1]);
    hipFree(out); hipFree(clk);
}
int main() {
    run<0>("v_fma_f64"); run<1>("v_mul_f64"); run<2>("v_add_f64"); run<3>("v_cvt_f32_f64"); run<4>("v_cvt_f64_f32");
    run<5>("v_rcp_f64"); run<6>("v_rndne_f64"); run<7>("v_cvt_i32_f64"); run<8>("v_fma_f32"); run<9>("v_cndmask_b32");
    run<10>("v_mul_f32"); run<11>("v_xor_b32"); run<12>("v_mov_b64"); run<13>("v_cmp_lt_f64"); run<14>("v_mul_hi_u32");
    run<15>("v_pk_mul_f32"); run<16>("v_mad_u64_u32"); run<17>("v_cndmask_e64 sgpr"); run<18>("v_bfi_b32"); run<19>("v_bfe_i32");
    run<20>("v_cmp+v_cndmask"); run<21>("v_and_b32"); run<22>("v_lshlrev_b32");
    return 0;
}
