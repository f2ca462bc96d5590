// util_kernels.hip — state marshalling and wavefront-ballot compaction of done envs (gfx950).
#include "launch.h"

namespace emei {

// [n,dim] float64 row-major (the reference's `self.state` rows) -> SoA of R
template <typename R>
__global__ void __launch_bounds__(kBlock) state_unpack_kernel(const double* aos, R* soa, int64_t n, int dim) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < dim; ++k) soa[(int64_t)k * n + i] = (R)aos[i * dim + k];
}
template <typename R>
__global__ void __launch_bounds__(kBlock) state_pack_kernel(const R* soa, double* aos, int64_t n, int dim) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < dim; ++k) aos[i * dim + k] = (double)soa[(int64_t)k * n + i];
}

int launch_state_unpack(const double* aos, void* soa, int precision, int64_t n, int dim, hipStream_t s) {
    dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    if (precision == EMEI_PRECISION_F32)
        hipLaunchKernelGGL(state_unpack_kernel<float>, grid, dim3(kBlock), 0, s, aos, (float*)soa, n, dim);
    else
        hipLaunchKernelGGL(state_unpack_kernel<double>, grid, dim3(kBlock), 0, s, aos, (double*)soa, n, dim);
    return hipGetLastError() == hipSuccess ? EMEI_OK : EMEI_ERR_HIP;
}
int launch_state_pack(const void* soa, double* aos, int precision, int64_t n, int dim, hipStream_t s) {
    dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    if (precision == EMEI_PRECISION_F32)
        hipLaunchKernelGGL(state_pack_kernel<float>, grid, dim3(kBlock), 0, s, (const float*)soa, aos, n, dim);
    else
        hipLaunchKernelGGL(state_pack_kernel<double>, grid, dim3(kBlock), 0, s, (const double*)soa, aos, n, dim);
    return hipGetLastError() == hipSuccess ? EMEI_OK : EMEI_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------
// Compaction.  The step kernels leave one ballot word per wave (bit b = env 64*w+b is done).  A
// single 1024-thread workgroup turns those words into the SORTED list of done env indices:
// popcount -> workgroup exclusive scan -> every wave expands 64 mask words cooperatively, lane b
// writing env 64*w+b at offset + popcount(mask & lanes_below(b)), so the index stores of one mask
// word are contiguous.  Deterministic (no atomics); output is ascending.
constexpr int kCompactBlock = 1024;

__global__ void __launch_bounds__(kCompactBlock)
    compact_done_kernel(const unsigned long long* masks, int64_t n_words, int64_t n_envs, int32_t* idx_out,
                        int32_t* count_out) {
    __shared__ int wave_tot[kCompactBlock / kWave];
    __shared__ int running_s;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
    if (tid == 0) running_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_words; base += kCompactBlock) {
        const int64_t w = base + tid;
        unsigned long long m = (w < n_words) ? masks[w] : 0ull;
        // the last word may carry lanes beyond n_envs only as zeros (inactive lanes never ballot)
        int cnt = __popcll(m);
        // inclusive scan inside the wave
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == kWave - 1) wave_tot[wv] = incl;
        __syncthreads();
        int wave_base = 0, block_tot = 0;
        for (int k = 0; k < kCompactBlock / kWave; ++k) {
            int t = wave_tot[k];
            if (k < wv) wave_base += t;
            block_tot += t;
        }
        const int running = running_s;
        int off = running + wave_base + incl - cnt;  // exclusive offset of this lane's word
        // cooperative expansion: word k of this wave is handled by all 64 lanes
        for (int k = 0; k < kWave; ++k) {
            unsigned long long mk = __shfl(m, k);
            if (mk == 0ull) continue;  // wave-uniform
            int offk = __shfl(off, k);
            int64_t wk = base + (int64_t)wv * kWave + k;
            if ((mk >> lane) & 1ull) {
                int pos = __popcll(mk & ((1ull << lane) - 1ull));
                idx_out[offk + pos] = (int32_t)(wk * kWave + lane);
            }
        }
        __syncthreads();
        if (tid == 0) running_s = running + block_tot;
        __syncthreads();
    }
    if (tid == 0) *count_out = running_s;
    (void)n_envs;
}

int launch_compact_done(const unsigned long long* masks, int64_t n, int32_t* idx_out, int32_t* count_out,
                        hipStream_t s) {
    int64_t n_words = (n + kWave - 1) / kWave;
    hipLaunchKernelGGL(compact_done_kernel, dim3(1), dim3(kCompactBlock), 0, s, masks, n_words, n, idx_out, count_out);
    return hipGetLastError() == hipSuccess ? EMEI_OK : EMEI_ERR_HIP;
}

}  // namespace emei
