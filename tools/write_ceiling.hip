// Micro-benchmark: the write stream of the staged rollout kernel WITHOUT its arithmetic — the same grid (one thread per env,
// 256-thread blocks), the same stores per step (one float4 of observations per lane at [t][i]; every 4 steps one float4 per lane
// covering 4 reward rows; every 16 steps one uint4 per lane covering 16 done rows), nothing else.  What this takes is the
// ceiling of the access pattern itself: bench.py's kernel cannot be faster than its own stores.
//   hipcc -O3 --offload-arch=gfx950 tools/write_ceiling.hip -o tools/write_ceiling && tools/write_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <bool ALL>
__global__ void __launch_bounds__(256) k(float4* obs, float* rew, uint8_t* done, int T, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t i0 = i - lane;
    float4 v = make_float4((float)i, 1.f, 2.f, 3.f);
    const int64_t rew_off = (int64_t)(lane >> 4) * n + i0 + ((lane & 15) << 2);
    const int64_t done_off = (int64_t)(lane >> 2) * n + i0 + ((lane & 3) << 4);
    for (int t = 0; t < T; ++t) {
        v.x += 1.f;  // a data dependence per step, so that the stores are not merged or hoisted
        obs[(int64_t)t * n + i] = v;
        if (ALL) {
            if ((t & 3) == 3) *(float4*)(rew + (int64_t)(t - 3) * n + rew_off) = v;
            if ((t & 15) == 15) *(uint4*)(done + (int64_t)(t - 15) * n + done_off) = make_uint4(t, t, t, t);
        }
    }
}

int main() {
    const int T = 1000;
    for (int64_t n : {65536LL, 131072LL, 262144LL, 1048576LL}) {
        float4* obs; float* rew; uint8_t* done;
        hipMalloc(&obs, sizeof(float4) * n * T); hipMalloc(&rew, 4 * n * T); hipMalloc(&done, n * T);
        for (int all = 0; all < 2; ++all) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float best = 1e9f, sum = 0;
            for (int rep = 0; rep < 12; ++rep) {
                hipEventRecord(e0);
                if (all) k<true><<<n / 256, 256>>>(obs, rew, done, T, n); else k<false><<<n / 256, 256>>>(obs, rew, done, T, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2) { best = ms < best ? ms : best; sum += ms; }
            }
            const double bytes = (all ? 21.0 : 16.0) * n * T;
            printf("n = %7lld, T = %d, %s: mean %.3f ms (%.2f TB/s), best %.3f ms (%.2f TB/s)\n", (long long)n, T,
                   all ? "obs + reward + done stores (21 B per env-step)" : "obs stores only (16 B per env-step)        ", sum / 10,
                   bytes / (sum / 10) * 1e-9, best, bytes / best * 1e-9);
        }
        hipFree(obs); hipFree(rew); hipFree(done);
    }
    return 0;
}
