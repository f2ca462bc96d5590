// cheetah.hip — HalfCheetah-style body (emei/envs/mujoco/half_cheetah.py).
#include "cheetah.h"

#include "emei_device.h"

namespace emei {

// half_cheetah.py:59-63 with step() semantics (B = 1 per env: the control cost is summed per env;
// the reference's batch form sums np.square(action) over the WHOLE batch, a quirk documented in
// DESIGN.md).  dt_env = real_time_scale * freq_rate (gym MujocoEnv.dt).  w_f = 1, w_c = 0.1 (:23-24).
__global__ void __launch_bounds__(kBlock)
    cheetah_reward_kernel(const float* obs, const float* pre_obs, const float* action, float inv_dt, float* reward,
                          int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float fwd = (obs[i * 18] - pre_obs[i * 18]) * inv_dt;
    float cost = 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float a = action[i * 6 + k];
        cost += a * a;
    }
    reward[i] = 1.0f * fwd - 0.1f * cost;
}

// half_cheetah.py:65-67: terminal = not all-finite(obs)
__global__ void __launch_bounds__(kBlock) cheetah_terminal_kernel(const float* obs, uint8_t* terminal, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    bool fin = true;
#pragma unroll
    for (int k = 0; k < 18; ++k) fin &= finite_r(obs[i * 18 + k]);
    terminal[i] = (uint8_t)!fin;
}

int cheetah_reward(int64_t n, const float* obs, const float* pre_obs, const float* action, double dt_env,
                   float* reward_out, hipStream_t s) {
    dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(cheetah_reward_kernel, grid, dim3(kBlock), 0, s, obs, pre_obs, action, (float)(1.0 / dt_env),
                       reward_out, n);
    return hipGetLastError() == hipSuccess ? EMEI_OK : EMEI_ERR_HIP;
}
int cheetah_terminal(int64_t n, const float* obs, uint8_t* terminal_out, hipStream_t s) {
    dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(cheetah_terminal_kernel, grid, dim3(kBlock), 0, s, obs, terminal_out, n);
    return hipGetLastError() == hipSuccess ? EMEI_OK : EMEI_ERR_HIP;
}

int cheetah_reset(void*, int32_t*, uint32_t*, int64_t, int, uint64_t, uint64_t, double, hipStream_t) {
    return EMEI_ERR_UNSUPPORTED;
}
int cheetah_rollout(void*, int32_t*, uint32_t*, unsigned long long*, int64_t, int, int32_t, int32_t, double, int32_t,
                    uint64_t, uint64_t, double, const float*, float*, float*, uint8_t*, uint32_t, hipStream_t) {
    return EMEI_ERR_UNSUPPORTED;
}

}  // namespace emei
