// cheetah.h — host entry points of the HalfCheetah-style 9-DoF planar body (cheetah.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace emei {

int cheetah_reset(void* state, int32_t* steps, uint32_t* episode, int64_t n, int precision, uint64_t seed,
                  uint64_t env_offset, double init_noise, hipStream_t s);
int cheetah_rollout(void* state, int32_t* steps, uint32_t* episode, unsigned long long* done_mask, int64_t n,
                    int precision, int32_t n_steps, int32_t freq_rate, double dt, int32_t max_episode_steps,
                    uint64_t seed, uint64_t env_offset, double init_noise, const float* actions, float* obs_out,
                    float* reward_out, uint8_t* done_out, uint32_t flags, hipStream_t s);
int cheetah_init_obs(int64_t count, const int64_t* env_index, const uint32_t* episode, uint64_t seed, uint64_t env_offset,
                     double init_noise, float* obs_out, hipStream_t s);
int cheetah_reward(int64_t n, const float* obs, const float* pre_obs, const float* action, double dt_env,
                   float* reward_out, hipStream_t s);
int cheetah_terminal(int64_t n, const float* obs, uint8_t* terminal_out, hipStream_t s);

}  // namespace emei
