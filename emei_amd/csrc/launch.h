// launch.h — host-side launch descriptors shared between the ABI layer and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pendulum_envs.h"

namespace emei {

enum PendOp { PEND_OP_ROLLOUT = 0, PEND_OP_RESET, PEND_OP_GET_OBS, PEND_OP_REWARD_TERMINAL, PEND_OP_NEXT_OBS, PEND_OP_INIT_OBS };

// emei_set_obs_peers: the gathered buffers the staged rollout kernel also writes every observation row to
struct ObsPeers {
    void* obs[EMEI_MAX_OBS_PEERS] = {};
    int64_t row_envs = 0, col = 0;
    int count = 0;
    int32_t max_steps = 0;  // rows of every buffer
};

struct PendLaunch {
    int op = PEND_OP_ROLLOUT;
    int env_id = 0, precision = 0;
    int ode_method = 0;  // enum emei_ode_method (CartPole only)
    void* state = nullptr;
    int32_t* steps = nullptr;
    uint32_t* episode = nullptr;
    unsigned long long* done_mask = nullptr;
    const void* actions = nullptr;
    const void* trig = nullptr;  // device {sin,cos} table (emei_trig_table)
    const void* obs_in = nullptr;  // stateless ops: [n,4] float32, or float64 when io_f64
    int io_f64 = 0;                // stateless ops: obs_in / obs_out / reward_out are float64 (EMEI_IO_F64)
    const int64_t* env_index = nullptr;  // PEND_OP_INIT_OBS
    const uint32_t* episode_in = nullptr;
    float* obs_out = nullptr;
    double* obs_f64 = nullptr;
    float* reward_out = nullptr;
    uint8_t* done_out = nullptr;
    int64_t n = 0;
    int32_t n_steps = 1, freq_rate = 1, action_dtype = 0, max_episode_steps = 0;
    uint32_t flags = 0;
    uint64_t seed = 0, env_offset = 0;
    PendParams p;
    hipStream_t stream = nullptr;
    int* selected = nullptr;  // out: enum emei_kernel_id of the rollout kernel launched (emei_last_rollout_kernel)
    // emei_step_host (PEND_OP_ROLLOUT with n_steps = 1): obs_f64 also receives the post-step observation of the STATE in
    // float64 (what emei_get_obs would return), and for n = 1 the kernel ends by storing flag_value to *host_flag (host
    // memory, system scope, after everything else it wrote)
    uint32_t* host_flag = nullptr;
    uint32_t flag_value = 0;
    ObsPeers peers;  // PEND_OP_ROLLOUT: count > 0 -> the staged peers kernel or EMEI_ERR_UNSUPPORTED
};

// pendulum_kernels.hip
int pend_launch(const PendLaunch& L);

// abi.hip: the per-device 256-entry {sin,cos} table (allocated and filled on first use)
const void* emei_trig_table(int device);

// util_kernels.hip
int launch_state_unpack(const double* aos, void* soa, int precision, int64_t n, int dim, hipStream_t s);
int launch_state_pack(const void* soa, double* aos, int precision, int64_t n, int dim, hipStream_t s);
int launch_compact_done(const unsigned long long* masks, int64_t n, int32_t* idx_out, int32_t* count_out, hipStream_t s);

}  // namespace emei
