// pendulum_kernels.hip — step / rollout kernels of the 4-state cart/pole family for gfx950.
//
// Layout: one thread per env instance; state is struct-of-arrays in HBM (4 arrays of n Reals + a
// step counter and an episode counter per env), so every state access of a wave is one fully
// coalesced 256 B (float) / 512 B (double) line group.  Observations leave as one float4 per lane
// (row-major [n,4] float32 = a 1 KiB contiguous store per wave).  The rollout keeps the state,
// the carried sin/cos and the counters in registers for all n_steps; per step it reads one action
// and writes obs + reward + done (22 B/env-step with uint8 actions).
// There is no dense contraction anywhere on this path, hence no MFMA: the roofline is HBM.
#include "pendulum_envs.h"
#include "launch.h"

namespace emei {

template <class Env>
struct RolloutArgs {
    typename Env::real* state;  // SoA: 4 arrays of n
    int32_t* steps;             // per-env step counter (TimeLimit)
    uint32_t* episode;          // per-env episode counter (RNG counter word)
    unsigned long long* done_mask;  // one ballot word per wave: done of the LAST step
    const void* actions;
    float4* obs_out;
    float* reward_out;
    uint8_t* done_out;
    int64_t n;
    int32_t n_steps, freq_rate, action_dtype, max_episode_steps;
    uint32_t flags;
    uint64_t seed, env_offset;
    PendParams p;
};

// emei_step (n_steps = 1) and emei_rollout (n_steps = T): base_control.py:61-83 /
// mujoco_env.py:157-167 for every env of the shard, T times, without leaving the registers.
template <class Env>
__global__ void __launch_bounds__(kBlock) pend_rollout_kernel(const RolloutArgs<Env> a) {
    using R = typename Env::real;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n) return;
    const int64_t n = a.n;

    R s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = a.state[k * n + i];
    int32_t steps = a.steps[i];
    uint32_t episode = a.episode[i];
    typename Env::Carry c;
    Env::prime(s, c, a.p);

    const bool auto_reset = (a.flags & EMEI_FLAG_AUTO_RESET) != 0;
    typename Env::Action act_next = Env::load_action(a.actions, a.action_dtype, i);
    uint32_t done = 0;
    for (int t = 0; t < a.n_steps; ++t) {
        typename Env::Action act = act_next;
        if (t + 1 < a.n_steps)  // prefetch the next action under this step's arithmetic
            act_next = Env::load_action(a.actions, a.action_dtype, (int64_t)(t + 1) * n + i);

        R o[4], rew;
        bool term;
        Env::step(s, c, act, a.p, a.freq_rate, o, rew, term);
        ++steps;
        bool trunc = (a.max_episode_steps > 0) & (steps >= a.max_episode_steps);
        done = (term ? EMEI_DONE_TERMINAL : 0u) | (trunc ? EMEI_DONE_TRUNCATED : 0u);

        const int64_t off = (int64_t)t * n + i;
        if (a.obs_out) a.obs_out[off] = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
        if (a.reward_out) a.reward_out[off] = (float)rew;
        if (a.done_out) a.done_out[off] = (uint8_t)done;

        // early-termination handling: the reset path (Philox + a fresh sincos) is skipped by the
        // whole wave unless the ballot says some lane is done
        if (auto_reset && __ballot(done != 0) != 0ull) {
            if (done != 0) {
                ++episode;
                steps = 0;
                Env::init(s, a.seed, a.env_offset + (uint64_t)i, episode, a.p);
                Env::prime(s, c, a.p);
            }
        }
    }

#pragma unroll
    for (int k = 0; k < 4; ++k) a.state[k * n + i] = s[k];
    a.steps[i] = steps;
    a.episode[i] = episode;
    // done bits of the last step, one 64-bit word per wave, for emei_compact_done
    unsigned long long m = __ballot(done != 0);
    if ((threadIdx.x & (kWave - 1)) == 0) a.done_mask[i / kWave] = m;
}

// Env.reset on the device
template <class Env>
__global__ void __launch_bounds__(kBlock)
    pend_reset_kernel(typename Env::real* state, int32_t* steps, uint32_t* episode, int64_t n, uint64_t seed,
                      uint64_t env_offset, PendParams p) {
    using R = typename Env::real;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    R s[4];
    Env::init(s, seed, env_offset + (uint64_t)i, 0u, p);
#pragma unroll
    for (int k = 0; k < 4; ++k) state[k * n + i] = s[k];
    steps[i] = 0;
    episode[i] = 0;
}

// current_obs as float64 [n,4]
template <class Env>
__global__ void __launch_bounds__(kBlock)
    pend_get_obs_kernel(const typename Env::real* state, double* obs, int64_t n) {
    using R = typename Env::real;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    R s[4], o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = state[k * n + i];
    Env::obs_of(s, o);
    double2* dst = (double2*)(obs + 4 * i);
    dst[0] = make_double2((double)o[0], (double)o[1]);
    dst[1] = make_double2((double)o[2], (double)o[3]);
}

// ---------------------------------------------------------------------------------------------
// stateless batched functions on float32 [n,4] observations
// get_batch_reward / get_batch_terminal (core.py:182-188)
template <class Env>
__global__ void __launch_bounds__(kBlock)
    pend_reward_terminal_kernel(const float4* obs, float* reward, uint8_t* terminal, int64_t n, PendParams p) {
    using R = typename Env::real;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 v = obs[i];
    R o[4] = {(R)v.x, (R)v.y, (R)v.z, (R)v.w};
    // the observation IS the state for reward/terminal purposes (wrapped angle has the same cosine);
    // build the carry from it.  For InvertedPendulum o[1] is theta, prime() adds phi_off itself.
    typename Env::Carry c;
    Env::prime(o, c, p);
    if (reward) reward[i] = (float)Env::reward(o, c, p);
    if (terminal) terminal[i] = (uint8_t)Env::terminal(o, c, p);
}

// get_batch_next_obs (core.py:190-193): one step from caller-supplied observations
template <class Env>
__global__ void __launch_bounds__(kBlock)
    pend_next_obs_kernel(const float4* obs, const void* actions, int action_dtype, float4* next_obs, int64_t n,
                         int freq_rate, PendParams p) {
    using R = typename Env::real;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 v = obs[i];
    R s[4] = {(R)v.x, (R)v.y, (R)v.z, (R)v.w}, o[4], rew;
    bool term;
    typename Env::Carry c;
    Env::prime(s, c, p);
    Env::step(s, c, Env::load_action(actions, action_dtype, i), p, freq_rate, o, rew, term);
    next_obs[i] = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
}

// ---------------------------------------------------------------------------------------------
// host-side dispatch over (env id, precision)
template <template <int, typename> class Fam, int V, typename R>
static int launch_rollout_t(const PendLaunch& L) {
    using Env = Fam<V, R>;
    RolloutArgs<Env> a;
    a.state = (R*)L.state;
    a.steps = L.steps;
    a.episode = L.episode;
    a.done_mask = L.done_mask;
    a.actions = L.actions;
    a.obs_out = (float4*)L.obs_out;
    a.reward_out = L.reward_out;
    a.done_out = L.done_out;
    a.n = L.n;
    a.n_steps = L.n_steps;
    a.freq_rate = L.freq_rate;
    a.action_dtype = L.action_dtype;
    a.max_episode_steps = L.max_episode_steps;
    a.flags = L.flags;
    a.seed = L.seed;
    a.env_offset = L.env_offset;
    a.p = L.p;
    dim3 grid((unsigned)((L.n + kBlock - 1) / kBlock));
    switch (L.op) {
        case PEND_OP_ROLLOUT:
            hipLaunchKernelGGL(pend_rollout_kernel<Env>, grid, dim3(kBlock), 0, L.stream, a);
            break;
        case PEND_OP_RESET:
            hipLaunchKernelGGL(pend_reset_kernel<Env>, grid, dim3(kBlock), 0, L.stream, (R*)L.state, L.steps,
                               L.episode, L.n, L.seed, L.env_offset, L.p);
            break;
        case PEND_OP_GET_OBS:
            hipLaunchKernelGGL(pend_get_obs_kernel<Env>, grid, dim3(kBlock), 0, L.stream, (const R*)L.state,
                               L.obs_f64, L.n);
            break;
        case PEND_OP_REWARD_TERMINAL:
            hipLaunchKernelGGL(pend_reward_terminal_kernel<Env>, grid, dim3(kBlock), 0, L.stream,
                               (const float4*)L.obs_in, L.reward_out, L.done_out, L.n, L.p);
            break;
        case PEND_OP_NEXT_OBS:
            hipLaunchKernelGGL(pend_next_obs_kernel<Env>, grid, dim3(kBlock), 0, L.stream, (const float4*)L.obs_in,
                               L.actions, L.action_dtype, (float4*)L.obs_out, L.n, L.freq_rate, L.p);
            break;
        default: return EMEI_ERR_INVALID;
    }
    return hipGetLastError() == hipSuccess ? EMEI_OK : EMEI_ERR_HIP;
}

template <template <int, typename> class Fam, int V>
static int launch_prec(const PendLaunch& L) {
    return L.precision == EMEI_PRECISION_F32 ? launch_rollout_t<Fam, V, float>(L) : launch_rollout_t<Fam, V, double>(L);
}

int pend_launch(const PendLaunch& L) {
    switch (L.env_id) {
        case EMEI_CARTPOLE_SWINGUP: return launch_prec<CartPole, 0>(L);
        case EMEI_CARTPOLE_BALANCING: return launch_prec<CartPole, 1>(L);
        case EMEI_IP_REBOUND_BALANCING: return launch_prec<InvPend, 0>(L);
        case EMEI_IP_BOUNDARY_BALANCING: return launch_prec<InvPend, 1>(L);
        case EMEI_IP_REBOUND_SWINGUP: return launch_prec<InvPend, 2>(L);
        case EMEI_IP_BOUNDARY_SWINGUP: return launch_prec<InvPend, 3>(L);
        default: return EMEI_ERR_UNSUPPORTED;
    }
}

}  // namespace emei
