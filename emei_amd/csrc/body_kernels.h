// body_kernels.h — step / rollout / reset kernels shared by the MuJoCo-backed bodies (HalfCheetah-style
// 9-DoF body, Hopper, InvertedDoublePendulum, InvertedPendulum with a non-default integrator / noise).
// The integrators (mujoco_env.py:70-79,86-97), the Gaussian init / observation noise (:98-104,:197-249),
// the device reset and the LDS staging are generic and live here; a `Body` type supplies:
//
//   using real;  static constexpr int NS (state = qpos ++ qvel), NO (obs), NA (action);
//   struct Model; make_model(dt, env_params)        // run-time constants, passed by value as a kernel argument
//   struct Warm; begin_stages(warm); accel(q, v, ctrl, m, hd, qacc, trig, warm)   // forward dynamics incl. soft constraints; `warm` carries the
//                                                   // constraint solver's start from one evaluation to the next
//                                                   // WITHIN an env-step (reset at every step); `hd` = dt when
//                                                   // joint damping is integrated implicitly (MuJoCo Euler), else 0;
//                                                   // `trig` = the {sin,cos} table staged in LDS (emei_device.h)
//   outputs(s, pre, ctrl, m, freq_rate, obs, rew, terminal, trig)   // obs / reward / terminal of a finished step
//   init_base(s)                                    // non-zero entries of init_qpos (added after the init noise)
//   obs_of(s, o)                                    // observation of a state (float64, for emei_get_obs)
//   batch_reward(obs, pre_obs, act, m, freq_rate) / batch_terminal(obs, m)   // stateless, float32 or float64 rows
//   kHasCtrlCost, ctrl_cost(act)                    // the reward has a control-cost term w_ctrl * sum a^2 (whole-batch quirk mode)
//   kSpareReset                                     // episodes end per lane (terminal states): keep a spare init state
//   kMinWavesPerEU                                  // register cap of the rollout kernel (1 = none)
//   park(s)                                         // state of the padding lanes of a ragged last wave
//   kObsIsState                                     // the observation determines the state (get_batch_next_obs)
//   kUnrollRK4                                      // RK4 stages as straight-line code (see body_substep)
//   kScratchPerLane                                 // elements of `real` of block LDS per lane that accel() may use (0: none)
//
// Layout: one thread per env, state SoA in HBM ([NS][n] Reals), registers across a rollout.  An env's
// observation (NO floats) and action (NA floats) are wider than one lane access, so each wave stages
// its 64 envs through a private LDS slice at every step boundary: the wave's action block
// (64*NA contiguous floats) arrives as 16 B-per-lane loads and is read back per lane; the 64 x NO
// observation block is written to LDS lane-wise and leaves as contiguous 16 B-per-lane stores
// (4.5 KiB per wave and step for the cheetah).  LDS slices are wave-private: no barrier.
#pragma once
#include "emei_device.h"

namespace emei {

// constructor parameters of the reward / terminal functions that differ from the reference's defaults
// (emei_hip.h: enum emei_env_param)
struct EnvParams {
    uint32_t mask = 0;
    double v[EMEI_MAX_ENV_PARAMS] = {0};
    double get(int k, double dflt) const { return (mask >> k) & 1u ? v[k] : dflt; }
};

// per-coordinate sigmas (state order: qpos then qvel) of the Gaussian init noise and of the
// per-substep observation noise, float32 draws; host form (any body) and kernel-argument form
struct NoiseSpec {
    float init[EMEI_MAX_STATE_DIM] = {0}, obs[EMEI_MAX_STATE_DIM] = {0};
    int32_t shared = 0;  // EMEI_NOISE_SHARED: one draw for all of qpos, one for all of qvel (sigmas of joint 0)
};
template <int NS>
struct NoiseArgs {
    float init[NS], obs[NS];
    int32_t shared, obs_on;
    NoiseArgs() = default;
    explicit NoiseArgs(const NoiseSpec& h) : shared(h.shared), obs_on(0) {
        for (int i = 0; i < NS; ++i) init[i] = h.init[i], obs[i] = h.obs[i], obs_on |= h.obs[i] != 0.f;
    }
};
// key tweak of the observation-noise stream (the reset stream uses the plain seed)
constexpr uint64_t kObsNoiseKey = 0x6F62736E6F697365ull;

// s (+)= sig[i] * N(0,1).  Draws: Philox counter (env, episode, blk0 + b), four normals per block in
// coordinate order; SHARED uses the first two normals of block blk0 for every position / velocity
// coordinate, scaled by the sigmas of joint 0 (the reference's B = 1 behaviour, mujoco_env.py:243-244).
template <typename R, int NS, bool ASSIGN>
__device__ __forceinline__ void gauss_state(R (&s)[NS], uint64_t key, uint64_t env, uint32_t episode, uint32_t blk0,
                                            const float (&sig)[NS], bool shared) {
    constexpr int NB = (NS + 3) / 4;
    if (shared) {
        u32x4 r = philox4x32_10(key, env, episode, blk0);
        float z0, z1;
        boxmuller(r.v[0], r.v[1], z0, z1);
        const R dp = (R)__fmul_rn(sig[0], z0), dv = (R)__fmul_rn(sig[NS / 2], z1);
#pragma unroll
        for (int i = 0; i < NS; ++i) s[i] = (ASSIGN ? R(0) : s[i]) + (i < NS / 2 ? dp : dv);
        return;
    }
    float z[4 * NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        u32x4 r = philox4x32_10(key, env, episode, blk0 + (uint32_t)b);
        boxmuller(r.v[0], r.v[1], z[4 * b], z[4 * b + 1]);
        boxmuller(r.v[2], r.v[3], z[4 * b + 2], z[4 * b + 3]);
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const R d = (R)__fmul_rn(sig[i], z[i]);
        s[i] = ASSIGN ? d : s[i] + d;
    }
}
// device reset: init_qpos / init_qvel + init noise (mujoco_env.py:130-140); Body::init_base adds the
// non-zero entries of init_qpos (the Hopper's rootz ref)
template <class Body>
__device__ __forceinline__ void body_init(typename Body::real (&s)[Body::NS], uint64_t seed, uint64_t env, uint32_t episode,
                                          const NoiseArgs<Body::NS>& ns) {
    gauss_state<typename Body::real, Body::NS, true>(s, seed, env, episode, 0u, ns.init, ns.shared != 0);
    Body::init_base(s);
}

// One substep of mujoco_env.py:91-97.  RK4 = false: MuJoCo's Euler velocity update, then either emei's
// position override from the OLD velocity (`semi` false, :94-97,189-191) or MuJoCo's own
// semi-implicit position update from the NEW velocity.  RK4 = true: mj_RungeKutta(4) — stage
// states X_i = X_0 + dt*a_i*F_{i-1} (a = 1/2, 1/2, 1), X' = X_0 + dt*sum b_i F_i (b = 1/6, 1/3, 1/3, 1/6),
// F = (v, qacc(q, v)) with the full forward dynamics (constraints included) at every stage.
template <class Body, bool RK4>
__device__ __forceinline__ void body_substep(typename Body::real (&s)[Body::NS], const typename Body::real (&ctrl)[Body::NA],
                                             const typename Body::Model& m, bool semi, const TrigCtx& trig,
                                             typename Body::Warm& warm) {
    using R = typename Body::real;
    constexpr int NV = Body::NS / 2;
    const R dt = (R)m.dt;
    R q[NV], v[NV], acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) q[i] = s[i], v[i] = s[NV + i];
    if constexpr (!RK4) {
        // Euler: every evaluation starts cold.  A/B on one box (config 4 / Hopper): 10.1 vs 10.9 ms and 8.0 vs 8.2 ms per
        // 100 steps — the previous substep's minimiser is no nearer to the new one than the unconstrained acceleration
        // is (either way a lane needs one step and the pass that confirms it), and carrying it costs registers (round 3, with the
        // Hopper's verify sweep switched on for it: 8.00 vs 8.52 ms; config 4: 8.05 vs 8.16).  The RK4
        // stages below DO share it: 25.6 vs 28.9 ms (Hopper), 37.3 vs 41.4 ms (cheetah); resetting it per substep loses half
        // of that (27.6 / 39.8).
        typename Body::Warm cold{};
        Body::accel(q, v, ctrl, m, dt, acc, trig, cold);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const R vn = fma_r(dt, acc[i], v[i]);
            s[i] = fma_r(dt, semi ? vn : v[i], q[i]);
            s[NV + i] = vn;
        }
    } else {
        R qs[NV], vs[NV], dq[NV], dv[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) qs[i] = q[i], vs[i] = v[i], dq[i] = R(0), dv[i] = R(0);
        // Body::kUnrollRK4: the four stages as straight-line code or as a rolled loop.  The cheetah MUST unroll:
        // with `#pragma unroll 1` this hipcc produced wrong front-leg accelerations for that instantiation (256 VGPR
        // + 162 AGPR, 326 spilled SGPRs) while the same source is correct for every other body and for the unrolled
        // form (caught by tests/test_gpu_integrators.py).  The Hopper must NOT: unrolled under its 256-register cap it
        // spills six times the algorithmic bytes to scratch.
        Body::begin_stages(warm);  // the evaluations of this substep share `warm` (a body may iterate differently then)
        auto stage = [&](int st) __attribute__((always_inline)) {
            Body::accel(qs, vs, ctrl, m, R(0), acc, trig, warm);
            const R b = (st == 0 || st == 3) ? R(1.0 / 6.0) : R(1.0 / 3.0);
            const R h = dt * (st == 2 ? R(1) : R(0.5));  // step to the NEXT stage state
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                dq[i] = fma_r(b, vs[i], dq[i]), dv[i] = fma_r(b, acc[i], dv[i]);
                qs[i] = fma_r(h, vs[i], q[i]);
                vs[i] = fma_r(h, acc[i], v[i]);
            }
        };
        if constexpr (Body::kUnrollRK4) {
            stage(0), stage(1), stage(2), stage(3);
        } else {
#pragma unroll 1
            for (int st = 0; st < 4; ++st) stage(st);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) s[i] = fma_r(dt, dq[i], q[i]), s[NV + i] = fma_r(dt, dv[i], v[i]);
    }
}

enum BodyOp { BODY_OP_ROLLOUT = 0, BODY_OP_RESET, BODY_OP_GET_OBS, BODY_OP_INIT_OBS, BODY_OP_REWARD, BODY_OP_TERMINAL, BODY_OP_NEXT_OBS };

// host-side launch descriptor (abi.hip -> body_dispatch.hip -> body_tu.hip)
struct BodyLaunch {
    int op = BODY_OP_ROLLOUT, env_id = 0, precision = 0;
    void* state = nullptr;
    int32_t* steps = nullptr;
    uint32_t* episode = nullptr;
    unsigned long long* done_mask = nullptr;
    const void* actions = nullptr;     // rollout / next_obs: float32 [.., n, NA]; BODY_OP_REWARD: dtype of obs_in
    const void* obs_in = nullptr;      // stateless ops: [n, NO] float32, or float64 when io_f64
    const void* pre_obs_in = nullptr;
    int io_f64 = 0;                    // stateless ops: obs_in / pre_obs_in / obs_out / reward_out (and the reward's actions) are float64
    double* batch_cost_scratch = nullptr;  // BODY_OP_REWARD with EMEI_REWARD_BATCH_CTRL_COST: 8 B of device scratch
    const int64_t* env_index = nullptr;
    const uint32_t* episode_in = nullptr;
    float* obs_out = nullptr;
    double* obs_f64 = nullptr;
    float* reward_out = nullptr;
    uint8_t* done_out = nullptr;
    int64_t n = 0;
    int32_t n_steps = 1, freq_rate = 1, max_episode_steps = 0;
    uint32_t flags = 0;
    uint64_t seed = 0, env_offset = 0;
    double dt = 0.002;
    int32_t integrator = 0;
    int32_t solver = 0;  // enum emei_solver (bodies with several simultaneous constraints)
    NoiseSpec noise;
    EnvParams env_params;
    const void* trig = nullptr;
    unsigned long long* cap_hits = nullptr;  // device counter of the handle
    hipStream_t stream = nullptr;
    int* selected = nullptr;  // out: enum emei_kernel_id of the rollout kernel launched
};
int body_launch(const BodyLaunch& L);  // body_dispatch.hip

template <class Body>
struct BodyArgs {
    typename Body::real* state;
    int32_t* steps;
    uint32_t* episode;
    unsigned long long* done_mask;
    const float* actions;
    float* obs_out;
    float* reward_out;
    uint8_t* done_out;
    int64_t n;
    int32_t n_steps, freq_rate, max_episode_steps;
    uint32_t flags;
    uint64_t seed, env_offset;
    int32_t semi;  // EMEI_INTEG_SEMI_IMPLICIT
    NoiseArgs<Body::NS> noise;
    const SinCosEntry* trig;  // 256-entry {sin,cos} table of this device (abi.hip:emei_trig_table)
    unsigned long long* cap_hits;  // handle counter (emei_device.h:report_cap_hit)
    typename Body::Model m;
};

// emei_step / emei_rollout (mujoco_env.py:157-167) for every env of the shard
// Body::kMinWavesPerEU = 2 caps the kernel at 256 registers so that two waves share a SIMD (measured per
// body: it pays for the Hopper, whose RK4 working set then spills little; the cheetah spills 660 B/lane
// to scratch at that cap and is faster with one resident wave and AGPRs as spill space)
// Threads per block of the rollout kernel.  The bodies that run at ONE wave per SIMD (the register file is the limit) use
// one-wave blocks: the dispatcher then refills a SIMD as soon as its wave ends, instead of a CU waiting for the slowest of a
// four-wave block (131 072 envs = two rounds of 1024 waves: with 256-thread blocks a CU runs exactly two blocks back to back).
// A/B on one box, 256 -> 64 threads: cheetah 8.06 -> 7.67 ms per 100 steps, Hopper RK4 26.46 -> 25.99, Euler 8.19 -> 8.11, the
// double pendulum unchanged; results bit-identical (a lane's arithmetic does not depend on its block).  -DEMEI_BODY_BLOCK=n
// overrides it for experiments.
template <class Body>
__host__ __device__ constexpr int rollout_block() {
#ifdef EMEI_BODY_BLOCK
    return EMEI_BODY_BLOCK;
#else
    return Body::kMinWavesPerEU == 1 ? kWave : kBlock;
#endif
}

template <class Body, bool RK4>
__global__ void __launch_bounds__(rollout_block<Body>()) __attribute__((amdgpu_waves_per_eu(Body::kMinWavesPerEU)))
    body_rollout_kernel(const BodyArgs<Body> a) {
    using R = typename Body::real;
    constexpr int NS = Body::NS, NO = Body::NO, NA = Body::NA;
    constexpr int kBlock = rollout_block<Body>();  // shadows the library-wide 256 inside this kernel
    constexpr int kWaves = kBlock / kWave;
    constexpr int kActVec = (kWave * NA + 3) / 4, kObsVec = (kWave * NO + 3) / 4;  // 16-byte vectors per wave block
    constexpr int kActIt = (kActVec + kWave - 1) / kWave, kObsIt = (kObsVec + kWave - 1) / kWave;
    __shared__ __attribute__((aligned(16))) float act_s[kWaves][kActIt * kWave * 4];
    __shared__ __attribute__((aligned(16))) float obs_s[kWaves][kObsIt * kWave * 4];
    __shared__ SinCosEntry trig_s[kTrigTableSize];
    stage_trig_table<kBlock>(trig_s, a.trig);  // every thread reaches the barrier: inactive lanes stay in the kernel
    TrigCtx trig;
    trig.tab = trig_s;
    __shared__ R scratch_s[(Body::kScratchPerLane > 0 ? Body::kScratchPerLane : 1) * (Body::kScratchPerLane > 0 ? kBlock : 1)];
    if constexpr (Body::kScratchPerLane > 0) trig.scratch = scratch_s;
    trig.scratch_stride = kBlock;
    trig.cap_hits = a.cap_hits;
    EMEI_PROFILE_BEGIN();
    EMEI_CLOCK_BEGIN();
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t n = a.n;
    const int64_t i0 = i - lane;                             // first env of this wave
    const int wave_envs = (int)min((int64_t)kWave, n - i0);  // ragged last wave
    const bool active = i < n;

    R s[NS];
    int32_t steps = 0;
    uint32_t episode = 0;
    if (active) {
#pragma unroll
        for (int k = 0; k < NS; ++k) s[k] = a.state[(int64_t)k * n + i];
        steps = a.steps[i];
        episode = a.episode[i];
    } else {
        // Padding lanes of a ragged last wave run the arithmetic of their wave (its branches are wave-uniform) on a state of their
        // own, and the cheetah's three-block lanes borrow a constraint slot from a wave-mate with at most one row block
        // (cheetah_model.h: `donor`).  From the zero state a padding cheetah stands on both feet — two blocks, no slot to lend —
        // and it fell differently in a fused rollout than in the same rollout cut into chunks (every launch parks it anew):
        // whether the one real env of an n = 1 engine found a donor, and so which of two solvers it ran, depended on the
        // chunking (tools/stress.py, round 4; the two agree to ~1e-13, not bit for bit).  Parked in the air a padding lane has
        // no row, ever.
        Body::park(s);
    }
    const bool auto_reset = (a.flags & EMEI_FLAG_AUTO_RESET) != 0;
    const bool obs_noise = a.noise.obs_on != 0;
    uint32_t done = 0;
    R spare[Body::kSpareReset ? NS : 1];
    bool have_spare = false;

    // this wave's action block of step t: wave_envs*NA contiguous floats starting at (t*n + i0)*NA
    float4 av[kActIt];
    auto fetch_actions = [&](int t) __attribute__((always_inline)) {
        const float* base = a.actions + ((int64_t)t * n + i0) * NA;
        const bool fast = wave_envs == kWave && (((uintptr_t)base) & 15u) == 0;
#pragma unroll
        for (int c = 0; c < kActIt; ++c) {
            const int vec = c * kWave + lane;
            av[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (fast) {
                if (vec < kActVec) av[c] = ((const float4*)base)[vec];
            } else {  // ragged / unaligned tail: scalar loads
                float* f = (float*)&av[c];
                for (int e = 0; e < 4; ++e)
                    if (vec * 4 + e < wave_envs * NA) f[e] = base[vec * 4 + e];
            }
        }
    };
    // a one-float action is already one coalesced dword per lane: no staging, fetched a step ahead
    float a_next = 0.f;
    if constexpr (NA == 1) {
        if (active) a_next = a.actions[i];
    } else {
        fetch_actions(0);
    }
    for (int t = 0; t < a.n_steps; ++t) {
        R ctrl[NA];
        if constexpr (NA == 1) {
            ctrl[0] = (R)a_next;
            if (active && t + 1 < a.n_steps) a_next = a.actions[(int64_t)(t + 1) * n + i];
        } else {
            // stage this step's actions through LDS, then prefetch the next step's block
#pragma unroll
            for (int c = 0; c < kActIt; ++c) ((float4*)act_s[wv])[c * kWave + lane] = av[c];
            wave_lds_fence();  // other lanes' vectors hold this lane's action
#pragma unroll
            for (int k = 0; k < NA; ++k) ctrl[k] = (R)act_s[wv][lane * NA + k];
            wave_lds_fence();  // the block is consumed before the next step overwrites it
            if (t + 1 < a.n_steps) fetch_actions(t + 1);
        }

        R pre[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) pre[k] = s[k];
        typename Body::Warm warm{};  // per env-step: a fused rollout and repeated emei_step calls iterate identically
        for (int k = 0; k < a.freq_rate; ++k) {  // mujoco_env.py:91-104
            body_substep<Body, RK4>(s, ctrl, a.m, a.semi != 0, trig, warm);
            if (obs_noise)
                gauss_state<R, NS, false>(s, a.seed ^ kObsNoiseKey, a.env_offset + (uint64_t)i, episode,
                                          ((uint32_t)steps * (uint32_t)a.freq_rate + (uint32_t)k) * (uint32_t)((NS + 3) / 4),
                                          a.noise.obs, a.noise.shared != 0);
        }
        EMEI_MARK(step_io);  // (the integrator's own arithmetic between two forward-dynamics evaluations is charged to nw_out)
        float o[NO];
        R rew;
        bool term;
        Body::outputs(s, pre, ctrl, a.m, a.freq_rate, o, rew, term, trig);
        ++steps;
        const bool trunc = (a.max_episode_steps > 0) & (steps >= a.max_episode_steps);
        done = active ? ((term ? EMEI_DONE_TERMINAL : 0u) | (trunc ? EMEI_DONE_TRUNCATED : 0u)) : 0u;

        if (a.obs_out) {  // lane-wise into LDS, linear out
            float* mine = &obs_s[wv][lane * NO];
#pragma unroll
            for (int k = 0; k < NO; ++k) mine[k] = o[k];
            wave_lds_fence();
            float* dst = a.obs_out + ((int64_t)t * n + i0) * NO;
            if (wave_envs == kWave && (((uintptr_t)dst) & 15u) == 0) {
#pragma unroll
                for (int c = 0; c < kObsIt; ++c) {
                    const int vec = c * kWave + lane;
                    if (vec < kObsVec) ((float4*)dst)[vec] = ((const float4*)obs_s[wv])[vec];
                }
            } else {
                for (int e = lane; e < wave_envs * NO; e += kWave) dst[e] = obs_s[wv][e];
            }
            wave_lds_fence();
        }
        if (active) {
            if (a.reward_out) a.reward_out[(int64_t)t * n + i] = (float)rew;
            if (a.done_out) a.done_out[(int64_t)t * n + i] = (uint8_t)done;
        }
        EMEI_MARK(step_reset);
        if (__builtin_expect(auto_reset && __ballot(done != 0) != 0ull, 0)) {
            if constexpr (Body::kSpareReset) {
                // spare initial state per lane, re-drawn for every lane that lacks one when a resetting lane
                // has none (see pendulum_kernels.h:maybe_reset): bodies whose episodes end at different times
                if (__ballot((done != 0) & !have_spare) != 0ull) {
                    if (!have_spare) {
                        body_init<Body>(spare, a.seed, a.env_offset + (uint64_t)i, episode + 1u, a.noise);
                        have_spare = true;
                    }
                }
                if (done != 0) {
                    ++episode;
                    steps = 0;
#pragma unroll
                    for (int k = 0; k < NS; ++k) s[k] = spare[k];
                    have_spare = false;
                }
            } else if (done != 0) {  // episodes only end by TimeLimit, for all lanes at once
                ++episode;
                steps = 0;
                body_init<Body>(s, a.seed, a.env_offset + (uint64_t)i, episode, a.noise);
            }
        }
    }
    if (active) {
#pragma unroll
        for (int k = 0; k < NS; ++k) a.state[(int64_t)k * n + i] = s[k];
        a.steps[i] = steps;
        a.episode[i] = episode;
    }
    unsigned long long mk = __ballot(done != 0);
    if (lane == 0 && active) a.done_mask[i / kWave] = mk;
    EMEI_CLOCK_END();
    EMEI_PROFILE_END();
}

template <class Body>
__global__ void __launch_bounds__(kBlock)
    body_reset_kernel(typename Body::real* state, int32_t* steps, uint32_t* episode, int64_t n, uint64_t seed,
                      uint64_t env_offset, NoiseArgs<Body::NS> noise) {
    using R = typename Body::real;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    R s[Body::NS];
    body_init<Body>(s, seed, env_offset + (uint64_t)i, 0u, noise);
#pragma unroll
    for (int k = 0; k < Body::NS; ++k) state[(int64_t)k * n + i] = s[k];
    steps[i] = 0;
    episode[i] = 0;
}

// current_obs as float64 [n, NO]
template <class Body>
__global__ void __launch_bounds__(kBlock)
    body_get_obs_kernel(const typename Body::real* state, double* obs, int64_t n, typename Body::Model m) {
    using R = typename Body::real;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    R s[Body::NS];
    double o[Body::NO];
#pragma unroll
    for (int k = 0; k < Body::NS; ++k) s[k] = state[(int64_t)k * n + i];
    Body::obs_of(s, o, m);
#pragma unroll
    for (int k = 0; k < Body::NO; ++k) obs[i * Body::NO + k] = o[k];
}

// initial observation of (env, episode) pairs under the device reset generator
template <class Body>
__global__ void __launch_bounds__(kBlock)
    body_init_obs_kernel(const int64_t* env_index, const uint32_t* episode, float* obs, int64_t count, uint64_t seed,
                         uint64_t env_offset, NoiseArgs<Body::NS> noise, typename Body::Model m) {
    using R = typename Body::real;
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= count) return;
    R s[Body::NS];
    double o[Body::NO];
    body_init<Body>(s, seed, env_offset + (uint64_t)env_index[k], episode[k], noise);
    Body::obs_of(s, o, m);
#pragma unroll
    for (int j = 0; j < Body::NO; ++j) obs[k * Body::NO + j] = (float)o[j];
}

// get_batch_reward / get_batch_terminal on rows of the caller's dtype T (float or double; float64 rows are not narrowed:
// the reference evaluates these on float64 arrays, half_cheetah.py:59-67, hopper.py:95-106)
template <class Body, typename T>
__global__ void __launch_bounds__(kBlock)
    body_reward_kernel(const T* obs, const T* pre_obs, const T* action, T* reward, int64_t n, int freq_rate,
                       typename Body::Model m, const double* batch_cost) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double r = Body::batch_reward(obs + i * Body::NO, pre_obs ? pre_obs + i * Body::NO : nullptr,
                                  action ? action + i * Body::NA : nullptr, m, freq_rate);
    if constexpr (Body::kHasCtrlCost) {
        // EMEI_REWARD_BATCH_CTRL_COST (half_cheetah.py:61): replace this row's control cost by the whole batch's
        if (batch_cost) r += m.w_ctrl * (Body::ctrl_cost(action + i * Body::NA) - *batch_cost);
    }
    reward[i] = (T)r;
}
// sum over the WHOLE batch of action^2 in one workgroup, in a fixed order (deterministic): np.sum(np.square(action)) of
// half_cheetah.py:61
template <typename T>
__global__ void __launch_bounds__(1024) batch_sumsq_kernel(const T* x, int64_t count, double* out) {
    __shared__ double part[1024];
    double acc = 0.0;
    for (int64_t k = threadIdx.x; k < count; k += 1024) acc += (double)x[k] * (double)x[k];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = part[0];
}
template <class Body, typename T>
__global__ void __launch_bounds__(kBlock)
    body_terminal_kernel(const T* obs, uint8_t* terminal, int64_t n, typename Body::Model m) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    terminal[i] = (uint8_t)Body::batch_terminal(obs + i * Body::NO, m);
}

// EmeiEnv.get_batch_next_obs (core.py:190-193; abstract in the reference): one env-step from caller-supplied
// observations of dtype T, for bodies whose observation determines the state (Body::kObsIsState)
template <class Body, bool RK4, typename T>
__global__ void __launch_bounds__(kBlock)
    body_next_obs_kernel(const T* obs, const float* actions, T* next_obs, int64_t n, int freq_rate, int semi,
                         typename Body::Model m, const SinCosEntry* trig_tab) {
    using R = typename Body::real;
    constexpr int NS = Body::NS, NO = Body::NO, NA = Body::NA;
    static_assert(NS == NO, "observation and state must have the same layout");
    __shared__ SinCosEntry trig_s[kTrigTableSize];
    stage_trig_table(trig_s, trig_tab);
    TrigCtx trig;
    trig.tab = trig_s;
    __shared__ R scratch_s[(Body::kScratchPerLane > 0 ? Body::kScratchPerLane : 1) * (Body::kScratchPerLane > 0 ? kBlock : 1)];
    if constexpr (Body::kScratchPerLane > 0) trig.scratch = scratch_s;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    R s[NS], pre[NS], ctrl[NA], rew;
#pragma unroll
    for (int k = 0; k < NS; ++k) s[k] = pre[k] = (R)obs[i * NO + k];
#pragma unroll
    for (int k = 0; k < NA; ++k) ctrl[k] = (R)actions[i * NA + k];
    typename Body::Warm warm{};
    for (int k = 0; k < freq_rate; ++k) body_substep<Body, RK4>(s, ctrl, m, semi != 0, trig, warm);
    if constexpr (sizeof(T) == 8) {  // float64 rows: the observation of the float64 state, unrounded
        double o[NO];
        Body::obs_of(s, o, m);
#pragma unroll
        for (int k = 0; k < NO; ++k) next_obs[i * NO + k] = o[k];
    } else {
        float o[NO];
        bool term;
        Body::outputs(s, pre, ctrl, m, freq_rate, o, rew, term, trig);
#pragma unroll
        for (int k = 0; k < NO; ++k) next_obs[i * NO + k] = o[k];
    }
}

// every launch of one Body type (one translation unit instantiates exactly one Body: body_tu.hip)
template <class Body>
static int launch_body(const BodyLaunch& L) {
    using R = typename Body::real;
    const typename Body::Model m = Body::make_model(L.dt, L.env_params);
    dim3 grid((unsigned)((L.n + kBlock - 1) / kBlock));
    dim3 rgrid((unsigned)((L.n + rollout_block<Body>() - 1) / rollout_block<Body>()));  // the rollout kernel's own block size
    switch (L.op) {
        case BODY_OP_ROLLOUT: {
            BodyArgs<Body> a;
            a.state = (R*)L.state, a.steps = L.steps, a.episode = L.episode, a.done_mask = L.done_mask;
            a.actions = (const float*)L.actions, a.obs_out = L.obs_out, a.reward_out = L.reward_out, a.done_out = L.done_out;
            a.n = L.n, a.n_steps = L.n_steps, a.freq_rate = L.freq_rate, a.max_episode_steps = L.max_episode_steps;
            a.flags = L.flags, a.seed = L.seed, a.env_offset = L.env_offset, a.m = m;
            a.semi = L.integrator == EMEI_INTEG_SEMI_IMPLICIT, a.noise = NoiseArgs<Body::NS>(L.noise);
            a.trig = (const SinCosEntry*)L.trig;
            a.cap_hits = L.cap_hits;
            if (L.integrator == EMEI_INTEG_RK4)
                hipLaunchKernelGGL((body_rollout_kernel<Body, true>), rgrid, dim3(rollout_block<Body>()), 0, L.stream, a);
            else
                hipLaunchKernelGGL((body_rollout_kernel<Body, false>), rgrid, dim3(rollout_block<Body>()), 0, L.stream, a);
            if (L.selected) *L.selected = L.integrator == EMEI_INTEG_RK4 ? EMEI_KERNEL_BODY_RK4 : EMEI_KERNEL_BODY;
            break;
        }
        case BODY_OP_RESET:
            hipLaunchKernelGGL(body_reset_kernel<Body>, grid, dim3(kBlock), 0, L.stream, (R*)L.state, L.steps, L.episode,
                               L.n, L.seed, L.env_offset, NoiseArgs<Body::NS>(L.noise));
            break;
        case BODY_OP_GET_OBS:
            hipLaunchKernelGGL(body_get_obs_kernel<Body>, grid, dim3(kBlock), 0, L.stream, (const R*)L.state, L.obs_f64,
                               L.n, m);
            break;
        case BODY_OP_INIT_OBS:
            hipLaunchKernelGGL(body_init_obs_kernel<Body>, grid, dim3(kBlock), 0, L.stream, L.env_index, L.episode_in,
                               L.obs_out, L.n, L.seed, L.env_offset, NoiseArgs<Body::NS>(L.noise), m);
            break;
        case BODY_OP_REWARD: {
            const double* bc = nullptr;
            if (L.batch_cost_scratch) {  // EMEI_REWARD_BATCH_CTRL_COST: whole-batch sum of action^2 first (stream-ordered)
                if constexpr (!Body::kHasCtrlCost) return EMEI_ERR_UNSUPPORTED;
                bc = L.batch_cost_scratch;
                if (L.io_f64)
                    hipLaunchKernelGGL(batch_sumsq_kernel<double>, dim3(1), dim3(1024), 0, L.stream, (const double*)L.actions,
                                       L.n * Body::NA, L.batch_cost_scratch);
                else
                    hipLaunchKernelGGL(batch_sumsq_kernel<float>, dim3(1), dim3(1024), 0, L.stream, (const float*)L.actions,
                                       L.n * Body::NA, L.batch_cost_scratch);
            }
            if (L.io_f64)
                hipLaunchKernelGGL((body_reward_kernel<Body, double>), grid, dim3(kBlock), 0, L.stream, (const double*)L.obs_in,
                                   (const double*)L.pre_obs_in, (const double*)L.actions, (double*)L.reward_out, L.n, L.freq_rate, m, bc);
            else
                hipLaunchKernelGGL((body_reward_kernel<Body, float>), grid, dim3(kBlock), 0, L.stream, (const float*)L.obs_in,
                                   (const float*)L.pre_obs_in, (const float*)L.actions, (float*)L.reward_out, L.n, L.freq_rate, m, bc);
            break;
        }
        case BODY_OP_TERMINAL:
            if (L.io_f64)
                hipLaunchKernelGGL((body_terminal_kernel<Body, double>), grid, dim3(kBlock), 0, L.stream, (const double*)L.obs_in,
                                   L.done_out, L.n, m);
            else
                hipLaunchKernelGGL((body_terminal_kernel<Body, float>), grid, dim3(kBlock), 0, L.stream, (const float*)L.obs_in,
                                   L.done_out, L.n, m);
            break;
        case BODY_OP_NEXT_OBS:
            if constexpr (Body::kObsIsState) {
                const int semi = (int)(L.integrator == EMEI_INTEG_SEMI_IMPLICIT);
                const SinCosEntry* tt = (const SinCosEntry*)L.trig;
                const float* act = (const float*)L.actions;
                if (L.integrator == EMEI_INTEG_RK4) {
                    if (L.io_f64)
                        hipLaunchKernelGGL((body_next_obs_kernel<Body, true, double>), grid, dim3(kBlock), 0, L.stream,
                                           (const double*)L.obs_in, act, (double*)L.obs_out, L.n, L.freq_rate, 0, m, tt);
                    else
                        hipLaunchKernelGGL((body_next_obs_kernel<Body, true, float>), grid, dim3(kBlock), 0, L.stream,
                                           (const float*)L.obs_in, act, (float*)L.obs_out, L.n, L.freq_rate, 0, m, tt);
                } else {
                    if (L.io_f64)
                        hipLaunchKernelGGL((body_next_obs_kernel<Body, false, double>), grid, dim3(kBlock), 0, L.stream,
                                           (const double*)L.obs_in, act, (double*)L.obs_out, L.n, L.freq_rate, semi, m, tt);
                    else
                        hipLaunchKernelGGL((body_next_obs_kernel<Body, false, float>), grid, dim3(kBlock), 0, L.stream,
                                           (const float*)L.obs_in, act, (float*)L.obs_out, L.n, L.freq_rate, semi, m, tt);
                }
                break;
            } else {
                return EMEI_ERR_UNSUPPORTED;  // e.g. the double pendulum's observation "wrap" is not invertible
            }
        default: return EMEI_ERR_INVALID;
    }
    return hipGetLastError() == hipSuccess ? EMEI_OK : EMEI_ERR_HIP;
}

}  // namespace emei
