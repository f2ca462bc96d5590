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
//   kStreamOutputs                                  // per-step outputs leave with the non-temporal hint (emei_device.h:store_body_out)
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

enum BodyOp { BODY_OP_ROLLOUT = 0, BODY_OP_RESET, BODY_OP_GET_OBS, BODY_OP_INIT_OBS, BODY_OP_REWARD, BODY_OP_TERMINAL, BODY_OP_NEXT_OBS,
              BODY_OP_OCCUPANCY /* host only: *selected = waves of the rollout kernel (of L.integrator) the current device holds at once */ };

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
    // chunked rollout (emei_config.rollout_chunk_steps): the handle's work-queue words (see WorkQueue) and its policy
    uint32_t* work = nullptr;
    int32_t chunk_steps = 0;   // 0 = automatic, -1 = off, k > 0 = k steps per work item
    int32_t resident_waves = 0;  // waves of the rollout kernel the device holds at once (automatic policy); 0 = unknown
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
    uint32_t* work;        // chunked rollout: WorkQueue words of the handle (null = one-piece launch, block b = env-wave b)
    int32_t chunk_steps;   // > 0: steps per work item; < 0: the guided schedule (WorkQueue::item_steps)
    uint32_t n_waves;      // env-waves of the shard = work items per chunk
    uint32_t n_items;      // n_waves x chunks: tickets at or beyond it end a worker
    typename Body::Model m;
};

// Chunked rollout of the bodies that run ONE wave per SIMD (one-wave blocks).  Why: at 131 072 envs a launch is two rounds of 1024
// waves whose durations differ by their lanes' contact histories; a SIMD that finishes its two waves early idles until the
// slowest pair of the launch ends (config 4: 10-14 % of the SIMD-time, profiles/r05_cheetah_tail.txt).  The launch is cut
// into items of (one env-wave) x (chunk_steps steps) and launched as PERSISTENT one-wave workers, as many as the device
// holds at once.  A worker draws a TICKET (atomic counter), works on item `ticket`, chunk-major:
//     chunk = ticket / n_waves,  env-wave = ticket % n_waves,
// and draws again until the tickets run out.  (First built as one block per item, refilled by the hardware dispatcher: the
// hand-over from a finished block to the next cost the cheetah ~6 us of idle SIMD per item, profiles/r05_cheetah_tail.txt.)
// Item (c, w) needs the state item (c - 1, w) leaves in the handle's arrays: it waits until progress[w] >= c, which (c - 1, w)
// publishes — release at agent scope, the state stores of all 64 lanes ordered before it — when it is done.  No deadlock:
// tickets are handed out in START order, so the item a block waits for has a smaller ticket, has therefore started, and itself
// only ever waits for still smaller tickets (a worker holds ONE ticket at a time: drawing the next one early would let a
// drawn-but-unstarted item block others).  The wait is bounded all the same (2 s of the 100 MHz counter, then the item is
// skipped, counted in `faults`, and its progress word published so that its successors do not wait in turn), and a worker
// ends at the first ticket beyond the last item: every wave reaches an exit.  A worker that finishes an item early takes the
// next ticket: that is the dynamic balancing; the state round trip between items is 2 x 18 doubles per env per chunk against
// 101 B x chunk_steps of outputs.
// Results are bit-identical to the one-piece launch: an item runs the same per-step code from the same state
// (tests/test_gpu_shapes.py: a fused rollout, its chunks and its single steps agree bit for bit).
constexpr int kGuidedShift = 1, kGuidedMin = 3;  // the automatic schedule: half of what remains, at least 3 steps (the sweep in profiles/r05_cheetah_tail.txt)
struct WorkQueue {
    // word offsets; [kProgress + w] = chunks of env-wave w that are done.  The fault count is sticky, the words behind it are
    // zeroed (one memset) in front of every chunked launch
    enum { kFaults = 0, kTicket = 1, kProgress = 2 };
    static constexpr unsigned long long kWaitTicks = 200000000ull;  // 2 s of s_memrealtime (100 MHz)
    // Step range [t_begin, t_end) of chunk `chunk`.  chunk_steps > 0: fixed length.  chunk_steps = -((m << 4) | g): the GUIDED
    // schedule — every chunk takes 1 / 2^g of the steps that remain, at least m: long items first, where a worker's finishing time
    // does not matter and the per-item cost (ticket, state round trip, release: ~3 us) is paid rarely, short items last, where
    // the launch ends with the slowest worker's LAST item (100 steps, g = 1, m = 3: 50 25 13 6 3 3 = 6 items per wave instead of
    // the 20 of a fixed length 5).  Items of one or two steps at the very end cost more than they save: a slow wave's step takes
    // longer than a row of such items takes to hand out, so its items queue up behind each other (waiting workers).  Host and
    // device run the same recurrence.
    __host__ __device__ static void item_steps(int n_steps, int chunk_steps, uint32_t chunk, int& t_begin, int& t_end) {
        if (chunk_steps > 0) {
            t_begin = (int)chunk * chunk_steps;
            t_end = t_begin + chunk_steps < n_steps ? t_begin + chunk_steps : n_steps;
            return;
        }
        const int g = (-chunk_steps) & 15, min_len = (-chunk_steps) >> 4;  // encoded by launch_body: (shortest item << 4) | g
        int t = 0, len = 0;
        for (uint32_t c = 0;; ++c) {
            len = (n_steps - t + (1 << g) - 1) >> g;
            len = len < min_len ? min_len : len;
            if (c == chunk) break;
            t += len;
        }
        t_begin = t, t_end = t + len < n_steps ? t + len : n_steps;
    }
    __host__ static unsigned count_chunks(int n_steps, int chunk_steps) {
        if (chunk_steps > 0) return (unsigned)((n_steps + chunk_steps - 1) / chunk_steps);
        unsigned c = 0;
        for (int b = 0, e = 0; e < n_steps; ++c) item_steps(n_steps, chunk_steps, c, b, e);
        return c;
    }
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

#ifndef EMEI_WORKQUEUE
#define EMEI_WORKQUEUE 1  // -DEMEI_WORKQUEUE=0: a variant build without the chunked-launch code, for A/B runs (tools/ab.sh)
#endif
// QUEUE: the persistent-worker form of a chunked launch (WorkQueue), its own instantiation so that the one-piece kernel is
// instruction for instruction what it was without it (the item loop around the rollout costs the cheetah 1.7 %: scalar
// registers held across it are spilled to VGPR lanes)
template <class Body, bool RK4, bool QUEUE = false>
__global__ void __launch_bounds__(rollout_block<Body>()) __attribute__((amdgpu_waves_per_eu(Body::kMinWavesPerEU)))
    body_rollout_kernel(const BodyArgs<Body> a) {
    static_assert(!QUEUE || rollout_block<Body>() == kWave, "work items are one-wave blocks");
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
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    constexpr bool kQueue = QUEUE;
    // One pass of this loop = one (env block, step range): the whole horizon of block blockIdx.x, then out — or (QUEUE) one
    // drawn item after the other until the tickets run out (WorkQueue: the block is a persistent worker).
    uint32_t ticket = 0;
    [[maybe_unused]] bool first_item = true;  // probe builds only
    if constexpr (kQueue) {
        if (lane == 0) ticket = __hip_atomic_fetch_add(a.work + WorkQueue::kTicket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma nounroll
    for (;;) {
    EMEI_CLOCK_BEGIN();
    uint32_t blk = blockIdx.x, chunk = 0;
    int t_begin = 0, t_end = a.n_steps;
    if constexpr (kQueue) {
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket >= a.n_items) {  // the exit every worker reaches: tickets only grow
            EMEI_CLOCK_WORKER_EXIT();
            break;
        }
        if (first_item) EMEI_CLOCK_FIRST_ITEM();
        first_item = false;
        chunk = ticket / a.n_waves, blk = ticket - chunk * a.n_waves;
        WorkQueue::item_steps(a.n_steps, a.chunk_steps, chunk, t_begin, t_end);
        bool gave_up = false;
        if (chunk > 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            // the acquire (agent scope: the predecessor may have run on another XCD, behind another L2) orders every
            // load below after the predecessor's state stores
            while (__hip_atomic_load(a.work + WorkQueue::kProgress + blk, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < chunk) {
                __builtin_amdgcn_s_sleep(16);
                if (__builtin_amdgcn_s_memrealtime() - t0 > WorkQueue::kWaitTicks) {
                    gave_up = true;
                    break;
                }
            }
            EMEI_CLOCK_WAITED(__builtin_amdgcn_s_memrealtime() - t0);
        }
        if (__builtin_expect(gave_up, 0)) {  // wave-uniform
            if (lane == 0) {
                __hip_atomic_fetch_add(a.work + WorkQueue::kFaults, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.work + WorkQueue::kProgress + blk, chunk + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                ticket = __hip_atomic_fetch_add(a.work + WorkQueue::kTicket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            continue;
        }
    }
    const int64_t i = (int64_t)blk * kBlock + threadIdx.x;
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
        if (active) a_next = a.actions[(int64_t)t_begin * n + i];
    } else {
        fetch_actions(t_begin);
    }
    for (int t = t_begin; t < t_end; ++t) {
        R ctrl[NA];
        if constexpr (NA == 1) {
            ctrl[0] = (R)a_next;
            if (active && t + 1 < t_end) a_next = a.actions[(int64_t)(t + 1) * n + i];
        } else {
            // stage this step's actions through LDS, then prefetch the next step's block
#pragma unroll
            for (int c = 0; c < kActIt; ++c) ((float4*)act_s[wv])[c * kWave + lane] = av[c];
            wave_lds_fence();  // other lanes' vectors hold this lane's action
#pragma unroll
            for (int k = 0; k < NA; ++k) ctrl[k] = (R)act_s[wv][lane * NA + k];
            wave_lds_fence();  // the block is consumed before the next step overwrites it
            if (t + 1 < t_end) fetch_actions(t + 1);
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
                    if (vec < kObsVec) store_body_out<Body::kStreamOutputs>((float4*)dst + vec, ((const float4*)obs_s[wv])[vec]);
                }
            } else {
                for (int e = lane; e < wave_envs * NO; e += kWave) dst[e] = obs_s[wv][e];
            }
            wave_lds_fence();
        }
        if (active) {
            if (a.reward_out) store_body_out<Body::kStreamOutputs>(a.reward_out + (int64_t)t * n + i, (float)rew);
            if (a.done_out) store_body_out<Body::kStreamOutputs>(a.done_out + (int64_t)t * n + i, (uint8_t)done);
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
    if constexpr (kQueue) {
        // publish this item: every lane's state stores, then the progress word (WorkQueue).  The next ticket is drawn BEFORE the
        // release fence waits for those stores (its round trip to the L2 overlaps theirs): still one unfinished item per
        // worker — this one, which nothing can hold up any more.
        if (lane == 0) ticket = __hip_atomic_fetch_add(a.work + WorkQueue::kTicket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) __hip_atomic_store(a.work + WorkQueue::kProgress + blk, chunk + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#if defined(EMEI_CLOCK_PROBE) && defined(EMEI_CLOCK_HIST_CORR)
    // probe builds: how well does the time per step of a wave's previous item predict that of this one?  (slots 0-5: n, sum x, sum y,
    // sum xy, sum xx, sum yy of ticks per step; x = previous item of the same env-wave, y = this one; tools/tail_probe.py --corr)
    if constexpr (kQueue) {
        const unsigned long long d = (__builtin_amdgcn_s_memrealtime() - clock_probe_.r0) / (unsigned long long)(t_end - t_begin);
        uint32_t* const slot = a.work + WorkQueue::kProgress + a.n_waves + blk;
        const unsigned long long prev = *slot;
        if (lane == 0) {
            if (chunk > 0 && prev > 0) {
                atomicAdd(&g_debug_stats[0], 1ull), atomicAdd(&g_debug_stats[1], prev), atomicAdd(&g_debug_stats[2], d);
                atomicAdd(&g_debug_stats[3], prev * d), atomicAdd(&g_debug_stats[4], prev * prev), atomicAdd(&g_debug_stats[5], d * d);
            }
            *slot = (uint32_t)d;
        }
    }
#endif
    EMEI_CLOCK_END();
    if constexpr (!kQueue) break;
    }  // item loop
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
            a.work = nullptr, a.chunk_steps = 0, a.n_waves = rgrid.x, a.n_items = 0;
            if constexpr (rollout_block<Body>() == kWave && EMEI_WORKQUEUE != 0) {
                // chunked launch (WorkQueue).  emei_config.rollout_chunk_steps: k > 0 fixed length; -1 off; -(100 m + g) the guided
                // schedule with 1 / 2^g of the remaining steps per chunk, at least m; 0 automatic: only when the shard has more
                // waves than the device holds at once (otherwise every wave is resident from the start and there is nothing to
                // balance), guided with kGuidedShift / kGuidedMin
                int cs = L.chunk_steps;
                // (device encoding of a guided schedule: -((shortest item << 4) | g))
                if (cs == 0) cs = (L.resident_waves > 0 && (int64_t)rgrid.x > L.resident_waves) ? -((kGuidedMin << 4) | kGuidedShift) : 0;
                else if (cs <= -100) cs = -(((-cs / 100) << 4) | max(1, min(-cs % 100, 6)));  // -(100 m + g): shortest item m, 1 / 2^g
                else if (cs < 0) cs = 0;
                const unsigned n_chunks = cs != 0 ? WorkQueue::count_chunks(L.n_steps, cs) : 1u;
                if (L.work && n_chunks > 1 && (uint64_t)rgrid.x * n_chunks < (1ull << 31)) {
                    // ticket, progress words back to zero (the fault count is sticky), in stream order: capturable
                    static_assert(WorkQueue::kProgress == WorkQueue::kTicket + 1, "one memset");
                    if (hipMemsetAsync(L.work + WorkQueue::kTicket, 0, (size_t)(1 + rgrid.x) * sizeof(uint32_t), L.stream) != hipSuccess)
                        return EMEI_ERR_HIP;
                    a.work = L.work, a.chunk_steps = cs, a.n_items = rgrid.x * n_chunks;
                    // persistent workers: as many one-wave blocks as the device holds at once (any count is correct: a
                    // worker that starts late simply finds fewer tickets left)
                    const unsigned workers = L.resident_waves > 0 ? (unsigned)L.resident_waves : 1024u;
                    rgrid.x = min(workers, a.n_items);
                    if (L.integrator == EMEI_INTEG_RK4)
                        hipLaunchKernelGGL((body_rollout_kernel<Body, true, true>), rgrid, dim3(kWave), 0, L.stream, a);
                    else
                        hipLaunchKernelGGL((body_rollout_kernel<Body, false, true>), rgrid, dim3(kWave), 0, L.stream, a);
                }
            }
            if (!a.work) {
                if (L.integrator == EMEI_INTEG_RK4)
                    hipLaunchKernelGGL((body_rollout_kernel<Body, true>), rgrid, dim3(rollout_block<Body>()), 0, L.stream, a);
                else
                    hipLaunchKernelGGL((body_rollout_kernel<Body, false>), rgrid, dim3(rollout_block<Body>()), 0, L.stream, a);
            }
            if (L.selected)
                *L.selected = a.work ? (L.integrator == EMEI_INTEG_RK4 ? EMEI_KERNEL_BODY_RK4_CHUNKED : EMEI_KERNEL_BODY_CHUNKED)
                                     : (L.integrator == EMEI_INTEG_RK4 ? EMEI_KERNEL_BODY_RK4 : EMEI_KERNEL_BODY);
            break;
        }
        case BODY_OP_OCCUPANCY: {
            int dev = 0, cus = 0, nb = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
                return EMEI_ERR_HIP;
            hipError_t e;
            if constexpr (rollout_block<Body>() == kWave && EMEI_WORKQUEUE != 0) {  // the persistent workers' own instantiation
                e = L.integrator == EMEI_INTEG_RK4
                        ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, body_rollout_kernel<Body, true, true>, rollout_block<Body>(), 0)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, body_rollout_kernel<Body, false, true>, rollout_block<Body>(), 0);
            } else {
                e = L.integrator == EMEI_INTEG_RK4
                        ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, body_rollout_kernel<Body, true>, rollout_block<Body>(), 0)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, body_rollout_kernel<Body, false>, rollout_block<Body>(), 0);
            }
            if (e != hipSuccess) return EMEI_ERR_HIP;
            if (L.selected) *L.selected = nb * cus * (rollout_block<Body>() / kWave);
            return EMEI_OK;
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
