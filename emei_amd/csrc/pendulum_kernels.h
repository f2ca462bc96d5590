// pendulum_kernels.h — step / rollout kernels of the 4-state cart/pole family for gfx950.
//
// Layout: one thread per env instance; state is struct-of-arrays in HBM (4 arrays of n Reals + a
// step counter and an episode counter per env), so every state access of a wave is one fully
// coalesced 256 B (float) / 512 B (double) line group.  Observations leave as one float4 per lane
// (row-major [n,4] float32 = a 1 KiB contiguous store per wave).  The rollout keeps the state,
// the carried sin/cos and the counters in registers for all n_steps; per step it reads one action
// and writes obs + reward + done (22 B/env-step with uint8 actions).
// There is no dense contraction anywhere on this path, hence no MFMA: the roofline is HBM.
#pragma once
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
    const SinCosEntry* trig;  // 256-entry {sin,cos} table in device memory (abi.hip: emei_trig_table)
    float4* obs_out;
    float* reward_out;
    uint8_t* done_out;
    int64_t n;
    int32_t n_steps, freq_rate, action_dtype, max_episode_steps;
    uint32_t flags;
    uint64_t seed, env_offset;
    typename Env::Params p;
    // the generic kernel only (emei_step_host; appended so that no other field moves): see launch.h
    double* obs_f64;
    uint32_t* host_flag;
    uint32_t flag_value;
    uint32_t xcd_contiguous;  // staged kernel: see the env_block map (set by launch_rollout_full when gridDim.x % 8 == 0)
    uint32_t first_block;  // staged kernel: a launch may cover the blocks [first_block, first_block + gridDim.x) of the shard (launch_rollout_full)
};

// emei_set_obs_peers (multi-GPU observation return by peer writes): where the staged rollout kernel ALSO stores every step's
// observation — up to EMEI_MAX_OBS_PEERS gathered buffers [n_steps, row_envs, 4] float32 (hipIpc-mapped memory of the other ranks, and
// this rank's own), this shard's envs at columns [col, col + n).  A kernel argument of its own, so that the kernel without peers keeps
// its arguments (and its code) unchanged.
struct PeerObs {
    float4* obs[EMEI_MAX_OBS_PEERS];
    int64_t row_envs, col;
    uint32_t count;
};

// emei_step (n_steps = 1) and emei_rollout (n_steps = T): base_control.py:61-83 /
// mujoco_env.py:157-167 for every env of the shard, T times, without leaving the registers.
// FULL = all three outputs are non-null: the stores are then unconditional, which lets the compiler
// count them and wait for a chunk's action loads with vmcnt(3*kChunk) instead of vmcnt(0).
template <class Env, typename ActT, bool FULL>
__global__ void __launch_bounds__(kBlock) pend_rollout_kernel(const RolloutArgs<Env> a) {
    using R = typename Env::real;
    __shared__ SinCosEntry trig_s[kTrigTableSize];
    stage_trig_table(trig_s, a.trig, Env::trig_rot_c(), Env::trig_rot_s());
    const ActT* __restrict__ actions = (const ActT*)a.actions;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n) return;
    const int64_t n = a.n;
    const uint32_t li = (uint32_t)i;  // n_envs < 2^31 (checked in emei_create)

    R s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = a.state[k * n + i];
    int32_t steps = a.steps[i];
    uint32_t episode = a.episode[i];
    typename Env::Carry c;
    trig_ctx_init(c.trig, trig_s);
    Env::prime(s, c, a.p);

    const bool auto_reset = (a.flags & EMEI_FLAG_AUTO_RESET) != 0;
    // Actions are fetched a whole chunk of kChunk steps ahead of their use.  On gfx950 loads and
    // stores share one in-order counter (vmcnt), so waiting for a load also waits for every OLDER
    // store: with the loads issued one chunk early, the only stores older than them were issued
    // >= kChunk steps ago and have long retired, and the 3*kChunk stores of the current chunk stay in
    // flight behind the wait.  (Fetching one step ahead put the HBM write latency of the previous
    // step's stores on the serial per-env chain: 2.3x slower.)
    // The action dtype is a template parameter and out-of-range steps are clamped rather than
    // branched around, so the kChunk loads are one straight-line burst and the compiler can count
    // them (a dtype switch around each load made it fall back to vmcnt(0) after every load).
    constexpr int kChunk = 8;
    const int last = a.n_steps - 1;
    ActT cur[kChunk], nxt[kChunk];
#pragma unroll
    for (int j = 0; j < kChunk; ++j) cur[j] = (actions + (int64_t)min(j, last) * n)[li];
    uint32_t done = 0;
    auto do_step = [&](int t, ActT raw) __attribute__((always_inline)) {
        R o[4], rew;
        bool term;
        Env::step(s, c, Env::decode_t(raw), a.p, a.freq_rate, o, rew, term);
        ++steps;
        bool trunc = (a.max_episode_steps > 0) & (steps >= a.max_episode_steps);
        done = (term ? EMEI_DONE_TERMINAL : 0u) | (trunc ? EMEI_DONE_TRUNCATED : 0u);

        // uniform (scalar) row base + 32-bit lane index: the address costs no vector instruction
        const int64_t row = (int64_t)t * n;
        if (FULL || a.obs_out) (a.obs_out + row)[li] = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
        if (FULL || a.reward_out) (a.reward_out + row)[li] = (float)rew;
        if (FULL || a.done_out) (a.done_out + row)[li] = (uint8_t)done;

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
    };
    int t0 = 0;
    // full chunks: no per-step bounds check, so every path through the body issues exactly
    // 3*kChunk stores after the prefetch and the wait for it can leave them all in flight
    for (; t0 + kChunk <= a.n_steps; t0 += kChunk) {
#pragma unroll
        for (int j = 0; j < kChunk; ++j) nxt[j] = (actions + (int64_t)min(t0 + kChunk + j, last) * n)[li];
#pragma unroll
        for (int j = 0; j < kChunk; ++j) do_step(t0 + j, cur[j]);
#pragma unroll
        for (int j = 0; j < kChunk; ++j) cur[j] = nxt[j];
    }
    // tail (< kChunk steps); also the whole of emei_step (n_steps = 1)
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {
        if (t0 + j >= a.n_steps) break;
        do_step(t0 + j, cur[j]);
    }

#pragma unroll
    for (int k = 0; k < 4; ++k) a.state[k * n + i] = s[k];
    a.steps[i] = steps;
    a.episode[i] = episode;
    // done bits of the last step, one 64-bit word per wave, for emei_compact_done
    unsigned long long m = __ballot(done != 0);
    if ((threadIdx.x & (kWave - 1)) == 0) a.done_mask[i / kWave] = m;
    if (a.obs_f64) {  // emei_step_host: the state's observation in float64 (after an auto-reset: the new episode's), as emei_get_obs
        R o[4];
        Env::obs_of(s, o);
        double2* dst = (double2*)(a.obs_f64 + 4 * i);
        dst[0] = make_double2((double)o[0], (double)o[1]);
        dst[1] = make_double2((double)o[2], (double)o[3]);
    }
    if (a.host_flag) {  // n = 1 (checked on the host): this thread wrote every result; release them to the host, then the flag
        __threadfence_system();
        __hip_atomic_store(a.host_flag, a.flag_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------------------------------------
// Staged rollout: the fast path of emei_rollout (all outputs present, n a multiple of 64,
// 16-byte aligned buffers).  Same arithmetic and results as pend_rollout_kernel; what changes is
// how bytes move.  A wave owns 64 consecutive envs, i.e. a 64-column stripe of the [T, N] action,
// reward and done arrays, and stages kStage = 16 steps of that stripe through its private slice of
// LDS so that every global access is a full 16 B per lane:
//   actions : one global_load_dwordx4 per lane and sizeof(ActT) fetches a 16-step tile, LDS holds it
//             row-major, each step reads its own element back (ds_read, lgkm counter: the wait no
//             longer sits behind the stores, which share vmcnt with loads on gfx950)
//   reward  : ds_write_b32 per step; every 4 steps one ds_read_b128 + global_store_dwordx4 (4 rows)
//   done    : ds_write_b8 per step; every 16 steps one ds_read_b128 + global_store_dwordx4 (16 rows)
//   obs     : already 16 B per lane, stored directly
// => 1.31 vector stores + 1/16 loads per env-step instead of 3 stores + 1 load, and no 64 B
// partial-line byte stores.  LDS slices are wave-private (LDS operations of one wave execute in
// order), so there is no barrier anywhere (an optional one per tile, Env::kTileBarrier, is off: see its comment).
// steps per staged tile: 16 for one-byte actions (1 KiB of LDS per wave and buffer); 8 for 4- and 8-byte actions, so that
// the double-buffered tiles of a 256-thread block stay at 16 / 32 KiB and four blocks (with the other slices) fit a CU's
// 160 KiB: config 3 runs 4 waves per SIMD
#ifndef EMEI_TILE_BARRIER
#define EMEI_TILE_BARRIER 1  // with Env::kTileBarrier: the per-tile block barrier of the staged kernel (A/B history: see its comment)
#endif
#ifndef EMEI_PRIO_ROTATE
#define EMEI_PRIO_ROTATE 1  // 0: a variant build without the priority rotation of the staged kernel, for A/B runs
#endif
template <typename ActT>
constexpr int stage_steps() { return sizeof(ActT) == 1 ? 16 : 8; }

// The body of the two staged kernels below (PEERS: with the peer stores of emei_set_obs_peers).
template <class Env, typename ActT, bool FREQ1, bool PEERS>
__device__ __forceinline__ void pend_rollout_staged(const RolloutArgs<Env>& a, const PeerObs& peers) {
    using R = typename Env::real;
    constexpr int kWavesPerBlock = kBlock / kWave;
    constexpr int kStage = stage_steps<ActT>();
    constexpr int kTileWaitKeep = kStage;  // VMEM operations left in flight when a staged action tile is retired
    constexpr int kActVec = kStage * sizeof(ActT) / 16;  // 1 KiB wave-loads (16 B per lane) per kStage x 64 tile of ActT
    static_assert(kStage % 4 == 0 && kStage * kWave * sizeof(ActT) % 1024 == 0, "whole wave-loads, whole reward groups");
    __shared__ uint4 act_s[kWavesPerBlock][2][kStage * kWave * sizeof(ActT) / 16];
    __shared__ float rew_s[kWavesPerBlock][4][kWave];
    __shared__ uint8_t done_s[kWavesPerBlock][kStage][kWave];
    // spare initial states (see maybe_reset): in LDS for the envs that run several waves per SIMD (registers are their
    // occupancy limit), in registers otherwise
    constexpr bool kLdsSpare = Env::kSpareInLds;
    __shared__ R spare_s[kLdsSpare ? kWavesPerBlock : 1][6][kLdsSpare ? kWave : 1];

    __shared__ SinCosEntry trig_s[kTrigTableSize];
    stage_trig_table(trig_s, a.trig, Env::trig_rot_c(), Env::trig_rot_s());
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
#if defined(EMEI_BLOCK_SWIZZLE)  // experiment: which env block a workgroup (and so an XCD) works on
    const int64_t i = (int64_t)((blockIdx.x + a.first_block) ^ (unsigned)EMEI_BLOCK_SWIZZLE) * kBlock + threadIdx.x;
#else
    // Which env block a workgroup works on.  Workgroups go to the 8 XCDs round-robin (workgroup b -> XCD b mod 8).  With
    // a.xcd_contiguous (shards of more than one wave per SIMD: launch_rollout_full) XCD k gets ONE contiguous eighth of the launch's
    // envs, i.e. of every output row, instead of every eighth 4 KiB piece: 131 072 envs 0.676 -> 0.615 ms, 1 048 576 envs 5.61 ->
    // 5.01 ms on one box; nothing (or slightly worse) at 65 536 envs, where the identity stays (profiles/r05_split_launch.txt).
    // Results do not depend on it (envs are independent); the map is fixed, so an env's lines stay in one XCD's L2 across launches.
    const unsigned env_block = a.xcd_contiguous ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int64_t i = (int64_t)(env_block + a.first_block) * kBlock + threadIdx.x;  // n % 64 == 0: whole waves only
#endif
    if (i >= a.n) return;
#if defined(EMEI_XCD_ONLY)  // experiment (timing only, results of the skipped envs are not produced): only the workgroups of the XCDs in this bit mask work
    if (!(((unsigned)EMEI_XCD_ONLY >> (blockIdx.x & 7u)) & 1u)) return;
#endif
    EMEI_CLOCK_BEGIN();
    const int64_t n = a.n;
    const uint32_t li = (uint32_t)i, i0 = li - (uint32_t)lane;
    const ActT* __restrict__ actions = (const ActT*)a.actions;

    R s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = a.state[k * n + i];
    int32_t steps = a.steps[i];
    uint32_t episode = a.episode[i];
    typename Env::Carry c;
    trig_ctx_init(c.trig, trig_s);
    Env::prime(s, c, a.p);
    const bool auto_reset = (a.flags & EMEI_FLAG_AUTO_RESET) != 0;
    const int last = a.n_steps - 1;
    uint32_t done = 0;

    // tile geometry of one 16-byte vector v (0..kActVec-1) of this lane inside a 16-step tile
    constexpr int kLanesPerRow = kWave * sizeof(ActT) / 16;  // lanes that cover one 64-env row
    constexpr int kRowsPerVec = kWave / kLanesPerRow;        // rows one wave-wide vector load covers
    const int rsub = lane / kLanesPerRow, cbyte = (lane % kLanesPerRow) * 16;
    const char* act_lane_base = (const char*)(actions + i0) + cbyte;  // this lane's column inside a row
    const int64_t act_row_bytes = n * (int64_t)sizeof(ActT);
    // Action tiles go global -> LDS directly (LDS-DMA, global_load_lds_dwordx4): one wave-instruction
    // writes 1 KiB of LDS linearly (lane l -> base + 16 l), which is exactly the row-major tile image,
    // and no VGPR carries the tile across the 16 steps it is in flight (held in registers, hipcc
    // parked it in scratch and stalled on the load at once).  hipcc does not count asm memory
    // operations, so the wait that retires a tile is written out below (EMEI_TILE_WAIT).
    const uint32_t act_lds0 = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)&act_s[wv][0][0]);
    constexpr uint32_t kTileBytes = kStage * kWave * sizeof(ActT);
#define EMEI_LOAD_TILE(T0, BUF)                                                                           \
    _Pragma("unroll") for (int k_ = 0; k_ < kActVec; ++k_) {                                             \
        const int row_ = min((T0) + k_ * kRowsPerVec + rsub, last); /* never read past step T-1 */        \
        const char* src_ = act_lane_base + row_ * act_row_bytes;                                          \
        const uint32_t dst_ = act_lds0 + (uint32_t)(BUF) * kTileBytes + (uint32_t)k_ * 1024u;              \
        uint32_t keep_;                                                                                   \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(src_), "s"(dst_) : "memory");                                    \
    }

    // FREQ1: freq_rate == 1 is resolved at compile time (no substep loop, no loop branches: every
    // taken branch costs a lone wave an instruction-buffer refill)
    const int freq_rate = FREQ1 ? 1 : a.freq_rate;
    auto advance = [&](ActT raw, R (&o)[4], R& rew) __attribute__((always_inline)) {
        bool term;
        Env::step(s, c, Env::decode_t(raw), a.p, freq_rate, o, rew, term);
        ++steps;
        bool trunc = (a.max_episode_steps > 0) & (steps >= a.max_episode_steps);
        done = (term ? EMEI_DONE_TERMINAL : 0u) | (trunc ? EMEI_DONE_TRUNCATED : 0u);
    };
    // Early termination.  The reset of a lane needs a fresh initial state (Philox4x32-10 + the env's
    // init distribution + its sin/cos): ~150-300 vector instructions that the whole wave executes for
    // the one or two lanes that are done.  Each lane therefore keeps a SPARE initial state for its next
    // episode: a reset just copies it, and spares are re-drawn only when a resetting lane has none —
    // then for every lane without one at once (the first reset of a wave draws all 64).  The draw is a
    // pure function of (seed, global env, episode), so results do not depend on when it is computed.
    R sp[4] = {R(0), R(0), R(0), R(0)}, sp_sn = R(0), sp_cs = R(0);  // dead outside the redraw when the spare lives in LDS
    float sp_e0 = 0.f, sp_e1 = 0.f;  // the non-trigonometric part of the spare's carry (Env::save_extra / load_extra)
    const unsigned long long reset_mask = auto_reset ? ~0ull : 0ull;
    // which lanes hold a spare: a wave-uniform 64-bit mask in scalar registers (ballot results and scalar logic only), so
    // that the bookkeeping of a reset costs no vector instruction; `inverse_ballot` turns it back into a lane predicate.
    // Env::kSpareFlagInVgpr (InvPend: its float64 literals fill the scalar file, hipcc kept the mask in two lanes of a
    // spill register — 2 v_readlane + 2 v_writelane + 2 v_readlane per resetting step, and some lane resets in 95 % of
    // config 3's wave-steps): a per-lane flag instead, 3 (>= every done code) where the lane holds a spare, else 0 — the
    // "a resetting lane has no spare" test is then ONE comparison, done > flag, and taking the spare one v_mov
    constexpr bool kVFlag = Env::kSpareFlagInVgpr;
    unsigned long long spare_mask = 0ull;
    uint32_t spare_flag = 0u;
    auto maybe_reset = [&]() __attribute__((always_inline)) {
#if defined(EMEI_NO_RESET)  // experiment (timing only, wrong results after the first done): what the per-step reset check and its block boundary cost
        return;
#endif
        EMEI_STAT_WAVE(16);  // staged-kernel event counters of a -DEMEI_NEWTON_STATS build (tools/pend_stats.py): env-steps (waves)
        const unsigned long long done_mask = __ballot(done != 0) & reset_mask;
        // cold for most envs: laid out of line so that the usual case falls through (Env::kResetLikely: in line)
        if (__builtin_expect(done_mask != 0ull, Env::kResetLikely ? 1 : 0)) {  // scalar test: no vector instruction
            EMEI_STAT_WAVE(17);  // ... steps in which some lane resets
            const bool refill = kVFlag ? (__ballot(done > spare_flag) & reset_mask) != 0ull : (done_mask & ~spare_mask) != 0ull;
            if (__builtin_expect(refill, 0)) {  // a resetting lane has no spare: redraw for every lane without one
                EMEI_STAT_WAVE(18);  // ... spare refills
                if (kVFlag ? spare_flag == 0u : __builtin_amdgcn_inverse_ballot_w64(~spare_mask)) {
                    typename Env::Carry sc;
                    sc.trig = c.trig;
                    Env::init(sp, a.seed, a.env_offset + (uint64_t)i, episode + 1u, a.p);
                    Env::prime(sp, sc, a.p);
                    if constexpr (kLdsSpare) {
                        spare_s[wv][0][lane] = sp[0], spare_s[wv][1][lane] = sp[1], spare_s[wv][2][lane] = sp[2];
                        spare_s[wv][3][lane] = sp[3], spare_s[wv][4][lane] = sc.sn, spare_s[wv][5][lane] = sc.cs;
                    } else {
                        sp_sn = sc.sn, sp_cs = sc.cs;
                        Env::save_extra(sc, sp_e0, sp_e1);
                    }
                }
                spare_mask = ~0ull, spare_flag = 3u;
            }
            if (__builtin_amdgcn_inverse_ballot_w64(done_mask)) {
                ++episode;
                steps = 0;
                if constexpr (kLdsSpare) {  // a lane only ever reads what it wrote itself: no fence
                    s[0] = spare_s[wv][0][lane], s[1] = spare_s[wv][1][lane], s[2] = spare_s[wv][2][lane];
                    s[3] = spare_s[wv][3][lane], c.sn = spare_s[wv][4][lane], c.cs = spare_s[wv][5][lane];
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) s[k] = sp[k];
                    c.sn = sp_sn, c.cs = sp_cs;
                    Env::load_extra(c, sp_e0, sp_e1);
                }
                spare_flag = 0u;
            }
            spare_mask &= ~done_mask;
        }
    };
    // per-lane element offsets inside a flushed block of rows, computed once (the per-step address
    // is then a scalar row base plus this constant: no 64-bit vector multiply on the hot path)
    const int64_t rew_lane_off = (int64_t)(lane >> 4) * n + i0 + ((lane & 15) << 2);  // 16 lanes x 16 B per row
    const int64_t done_lane_off = (int64_t)(lane >> 2) * n + i0 + ((lane & 3) << 4);  // 4 lanes x 16 B per row
#if defined(EMEI_EXP_FLUSH_ROW0)  // experiment (timing only): the reward / done flushes always hit the first rows
#define EMEI_FLUSH_ROW(r) ((r) & 0)
#else
#define EMEI_FLUSH_ROW(r) (r)
#endif
    auto store_rew_rows = [&](int64_t row0, const float4& v) __attribute__((always_inline)) {
        store16_stream((float4*)(a.reward_out + EMEI_FLUSH_ROW(row0) * n + rew_lane_off), v);
    };
    auto store_done_rows = [&](int64_t row0, const uint4& v) __attribute__((always_inline)) {
        if (kStage == 16 || lane < 4 * kStage) store16_stream((uint4*)(a.done_out + EMEI_FLUSH_ROW(row0) * n + done_lane_off), v);  // kStage rows x 64 B
    };

    // Issue priority (Env::kRotatePriority: the InvertedPendulum kernels).  The SIMD's arbiter serves the OLDEST of its ready waves
    // first: of the four waves that share a SIMD (config 3) the first finishes after 45 % of the launch, the second after 60 %,
    // the third after 80 % (lifetimes in four clean groups of 1024 waves, profiles/r05_clock_probe.txt) and the SIMD ends the
    // launch with one wave.  Every tile the wave sets its priority to (tile + its wave slot) mod 4 — slots are 0..3, measured —
    // so that each of the four is the favoured one a quarter of the time.  Worth 6 % for the Balancing variants (0.683 -> 0.643 ms),
    // 0-3 % for SwingUp; finer rotation (every step, every two) is no better; for the HBM-bound CartPole kernel at two waves per
    // SIMD it was -8 % on one box and +4 % on another: not applied there.  s_setprio takes an immediate, hence the switch.
    const uint32_t wave_slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4) & 3u;  // HW_REG_HW_ID (4), WAVE_ID = bits [3:0]
    auto rotate_priority = [&](uint32_t tile) __attribute__((always_inline)) {
#if EMEI_PRIO_ROTATE
        if constexpr (Env::kRotatePriority) switch ((tile + wave_slot) & 3u) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
#endif
    };
    // PEERS: the observation row of step t also goes to every peer's gathered buffer (remote stores over xGMI: the launch then runs at the
    // links' rate, which is what the separate all-gather would take).  They are further stores AFTER the tile's loads in the shared in-order
    // counter, so the retire wait below (at most kStage operations left in flight) still covers every load.
    auto store_obs_peers = [&](int t, const float4& v) __attribute__((always_inline)) {
        if constexpr (PEERS) {
            const int64_t at = (int64_t)t * peers.row_envs + peers.col + li;
            for (uint32_t p = 0; p < peers.count; ++p) peers.obs[p][at] = v;
        }
    };
    EMEI_LOAD_TILE(0, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int t0 = 0, buf = 0;
    // The LDS read of a flush is issued at the end of one step and its global store after the
    // arithmetic of the NEXT step (each step is its own scheduling region because of the reset
    // check, so a read placed next to its use would expose the LDS latency on the serial chain).
    // Which step flushes what is known at compile time inside the unrolled tile; only the hand-over
    // from the previous tile (step 0) is a run-time, wave-uniform condition.
    float4 rew_pend = make_float4(0.f, 0.f, 0.f, 0.f);
    uint4 done_pend = make_uint4(0u, 0u, 0u, 0u);
    for (; t0 + kStage <= a.n_steps; t0 += kStage, buf ^= 1) {
        EMEI_LOAD_TILE(t0 + kStage, buf ^ 1)  // next tile: in flight under the 16 steps below
        rotate_priority((uint32_t)(t0 / kStage));
        const ActT* act_l = (const ActT*)&act_s[wv][buf][0];
        ActT act_cur = act_l[lane];
        // 16 steps = 4 groups x 4 unrolled steps (one reward flush period), the group loop unrolled by two: at the loop's
        // back-edge hipcc moves the carried state back into its canonical registers (~18 copies), and with 8 steps per trip
        // that costs half as much per step — A/B on one box: 0.279 -> 0.270 ms (SwingUp), 0.165 -> 0.160 (Balancing),
        // 0.613 -> 0.605 (InvertedPendulum).  All 16 unrolled is faster still for Balancing but bimodal for SwingUp
        // (0.267 / 0.291 ms: the instruction cache two CUs share).
#ifndef EMEI_GROUP_UNROLL
#define EMEI_GROUP_UNROLL 2
#endif
#pragma unroll EMEI_GROUP_UNROLL
        for (int g = 0; g < kStage / 4; ++g) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = 4 * g + q;
                // the next step's action leaves LDS while this step computes
                const ActT act_now = act_cur;
                act_cur = act_l[min(j + 1, kStage - 1) * kWave + lane];
                R o[4], rew;
                advance(act_now, o, rew);
                const float4 obs4 = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
#if defined(EMEI_EXP_OBS_ROW0)  // experiment (timing only): the observation stores alternate between rows 0 and 1 — the same instructions, no HBM write stream
                (a.obs_out + (int64_t)((t0 + j) & 1) * n)[li] = obs4;
#else
                store16_stream(a.obs_out + (int64_t)(t0 + j) * n + li, obs4);
#endif
                store_obs_peers(t0 + j, obs4);
                if (q == 0) {  // rows staged during the previous group: their LDS read has landed
                    if (t0 + j > 0) store_rew_rows(t0 + j - 4, rew_pend);
                    if (g == 0 && t0 > 0) store_done_rows(t0 - kStage, done_pend);
                }
                rew_s[wv][q][lane] = (float)rew;
                done_s[wv][j][lane] = (uint8_t)done;
                if (q == 3) {
                    wave_lds_fence();  // the flush reads what OTHER lanes staged
                    rew_pend = ((const float4*)&rew_s[wv][0][0])[lane];                             // 4 rows x 256 B
                    if (g == kStage / 4 - 1) done_pend = ((const uint4*)&done_s[wv][0][0])[lane & (4 * kStage - 1)];  // kStage rows x 64 B
                    wave_lds_fence();  // ... and is done before the next group overwrites the slice
                }
                maybe_reset();
            }
        }
        // Retire the tile.  Loads, stores and LDS-DMA share one in-order counter, and the tile's loads were
        // issued before ALL of this tile's stores, of which the kStage observation stores are unconditional
        // (the reward / done flushes add 1 to 5 more): leaving the kTileWaitKeep = kStage youngest operations
        // in flight therefore covers every LDS-DMA load, and what it additionally waits for was issued at
        // least kStage - 3 steps ago.  The count is DERIVED from the tile constants — a hand-written 19
        // (16 + the first tile's 3 reward flushes) was exact with zero margin — and tests/test_isa_guards.py
        // checks the emitted immediate and the stores of the loop body in the shipped code object.
        static_assert(kTileWaitKeep == kStage && kTileWaitKeep < 64, "one unconditional obs store per staged step");
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(kTileWaitKeep) : "memory");
        // Env::kTileBarrier (no env sets it any more): the four waves of a block meet here, once per tile.  They work on 256 neighbouring
        // envs — 4 KiB of every output row — and otherwise drift apart.  With plain output stores this was worth 7-11 % for CartPoleSwingUp
        // on the slower boxes (0.287-0.294 -> 0.260-0.267 ms; nothing on the fast ones; CartPoleBalancing lost 4-5 % to the waiting).  The
        // non-temporal stores (emei_device.h:store16_stream) remove what it worked around, and on top of them the barrier COSTS 3-4 %
        // (0.2348 -> 0.2253 ms without it, 131 072 envs 0.4669 -> 0.4518): profiles/EXPERIMENTS.md.
        if constexpr (EMEI_TILE_BARRIER != 0 && Env::kTileBarrier) __builtin_amdgcn_s_barrier();
    }
    if (t0 > 0) {
        store_rew_rows(t0 - 4, rew_pend);
        store_done_rows(t0 - kStage, done_pend);
    }
    // tail (< kStage steps): direct loads and stores
    for (int t = t0; t < a.n_steps; ++t) {
        R o[4], rew;
        advance((actions + (int64_t)t * n)[li], o, rew);
        const float4 obs4 = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
        (a.obs_out + (int64_t)t * n)[li] = obs4;
        store_obs_peers(t, obs4);
        (a.reward_out + (int64_t)t * n)[li] = (float)rew;
        (a.done_out + (int64_t)t * n)[li] = (uint8_t)done;
        maybe_reset();
    }

#pragma unroll
    for (int k = 0; k < 4; ++k) a.state[k * n + i] = s[k];
    a.steps[i] = steps;
    a.episode[i] = episode;
    unsigned long long m = __ballot(done != 0);
    if (lane == 0) a.done_mask[i / kWave] = m;
    EMEI_CLOCK_END();
#undef EMEI_LOAD_TILE
}

template <class Env, typename ActT, bool FREQ1>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(Env::kMinWavesPerEU)))
    pend_rollout_staged_kernel(const RolloutArgs<Env> a) {
    pend_rollout_staged<Env, ActT, FREQ1, false>(a, PeerObs{});
}

// ... and with the peer stores (emei_set_obs_peers; Env::kPeerWrite: the CartPole family, BASELINE configs[4]'s env)
template <class Env, typename ActT, bool FREQ1>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(Env::kMinWavesPerEU)))
    pend_rollout_staged_peers_kernel(const RolloutArgs<Env> a, const PeerObs peers) {
    pend_rollout_staged<Env, ActT, FREQ1, true>(a, peers);
}

// Env.reset on the device
template <class Env>
__global__ void __launch_bounds__(kBlock)
    pend_reset_kernel(typename Env::real* state, int32_t* steps, uint32_t* episode, int64_t n, uint64_t seed,
                      uint64_t env_offset, typename Env::Params p) {
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

// initial observation of (env, episode) pairs under the device reset generator
template <class Env>
__global__ void __launch_bounds__(kBlock)
    pend_init_obs_kernel(const int64_t* env_index, const uint32_t* episode, float4* obs, int64_t count, uint64_t seed,
                         uint64_t env_offset, typename Env::Params p) {
    using R = typename Env::real;
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= count) return;
    R s[4], o[4];
    Env::init(s, seed, env_offset + (uint64_t)env_index[k], episode[k], p);
    Env::obs_of(s, o);
    obs[k] = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
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
// stateless batched functions on [n,4] observations of the caller's dtype T (float or double: emei_io_dtype)
// get_batch_reward / get_batch_terminal (core.py:182-188).  float64 rows are not narrowed: the reference evaluates
// these functions on float64 arrays (cartpole.py:124-129,145-151; inverted_pendulum.py:73-183)
template <typename T>
__device__ __forceinline__ void load_row4(const T* base, int64_t i, T (&v)[4]) {
    if constexpr (sizeof(T) == 4) {
        const float4 x = ((const float4*)base)[i];
        v[0] = x.x, v[1] = x.y, v[2] = x.z, v[3] = x.w;
    } else {
        const double2 x = ((const double2*)base)[2 * i], y = ((const double2*)base)[2 * i + 1];
        v[0] = x.x, v[1] = x.y, v[2] = y.x, v[3] = y.y;
    }
}
template <typename T, typename R>
__device__ __forceinline__ void store_row4(T* base, int64_t i, const R (&o)[4]) {
    if constexpr (sizeof(T) == 4) {
        ((float4*)base)[i] = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
    } else {
        ((double2*)base)[2 * i] = make_double2((double)o[0], (double)o[1]);
        ((double2*)base)[2 * i + 1] = make_double2((double)o[2], (double)o[3]);
    }
}

template <class Env, typename T>
__global__ void __launch_bounds__(kBlock)
    pend_reward_terminal_kernel(const T* obs, T* reward, uint8_t* terminal, int64_t n,
                                typename Env::Params p, const SinCosEntry* trig) {
    using R = typename Env::real;
    __shared__ SinCosEntry trig_s[kTrigTableSize];
    stage_trig_table(trig_s, trig, Env::trig_rot_c(), Env::trig_rot_s());
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    T v[4];
    load_row4(obs, i, v);
    R o[4] = {(R)v[0], (R)v[1], (R)v[2], (R)v[3]};
    // the observation IS the state for reward/terminal purposes (wrapped angle has the same cosine);
    // build the carry from it.  For InvertedPendulum o[1] is theta, prime() adds phi_off itself.
    typename Env::Carry c;
    trig_ctx_init(c.trig, trig_s);
    Env::prime(o, c, p);
    if (reward) {
        // float64 rows: the reward in the working precision (the fused kernels round it to the float32 they store)
        if constexpr (sizeof(T) == 8) reward[i] = (T)Env::reward_exact(o, c, p);
        else reward[i] = (T)Env::reward(o, c, p);
    }
    if (terminal) terminal[i] = (uint8_t)Env::terminal(o, c, p);
}

// get_batch_next_obs (core.py:190-193): one step from caller-supplied observations
template <class Env, typename T>
__global__ void __launch_bounds__(kBlock)
    pend_next_obs_kernel(const T* obs, const void* actions, int action_dtype, T* next_obs, int64_t n,
                         int freq_rate, typename Env::Params p, const SinCosEntry* trig) {
    using R = typename Env::real;
    __shared__ SinCosEntry trig_s[kTrigTableSize];
    stage_trig_table(trig_s, trig, Env::trig_rot_c(), Env::trig_rot_s());
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    T v[4];
    load_row4(obs, i, v);
    R s[4] = {(R)v[0], (R)v[1], (R)v[2], (R)v[3]}, o[4], rew;
    bool term;
    typename Env::Carry c;
    trig_ctx_init(c.trig, trig_s);
    Env::prime(s, c, p);
    Env::step(s, c, Env::load_action(actions, action_dtype, i), p, freq_rate, o, rew, term);
    store_row4(next_obs, i, o);
}

// ---------------------------------------------------------------------------------------------
static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

#ifndef EMEI_XCD_CONTIGUOUS
#define EMEI_XCD_CONTIGUOUS 1  // 0: a variant build with the identity workgroup -> env block map, for A/B runs
#endif
#ifndef EMEI_SPLIT_LAUNCH
#define EMEI_SPLIT_LAUNCH 1  // 0: a variant build without the split of large HBM-bound launches, for A/B runs
#endif
// CUs of the current device (cached: the launch path neither allocates nor synchronises; an attribute query is neither)
static inline int device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev] && hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus[dev] = 256;
    return cus[dev];
}

// Returns the enum emei_kernel_id launched, or a negative EMEI_ERR_* (peers set on a path that has no peer stores).
template <class Env, typename ActT>
static int launch_rollout_full(const RolloutArgs<Env>& a, dim3 grid, hipStream_t stream, const ObsPeers& peers) {
    const bool full = a.obs_out && a.reward_out && a.done_out;
    PeerObs po{};
    if (peers.count > 0) {
        if (!Env::kPeerWrite) return EMEI_ERR_UNSUPPORTED;
        for (int p = 0; p < peers.count; ++p) {
            if (!aligned16(peers.obs[p])) return EMEI_ERR_UNSUPPORTED;
            po.obs[p] = (float4*)peers.obs[p];
        }
        po.row_envs = peers.row_envs, po.col = peers.col, po.count = (uint32_t)peers.count;
    }
    if (full && a.n_steps >= stage_steps<ActT>() && a.n % kWave == 0 && aligned16(a.actions) && aligned16(a.obs_out) &&
        aligned16(a.reward_out) && aligned16(a.done_out))
    {
        // Env::kSplitLaunch (the HBM-bound CartPole kernels): a shard of 2 or 3 waves per SIMD (BASELINE configs[4]'s 131 072 envs per GPU)
        // runs as consecutive launches of ONE wave per SIMD each (one 256-thread block per CU): co-resident waves' write streams
        // get in each other's way at that occupancy.  Same box: 131 072 envs 0.706 -> 0.671 ms, 196 608 0.976 -> 0.912; nothing at
        // 4 waves per SIMD, +8 % (worse) at 8, nothing at 16 — hence kSplitMaxRounds = 3 (profiles/r05_split_launch.txt).  Envs are
        // independent: the same bits.
        unsigned per_launch = grid.x;
        if constexpr (EMEI_SPLIT_LAUNCH != 0 && Env::kSplitLaunch) {
            const unsigned round = (unsigned)device_cus();  // blocks of one wave per SIMD
            if (grid.x > round && grid.x <= Env::kSplitMaxRounds * round) per_launch = round;
        }
        RolloutArgs<Env> b = a;
#ifdef EMEI_XCD_ALWAYS  // experiment: the XCD-contiguous map at every size
        const bool big = true;
#else
        const bool big = grid.x > (unsigned)device_cus();  // more than one wave per SIMD
#endif
        for (unsigned first = 0; first < grid.x; first += per_launch) {
            b.first_block = first;
            const dim3 g(min(per_launch, grid.x - first));
            b.xcd_contiguous = EMEI_XCD_CONTIGUOUS != 0 && Env::kXcdContiguous && big && g.x % 8 == 0;
            if constexpr (Env::kPeerWrite) {
                if (po.count > 0) {
                    if (a.freq_rate == 1)
                        hipLaunchKernelGGL((pend_rollout_staged_peers_kernel<Env, ActT, true>), g, dim3(kBlock), 0, stream, b, po);
                    else
                        hipLaunchKernelGGL((pend_rollout_staged_peers_kernel<Env, ActT, false>), g, dim3(kBlock), 0, stream, b, po);
                    continue;
                }
            }
            if (a.freq_rate == 1)
                hipLaunchKernelGGL((pend_rollout_staged_kernel<Env, ActT, true>), g, dim3(kBlock), 0, stream, b);
            else
                hipLaunchKernelGGL((pend_rollout_staged_kernel<Env, ActT, false>), g, dim3(kBlock), 0, stream, b);
        }
        if (po.count > 0) return a.freq_rate == 1 ? EMEI_KERNEL_PEND_STAGED_PEERS_FREQ1 : EMEI_KERNEL_PEND_STAGED_PEERS;
        return a.freq_rate == 1 ? EMEI_KERNEL_PEND_STAGED_FREQ1 : EMEI_KERNEL_PEND_STAGED;
    }
    if (po.count > 0) return EMEI_ERR_UNSUPPORTED;  // ragged n, a short horizon, unaligned or missing outputs: the generic kernels have no peer stores
    if (full) {
        hipLaunchKernelGGL((pend_rollout_kernel<Env, ActT, true>), grid, dim3(kBlock), 0, stream, a);
        return EMEI_KERNEL_PEND_GENERIC_FULL;
    }
    hipLaunchKernelGGL((pend_rollout_kernel<Env, ActT, false>), grid, dim3(kBlock), 0, stream, a);
    return EMEI_KERNEL_PEND_GENERIC;
}

// host-side dispatch over (env id, precision)
// every launch of one Env type (one translation unit instantiates exactly one Env: pendulum_tu.hip)
template <class Env>
static int launch_env(const PendLaunch& L) {
    using R = typename Env::real;
    RolloutArgs<Env> a;
    a.state = (R*)L.state;
    a.steps = L.steps;
    a.episode = L.episode;
    a.done_mask = L.done_mask;
    a.actions = L.actions;
    a.trig = (const SinCosEntry*)L.trig;
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
    a.p = Env::make_params(L.p);
    a.obs_f64 = L.op == PEND_OP_ROLLOUT ? L.obs_f64 : nullptr;
    a.host_flag = L.op == PEND_OP_ROLLOUT ? L.host_flag : nullptr;
    a.flag_value = L.flag_value;
    a.first_block = 0, a.xcd_contiguous = 0;
    dim3 grid((unsigned)((L.n + kBlock - 1) / kBlock));
    switch (L.op) {
        case PEND_OP_ROLLOUT: {
            // discrete envs take uint8/int32/int64 actions, continuous envs float32
            int sel;
            if constexpr (Env::kDiscrete) {
                if (L.action_dtype == EMEI_ACT_U8) sel = launch_rollout_full<Env, uint8_t>(a, grid, L.stream, L.peers);
                else if (L.action_dtype == EMEI_ACT_I32) sel = launch_rollout_full<Env, int32_t>(a, grid, L.stream, L.peers);
                else if (L.action_dtype == EMEI_ACT_I64) sel = launch_rollout_full<Env, int64_t>(a, grid, L.stream, L.peers);
                else return EMEI_ERR_INVALID;
            } else {
                if (L.action_dtype != EMEI_ACT_F32) return EMEI_ERR_INVALID;
                sel = launch_rollout_full<Env, float>(a, grid, L.stream, L.peers);
            }
            if (sel < 0) return sel;
            if (L.selected) *L.selected = sel;
            break;
        }
        case PEND_OP_RESET:
            hipLaunchKernelGGL(pend_reset_kernel<Env>, grid, dim3(kBlock), 0, L.stream, (R*)L.state, L.steps,
                               L.episode, L.n, L.seed, L.env_offset, a.p);
            break;
        case PEND_OP_INIT_OBS:
            hipLaunchKernelGGL(pend_init_obs_kernel<Env>, grid, dim3(kBlock), 0, L.stream, L.env_index, L.episode_in,
                               (float4*)L.obs_out, L.n, L.seed, L.env_offset, a.p);
            break;
        case PEND_OP_GET_OBS:
            hipLaunchKernelGGL(pend_get_obs_kernel<Env>, grid, dim3(kBlock), 0, L.stream, (const R*)L.state,
                               L.obs_f64, L.n);
            break;
        case PEND_OP_REWARD_TERMINAL:
            if (L.io_f64)
                hipLaunchKernelGGL((pend_reward_terminal_kernel<Env, double>), grid, dim3(kBlock), 0, L.stream,
                                   (const double*)L.obs_in, (double*)L.reward_out, L.done_out, L.n, a.p, a.trig);
            else
                hipLaunchKernelGGL((pend_reward_terminal_kernel<Env, float>), grid, dim3(kBlock), 0, L.stream,
                                   (const float*)L.obs_in, (float*)L.reward_out, L.done_out, L.n, a.p, a.trig);
            break;
        case PEND_OP_NEXT_OBS:
            if (L.io_f64)
                hipLaunchKernelGGL((pend_next_obs_kernel<Env, double>), grid, dim3(kBlock), 0, L.stream, (const double*)L.obs_in,
                                   L.actions, L.action_dtype, (double*)L.obs_out, L.n, L.freq_rate, a.p, a.trig);
            else
                hipLaunchKernelGGL((pend_next_obs_kernel<Env, float>), grid, dim3(kBlock), 0, L.stream, (const float*)L.obs_in,
                                   L.actions, L.action_dtype, (float*)L.obs_out, L.n, L.freq_rate, a.p, a.trig);
            break;
        default: return EMEI_ERR_INVALID;
    }
    return hipGetLastError() == hipSuccess ? EMEI_OK : EMEI_ERR_HIP;
}


}  // namespace emei
