// cheetah.hip — HalfCheetah-style body (emei/envs/mujoco/half_cheetah.py on mujoco_env.py):
// model constants from assets/half_cheetah.xml, step / rollout / reset kernels, reward / terminal.
//
// One thread per env; state SoA in HBM ([18][n] Reals: qpos then qvel), kept in registers across a
// rollout.  Unlike the 4-state family, an env's observation is 72 B and its action 24 B, so lane-wise
// global accesses would be 18 strided dword stores per step.  Each wave therefore stages its 64 envs
// through a private LDS slice every (sub)step boundary: actions arrive as 16 B-per-lane loads of the
// wave's contiguous 1536 B block and are read back per lane; the 64x18 observation block is written
// to LDS lane-wise and leaves as 4.5 KiB of contiguous 16 B-per-lane stores.  The arithmetic
// (cheetah_model.h, ~3-6 k flops per substep, data-dependent contact branches) dominates: this
// kernel is VALU/latency-bound and its HBM-roofline fraction is low by construction.
#include "cheetah.h"

#include <cmath>
#include <cstring>

#include "cheetah_model.h"

namespace emei {

using cheetah::Model;
using cheetah::NV;

// ---------------------------------------------------------------------------------------------
// host: model constants from assets/half_cheetah.xml (inertiafromgeom, settotalmass = 14, xml:35)
namespace {
struct H2 {
    double x, z;
};
H2 hrot(double a, H2 v) { return {v.x * std::cos(a) + v.z * std::sin(a), -v.x * std::sin(a) + v.z * std::cos(a)}; }
double capsule_mass(double rho, double r, double half) { return rho * (M_PI * r * r * 2 * half + 4.0 / 3.0 * M_PI * r * r * r); }
double capsule_inertia_perp(double rho, double r, double half) {
    double h = 2 * half, mcyl = rho * M_PI * r * r * h, msph = rho * 4.0 / 3.0 * M_PI * r * r * r;
    return mcyl * (3 * r * r + h * h) / 12 + msph * (2 * r * r / 5 + h * h / 4 + 3 * h * r / 8);
}
}  // namespace

Model cheetah_make_model(double dt, double init_noise) {
    Model m;
    memset(&m, 0, sizeof(m));
    const double r = 0.046, rho = 1000.0;
    // bodies in the xml's order: torso, bthigh, bshin, bfoot, fthigh, fshin, ffoot
    const int parent[7] = {-1, 0, 1, 2, 0, 4, 5};
    const H2 bpos[7] = {{0, 0.7}, {-0.5, 0}, {0.16, -0.25}, {-0.28, -0.14}, {0.5, 0}, {-0.14, -0.24}, {0.13, -0.18}};
    struct G {
        int body;
        H2 c;
        double ang, half;
    } g[8] = {{0, {0, 0}, M_PI / 2, 0.5},       {0, {0.6, 0.1}, 0.87, 0.15},       {1, {0.1, -0.13}, -3.8, 0.145},
              {2, {-0.14, -0.07}, -2.03, 0.15}, {3, {0.03, -0.097}, -0.27, 0.094}, {4, {-0.07, -0.12}, 0.52, 0.133},
              {5, {0.065, -0.09}, -0.6, 0.106}, {6, {0.045, -0.07}, -0.6, 0.07}};
    double gm[8], gi[8], total = 0;
    for (int k = 0; k < 8; ++k) gm[k] = capsule_mass(rho, r, g[k].half), gi[k] = capsule_inertia_perp(rho, r, g[k].half), total += gm[k];
    double mass[7], inertia[7];
    H2 com[7];
    for (int b = 0; b < 7; ++b) {
        double mb = 0;
        H2 c = {0, 0};
        for (int k = 0; k < 8; ++k)
            if (g[k].body == b) mb += gm[k], c.x += gm[k] * g[k].c.x, c.z += gm[k] * g[k].c.z;
        c.x /= mb, c.z /= mb;
        double I = 0;
        for (int k = 0; k < 8; ++k)
            if (g[k].body == b) {
                double dx = g[k].c.x - c.x, dz = g[k].c.z - c.z;
                I += gi[k] + gm[k] * (dx * dx + dz * dz);
            }
        mass[b] = mb, com[b] = c, inertia[b] = I;
    }
    const double s = 14.0 / total;
    for (int b = 0; b < 7; ++b) mass[b] *= s, inertia[b] *= s;
    // subtree masses
    double sub[7];
    for (int b = 0; b < 7; ++b) sub[b] = mass[b];
    for (int b = 6; b > 0; --b) sub[parent[b]] += sub[b];
    // permuted link order: 0 bfoot 1 bshin 2 bthigh 3 ffoot 4 fshin 5 fthigh 6 torso  <- xml body index
    const int perm_body[7] = {3, 2, 1, 6, 5, 4, 0};
    for (int p = 0; p < 7; ++p) {
        const int b = perm_body[p];
        double sx = mass[b] * com[b].x, sz = mass[b] * com[b].z;
        double dg = inertia[b] + mass[b] * (com[b].x * com[b].x + com[b].z * com[b].z);
        for (int c = 1; c < 7; ++c)
            if (parent[c] == b) {
                sx += sub[c] * bpos[c].x, sz += sub[c] * bpos[c].z;
                dg += sub[c] * (bpos[c].x * bpos[c].x + bpos[c].z * bpos[c].z);
            }
        m.sx[p] = sx, m.sz[p] = sz, m.diag[p] = dg;
    }
    m.d_tb[0] = bpos[1].x, m.d_tb[1] = bpos[1].z;
    m.d_tf[0] = bpos[4].x, m.d_tf[1] = bpos[4].z;
    m.d_bt_bs[0] = bpos[2].x, m.d_bt_bs[1] = bpos[2].z;
    m.d_bs_bf[0] = bpos[3].x, m.d_bs_bf[1] = bpos[3].z;
    m.d_ft_fs[0] = bpos[5].x, m.d_ft_fs[1] = bpos[5].z;
    m.d_fs_ff[0] = bpos[6].x, m.d_fs_ff[1] = bpos[6].z;
    m.mtot = 14.0, m.gravity = 9.81, m.z0 = bpos[0].z;
    const double stiff[6] = {240, 180, 120, 180, 120, 60}, damp[6] = {6, 4.5, 3, 4.5, 3, 1.5};
    const double lo[6] = {-0.52, -0.785, -0.4, -1.0, -1.2, -0.5}, hi[6] = {1.05, 0.785, 0.785, 0.7, 0.87, 0.5};
    const double gear[6] = {120, 90, 60, 120, 60, 30};
    for (int k = 0; k < 6; ++k)
        m.stiff[k] = stiff[k], m.damp[k] = damp[k], m.arm[k] = 0.1, m.lo[k] = lo[k], m.hi[k] = hi[k], m.gear[k] = gear[k];
    for (int k = 0; k < 8; ++k) {  // capsule end spheres: centre -/+ half * axis, axis = +z rotated by ang about y
        H2 ax = hrot(g[k].ang, {0, 1});
        m.geom_end[2 * k][0] = g[k].c.x - g[k].half * ax.x, m.geom_end[2 * k][1] = g[k].c.z - g[k].half * ax.z;
        m.geom_end[2 * k + 1][0] = g[k].c.x + g[k].half * ax.x, m.geom_end[2 * k + 1][1] = g[k].c.z + g[k].half * ax.z;
    }
    m.radius = r, m.friction = 0.4;
    // solref (.02, 1) with MuJoCo's refsafe clamp timeconst >= 2 dt; solimp contacts (0,.8,.01), limits (0,.8,.03)
    const double tc = 0.02 < 2 * dt ? 2 * dt : 0.02, dmax = 0.8;
    m.cK = m.lK = 1.0 / (dmax * dmax * tc * tc), m.cB = m.lB = 2.0 / (dmax * tc);
    m.c_dmin = 0.0, m.c_dmax = dmax, m.c_width = 0.01;
    m.l_dmin = 0.0, m.l_dmax = dmax, m.l_width = 0.03;
    m.dt = dt;
    m.init_sigma = (float)init_noise;
    return m;
}

// ---------------------------------------------------------------------------------------------
constexpr int kObs = 18, kAct = 6;

// device reset: init_qpos/qvel (zeros) + sigma N(0,1) per coordinate (mujoco_env.py:137-140)
template <typename R>
__device__ __forceinline__ void cheetah_init(R (&q)[NV], R (&v)[NV], uint64_t seed, uint64_t env, uint32_t episode,
                                             float sigma) {
    float z[20];
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        u32x4 r = philox4x32_10(seed, env, episode, (uint32_t)b);
        boxmuller(r.v[0], r.v[1], z[4 * b], z[4 * b + 1]);
        boxmuller(r.v[2], r.v[3], z[4 * b + 2], z[4 * b + 3]);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) q[i] = (R)__fmul_rn(sigma, z[i]), v[i] = (R)__fmul_rn(sigma, z[NV + i]);
}

template <typename R>
__global__ void __launch_bounds__(kBlock)
    cheetah_reset_kernel(R* state, int32_t* steps, uint32_t* episode, int64_t n, uint64_t seed, uint64_t env_offset,
                         float sigma) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    R q[NV], v[NV];
    cheetah_init(q, v, seed, env_offset + (uint64_t)i, 0u, sigma);
#pragma unroll
    for (int k = 0; k < NV; ++k) state[(int64_t)k * n + i] = q[k], state[(int64_t)(NV + k) * n + i] = v[k];
    steps[i] = 0;
    episode[i] = 0;
}

__global__ void __launch_bounds__(kBlock)
    cheetah_init_obs_kernel(const int64_t* env_index, const uint32_t* episode, float* obs, int64_t count, uint64_t seed,
                            uint64_t env_offset, float sigma) {
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= count) return;
    double q[NV], v[NV];
    cheetah_init(q, v, seed, env_offset + (uint64_t)env_index[k], episode[k], sigma);
#pragma unroll
    for (int j = 0; j < NV; ++j) obs[k * kObs + j] = (float)q[j], obs[k * kObs + NV + j] = (float)v[j];
}

struct CheetahArgs {
    void* state;
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
    Model m;
};

// emei_step / emei_rollout for the cheetah (mujoco_env.py:157-167)
template <typename R>
__global__ void __launch_bounds__(kBlock) cheetah_rollout_kernel(const CheetahArgs a) {
    constexpr int kWaves = kBlock / kWave;
    __shared__ __attribute__((aligned(16))) float act_s[kWaves][kWave * kAct];  // 1536 B per wave
    __shared__ __attribute__((aligned(16))) float obs_s[kWaves][kWave * kObs];  // 4608 B per wave
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t n = a.n;
    const int64_t i0 = i - lane;                                     // first env of this wave
    const int wave_envs = (int)min((int64_t)kWave, n - i0);          // ragged last wave
    const bool active = i < n;
    R* state = (R*)a.state;

    R q[NV], v[NV];
    int32_t steps = 0;
    uint32_t episode = 0;
    if (active) {
#pragma unroll
        for (int k = 0; k < NV; ++k) q[k] = state[(int64_t)k * n + i], v[k] = state[(int64_t)(NV + k) * n + i];
        steps = a.steps[i];
        episode = a.episode[i];
    } else {
#pragma unroll
        for (int k = 0; k < NV; ++k) q[k] = R(0), v[k] = R(0);
    }
    const bool auto_reset = (a.flags & EMEI_FLAG_AUTO_RESET) != 0;
    uint32_t done = 0;

    // this wave's action block of step t: wave_envs*6 contiguous floats starting at (t*n + i0)*6
    auto fetch_actions = [&](int t, float4& lo, float4& hi) __attribute__((always_inline)) {
        const float* base = a.actions + ((int64_t)t * n + i0) * kAct;
        const int nvec = wave_envs * kAct / 4;  // 16-byte vectors in the block (wave_envs*6 is a multiple of 2 floats)
        lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
        if (wave_envs == kWave && (((uintptr_t)base) & 15u) == 0) {
            lo = ((const float4*)base)[lane];
            if (lane < 32) hi = ((const float4*)base)[kWave + lane];
        } else {  // ragged / unaligned tail: scalar loads
            float* l4 = (float*)&lo;
            float* h4 = (float*)&hi;
            for (int c = 0; c < 4; ++c) {
                int e0 = lane * 4 + c, e1 = (kWave + lane) * 4 + c;
                if (e0 < wave_envs * kAct) l4[c] = base[e0];
                if (lane < 32 && e1 < wave_envs * kAct) h4[c] = base[e1];
            }
        }
        (void)nvec;
    };
    float4 alo, ahi;
    fetch_actions(0, alo, ahi);
    for (int t = 0; t < a.n_steps; ++t) {
        // stage this step's actions through LDS, then prefetch the next step's block
        ((float4*)act_s[wv])[lane] = alo;
        if (lane < 32) ((float4*)act_s[wv])[kWave + lane] = ahi;
        R ctrl[kAct];
#pragma unroll
        for (int k = 0; k < kAct; ++k) ctrl[k] = (R)act_s[wv][lane * kAct + k];
        if (t + 1 < a.n_steps) fetch_actions(t + 1, alo, ahi);

        const R x_before = q[0];
        for (int s = 0; s < a.freq_rate; ++s) cheetah::substep(q, v, ctrl, a.m);  // mujoco_env.py:88-97
        // half_cheetah.py:59-63 (per env, = step() semantics): w_f (x' - x)/dt_env - w_c sum a^2, dt_env = dt*freq_rate
        R cost = R(0);
#pragma unroll
        for (int k = 0; k < kAct; ++k) cost = fma_r(ctrl[k], ctrl[k], cost);
        const R rew = (q[0] - x_before) / ((R)a.m.dt * (R)a.freq_rate) - R(0.1) * cost;
        bool fin = true;  // half_cheetah.py:65-67
#pragma unroll
        for (int k = 0; k < NV; ++k) fin &= finite_r(q[k]) & finite_r(v[k]);
        ++steps;
        const bool trunc = (a.max_episode_steps > 0) & (steps >= a.max_episode_steps);
        done = active ? ((fin ? 0u : EMEI_DONE_TERMINAL) | (trunc ? EMEI_DONE_TRUNCATED : 0u)) : 0u;

        if (a.obs_out) {  // obs = concat(qpos, qvel) (mujoco_env.py:153-155): lane-wise into LDS, linear out
            float* mine = &obs_s[wv][lane * kObs];
#pragma unroll
            for (int k = 0; k < NV; ++k) mine[k] = (float)q[k], mine[NV + k] = (float)v[k];
            float* dst = a.obs_out + ((int64_t)t * n + i0) * kObs;
            const int nflt = wave_envs * kObs;
            if (wave_envs == kWave && (((uintptr_t)dst) & 15u) == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) ((float4*)dst)[c * kWave + lane] = ((const float4*)obs_s[wv])[c * kWave + lane];
                if (lane < 32) ((float4*)dst)[4 * kWave + lane] = ((const float4*)obs_s[wv])[4 * kWave + lane];
            } else {
                for (int e = lane; e < nflt; e += kWave) dst[e] = obs_s[wv][e];
            }
        }
        if (active) {
            if (a.reward_out) a.reward_out[(int64_t)t * n + i] = (float)rew;
            if (a.done_out) a.done_out[(int64_t)t * n + i] = (uint8_t)done;
        }
        if (__builtin_expect(auto_reset && __ballot(done != 0) != 0ull, 0)) {
            if (done != 0) {
                ++episode;
                steps = 0;
                cheetah_init(q, v, a.seed, a.env_offset + (uint64_t)i, episode, a.m.init_sigma);
            }
        }
    }
    if (active) {
#pragma unroll
        for (int k = 0; k < NV; ++k) state[(int64_t)k * n + i] = q[k], state[(int64_t)(NV + k) * n + i] = v[k];
        a.steps[i] = steps;
        a.episode[i] = episode;
    }
    unsigned long long mk = __ballot(done != 0);
    if (lane == 0 && active) a.done_mask[i / kWave] = mk;
}

// ---------------------------------------------------------------------------------------------
// half_cheetah.py:59-63 with step() semantics (B = 1 per env: the control cost is summed per env;
// the reference's batch form sums np.square(action) over the WHOLE batch, a quirk documented in
// DESIGN.md).  dt_env = real_time_scale * freq_rate (gym MujocoEnv.dt).  w_f = 1, w_c = 0.1 (:23-24).
__global__ void __launch_bounds__(kBlock)
    cheetah_reward_kernel(const float* obs, const float* pre_obs, const float* action, double inv_dt, float* reward,
                          int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double fwd = ((double)obs[i * 18] - (double)pre_obs[i * 18]) * inv_dt;
    double cost = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double a = action[i * 6 + k];
        cost += a * a;
    }
    reward[i] = (float)(1.0 * fwd - 0.1 * cost);
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
    hipLaunchKernelGGL(cheetah_reward_kernel, grid, dim3(kBlock), 0, s, obs, pre_obs, action, 1.0 / dt_env, reward_out, n);
    return hipGetLastError() == hipSuccess ? EMEI_OK : EMEI_ERR_HIP;
}
int cheetah_terminal(int64_t n, const float* obs, uint8_t* terminal_out, hipStream_t s) {
    dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(cheetah_terminal_kernel, grid, dim3(kBlock), 0, s, obs, terminal_out, n);
    return hipGetLastError() == hipSuccess ? EMEI_OK : EMEI_ERR_HIP;
}

int cheetah_init_obs(int64_t count, const int64_t* env_index, const uint32_t* episode, uint64_t seed, uint64_t env_offset,
                     double init_noise, float* obs_out, hipStream_t s) {
    dim3 grid((unsigned)((count + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(cheetah_init_obs_kernel, grid, dim3(kBlock), 0, s, env_index, episode, obs_out, count, seed,
                       env_offset, (float)init_noise);
    return hipGetLastError() == hipSuccess ? EMEI_OK : EMEI_ERR_HIP;
}

int cheetah_reset(void* state, int32_t* steps, uint32_t* episode, int64_t n, int precision, uint64_t seed,
                  uint64_t env_offset, double init_noise, hipStream_t s) {
    dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    if (precision == EMEI_PRECISION_F32)
        hipLaunchKernelGGL(cheetah_reset_kernel<float>, grid, dim3(kBlock), 0, s, (float*)state, steps, episode, n, seed,
                           env_offset, (float)init_noise);
    else
        hipLaunchKernelGGL(cheetah_reset_kernel<double>, grid, dim3(kBlock), 0, s, (double*)state, steps, episode, n,
                           seed, env_offset, (float)init_noise);
    return hipGetLastError() == hipSuccess ? EMEI_OK : EMEI_ERR_HIP;
}

int cheetah_rollout(void* state, int32_t* steps, uint32_t* episode, unsigned long long* done_mask, int64_t n,
                    int precision, int32_t n_steps, int32_t freq_rate, double dt, int32_t max_episode_steps,
                    uint64_t seed, uint64_t env_offset, double init_noise, const float* actions, float* obs_out,
                    float* reward_out, uint8_t* done_out, uint32_t flags, hipStream_t s) {
    CheetahArgs a;
    a.state = state, a.steps = steps, a.episode = episode, a.done_mask = done_mask;
    a.actions = actions, a.obs_out = obs_out, a.reward_out = reward_out, a.done_out = done_out;
    a.n = n, a.n_steps = n_steps, a.freq_rate = freq_rate, a.max_episode_steps = max_episode_steps;
    a.flags = flags, a.seed = seed, a.env_offset = env_offset;
    a.m = cheetah_make_model(dt, init_noise);
    dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    if (precision == EMEI_PRECISION_F32)
        hipLaunchKernelGGL(cheetah_rollout_kernel<float>, grid, dim3(kBlock), 0, s, a);
    else
        hipLaunchKernelGGL(cheetah_rollout_kernel<double>, grid, dim3(kBlock), 0, s, a);
    return hipGetLastError() == hipSuccess ? EMEI_OK : EMEI_ERR_HIP;
}

}  // namespace emei
