// abi.hip — the extern "C" boundary of libemei_hip.so (declared in include/emei_hip.h).
// Host code only: argument checks, handle bookkeeping, launch dispatch.  No call in a launch path
// allocates, frees or synchronises (HIP-graph capturable).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <new>

#include "launch.h"
#include "body_kernels.h"
#include "cheetah_model.h"
#include "hopper_model.h"
#include "dpend_model.h"

using namespace emei;

struct emei_env {
    emei_config cfg;
    int obs_dim, act_dim, state_dim;
    size_t real_size;
    void* state;
    int32_t* steps;
    uint32_t* episode;
    unsigned long long* done_mask;
    void* frozen_state;
    int32_t* frozen_steps;
    uint32_t* frozen_episode;
    uint64_t frozen_seed;          // key of the reset generator at emei_freeze (restored by emei_unfreeze)
    unsigned long long* cap_hits;  // device counter: Newton solves that ended at the iteration cap (emei_get_solver_cap_hits)
    uint32_t* work;                // body_kernels.h:WorkQueue words of the chunked body rollout (2 + waves of the shard)
    int resident_waves;            // waves of this handle's body rollout kernel the device holds at once (0: not a body / unknown)
    bool has_state, frozen;
    int last_kernel;  // enum emei_kernel_id of the last emei_step / emei_rollout
    uint32_t* host_flag = nullptr;  // emei_step_host: page-locked completion word (allocated on first use) ...
    uint32_t flag_seq = 0;          // ... and the value the next launch stores into it
    PendParams pend;
    const void* trig;
    ObsPeers peers;  // emei_set_obs_peers
};

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(EMEI_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" EMEI_API const char* emei_last_error(void) { return g_err; }
extern "C" EMEI_API int emei_abi_version(void) { return EMEI_ABI_VERSION; }

extern "C" EMEI_API int emei_env_dims(int env_id, int* obs_dim, int* act_dim, int* state_dim) {
    int od, ad, sd;
    switch (env_id) {
        case EMEI_CARTPOLE_SWINGUP:
        case EMEI_CARTPOLE_BALANCING: od = 4, ad = 0, sd = 4; break;
        case EMEI_IP_REBOUND_BALANCING:
        case EMEI_IP_BOUNDARY_BALANCING:
        case EMEI_IP_REBOUND_SWINGUP:
        case EMEI_IP_BOUNDARY_SWINGUP: od = 4, ad = 1, sd = 4; break;
        case EMEI_HALFCHEETAH_RUNNING: od = 18, ad = 6, sd = 18; break;
        case EMEI_IDP_REBOUND_BALANCING:
        case EMEI_IDP_BOUNDARY_BALANCING:
        case EMEI_IDP_REBOUND_SWINGUP:
        case EMEI_IDP_BOUNDARY_SWINGUP: od = 6, ad = 1, sd = 6; break;
        case EMEI_HOPPER_RUNNING: od = 12, ad = 3, sd = 12; break;
        default: return fail(EMEI_ERR_INVALID, "unknown env_id %d", env_id);
    }
    if (obs_dim) *obs_dim = od;
    if (act_dim) *act_dim = ad;
    if (state_dim) *state_dim = sd;
    return EMEI_OK;
}

extern "C" EMEI_API int emei_model_constants(int env_id, double* out, int capacity) {
    if (!out || capacity < 0) return fail(EMEI_ERR_INVALID, "emei_model_constants: bad argument");
    double buf[256];
    int n;
    switch (env_id) {
        case EMEI_IP_REBOUND_BALANCING:
        case EMEI_IP_BOUNDARY_BALANCING:
        case EMEI_IP_REBOUND_SWINGUP:
        case EMEI_IP_BOUNDARY_SWINGUP: n = ip_xml_constants(buf); break;
        case EMEI_IDP_REBOUND_BALANCING:
        case EMEI_IDP_BOUNDARY_BALANCING:
        case EMEI_IDP_REBOUND_SWINGUP:
        case EMEI_IDP_BOUNDARY_SWINGUP: n = dpend::xml_constants(buf); break;
        case EMEI_HALFCHEETAH_RUNNING: n = cheetah::xml_constants(buf); break;
        case EMEI_HOPPER_RUNNING: n = hopper::xml_constants(buf); break;
        case EMEI_CARTPOLE_SWINGUP:
        case EMEI_CARTPOLE_BALANCING: return fail(EMEI_ERR_UNSUPPORTED, "emei_model_constants: CartPole has no model file (cartpole.py:22-27)");
        default: return fail(EMEI_ERR_INVALID, "unknown env_id %d", env_id);
    }
    if (n > capacity) return fail(EMEI_ERR_INVALID, "emei_model_constants: %d values, capacity %d", n, capacity);
    memcpy(out, buf, n * sizeof(double));
    return n;
}

extern "C" EMEI_API int emei_model_invweights(int env_id, double* out, int capacity) {
    if (!out || capacity < 0) return fail(EMEI_ERR_INVALID, "emei_model_invweights: bad argument");
    double buf[32];
    int n;
    switch (env_id) {
        case EMEI_IP_REBOUND_BALANCING:
        case EMEI_IP_BOUNDARY_BALANCING:
        case EMEI_IP_REBOUND_SWINGUP:
        case EMEI_IP_BOUNDARY_SWINGUP: {
            constexpr IpModel m = ip_make_model(false);
            buf[0] = m.invw, buf[1] = m.invw_hinge, n = 2;
            break;
        }
        case EMEI_IDP_REBOUND_BALANCING:
        case EMEI_IDP_BOUNDARY_BALANCING:
        case EMEI_IDP_REBOUND_SWINGUP:
        case EMEI_IDP_BOUNDARY_SWINGUP: buf[0] = dpend::make_model(false, 0.002).invw, n = 1; break;
        case EMEI_HALFCHEETAH_RUNNING: n = cheetah::xml_invweights(buf); break;
        case EMEI_HOPPER_RUNNING: n = hopper::xml_invweights(buf); break;
        case EMEI_CARTPOLE_SWINGUP:
        case EMEI_CARTPOLE_BALANCING: return fail(EMEI_ERR_UNSUPPORTED, "emei_model_invweights: CartPole has no constraint rows");
        default: return fail(EMEI_ERR_INVALID, "unknown env_id %d", env_id);
    }
    if (n > capacity) return fail(EMEI_ERR_INVALID, "emei_model_invweights: %d values, capacity %d", n, capacity);
    memcpy(out, buf, n * sizeof(double));
    return n;
}

// ---------------------------------------------------------------------------------------------
// {sin, cos}(k * 2pi/256), k = 0..255, correctly rounded from long double, one copy per device.
// Allocated on the first emei_create / stateless call for that device (never inside a hot launch
// path of an existing handle).
#include <cstddef>
#include <mutex>
namespace emei {
const void* emei_trig_table(int device) {
    static std::mutex mu;
    static void* tabs[64] = {nullptr};
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!tabs[device]) {
        SinCosEntry host[kTrigTableSize];
        for (int k = 0; k < kTrigTableSize; ++k) {
            const long double a = 2.0L * 3.141592653589793238462643383279502884L * k / kTrigTableSize;
            host[k].s = (double)sinl(a), host[k].c = (double)cosl(a);
        }
        int prev = 0;
        (void)hipGetDevice(&prev);
        void* d = nullptr;
        if (hipSetDevice(device) == hipSuccess && hipMalloc(&d, sizeof(host)) == hipSuccess &&
            hipMemcpy(d, host, sizeof(host), hipMemcpyHostToDevice) == hipSuccess)
            tabs[device] = d;
        (void)hipSetDevice(prev);
    }
    return tabs[device];
}
}  // namespace emei

static const void* current_device_trig() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    return emei_trig_table(dev);
}

static bool is_pend(int env_id) { return env_id >= EMEI_CARTPOLE_SWINGUP && env_id <= EMEI_IP_BOUNDARY_SWINGUP; }
static bool is_ip(int env_id) { return env_id >= EMEI_IP_REBOUND_BALANCING && env_id <= EMEI_IP_BOUNDARY_SWINGUP; }
// The classic-control envs ignore `integrator` like the reference (base_control.py:73) and have no
// observation noise; an InvertedPendulum with a non-default integrator or observation noise leaves
// the staged 4-state kernel for the generic Body rollout (ipend_model.h).
static bool steps_as_body(const emei_config& c) {
    if (!is_pend(c.env_id)) return true;
    bool obs_noise = false;
    for (int i = 0; i < EMEI_MAX_STATE_DIM; ++i) obs_noise |= c.obs_sigma[i] != 0.f;
    return is_ip(c.env_id) && (c.integrator != EMEI_INTEG_EULER || obs_noise);
}
static bool is_body(int env_id) { return env_id >= EMEI_HALFCHEETAH_RUNNING && env_id <= EMEI_HOPPER_RUNNING; }

// ---------------------------------------------------------------------------------------------
// Host twin of the InvertedPendulum constants for the launch descriptors: every number comes from pendulum_envs.h:
// ip_make_model (the one typed copy of emei/envs/mujoco/assets/inverted_pendulum.xml on the kernel side).
static PendParams pend_params(int env_id, double dt, const float* init_sigma = nullptr, int noise_shared = 0) {
    PendParams p;
    memset(&p, 0, sizeof(p));
    p.dt = dt;
    p.dt32 = (float)dt;
    for (int i = 0; i < 4; ++i) p.init_sigma[i] = init_sigma ? init_sigma[i] : 0.f;
    p.noise_shared = noise_shared;
    const bool swingup = env_id == EMEI_IP_REBOUND_SWINGUP || env_id == EMEI_IP_BOUNDARY_SWINGUP;
    const IpModel x = ip_make_model(swingup);
    p.M11 = x.M11, p.M22 = x.M22, p.mpr = x.mpr, p.mgr = x.mp * x.gravity * x.r;
    p.gear = x.gear, p.ctrl_lo = x.ctrl_lo, p.ctrl_hi = x.ctrl_hi, p.x_lo = x.x_lo, p.x_hi = x.x_hi;
    p.phi_off = x.phi_off, p.sin_off = x.sin_off, p.cos_off = x.cos_off;  // _update_model: pole body turned by pi about y
    p.invw = x.invw;                                                      // dof_invweight0 of the slider at qpos0
    p.tc = x.solref_tc < 2 * dt ? 2 * dt : x.solref_tc;                   // solref timeconst with MuJoCo's refsafe clamp
    p.dampratio = 1.0, p.dmin = x.dmin, p.dmax = x.dmax, p.width = x.width;
    p.limK = 1.0 / (p.dmax * p.dmax * p.tc * p.tc * p.dampratio * p.dampratio), p.limB = 2.0 / (p.dmax * p.tc);
    return p;
}

// ---------------------------------------------------------------------------------------------
extern "C" EMEI_API int emei_create(const emei_config* cfg, emei_env** out) {
    if (!cfg || !out) return fail(EMEI_ERR_INVALID, "emei_create: null argument");
    // older callers pass a shorter struct: every tail field has an all-zero default
    const uint32_t size_v2a = (uint32_t)offsetof(emei_config, env_param_mask);  // 328: before env_params existed
    const uint32_t size_v5 = (uint32_t)offsetof(emei_config, ode_method);       // 400: before ode_method / rollout_chunk_steps
    if (cfg->struct_size != sizeof(emei_config) && cfg->struct_size != size_v5 && cfg->struct_size != size_v2a &&
        cfg->struct_size != EMEI_CONFIG_SIZE_V1)
        return fail(EMEI_ERR_INVALID, "emei_create: emei_config size %u, library expects %zu (or the older sizes %u / %u / %u)",
                    cfg->struct_size, sizeof(emei_config), size_v5, size_v2a, EMEI_CONFIG_SIZE_V1);
    emei_config c2;  // the caller's struct, widened to this library's layout
    memset(&c2, 0, sizeof(c2));
    memcpy(&c2, cfg, cfg->struct_size);
    if (cfg->struct_size == EMEI_CONFIG_SIZE_V1)
        for (int i = 0; i < EMEI_MAX_STATE_DIM; ++i) c2.init_sigma[i] = (float)c2.init_noise;
    c2.struct_size = sizeof(emei_config);
    cfg = &c2;
    if (cfg->integrator < EMEI_INTEG_EULER || cfg->integrator > EMEI_INTEG_RK4)
        return fail(EMEI_ERR_UNSUPPORTED, "emei_create: integrator=%d", cfg->integrator);  // mujoco_env.py:78-79
    if (cfg->noise_layout != EMEI_NOISE_IID && cfg->noise_layout != EMEI_NOISE_SHARED)
        return fail(EMEI_ERR_INVALID, "emei_create: noise_layout=%d", cfg->noise_layout);
    for (int i = 0; i < EMEI_MAX_STATE_DIM; ++i)
        if (!(cfg->init_sigma[i] >= 0) || !(cfg->obs_sigma[i] >= 0))
            return fail(EMEI_ERR_INVALID, "emei_create: noise sigmas must be >= 0");
    if (cfg->ode_method != EMEI_ODE_EULER && cfg->ode_method != EMEI_ODE_RK4)
        return fail(EMEI_ERR_UNSUPPORTED, "emei_create: ode_method=%d", cfg->ode_method);  // base_control.py:171-172
    if (cfg->ode_method != EMEI_ODE_EULER && cfg->env_id != EMEI_CARTPOLE_SWINGUP && cfg->env_id != EMEI_CARTPOLE_BALANCING)
        return fail(EMEI_ERR_INVALID, "emei_create: ode_method is ODE_approximation's switch (classic control); env_id %d is integrated "
                    "according to `integrator`", cfg->env_id);
    if (cfg->rollout_chunk_steps < -1 && (cfg->rollout_chunk_steps > -101 || cfg->rollout_chunk_steps < -1506 ||
                                          -cfg->rollout_chunk_steps % 100 < 1 || -cfg->rollout_chunk_steps % 100 > 6)) return fail(EMEI_ERR_INVALID, "emei_create: rollout_chunk_steps=%d", cfg->rollout_chunk_steps);
    if (cfg->solver != EMEI_SOLVER_NEWTON && cfg->solver != EMEI_SOLVER_SWEEP1) return fail(EMEI_ERR_INVALID, "emei_create: solver=%u", cfg->solver);
    if (cfg->env_param_mask >> EMEI_MAX_ENV_PARAMS) return fail(EMEI_ERR_INVALID, "emei_create: env_param_mask=0x%x", cfg->env_param_mask);
    if (cfg->env_param_mask != 0 && cfg->env_id != EMEI_HALFCHEETAH_RUNNING && cfg->env_id != EMEI_HOPPER_RUNNING)
        return fail(EMEI_ERR_UNSUPPORTED, "emei_create: env_id %d takes no reward / health parameters", cfg->env_id);
    if (cfg->env_id == EMEI_HALFCHEETAH_RUNNING && (cfg->env_param_mask >> 2))
        return fail(EMEI_ERR_INVALID, "emei_create: HalfCheetahRunning has only the two reward weights");
    int od, ad, sd;
    if (emei_env_dims(cfg->env_id, &od, &ad, &sd) != EMEI_OK) return EMEI_ERR_INVALID;
    // the kernels index envs with uint32_t lane indices and int32 done lists: n_envs < 2^31
    if (cfg->n_envs <= 0 || cfg->n_envs >= (int64_t)1 << 31)
        return fail(EMEI_ERR_INVALID, "emei_create: n_envs=%lld out of range", (long long)cfg->n_envs);
    if (cfg->freq_rate < 1) return fail(EMEI_ERR_INVALID, "emei_create: freq_rate=%d < 1", cfg->freq_rate);
    if (!(cfg->real_time_scale > 0)) return fail(EMEI_ERR_INVALID, "emei_create: real_time_scale must be > 0");
    if (cfg->precision != EMEI_PRECISION_REF && cfg->precision != EMEI_PRECISION_F32)
        return fail(EMEI_ERR_INVALID, "emei_create: precision=%d", cfg->precision);
    if (cfg->max_episode_steps < 0) return fail(EMEI_ERR_INVALID, "emei_create: max_episode_steps < 0");
    int prev_device = 0;
    HIP_TRY(hipGetDevice(&prev_device));
    HIP_TRY(hipSetDevice(cfg->device));
    struct RestoreDevice {  // emei_create leaves the caller's current device as it found it
        int d;
        ~RestoreDevice() { (void)hipSetDevice(d); }
    } restore{prev_device};

    emei_env* h = new (std::nothrow) emei_env();
    if (!h) return fail(EMEI_ERR_INVALID, "emei_create: out of host memory");
    memset(h, 0, sizeof(*h));
    h->cfg = *cfg;
    h->obs_dim = od, h->act_dim = ad, h->state_dim = sd;
    h->real_size = cfg->precision == EMEI_PRECISION_F32 ? sizeof(float) : sizeof(double);
    const size_t n = (size_t)cfg->n_envs, n_words = (n + kWave - 1) / kWave;
    const size_t state_bytes = n * sd * h->real_size;
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMalloc(&h->state, state_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&h->steps, n * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&h->episode, n * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&h->done_mask, n_words * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(&h->frozen_state, state_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&h->frozen_steps, n * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&h->frozen_episode, n * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&h->cap_hits, sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(h->cap_hits, 0, sizeof(unsigned long long));
    if (e == hipSuccess && steps_as_body(*cfg)) {
        // (2 + waves words of the queue; a second `waves` words are scratch of probe builds: body_kernels.h, EMEI_CLOCK_HIST_CORR)
        e = hipMalloc((void**)&h->work, (2 + 2 * n_words) * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemset(h->work, 0, (2 + 2 * n_words) * sizeof(uint32_t));
    }
    if (e == hipSuccess) e = hipMemset(h->done_mask, 0, n_words * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(h->steps, 0, n * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(h->episode, 0, n * sizeof(uint32_t));
    if (e != hipSuccess) {
        emei_destroy(h);
        return fail(EMEI_ERR_HIP, "emei_create: allocation failed: %s", hipGetErrorString(e));
    }
    if (is_pend(cfg->env_id))
        h->pend = pend_params(cfg->env_id, cfg->real_time_scale, cfg->init_sigma, cfg->noise_layout);
    h->trig = emei_trig_table(cfg->device);
    if (!h->trig) {
        emei_destroy(h);
        return fail(EMEI_ERR_HIP, "emei_create: could not build the trig table on device %d", cfg->device);
    }
    if (steps_as_body(*cfg)) {  // occupancy of the rollout kernel: the automatic chunking policy (body_kernels.h:WorkQueue)
        BodyLaunch L;
        L.op = BODY_OP_OCCUPANCY, L.env_id = cfg->env_id, L.precision = cfg->precision, L.integrator = cfg->integrator;
        L.solver = (int)cfg->solver, L.selected = &h->resident_waves;
        if (body_launch(L) != EMEI_OK) h->resident_waves = 0;  // unknown: the automatic policy then stays with one-piece launches
    }
    *out = h;
    return EMEI_OK;
}

extern "C" EMEI_API int emei_destroy(emei_env* h) {
    if (!h) return EMEI_OK;
    (void)hipFree(h->state);
    (void)hipFree(h->steps);
    (void)hipFree(h->episode);
    (void)hipFree(h->done_mask);
    (void)hipFree(h->frozen_state);
    (void)hipFree(h->frozen_steps);
    (void)hipFree(h->frozen_episode);
    (void)hipFree(h->cap_hits);
    (void)hipFree(h->work);
    if (h->host_flag) (void)hipHostFree(h->host_flag);
    delete h;
    return EMEI_OK;
}

// One handle <-> one device: every launch of a handle goes to the CALLER's current device, so a handle used
// while another device is current would touch memory of the wrong GPU.  Refused up front.
static int wrong_device(const emei_env* h, const char* fn) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return fail(EMEI_ERR_HIP, "%s: hipGetDevice failed", fn);
    if (dev != h->cfg.device)
        return fail(EMEI_ERR_INVALID, "%s: the handle belongs to device %d but device %d is current", fn, h->cfg.device, dev);
    return EMEI_OK;
}
#define EMEI_ON_DEVICE(h, fn)                                  \
    do {                                                        \
        if (int rc_ = wrong_device((h), (fn))) return rc_;      \
    } while (0)

static PendLaunch pend_base(emei_env* h, void* stream) {
    PendLaunch L;
    L.env_id = h->cfg.env_id;
    L.precision = h->cfg.precision;
    L.ode_method = h->cfg.ode_method;
    L.state = h->state;
    L.steps = h->steps;
    L.episode = h->episode;
    L.done_mask = h->done_mask;
    L.n = h->cfg.n_envs;
    L.freq_rate = h->cfg.freq_rate;
    L.max_episode_steps = h->cfg.max_episode_steps;
    L.seed = h->cfg.seed;
    L.env_offset = h->cfg.env_index_offset;
    L.p = h->pend;
    L.trig = h->trig;
    L.stream = (hipStream_t)stream;
    L.peers = h->peers;
    return L;
}

static BodyLaunch body_base(emei_env* h, void* stream) {
    BodyLaunch L;
    L.env_id = h->cfg.env_id;
    L.precision = h->cfg.precision;
    L.state = h->state;
    L.steps = h->steps;
    L.episode = h->episode;
    L.done_mask = h->done_mask;
    L.n = h->cfg.n_envs;
    L.freq_rate = h->cfg.freq_rate;
    L.max_episode_steps = h->cfg.max_episode_steps;
    L.seed = h->cfg.seed;
    L.env_offset = h->cfg.env_index_offset;
    L.dt = h->cfg.real_time_scale;
    L.integrator = h->cfg.integrator;
    L.solver = (int)h->cfg.solver;
    memcpy(L.noise.init, h->cfg.init_sigma, sizeof(L.noise.init));
    memcpy(L.noise.obs, h->cfg.obs_sigma, sizeof(L.noise.obs));
    L.noise.shared = h->cfg.noise_layout == EMEI_NOISE_SHARED;
    L.env_params.mask = h->cfg.env_param_mask;
    memcpy(L.env_params.v, h->cfg.env_params, sizeof(L.env_params.v));
    L.trig = h->trig;
    L.cap_hits = h->cap_hits;
    L.work = h->work, L.chunk_steps = h->cfg.rollout_chunk_steps, L.resident_waves = h->resident_waves;
    L.stream = (hipStream_t)stream;
    return L;
}

extern "C" EMEI_API int emei_reset(emei_env* h, uint64_t seed, void* stream) {
    if (!h) return fail(EMEI_ERR_INVALID, "emei_reset: null handle");
    EMEI_ON_DEVICE(h, "emei_reset");
    h->cfg.seed = seed;
    int rc;
    if (is_pend(h->cfg.env_id)) {
        PendLaunch L = pend_base(h, stream);
        L.op = PEND_OP_RESET;
        rc = pend_launch(L);
    } else {
        BodyLaunch L = body_base(h, stream);
        L.op = BODY_OP_RESET;
        rc = body_launch(L);
    }
    if (rc != EMEI_OK) return fail(rc, "emei_reset: launch failed");
    h->has_state = true;
    return EMEI_OK;
}

extern "C" EMEI_API int emei_set_seed(emei_env* h, uint64_t seed) {
    if (!h) return fail(EMEI_ERR_INVALID, "emei_set_seed: null handle");
    h->cfg.seed = seed;
    return EMEI_OK;
}

extern "C" EMEI_API int emei_last_rollout_kernel(emei_env* h) {
    if (!h) return fail(EMEI_ERR_INVALID, "emei_last_rollout_kernel: null handle");
    return h->last_kernel;
}

extern "C" EMEI_API int emei_set_state(emei_env* h, const double* state_aos, int reset_counters, void* stream) {
    if (!h || !state_aos) return fail(EMEI_ERR_INVALID, "emei_set_state: null argument");
    EMEI_ON_DEVICE(h, "emei_set_state");
    int rc = launch_state_unpack(state_aos, h->state, h->cfg.precision, h->cfg.n_envs, h->state_dim, (hipStream_t)stream);
    if (rc != EMEI_OK) return fail(rc, "emei_set_state: launch failed");
    if (reset_counters) {
        HIP_TRY(hipMemsetAsync(h->steps, 0, (size_t)h->cfg.n_envs * sizeof(int32_t), (hipStream_t)stream));
        HIP_TRY(hipMemsetAsync(h->episode, 0, (size_t)h->cfg.n_envs * sizeof(uint32_t), (hipStream_t)stream));
    }
    h->has_state = true;
    return EMEI_OK;
}

extern "C" EMEI_API int emei_get_state(emei_env* h, double* state_aos, void* stream) {
    if (!h || !state_aos) return fail(EMEI_ERR_INVALID, "emei_get_state: null argument");
    EMEI_ON_DEVICE(h, "emei_get_state");
    if (!h->has_state) return fail(EMEI_ERR_STATE, "emei_get_state: call reset before using the state");
    int rc = launch_state_pack(h->state, state_aos, h->cfg.precision, h->cfg.n_envs, h->state_dim, (hipStream_t)stream);
    return rc == EMEI_OK ? rc : fail(rc, "emei_get_state: launch failed");
}

extern "C" EMEI_API int emei_get_obs(emei_env* h, double* obs_aos, void* stream) {
    if (!h || !obs_aos) return fail(EMEI_ERR_INVALID, "emei_get_obs: null argument");
    EMEI_ON_DEVICE(h, "emei_get_obs");
    if (!h->has_state) return fail(EMEI_ERR_STATE, "emei_get_obs: call reset before using the state");
    if (is_pend(h->cfg.env_id)) {
        PendLaunch L = pend_base(h, stream);
        L.op = PEND_OP_GET_OBS;
        L.obs_f64 = obs_aos;
        int rc = pend_launch(L);
        return rc == EMEI_OK ? rc : fail(rc, "emei_get_obs: launch failed");
    }
    BodyLaunch L = body_base(h, stream);
    L.op = BODY_OP_GET_OBS;
    L.obs_f64 = obs_aos;
    int rc = body_launch(L);
    return rc == EMEI_OK ? rc : fail(rc, "emei_get_obs: launch failed");
}

extern "C" EMEI_API int emei_freeze(emei_env* h, void* stream) {
    if (!h) return fail(EMEI_ERR_INVALID, "emei_freeze: null handle");
    EMEI_ON_DEVICE(h, "emei_freeze");
    if (!h->has_state) return fail(EMEI_ERR_STATE, "emei_freeze: no state to freeze (call reset first)");
    const size_t n = (size_t)h->cfg.n_envs;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(h->frozen_state, h->state, n * h->state_dim * h->real_size, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->frozen_steps, h->steps, n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->frozen_episode, h->episode, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    h->frozen_seed = h->cfg.seed;  // a reset(seed=) between freeze and unfreeze must not re-key the restored episodes
    h->frozen = true;
    return EMEI_OK;
}

extern "C" EMEI_API int emei_unfreeze(emei_env* h, void* stream) {
    if (!h) return fail(EMEI_ERR_INVALID, "emei_unfreeze: null handle");
    EMEI_ON_DEVICE(h, "emei_unfreeze");
    if (!h->frozen) return fail(EMEI_ERR_STATE, "emei_unfreeze: env has not been frozen");
    const size_t n = (size_t)h->cfg.n_envs;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(h->state, h->frozen_state, n * h->state_dim * h->real_size, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->steps, h->frozen_steps, n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->episode, h->frozen_episode, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    h->cfg.seed = h->frozen_seed;
    h->frozen = false;
    return EMEI_OK;
}

extern "C" EMEI_API int emei_get_solver_cap_hits(emei_env* h, uint64_t* count_out, void* stream) {
    if (!h || !count_out) return fail(EMEI_ERR_INVALID, "emei_get_solver_cap_hits: null argument");
    EMEI_ON_DEVICE(h, "emei_get_solver_cap_hits");
    HIP_TRY(hipMemcpyAsync(count_out, h->cap_hits, sizeof(uint64_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return EMEI_OK;
}

extern "C" EMEI_API int emei_get_rollout_faults(emei_env* h, uint64_t* count_out, void* stream) {
    if (!h || !count_out) return fail(EMEI_ERR_INVALID, "emei_get_rollout_faults: null argument");
    EMEI_ON_DEVICE(h, "emei_get_rollout_faults");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(count_out, 0, sizeof(uint64_t), s));
    if (h->work) HIP_TRY(hipMemcpyAsync(count_out, h->work + WorkQueue::kFaults, sizeof(uint32_t), hipMemcpyDeviceToDevice, s));  // little endian
    return EMEI_OK;
}

extern "C" EMEI_API int emei_set_obs_peers(emei_env* h, int n_peers, void* const* peer_obs, int64_t row_envs, int64_t col_offset, int32_t max_steps) {
    if (!h) return fail(EMEI_ERR_INVALID, "emei_set_obs_peers: null handle");
    if (n_peers == 0) {
        h->peers = ObsPeers();
        return EMEI_OK;
    }
    if (n_peers < 0 || n_peers > EMEI_MAX_OBS_PEERS) return fail(EMEI_ERR_INVALID, "emei_set_obs_peers: n_peers=%d outside 0..%d", n_peers, EMEI_MAX_OBS_PEERS);
    if (!peer_obs) return fail(EMEI_ERR_INVALID, "emei_set_obs_peers: null pointer list");
    if (max_steps < 1) return fail(EMEI_ERR_INVALID, "emei_set_obs_peers: max_steps=%d < 1", max_steps);
    if (col_offset < 0 || row_envs < col_offset + h->cfg.n_envs)
        return fail(EMEI_ERR_INVALID, "emei_set_obs_peers: columns [%lld, %lld) do not fit a row of %lld envs", (long long)col_offset,
                    (long long)(col_offset + h->cfg.n_envs), (long long)row_envs);
    if (steps_as_body(h->cfg) || h->obs_dim != 4 || h->cfg.env_id > EMEI_CARTPOLE_BALANCING)
        return fail(EMEI_ERR_UNSUPPORTED, "emei_set_obs_peers: built for the CartPole family (env_id %d)", h->cfg.env_id);
    ObsPeers P;
    for (int p = 0; p < n_peers; ++p) {
        if (!peer_obs[p] || ((uintptr_t)peer_obs[p] & 15u)) return fail(EMEI_ERR_INVALID, "emei_set_obs_peers: peer %d is null or not 16-byte aligned", p);
        P.obs[p] = peer_obs[p];
    }
    P.row_envs = row_envs, P.col = col_offset, P.count = n_peers, P.max_steps = max_steps;
    h->peers = P;
    return EMEI_OK;
}

// Device memory another process of this node can map (hipIpc): see include/emei_hip.h
static_assert(sizeof(hipIpcMemHandle_t) == sizeof(emei_ipc_handle), "hipIpcMemHandle_t is 64 opaque bytes");
struct DeviceScope {  // the calls below act on the caller's device and leave the current device as they found it
    int prev = -1;
    hipError_t err;
    explicit DeviceScope(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) err = hipSetDevice(device);
    }
    ~DeviceScope() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

extern "C" EMEI_API int emei_peer_buffer_create(int device, uint64_t bytes, void** dev_ptr_out, emei_ipc_handle* handle_out) {
    if (!dev_ptr_out || !handle_out || bytes == 0) return fail(EMEI_ERR_INVALID, "emei_peer_buffer_create: bad argument");
    DeviceScope scope(device);
    HIP_TRY(scope.err);
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    hipIpcMemHandle_t hnd;
    hipError_t e = hipIpcGetMemHandle(&hnd, p);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return fail(EMEI_ERR_HIP, "hipIpcGetMemHandle: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 is needed where the host driver only has dmabuf IPC)", hipGetErrorString(e));
    }
    memcpy(handle_out->bytes, &hnd, sizeof(hnd));
    *dev_ptr_out = p;
    return EMEI_OK;
}

extern "C" EMEI_API int emei_peer_buffer_open(int device, const emei_ipc_handle* handle, void** dev_ptr_out) {
    if (!handle || !dev_ptr_out) return fail(EMEI_ERR_INVALID, "emei_peer_buffer_open: null argument");
    DeviceScope scope(device);
    HIP_TRY(scope.err);
    hipIpcMemHandle_t hnd;
    memcpy(&hnd, handle->bytes, sizeof(hnd));
    void* p = nullptr;
    HIP_TRY(hipIpcOpenMemHandle(&p, hnd, hipIpcMemLazyEnablePeerAccess));
    *dev_ptr_out = p;
    return EMEI_OK;
}

extern "C" EMEI_API int emei_peer_buffer_close(int device, void* dev_ptr) {
    if (!dev_ptr) return fail(EMEI_ERR_INVALID, "emei_peer_buffer_close: null pointer");
    DeviceScope scope(device);
    HIP_TRY(scope.err);
    HIP_TRY(hipIpcCloseMemHandle(dev_ptr));
    return EMEI_OK;
}

extern "C" EMEI_API int emei_peer_buffer_destroy(int device, void* dev_ptr) {
    if (!dev_ptr) return fail(EMEI_ERR_INVALID, "emei_peer_buffer_destroy: null pointer");
    DeviceScope scope(device);
    HIP_TRY(scope.err);
    HIP_TRY(hipFree(dev_ptr));
    return EMEI_OK;
}

static int check_action_dtype(const emei_env* h, int action_dtype) {
    if (action_dtype < EMEI_ACT_U8 || action_dtype > EMEI_ACT_F32) return fail(EMEI_ERR_INVALID, "bad action_dtype %d", action_dtype);
    int od, ad, sd;
    emei_env_dims(h->cfg.env_id, &od, &ad, &sd);
    if (ad > 0 && action_dtype != EMEI_ACT_F32)
        return fail(EMEI_ERR_INVALID, "continuous-action envs take float32 actions [n,%d]", ad);
    if (ad == 0 && action_dtype == EMEI_ACT_F32)
        return fail(EMEI_ERR_INVALID, "discrete-action envs take uint8/int32/int64 actions [n]");
    return EMEI_OK;
}

extern "C" EMEI_API int emei_rollout(emei_env* h, int32_t n_steps, const void* actions, int action_dtype, float* obs_out,
                            float* reward_out, uint8_t* done_out, uint32_t flags, void* stream) {
    if (!h || !actions) return fail(EMEI_ERR_INVALID, "emei_rollout: null argument");
    EMEI_ON_DEVICE(h, "emei_rollout");
    if (n_steps < 1) return fail(EMEI_ERR_INVALID, "emei_rollout: n_steps=%d < 1", n_steps);
    if (!h->has_state) return fail(EMEI_ERR_STATE, "Call reset before using step method.");  // base_control.py:67
    if (flags & ~EMEI_FLAG_AUTO_RESET) return fail(EMEI_ERR_INVALID, "emei_rollout: unknown flags 0x%x", flags);
    if (check_action_dtype(h, action_dtype) != EMEI_OK) return EMEI_ERR_INVALID;
    int rc;
    if (h->peers.count > 0 && n_steps > h->peers.max_steps)
        return fail(EMEI_ERR_INVALID, "emei_rollout: n_steps=%d but the registered observation peers hold %d rows (emei_set_obs_peers)", n_steps,
                    h->peers.max_steps);
    if (!steps_as_body(h->cfg)) {
        PendLaunch L = pend_base(h, stream);
        L.op = PEND_OP_ROLLOUT;
        L.actions = actions;
        L.action_dtype = action_dtype;
        L.obs_out = obs_out;
        L.reward_out = reward_out;
        L.done_out = done_out;
        L.n_steps = n_steps;
        L.flags = flags;
        L.selected = &h->last_kernel;
        rc = pend_launch(L);
        if (rc == EMEI_ERR_UNSUPPORTED && h->peers.count > 0)
            return fail(rc, "emei_rollout: observation peers are set (emei_set_obs_peers), which only the staged kernel of the CartPole family "
                            "serves: n_envs a multiple of 64, n_steps >= 16, all three outputs, 16-byte aligned buffers (n_envs=%lld, n_steps=%d)",
                        (long long)h->cfg.n_envs, n_steps);
    } else {
        if (h->peers.count > 0)
            return fail(EMEI_ERR_UNSUPPORTED, "emei_rollout: observation peers are set (emei_set_obs_peers), which the body kernels do not serve");
        BodyLaunch L = body_base(h, stream);
        L.op = BODY_OP_ROLLOUT;
        L.actions = actions;
        L.obs_out = obs_out;
        L.reward_out = reward_out;
        L.done_out = done_out;
        L.n_steps = n_steps;
        L.flags = flags;
        L.selected = &h->last_kernel;
        rc = body_launch(L);
    }
    return rc == EMEI_OK ? rc : fail(rc, "emei_rollout: launch failed (%s)", hipGetErrorString(hipGetLastError()));
}

extern "C" EMEI_API int emei_step_host(emei_env* h, const void* actions_host, int action_dtype, double* obs64_host, float* obs32_host,
                                       float* reward_host, uint8_t* done_host, uint32_t flags, void* stream) {
    if (!h || !actions_host || !obs64_host || !obs32_host || !reward_host || !done_host)
        return fail(EMEI_ERR_INVALID, "emei_step_host: null argument");
    EMEI_ON_DEVICE(h, "emei_step_host");
    if (!h->has_state) return fail(EMEI_ERR_STATE, "Call reset before using step method.");  // base_control.py:67
    if (flags & ~EMEI_FLAG_AUTO_RESET) return fail(EMEI_ERR_INVALID, "emei_step_host: unknown flags 0x%x", flags);
    if (check_action_dtype(h, action_dtype) != EMEI_OK) return EMEI_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (!steps_as_body(h->cfg) && h->cfg.n_envs == 1) {
        // ONE launch: the generic rollout kernel also writes the float64 observation and ends with a completion word
        if (!h->host_flag) {
            HIP_TRY(hipHostMalloc((void**)&h->host_flag, 64, hipHostMallocDefault));
            *h->host_flag = 0;
        }
        PendLaunch L = pend_base(h, stream);
        L.op = PEND_OP_ROLLOUT;
        L.actions = actions_host;
        L.action_dtype = action_dtype;
        L.obs_out = obs32_host, L.reward_out = reward_host, L.done_out = done_host, L.obs_f64 = obs64_host;
        L.n_steps = 1;
        L.flags = flags;
        L.selected = &h->last_kernel;
        L.host_flag = h->host_flag;
        L.flag_value = ++h->flag_seq;
        if (L.flag_value == 0) L.flag_value = ++h->flag_seq;  // 0 is the word's initial content
        int rc = pend_launch(L);
        if (rc != EMEI_OK) return fail(rc, "emei_step_host: launch failed (%s)", hipGetErrorString(hipGetLastError()));
        // poll the word (the kernel's last store, system scope); a launch that does not finish within ~5 ms is handed to the
        // runtime's own wait, which also surfaces a device fault
        volatile uint32_t* flag = h->host_flag;
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int spin = 0;; ++spin) {
            if (*flag == L.flag_value) {
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                return EMEI_OK;
            }
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#else
            __asm__ __volatile__("" ::: "memory");
#endif
            if ((spin & 1023) == 1023) {  // the bound is wall time (a pause is 40-140 cycles depending on the part), checked every 1024 spins
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec) > 5000000ll) break;
            }
        }
        HIP_TRY(hipStreamSynchronize(s));
        return *flag == L.flag_value ? EMEI_OK : fail(EMEI_ERR_HIP, "emei_step_host: the kernel finished without its completion word");
    }
    int rc = emei_rollout(h, 1, actions_host, action_dtype, obs32_host, reward_host, done_host, flags, stream);
    if (rc != EMEI_OK) return rc;
    rc = emei_get_obs(h, obs64_host, stream);
    if (rc != EMEI_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    return EMEI_OK;
}

extern "C" EMEI_API int emei_step(emei_env* h, const void* actions, int action_dtype, float* obs_out, float* reward_out,
                         uint8_t* done_out, uint32_t flags, void* stream) {
    return emei_rollout(h, 1, actions, action_dtype, obs_out, reward_out, done_out, flags, stream);
}

extern "C" EMEI_API int emei_compact_done(emei_env* h, int32_t* idx_out, int32_t* count_out, void* stream) {
    if (!h || !idx_out || !count_out) return fail(EMEI_ERR_INVALID, "emei_compact_done: null argument");
    EMEI_ON_DEVICE(h, "emei_compact_done");
    int rc = launch_compact_done(h->done_mask, h->cfg.n_envs, idx_out, count_out, (hipStream_t)stream);
    return rc == EMEI_OK ? rc : fail(rc, "emei_compact_done: launch failed");
}

extern "C" EMEI_API int emei_get_counters(emei_env* h, int32_t* steps_out, uint32_t* episode_out, void* stream) {
    if (!h) return fail(EMEI_ERR_INVALID, "emei_get_counters: null handle");
    EMEI_ON_DEVICE(h, "emei_get_counters");
    const size_t n = (size_t)h->cfg.n_envs;
    hipStream_t s = (hipStream_t)stream;
    if (steps_out) HIP_TRY(hipMemcpyAsync(steps_out, h->steps, n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    if (episode_out) HIP_TRY(hipMemcpyAsync(episode_out, h->episode, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    return EMEI_OK;
}

extern "C" EMEI_API int emei_episode_init_obs(emei_env* h, int64_t count, const int64_t* env_index, const uint32_t* episode,
                                              float* obs_out, void* stream) {
    if (!h || !env_index || !episode || !obs_out) return fail(EMEI_ERR_INVALID, "emei_episode_init_obs: null argument");
    EMEI_ON_DEVICE(h, "emei_episode_init_obs");
    if (count <= 0) return fail(EMEI_ERR_INVALID, "emei_episode_init_obs: count=%lld", (long long)count);
    int rc;
    if (is_pend(h->cfg.env_id)) {
        PendLaunch L = pend_base(h, stream);
        L.op = PEND_OP_INIT_OBS;
        L.n = count;
        L.env_index = env_index;
        L.episode_in = episode;
        L.obs_out = obs_out;
        rc = pend_launch(L);
    } else {
        BodyLaunch L = body_base(h, stream);
        L.op = BODY_OP_INIT_OBS;
        L.n = count;
        L.env_index = env_index;
        L.episode_in = episode;
        L.obs_out = obs_out;
        rc = body_launch(L);
    }
    return rc == EMEI_OK ? rc : fail(rc, "emei_episode_init_obs: launch failed");
}

// ---------------------------------------------------------------------------------------------
static int fill_env_params(int env_id, uint32_t mask, const double* params, EnvParams& ep, const char* fn) {
    if (mask == 0) return EMEI_OK;
    if (!params || (mask >> EMEI_MAX_ENV_PARAMS)) return fail(EMEI_ERR_INVALID, "%s: bad env_param_mask 0x%x / params", fn, mask);
    if (env_id != EMEI_HALFCHEETAH_RUNNING && env_id != EMEI_HOPPER_RUNNING)
        return fail(EMEI_ERR_UNSUPPORTED, "%s: env_id %d takes no reward / health parameters", fn, env_id);
    ep.mask = mask;
    memcpy(ep.v, params, sizeof(ep.v));
    return EMEI_OK;
}

static bool bad_io(int io_dtype) { return io_dtype != EMEI_IO_F32 && io_dtype != EMEI_IO_F64; }
static bool misaligned16(const void* p) { return ((uintptr_t)p & 15u) != 0; }

extern "C" EMEI_API int emei_reward_io(int env_id, int64_t n, int io_dtype, const void* obs, const void* pre_obs, const void* action,
                              double real_time_scale, int32_t freq_rate, uint32_t env_param_mask, const double* env_params,
                              uint32_t flags, void* reward_out, void* stream) {
    if (n <= 0 || !obs || !reward_out) return fail(EMEI_ERR_INVALID, "emei_reward: bad argument");
    if (bad_io(io_dtype)) return fail(EMEI_ERR_INVALID, "emei_reward: io_dtype=%d", io_dtype);
    if (flags & ~EMEI_REWARD_BATCH_CTRL_COST) return fail(EMEI_ERR_INVALID, "emei_reward: unknown flags 0x%x", flags);
    if ((flags & EMEI_REWARD_BATCH_CTRL_COST) && env_id != EMEI_HALFCHEETAH_RUNNING && env_id != EMEI_HOPPER_RUNNING)
        return fail(EMEI_ERR_UNSUPPORTED, "emei_reward: env_id %d has no control-cost term to sum over the batch", env_id);
    if (is_pend(env_id)) {
        if (misaligned16(obs)) return fail(EMEI_ERR_INVALID, "emei_reward: obs must be 16-byte aligned");
        PendLaunch L;
        L.op = PEND_OP_REWARD_TERMINAL;
        L.env_id = env_id;
        L.precision = EMEI_PRECISION_REF;  // evaluated in float64 whatever the row dtype
        L.obs_in = obs;
        L.io_f64 = io_dtype == EMEI_IO_F64;
        L.reward_out = (float*)reward_out;
        L.n = n;
        L.p = pend_params(env_id, real_time_scale > 0 ? real_time_scale : 0.02);
        L.trig = current_device_trig();
        L.stream = (hipStream_t)stream;
        int rc = pend_launch(L);
        return rc == EMEI_OK ? rc : fail(rc, "emei_reward: launch failed");
    }
    if (is_body(env_id)) {
        if ((env_id == EMEI_HALFCHEETAH_RUNNING || env_id == EMEI_HOPPER_RUNNING) && (!pre_obs || !action))
            return fail(EMEI_ERR_INVALID, "emei_reward: this env's reward needs pre_obs and action");
        if (!(real_time_scale > 0) || freq_rate < 1) return fail(EMEI_ERR_INVALID, "emei_reward: bad dt/freq_rate");
        BodyLaunch L;
        if (int rc_ = fill_env_params(env_id, env_param_mask, env_params, L.env_params, "emei_reward")) return rc_;
        L.op = BODY_OP_REWARD;
        L.env_id = env_id;
        L.precision = EMEI_PRECISION_REF;
        L.obs_in = obs, L.pre_obs_in = pre_obs, L.actions = action;
        L.io_f64 = io_dtype == EMEI_IO_F64;
        L.reward_out = (float*)reward_out;
        L.n = n;
        L.freq_rate = freq_rate;
        L.dt = real_time_scale;
        L.stream = (hipStream_t)stream;
        if (flags & EMEI_REWARD_BATCH_CTRL_COST)  // 8 B for the whole-batch sum, allocated and freed in stream order
            HIP_TRY(hipMallocAsync((void**)&L.batch_cost_scratch, sizeof(double), L.stream));
        int rc = body_launch(L);
        if (L.batch_cost_scratch) (void)hipFreeAsync(L.batch_cost_scratch, L.stream);
        return rc == EMEI_OK ? rc : fail(rc, "emei_reward: launch failed");
    }
    return fail(EMEI_ERR_INVALID, "emei_reward: unknown env_id %d", env_id);
}

extern "C" EMEI_API int emei_reward_ex(int env_id, int64_t n, const float* obs, const float* pre_obs, const float* action,
                              double real_time_scale, int32_t freq_rate, uint32_t env_param_mask, const double* env_params,
                              float* reward_out, void* stream) {
    return emei_reward_io(env_id, n, EMEI_IO_F32, obs, pre_obs, action, real_time_scale, freq_rate, env_param_mask, env_params, 0u,
                          reward_out, stream);
}

extern "C" EMEI_API int emei_reward(int env_id, int64_t n, const float* obs, const float* pre_obs, const float* action,
                           double real_time_scale, int32_t freq_rate, float* reward_out, void* stream) {
    return emei_reward_io(env_id, n, EMEI_IO_F32, obs, pre_obs, action, real_time_scale, freq_rate, 0u, nullptr, 0u, reward_out, stream);
}

extern "C" EMEI_API int emei_terminal_io(int env_id, int64_t n, int io_dtype, const void* obs, uint32_t env_param_mask,
                                const double* env_params, uint8_t* terminal_out, void* stream) {
    if (n <= 0 || !obs || !terminal_out) return fail(EMEI_ERR_INVALID, "emei_terminal: bad argument");
    if (bad_io(io_dtype)) return fail(EMEI_ERR_INVALID, "emei_terminal: io_dtype=%d", io_dtype);
    if (is_pend(env_id)) {
        if (misaligned16(obs)) return fail(EMEI_ERR_INVALID, "emei_terminal: obs must be 16-byte aligned");
        PendLaunch L;
        L.op = PEND_OP_REWARD_TERMINAL;
        L.env_id = env_id;
        L.precision = EMEI_PRECISION_REF;
        L.obs_in = obs;
        L.io_f64 = io_dtype == EMEI_IO_F64;
        L.done_out = terminal_out;
        L.n = n;
        L.p = pend_params(env_id, 0.02);
        L.trig = current_device_trig();
        L.stream = (hipStream_t)stream;
        int rc = pend_launch(L);
        return rc == EMEI_OK ? rc : fail(rc, "emei_terminal: launch failed");
    }
    if (is_body(env_id)) {
        BodyLaunch L;
        if (int rc_ = fill_env_params(env_id, env_param_mask, env_params, L.env_params, "emei_terminal")) return rc_;
        L.op = BODY_OP_TERMINAL;
        L.env_id = env_id;
        L.precision = EMEI_PRECISION_REF;
        L.obs_in = obs;
        L.io_f64 = io_dtype == EMEI_IO_F64;
        L.done_out = terminal_out;
        L.n = n;
        L.stream = (hipStream_t)stream;
        int rc = body_launch(L);
        return rc == EMEI_OK ? rc : fail(rc, "emei_terminal: launch failed");
    }
    return fail(EMEI_ERR_INVALID, "emei_terminal: unknown env_id %d", env_id);
}

extern "C" EMEI_API int emei_terminal_ex(int env_id, int64_t n, const float* obs, uint32_t env_param_mask,
                                const double* env_params, uint8_t* terminal_out, void* stream) {
    return emei_terminal_io(env_id, n, EMEI_IO_F32, obs, env_param_mask, env_params, terminal_out, stream);
}

extern "C" EMEI_API int emei_terminal(int env_id, int64_t n, const float* obs, uint8_t* terminal_out, void* stream) {
    return emei_terminal_io(env_id, n, EMEI_IO_F32, obs, 0u, nullptr, terminal_out, stream);
}

extern "C" EMEI_API int emei_next_obs_io(int env_id, int64_t n, int io_dtype, const void* obs, const void* actions, int action_dtype,
                                double real_time_scale, int32_t freq_rate, int32_t precision, int32_t integrator,
                                void* next_obs_out, void* stream) {
    if (n <= 0 || !obs || !actions || !next_obs_out) return fail(EMEI_ERR_INVALID, "emei_next_obs: bad argument");
    if (bad_io(io_dtype)) return fail(EMEI_ERR_INVALID, "emei_next_obs: io_dtype=%d", io_dtype);
    if (!(real_time_scale > 0) || freq_rate < 1) return fail(EMEI_ERR_INVALID, "emei_next_obs: bad dt/freq_rate");
    if (action_dtype < EMEI_ACT_U8 || action_dtype > EMEI_ACT_F32) return fail(EMEI_ERR_INVALID, "bad action_dtype");
    if (precision != EMEI_PRECISION_REF && precision != EMEI_PRECISION_F32) return fail(EMEI_ERR_INVALID, "emei_next_obs: precision=%d", precision);
    const int ode_method = (integrator & EMEI_NEXT_OBS_ODE_RK4) ? EMEI_ODE_RK4 : EMEI_ODE_EULER;  // classic control only
    integrator &= ~EMEI_NEXT_OBS_ODE_RK4;
    if (integrator < EMEI_INTEG_EULER || integrator > EMEI_INTEG_RK4)
        return fail(EMEI_ERR_UNSUPPORTED, "emei_next_obs: integrator=%d", integrator);
    int od, ad, sd;
    if (emei_env_dims(env_id, &od, &ad, &sd) != EMEI_OK) return EMEI_ERR_INVALID;
    const bool ip_as_body = is_ip(env_id) && integrator != EMEI_INTEG_EULER;
    if (is_pend(env_id) && !ip_as_body) {
        if (misaligned16(obs) || misaligned16(next_obs_out)) return fail(EMEI_ERR_INVALID, "emei_next_obs: obs / next_obs_out must be 16-byte aligned");
        PendLaunch L;
        L.op = PEND_OP_NEXT_OBS;
        L.env_id = env_id;
        L.precision = precision;
        L.ode_method = ode_method;
        L.obs_in = obs;
        L.io_f64 = io_dtype == EMEI_IO_F64;
        L.actions = actions;
        L.action_dtype = action_dtype;
        L.obs_out = (float*)next_obs_out;
        L.n = n;
        L.freq_rate = freq_rate;
        L.p = pend_params(env_id, real_time_scale);
        L.trig = current_device_trig();
        L.stream = (hipStream_t)stream;
        int rc = pend_launch(L);
        return rc == EMEI_OK ? rc : fail(rc, "emei_next_obs: launch failed");
    }
    if (action_dtype != EMEI_ACT_F32) return fail(EMEI_ERR_INVALID, "continuous-action envs take float32 actions [n,%d]", ad);
    BodyLaunch L;
    L.op = BODY_OP_NEXT_OBS;
    L.env_id = env_id;
    L.precision = precision;
    L.obs_in = obs;
    L.io_f64 = io_dtype == EMEI_IO_F64;
    L.actions = actions;
    L.obs_out = (float*)next_obs_out;
    L.n = n;
    L.freq_rate = freq_rate;
    L.dt = real_time_scale;
    L.integrator = integrator;
    L.trig = current_device_trig();
    L.stream = (hipStream_t)stream;
    int rc = body_launch(L);
    if (rc == EMEI_ERR_UNSUPPORTED)
        return fail(rc, "emei_next_obs: the observation of env_id %d does not determine its state", env_id);  // core.py:190-193
    return rc == EMEI_OK ? rc : fail(rc, "emei_next_obs: launch failed");
}

extern "C" EMEI_API int emei_next_obs_ex(int env_id, int64_t n, const float* obs, const void* actions, int action_dtype,
                                double real_time_scale, int32_t freq_rate, int32_t precision, int32_t integrator,
                                float* next_obs_out, void* stream) {
    return emei_next_obs_io(env_id, n, EMEI_IO_F32, obs, actions, action_dtype, real_time_scale, freq_rate, precision, integrator,
                            next_obs_out, stream);
}

extern "C" EMEI_API int emei_next_obs(int env_id, int64_t n, const float* obs, const void* actions, int action_dtype,
                             double real_time_scale, int32_t freq_rate, int32_t precision, float* next_obs_out,
                             void* stream) {
    return emei_next_obs_io(env_id, n, EMEI_IO_F32, obs, actions, action_dtype, real_time_scale, freq_rate, precision, EMEI_INTEG_EULER,
                            next_obs_out, stream);
}
