/*
 * emei_hip.h — C ABI of libemei_hip.so, the MI355X (gfx950) env-step engine.
 *
 * The reference (polixir/emei) has no FFI boundary: its "operator API" for this path is the Python
 * class surface.  Each entry point below names the reference interface it replaces (file:line under
 * /root/reference).  All pointers are caller-owned DEVICE-ACCESSIBLE pointers (device memory, e.g.
 * torch.Tensor.data_ptr(), or pinned host memory mapped into the device address space), contiguous,
 * never retained past the call (one exception: the gathered buffers handed to emei_set_obs_peers stay registered, and are written by
 * every following rollout, until the list is replaced or cleared).  The library never allocates outputs, never
 * synchronises the stream and never throws across the boundary.  `stream` is a hipStream_t passed
 * as void* (NULL = the default stream).  One handle <-> one device: calls on a handle must be made with
 * that device current (checked: EMEI_ERR_INVALID otherwise; emei_create itself restores the caller's
 * current device).  A handle is not thread-safe, distinct handles are.
 *
 * Return value: 0 = ok, negative = error (EMEI_ERR_*); emei_last_error() gives a thread-local text.
 */
#ifndef EMEI_HIP_H
#define EMEI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history.  1: emei_config of 64 B.  2: integrator / noise layout / per-coordinate sigmas (328 B), env_params (400 B),
 * emei_set_seed, emei_last_rollout_kernel, emei_config.solver (the former reserved0; reserved words MUST be zero).
 * 3: the *_io stateless entry points (float64 observations), emei_freeze / emei_unfreeze snapshot the reset key,
 *    emei_model_constants, emei_get_solver_cap_hits.
 * 4: emei_model_invweights.
 * 5: emei_step_host (page-locked host values in and out; for n_envs == 1 of the 4-state family one launch whose last store is a
 *    completion word the library polls).
 * 6: emei_config.ode_method (classic control's ODE_approximation(method="rk4"), opt-in) and emei_config.rollout_chunk_steps
 *    (struct_size 408; a 400-byte caller gets euler / automatic), emei_get_rollout_faults, EMEI_NEXT_OBS_ODE_RK4.
 * 7: emei_set_obs_peers + emei_peer_buffer_create / open / close / destroy (multi-GPU observation return by peer writes from the
 *    rollout kernel), EMEI_KERNEL_PEND_STAGED_PEERS_FREQ1 / EMEI_KERNEL_PEND_STAGED_PEERS. */
#define EMEI_ABI_VERSION 7

#if defined(__GNUC__)
#define EMEI_API __attribute__((visibility("default")))
#else
#define EMEI_API
#endif

/* Environments (reference classes). */
enum emei_env_id {
    EMEI_CARTPOLE_SWINGUP = 0,      /* emei/envs/classic_control/cartpole.py:135-156 */
    EMEI_CARTPOLE_BALANCING = 1,    /* emei/envs/classic_control/cartpole.py:115-132 */
    EMEI_IP_REBOUND_BALANCING = 2,  /* emei/envs/mujoco/inverted_pendulum.py:52-79   */
    EMEI_IP_BOUNDARY_BALANCING = 3, /* emei/envs/mujoco/inverted_pendulum.py:82-111  */
    EMEI_IP_REBOUND_SWINGUP = 4,    /* emei/envs/mujoco/inverted_pendulum.py:114-146 */
    EMEI_IP_BOUNDARY_SWINGUP = 5,   /* emei/envs/mujoco/inverted_pendulum.py:149-183 */
    EMEI_HALFCHEETAH_RUNNING = 6,   /* emei/envs/mujoco/half_cheetah.py:16-67        */
    EMEI_IDP_REBOUND_BALANCING = 7,  /* emei/envs/mujoco/inverted_double_pendulum.py:63-90   */
    EMEI_IDP_BOUNDARY_BALANCING = 8, /* emei/envs/mujoco/inverted_double_pendulum.py:93-123  */
    EMEI_IDP_REBOUND_SWINGUP = 9,    /* emei/envs/mujoco/inverted_double_pendulum.py:126-153 */
    EMEI_IDP_BOUNDARY_SWINGUP = 10,  /* emei/envs/mujoco/inverted_double_pendulum.py:156-196 */
    EMEI_HOPPER_RUNNING = 11,        /* emei/envs/mujoco/hopper.py:17-106 */
    EMEI_NUM_ENVS_IDS = 12
};

/* Arithmetic the kernels compute in.
 * REF  : float64 state and derivative, mirroring the reference's own precision (classic control:
 *        float64 accumulator += float32(derivative)*float32(dt), base_control.py:162-164 and
 *        cartpole.py:60; MuJoCo bodies: float64 throughout).  State arrays in HBM are float64.
 * F32  : float32 state and arithmetic (fast mode; not bit-faithful on long chaotic trajectories). */
enum emei_precision { EMEI_PRECISION_REF = 0, EMEI_PRECISION_F32 = 1 };

/* Time integrator of the MuJoCo-backed bodies (mujoco_env.py:70-79).  Classic control ignores the
 * kwarg exactly like the reference (base_control.py:73 never forwards `method`).
 * EULER          : MuJoCo Euler velocity update + emei's position override q += dt*v_old (:94-97,169-195)
 * SEMI_IMPLICIT  : MuJoCo Euler as it is: v' = v + dt*qacc, q += dt*v'
 * RK4            : MuJoCo's 4-stage Runge-Kutta (mjINT_RK4): full forward dynamics at every stage */
enum emei_integrator { EMEI_INTEG_EULER = 0, EMEI_INTEG_SEMI_IMPLICIT = 1, EMEI_INTEG_RK4 = 2 };

/* `method` of ODE_approximation for the classic-control envs (base_control.py:133-173).  The reference's step() never
 * passes it (:73), so every classic-control env is integrated with EULER whatever its `integrator` kwarg says — that stays the
 * default.  RK4 is the function's other branch (:165-170), reachable only by calling ODE_approximation directly; selecting it
 * here is an explicit opt-in and reproduces that branch's float32 / float64 promotion chain (k stages are float32 arrays,
 * the stage states float64).  Ignored by the MuJoCo-backed bodies (their switch is `integrator`). */
enum emei_ode_method { EMEI_ODE_EULER = 0, EMEI_ODE_RK4 = 1 };

/* How Gaussian init / observation noise is laid out over the coordinates of one env.
 * IID      : an independent draw per coordinate (the evident intent of additive_gaussian_noise)
 * SHARED   : what the reference actually does for its only working batch size, B = 1: the row
 *            slicing of mujoco_env.py:243-244 adds ONE draw to every qpos entry and ONE to every qvel */
enum emei_noise_layout { EMEI_NOISE_IID = 0, EMEI_NOISE_SHARED = 1 };

/* Constraint solver of the MuJoCo-backed bodies with several simultaneous constraints (HalfCheetah, Hopper):
 * NEWTON : MuJoCo's primal formulation as documented — one row per violated joint limit, the 2 (condim - 1) edges of the
 *          pyramidal friction cone per contact, regularisers from the qpos0 inverse weights — solved to convergence
 *          by Newton's method on the active set; Euler's implicit joint damping applied after the solve (the default)
 * SWEEP1 : one fixed-order Gauss-Seidel sweep with box-clamped friction (round 1; ~5x cheaper, kept for comparison) */
enum emei_solver { EMEI_SOLVER_NEWTON = 0, EMEI_SOLVER_SWEEP1 = 1 };

/* dtype of the `actions` argument of emei_step / emei_rollout. */
enum emei_action_dtype { EMEI_ACT_U8 = 0, EMEI_ACT_I32 = 1, EMEI_ACT_I64 = 2, EMEI_ACT_F32 = 3 };

/* flags of emei_step / emei_rollout */
#define EMEI_FLAG_AUTO_RESET 1u /* re-initialise an env on the device when done (terminal|truncated) */

/* done byte written by emei_step / emei_rollout */
#define EMEI_DONE_TERMINAL 1u  /* reference `terminal` (get_batch_terminal) */
#define EMEI_DONE_TRUNCATED 2u /* gym TimeLimit `truncated` (register_env.py max_episode_steps) */

/* error codes */
#define EMEI_OK 0
#define EMEI_ERR_INVALID -1     /* bad argument (the reference asserts / raises ValueError) */
#define EMEI_ERR_HIP -2         /* a HIP runtime call failed */
#define EMEI_ERR_UNSUPPORTED -3 /* the reference raises NotImplementedError here */
#define EMEI_ERR_STATE -4       /* call order: e.g. step before reset (base_control.py:67) */

#define EMEI_MAX_STATE_DIM 32
#define EMEI_MAX_ENV_PARAMS 8

/* Indices into env_params.  HalfCheetahRunning (half_cheetah.py:23-24) uses 0-1; HopperRunning
 * (hopper.py:25-31) uses 0-7.  Defaults: cheetah 1.0, 0.1; hopper 1.0, 1e-3, 1.0, 1 (True), -100, 100,
 * 0.7, +inf.  The Hopper's healthy_angle_range is accepted by the Python class and, as in the reference
 * (np.logical_and's third argument is `out=`, hopper.py:91), never applied. */
enum emei_env_param {
    EMEI_PARAM_FORWARD_REWARD_WEIGHT = 0,
    EMEI_PARAM_CTRL_COST_WEIGHT = 1,
    EMEI_PARAM_HEALTHY_REWARD = 2,
    EMEI_PARAM_TERMINATE_WHEN_UNHEALTHY = 3, /* 0 / 1 */
    EMEI_PARAM_HEALTHY_STATE_LO = 4,
    EMEI_PARAM_HEALTHY_STATE_HI = 5,
    EMEI_PARAM_HEALTHY_Z_LO = 6,
    EMEI_PARAM_HEALTHY_Z_HI = 7
};

typedef struct emei_env emei_env; /* opaque */

typedef struct emei_config {
    uint32_t struct_size;       /* = sizeof(emei_config), ABI evolution */
    int32_t env_id;             /* enum emei_env_id */
    int64_t n_envs;             /* env instances owned by this handle (this rank's shard) */
    int32_t freq_rate;          /* substeps per step (base_control.py:17, mujoco_env.py:42) */
    int32_t precision;          /* enum emei_precision */
    double real_time_scale;     /* dt of ONE substep; NOT divided by freq_rate (base_control.py:73) */
    int32_t max_episode_steps;  /* TimeLimit (register_env.py:14-66); 0 = never truncate */
    int32_t device;             /* HIP device ordinal */
    uint64_t seed;              /* key of the device-side reset generator */
    uint64_t env_index_offset;  /* global index of env 0 of this shard (results independent of sharding) */
    double init_noise;          /* sigma of the Gaussian init noise on qpos of the MuJoCo bodies (mujoco_env.py:31) */
    /* -- fields below exist from ABI version 2 (struct_size 328); a version-1 caller (struct_size 64)
     *    gets integrator = EULER, IID layout, init_sigma[*] = init_noise, no observation noise.  A
     *    version-2 caller states every sigma in the arrays; `init_noise` above is then ignored. ------ */
    int32_t integrator;         /* enum emei_integrator (mujoco_env.py:70-79) */
    int32_t noise_layout;       /* enum emei_noise_layout */
    /* Per-coordinate sigmas in state order (qpos entries, then qvel entries): the float, (pos, vel)
     * tuple and {joint: (pos, vel)} dict forms of init_noise_params / obs_noise_params
     * (mujoco_env.py:218-227) all reduce to this.  obs_sigma is the noise added to the state after
     * EVERY substep (mujoco_env.py:98-104); all zero = off.  With the SHARED layout only the entries
     * of joint 0 (index 0 and state_dim/2) are used, as in the reference. */
    float init_sigma[EMEI_MAX_STATE_DIM];
    float obs_sigma[EMEI_MAX_STATE_DIM];
    /* -- from struct_size 400: constructor parameters of the reward / terminal functions.  Bit k of
     *    env_param_mask set = env_params[k] overrides the reference's default (enum emei_env_param). ---- */
    uint32_t env_param_mask;
    uint32_t solver;            /* enum emei_solver (this field was `reserved0`, always 0, before) */
    double env_params[EMEI_MAX_ENV_PARAMS];
    /* -- from struct_size 408 (ABI 6) --------------------------------------------------------------------------------- */
    int32_t ode_method;          /* enum emei_ode_method: CartPole only (base_control.py:165-170); 0 = what step() runs */
    /* emei_rollout of the bodies that run one-wave blocks (HalfCheetah, Hopper with the NEWTON solver, the double pendulum):
     * the launch is cut into work items of (64 envs) x (a chunk of steps), handed out chunk-major by an atomic ticket to
     * persistent one-wave workers, so that a SIMD that finishes early takes the next item instead of idling until the slowest
     * wave of the launch ends; an env's state travels between its items through the handle's state arrays, results are
     * bit-identical to the one-piece launch.
     *   0 = automatic: on when the shard has more waves than the device holds at once, with the guided schedule (each chunk
     *       half of the steps that remain, at least three: long items first, short items last)
     *  -1 = off (one-piece launches)
     *   k > 0 = k steps per item
     *  -(100 m + g), m = 1 .. 15, g = 1 .. 6 = the guided schedule with 1 / 2^g of the remaining steps per chunk, at least m (tuning) */
    int32_t rollout_chunk_steps;
} emei_config;
#define EMEI_CONFIG_SIZE_V1 64u

/* -- lifecycle ------------------------------------------------------------------------------- */
/* Replaces Env.__init__(freq_rate, real_time_scale, integrator, ...) (base_control.py:14-30;
 * mujoco_env.py:24-62).  Allocates the state SoA on cfg->device. */
EMEI_API int emei_create(const emei_config* cfg, emei_env** out);
EMEI_API int emei_destroy(emei_env* h);
EMEI_API const char* emei_last_error(void);
EMEI_API int emei_abi_version(void);

/* Debug / test getter: the model constants the kernels of `env_id` are COMPILED from, in physical terms, so that a test
 * can pin them to the reference's only data for the MuJoCo-backed bodies (the XML files under emei/envs/mujoco/assets;
 * tests/test_model_constants.py).  Writes at most `capacity` doubles to `out`, returns the count (negative = error;
 * EMEI_ERR_UNSUPPORTED for the classic-control CartPole, whose constants are cartpole.py:22-27 literals).  Host only.
 *   InvertedPendulum x4 (17): gravity, cart mass, pole mass, pole inertia about its com, com distance from the hinge, tilt of
 *       the pole axis at theta = 0, gear, ctrlrange lo, hi, slider range lo, hi, limit solref time constant, solimp dmin, dmax, width,
 *       hinge range lo, hi (rad; a limit row of the Balancing variants, freed by the SwingUp variants)
 *   InvertedDoublePendulum x4 (17): gravity x, gravity z, cart mass, pole mass, pole inertia about its com, pole com distance,
 *       pole-1 length, gear, ctrlrange lo, hi, slider range lo, hi, joint margin, limit solref time constant, solimp dmin, dmax, width
 *   HalfCheetahRunning (148) / HopperRunning (84), nb = 7 / 4 bodies, ng = 8 / 4 capsules, nj = 6 / 3 actuated hinges:
 *       gravity; per body in XML order {mass, com x, com z (body frame), inertia about the com, origin x, origin z (parent frame;
 *       root: world)}; per capsule in XML order {body, end-sphere centre 0 x, z, centre 1 x, z (body frame), radius, pair friction
 *       with the floor}; per actuated joint {stiffness, damping, armature, range lo, hi (rad), gear}; contact margin, contact
 *       solref time constant, solimp dmin, dmax, width, limit solref time constant, solimp dmin, dmax, width, ctrlrange lo, hi,
 *       rootz ref, sign of the leg hinges' axis (+1: +y, -1: -y)
 *
 * WHICH FILE: these are the constants of emei's OWN XML files under emei/envs/mujoco/assets.  An installed reference passes a RELATIVE
 * model_path (inverted_pendulum.py:28, half_cheetah.py:48) to gym's MujocoEnv, which resolves it against GYM's assets directory —
 * gym's files are not in the build image, the fields in which they differ from emei's are unknown. */
EMEI_API int emei_model_constants(int env_id, double* out, int capacity);

/* Debug / test getter: the qpos0 inverse weights (MuJoCo's mj_setConst: dof_invweight0 = (M0^-1)_jj of a joint, body_invweight0 =
 * trace(J_com M0^-1 J_com') / 3 of a body) that scale the constraint regularisers R = (1 - d) / d * diagApprox of the kernels of
 * `env_id` — computed at compile time from the kernels' OWN model (absolute-angle inertia at qpos0), so that a test can compare
 * them with the oracle's joint-space derivation as two independent computations (tests/test_oracle_solver.py; what is being
 * restated: mujoco.mj_step behind mujoco_env.py:93).  Same calling convention as emei_model_constants.  Host only.
 *   InvertedPendulum x4 (2): slider, hinge.   InvertedDoublePendulum x4 (1): slider.
 *   HalfCheetahRunning (13): the six leg hinges in XML order, then the seven bodies in XML order.
 *   HopperRunning (7): thigh, leg, foot hinges, then torso, thigh, leg, foot bodies. */
EMEI_API int emei_model_invweights(int env_id, double* out, int capacity);

/* Static facts about an env id: obs_dim, act_dim (0 = discrete scalar action), state_dim. */
EMEI_API int emei_env_dims(int env_id, int* obs_dim, int* act_dim, int* state_dim);

/* -- reset / state --------------------------------------------------------------------------- */
/* Env.reset(seed=) on the device (base_control.py:38-47; mujoco_env.py:130-140): every env gets a
 * fresh initial state from the counter-based generator keyed by `seed`; step counters and episode
 * counters are zeroed.  (The reference's host PCG64 / MT19937 streams are reproduced on the host and
 * uploaded with emei_set_state when bit parity of the initial state is wanted.) */
EMEI_API int emei_reset(emei_env* h, uint64_t seed, void* stream);

/* Upload / download the internal state as a row-major [n_envs, state_dim] float64 array
 * (`self.state` of base_control.py:28,46,74; (qpos,qvel) of mujoco_env.py:116,132).
 * emei_set_state zeroes the step counters when reset_counters != 0. */
EMEI_API int emei_set_state(emei_env* h, const double* state_aos, int reset_counters, void* stream);
EMEI_API int emei_get_state(emei_env* h, double* state_aos, void* stream);
/* current observation as float64 [n_envs, obs_dim] (current_obs, mujoco_env.py:153-155;
 * inverted_pendulum.py:45-49) */
EMEI_API int emei_get_obs(emei_env* h, double* obs_aos, void* stream);

/* Freezable.freeze / unfreeze (base_control.py:32-36; mujoco_env.py:114-120): device-to-device
 * snapshot / restore of the state SoA, the counters and the key of the reset generator (so that the auto-reset episodes of
 * the restored trajectory draw what they would have drawn, whatever reset(seed=) happened in between). */
EMEI_API int emei_freeze(emei_env* h, void* stream);
EMEI_API int emei_unfreeze(emei_env* h, void* stream);

/* Newton solves of this handle's HalfCheetah / Hopper rollouts that ended at the iteration cap (24 passes) WITHOUT meeting their
 * stopping rule, since emei_create: count_out [1] uint64 (device-accessible).  The kernels' iteration takes unit Newton steps
 * without a line search; it has never been seen to need more than 9 passes, and this counter is how that is checked
 * (tests/test_gpu_parity_sweeps.py, tests/test_gpu_long_horizon.py assert 0).  Always 0 for the other envs and the SWEEP1 solver. */
EMEI_API int emei_get_solver_cap_hits(emei_env* h, uint64_t* count_out, void* stream);

/* Work items of this handle's chunked body rollouts (emei_config.rollout_chunk_steps) that gave up waiting for the item that
 * precedes them on the same envs (a bounded wait: 2 s of the device's 100 MHz counter) and were skipped, since emei_create:
 * count_out [1] uint64 (device-accessible).  Must stay 0 — an item's predecessor has always started before it (tickets are
 * handed out in start order); a non-zero count means a rollout's outputs are incomplete. */
EMEI_API int emei_get_rollout_faults(emei_env* h, uint64_t* count_out, void* stream);

/* Re-key the device reset generator without touching the state: the seed of Env.reset(seed=) reaches the
 * handle also when the initial state itself is drawn on the host and uploaded with emei_set_state
 * (base_control.py:38-47: gym seeds np_random, every later auto-reset episode must depend on it too). */
EMEI_API int emei_set_seed(emei_env* h, uint64_t seed);

/* -- the hot path ---------------------------------------------------------------------------- */
/* Env.step(action) for all n_envs instances (base_control.py:61-83; mujoco_env.py:157-167):
 *   actions   [n_envs] (discrete) or [n_envs, act_dim] (continuous), dtype per action_dtype
 *   obs_out   [n_envs, obs_dim] float32  next observation (before any auto-reset)
 *   reward_out[n_envs] float32
 *   done_out  [n_envs] uint8   EMEI_DONE_* bits
 * Any output pointer may be NULL to skip that output. */
EMEI_API int emei_step(emei_env* h, const void* actions, int action_dtype, float* obs_out, float* reward_out,
              uint8_t* done_out, uint32_t flags, void* stream);

/* The gym single-env call itself — `obs, reward, terminal, truncated, info = env.step(action)` of base_control.py:61-83 /
 * mujoco_env.py:157-167 with HOST values in and out — as one synchronous call: every pointer is page-locked host memory
 * that the device can address (hipHostMalloc / hipHostRegister / torch pin_memory), the kernels read the action from it
 * and write the results into it, and the function returns when the results are visible to the host.
 *   actions_host  as `actions` of emei_step
 *   obs64_host    [n_envs, obs_dim] float64: the observation of the STATE after the step (the float64 array the reference
 *                 returns; after an auto-reset the new episode's first observation, as emei_get_obs)
 *   obs32_host, reward_host, done_host: as emei_step (the step's own observation, before any auto-reset)
 * One launch and no stream synchronisation for a single env of the 4-state family (the kernel's last store is a completion
 * word the host polls); step + emei_get_obs + a stream synchronisation otherwise.  None of the pointers may be NULL. */
EMEI_API int emei_step_host(emei_env* h, const void* actions_host, int action_dtype, double* obs64_host, float* obs32_host,
                   float* reward_host, uint8_t* done_host, uint32_t flags, void* stream);

/* n_steps fused steps in ONE launch, state kept in registers (the caller loops of zoo/util.py:54-73
 * and emei/util.py:14-36 collapsed): actions [n_steps, n_envs(, act_dim)], obs_out
 * [n_steps, n_envs, obs_dim], reward_out [n_steps, n_envs], done_out [n_steps, n_envs].
 * Results are identical to n_steps calls of emei_step. */
EMEI_API int emei_rollout(emei_env* h, int32_t n_steps, const void* actions, int action_dtype, float* obs_out,
                 float* reward_out, uint8_t* done_out, uint32_t flags, void* stream);

/* Which kernel the LAST emei_step / emei_rollout of this handle launched (enum emei_kernel_id): a debug /
 * test getter, so that a parity test can assert that the path it checked is the path bench.py times. */
enum emei_kernel_id {
    EMEI_KERNEL_NONE = 0,
    EMEI_KERNEL_PEND_STAGED_FREQ1 = 1, /* pend_rollout_staged_kernel<Env, ActT, true>  (freq_rate == 1) */
    EMEI_KERNEL_PEND_STAGED = 2,       /* pend_rollout_staged_kernel<Env, ActT, false> */
    EMEI_KERNEL_PEND_GENERIC_FULL = 3, /* pend_rollout_kernel<Env, ActT, true>: ragged n, short T or unaligned buffers */
    EMEI_KERNEL_PEND_GENERIC = 4,      /* pend_rollout_kernel<Env, ActT, false>: some output pointer is NULL */
    EMEI_KERNEL_BODY = 5,              /* body_rollout_kernel<Body, false> (euler / semi-implicit euler) */
    EMEI_KERNEL_BODY_RK4 = 6,          /* body_rollout_kernel<Body, true> */
    EMEI_KERNEL_BODY_CHUNKED = 7,      /* body_rollout_kernel<Body, false> as (64 envs) x (chunk of steps) work items (rollout_chunk_steps) */
    EMEI_KERNEL_BODY_RK4_CHUNKED = 8,  /* body_rollout_kernel<Body, true> likewise */
    EMEI_KERNEL_PEND_STAGED_PEERS_FREQ1 = 9, /* pend_rollout_staged_peers_kernel<Env, ActT, true>: with the peer stores of emei_set_obs_peers */
    EMEI_KERNEL_PEND_STAGED_PEERS = 10       /* pend_rollout_staged_peers_kernel<Env, ActT, false> */
};
EMEI_API int emei_last_rollout_kernel(emei_env* h);

/* --- Multi-GPU observation return by PEER WRITES (one process per GPU; north_star: "shards env instances across up to 8 GPUs of one node
 * ... only for the batched observation return"; SURVEY §5: "hipIpc peer writes from the step kernel's epilogue").  The reference has no
 * counterpart (it is a single-process CPU library: zoo/util.py:54-73 steps one env); this replaces the RCCL all-gather that follows a
 * rollout launch: the rollout kernel itself stores every step's observation row into the gathered buffer of every rank, so the block is
 * neither re-read from HBM nor moved by a separate collective.
 *
 * emei_set_obs_peers: from now on emei_rollout (emei_step likewise: it is a rollout of one step, hence refused) ALSO stores the observation of step t, env i at
 *     ((float*)peer_obs[p])[((int64_t)t * row_envs + col_offset + i) * obs_dim + k],  k < obs_dim,  for every p < n_peers
 * i.e. each peer buffer is a gathered block [n_steps, row_envs, obs_dim] float32 and this handle's envs are its columns
 * [col_offset, col_offset + n_envs).  peer_obs[p] are device pointers valid on this handle's device: memory of this process, or of another
 * rank mapped with emei_peer_buffer_open (16-byte aligned; row_envs >= col_offset + n_envs; each holds max_steps rows: a rollout of
 * more steps is refused with EMEI_ERR_INVALID instead of writing past them).  n_peers = 0 switches it off; at most EMEI_MAX_OBS_PEERS (a rank's own gathered buffer counts as one).
 * Built for the staged kernel of the CartPole family (BASELINE configs[4]'s env; n_envs a multiple of 64, n_steps >= 16, 16-byte
 * aligned buffers): any other rollout with peers set FAILS with EMEI_ERR_UNSUPPORTED — nothing is skipped silently.
 * Ordering is the caller's: a peer may read a block once the writer's launch has completed (stream / event synchronisation on the
 * writer, then any host-side barrier between the ranks), and the writer may reuse a buffer once its readers are done. */
#define EMEI_MAX_OBS_PEERS 8
EMEI_API int emei_set_obs_peers(emei_env* h, int n_peers, void* const* peer_obs, int64_t row_envs, int64_t col_offset, int32_t max_steps);

/* Device memory another process of this node can map: hipMalloc + hipIpcGetMemHandle / hipIpcOpenMemHandle (the handle is 64 opaque
 * bytes to be carried to the other rank by whatever the host side has: torch.distributed.all_gather_object, MPI, a pipe).  `device` is
 * the HIP device ordinal of the CALLER (the creator allocates there; the opener maps the creator's memory into that device's address
 * space, peer access over xGMI).  A buffer is closed by every opener before its creator destroys it. */
typedef struct emei_ipc_handle { unsigned char bytes[64]; } emei_ipc_handle;
EMEI_API int emei_peer_buffer_create(int device, uint64_t bytes, void** dev_ptr_out, emei_ipc_handle* handle_out);
EMEI_API int emei_peer_buffer_open(int device, const emei_ipc_handle* handle, void** dev_ptr_out);
EMEI_API int emei_peer_buffer_close(int device, void* dev_ptr);
EMEI_API int emei_peer_buffer_destroy(int device, void* dev_ptr);

/* Sorted indices of the envs whose last step reported done (wavefront-ballot compaction of the done
 * masks the step kernels leave behind): idx_out [n_envs] int32 (first *count_out entries valid),
 * count_out [1] int32. */
EMEI_API int emei_compact_done(emei_env* h, int32_t* idx_out, int32_t* count_out, void* stream);

/* Per-env counters of the handle: steps since the last reset (TimeLimit) and the episode index (the
 * counter word of the device reset generator).  Either pointer may be NULL. */
EMEI_API int emei_get_counters(emei_env* h, int32_t* steps_out, uint32_t* episode_out, void* stream);

/* Initial observation the DEVICE reset gives env `env_index[k]` (local index in this handle) at the
 * start of episode `episode[k]`: what Env.reset() would return for that (env, episode) pair.  Used to
 * rebuild `observations` after an auto-reset when a rollout is turned into an offline dataset
 * (schema of zoo/util.py:16-30,62-67): obs_out [count, obs_dim] float32. */
EMEI_API int emei_episode_init_obs(emei_env* h, int64_t count, const int64_t* env_index, const uint32_t* episode,
                                   float* obs_out, void* stream);

/* -- stateless batched reward / terminal (model-based-RL callers) ---------------------------- */
/* EmeiEnv.get_batch_reward / get_batch_terminal (core.py:182-188; cartpole.py:124-129,145-151;
 * inverted_pendulum.py:73-183; half_cheetah.py:59-67): obs, pre_obs [n, obs_dim] float32,
 * action [n, act_dim] float32 (may be NULL where the env ignores it), out [n]. */
EMEI_API int emei_reward(int env_id, int64_t n, const float* obs, const float* pre_obs, const float* action,
                double real_time_scale, int32_t freq_rate, float* reward_out, void* stream);
EMEI_API int emei_terminal(int env_id, int64_t n, const float* obs, uint8_t* terminal_out, void* stream);

/* The same with constructor parameters (enum emei_env_param; mask 0 = the defaults above). */
EMEI_API int emei_reward_ex(int env_id, int64_t n, const float* obs, const float* pre_obs, const float* action,
                   double real_time_scale, int32_t freq_rate, uint32_t env_param_mask, const double* env_params,
                   float* reward_out, void* stream);
EMEI_API int emei_terminal_ex(int env_id, int64_t n, const float* obs, uint32_t env_param_mask, const double* env_params,
                     uint8_t* terminal_out, void* stream);

/* The same three functions on the dtype the CALLER holds (enum emei_io_dtype).  The reference evaluates
 * get_batch_reward / get_batch_terminal on float64 arrays (cartpole.py:124-129,145-151; inverted_pendulum.py:73-183;
 * half_cheetah.py:59-67; hopper.py:95-106): with EMEI_IO_F64 obs / pre_obs / action are float64 [n, dim] and are
 * NOT narrowed — thresholds (|x| < 5, x_l < x < x_r, z ranges) and the x-difference of the forward reward are evaluated
 * on the caller's float64 values; reward_out is float64 [n].  EMEI_IO_F32 is the float32 form above. */
enum emei_io_dtype { EMEI_IO_F32 = 0, EMEI_IO_F64 = 1 };
/* flags of emei_reward_io */
#define EMEI_REWARD_BATCH_CTRL_COST 1u /* HalfCheetahRunning / HopperRunning only: the control cost is np.sum(np.square(action)) over the
                                          WHOLE batch, as half_cheetah.py:61 executes for B > 1 (no axis argument); the
                                          default (0) sums per env = step() semantics */
EMEI_API int emei_reward_io(int env_id, int64_t n, int io_dtype, const void* obs, const void* pre_obs, const void* action,
                   double real_time_scale, int32_t freq_rate, uint32_t env_param_mask, const double* env_params,
                   uint32_t flags, void* reward_out, void* stream);
EMEI_API int emei_terminal_io(int env_id, int64_t n, int io_dtype, const void* obs, uint32_t env_param_mask,
                     const double* env_params, uint8_t* terminal_out, void* stream);

/* EmeiEnv.get_batch_next_obs (core.py:190-193; abstract in the reference, no env implements it):
 * one step from caller-supplied float32 observations without touching any handle state.
 * obs [n, obs_dim], action as in emei_step, next_obs_out [n, obs_dim]. */
EMEI_API int emei_next_obs(int env_id, int64_t n, const float* obs, const void* actions, int action_dtype,
                  double real_time_scale, int32_t freq_rate, int32_t precision, float* next_obs_out,
                  void* stream);

/* The same for every env whose observation determines its state (CartPole, InvertedPendulum, HalfCheetah,
 * Hopper; the InvertedDoublePendulum's observation "wrap", inverted_double_pendulum.py:59, is not
 * invertible -> EMEI_ERR_UNSUPPORTED), with the integrator of mujoco_env.py:70-79 (classic control
 * ignores it).  No observation noise is added. */
EMEI_API int emei_next_obs_ex(int env_id, int64_t n, const float* obs, const void* actions, int action_dtype,
                     double real_time_scale, int32_t freq_rate, int32_t precision, int32_t integrator,
                     float* next_obs_out, void* stream);

/* emei_next_obs_ex on the caller's dtype: obs and next_obs_out are [n, obs_dim] of io_dtype (float64 observations enter the
 * float64 state unrounded); actions as in emei_step.  `integrator` may carry EMEI_NEXT_OBS_ODE_RK4 for the classic-control
 * envs (enum emei_ode_method RK4: ODE_approximation(method="rk4"), base_control.py:165-170); ignored by the other envs. */
#define EMEI_NEXT_OBS_ODE_RK4 0x100
EMEI_API int emei_next_obs_io(int env_id, int64_t n, int io_dtype, const void* obs, const void* actions, int action_dtype,
                     double real_time_scale, int32_t freq_rate, int32_t precision, int32_t integrator,
                     void* next_obs_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EMEI_HIP_H */
