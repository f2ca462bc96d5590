"""HipEnv — the GPU-backed base of every emei_amd env: EmeiEnv's surface on top of the C-ABI engine.

``num_envs=1`` (default) behaves like the reference's single env: ``reset()`` returns
``(np.ndarray float64 [obs_dim], {})`` and ``step(a)`` returns
``(obs, np.float64 reward, np.bool_ terminal, truncated, {})`` (base_control.py:38-83;
mujoco_env.py:157-167).  ``num_envs=N`` is the vectorised form: tensors on the device, shapes
``[N, obs_dim]`` / ``[N]``.  Every computation runs in libemei_hip.so; there is no CPU path.
"""
from typing import Optional

import numpy as np

from ..core import EmeiEnv

_Tensor = None


def _is_tensor(x):
    """isinstance(x, torch.Tensor) without importing torch on every step() (the import is a dictionary lookup, but the
    single-env path counts microseconds)"""
    global _Tensor
    if _Tensor is None:
        import torch

        _Tensor = torch.Tensor
    return isinstance(x, _Tensor)


class HipEnv(EmeiEnv):
    ENGINE_NAME = None  # key of emei_amd._lib.ENV_IDS
    ENGINE_INTEGRATOR = None  # set per instance by the MuJoCo-backed envs; classic control ignores the kwarg
    ODE_METHOD = "euler"  # set per instance by the classic-control envs (`ode_method=`): ODE_approximation's `method`
    metadata = {"render_modes": [], "render_fps": 50}

    def __init__(self, freq_rate: int = 1, real_time_scale: float = 0.02, integrator: str = "euler",
                 num_envs: int = 1, precision: str = "ref", device: Optional[int] = None,
                 max_episode_steps: Optional[int] = None, auto_reset: bool = False, init_noise=0.0,
                 env_index_offset: int = 0, obs_noise=0.0, noise_layout: Optional[str] = None,
                 render_mode: Optional[str] = None, engine_env_params: Optional[dict] = None, solver: str = "newton",
                 rollout_chunk_steps: int = 0):
        if render_mode is not None:
            # base_control.py:15,21 / mujoco_env.py:33 accept "human" / "rgb_array"; pygame and the MuJoCo viewer
            # are outside the env-step path
            raise NotImplementedError(f"render_mode={render_mode!r}: rendering is not part of the HIP engine")
        self.render_mode = None
        self.freq_rate = freq_rate
        self.real_time_scale = real_time_scale
        self.integrator = integrator
        self.num_envs = int(num_envs)
        self.precision = precision
        self.device_index = device
        self.max_episode_steps = int(max_episode_steps or 0)
        self.auto_reset = bool(auto_reset)
        self._init_noise = init_noise
        self._engine_env_params = dict(engine_env_params or {})  # non-default reward / health parameters (_lib.ENV_PARAMS)
        self._obs_noise = obs_noise
        self._rollout_chunk_steps = int(rollout_chunk_steps)  # Engine: work items of the one-wave-per-SIMD bodies' rollouts
        self._solver = solver  # constraint solver of the multi-constraint bodies (HalfCheetah, Hopper): "newton" | "sweep1"
        # the reference only works for B = 1, where its row slicing shares one draw over all of qpos and
        # one over all of qvel (mujoco_env.py:243-244); the vectorised form draws per coordinate
        self._noise_layout = noise_layout or ("shared" if self.num_envs == 1 else "iid")
        self._env_index_offset = int(env_index_offset)
        self._engine = None
        self._np_random = None
        self._dev_seed_base, self._dev_seed_count = None, 0  # key of the device reset generator (see _next_device_seed)
        self.state = None  # `self.state is not None` after reset (base_control.py:67)
        EmeiEnv.__init__(self, env_params=dict(freq_rate=freq_rate, real_time_scale=real_time_scale, integrator=integrator))

    # -- gym.Env plumbing the reference inherits ---------------------------------------------------
    @property
    def np_random(self):
        """gym 0.26: a PCG64 Generator, re-seeded by reset(seed=...)."""
        if self._np_random is None:
            self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence()))
        return self._np_random

    @property
    def unwrapped(self):
        return self

    @property
    def dt(self):
        """gym MujocoEnv.dt = timestep * frame_skip."""
        return self.real_time_scale * self.freq_rate

    def render(self):
        raise NotImplementedError("rendering (pygame / MuJoCo viewer) is outside the env-step path")

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    # -- engine -------------------------------------------------------------------------------------
    @property
    def engine(self):
        if self._engine is None:
            from ..engine import Engine  # raises loudly when the HIP library or a GPU is missing

            self._engine = Engine(self.ENGINE_NAME, self.num_envs, freq_rate=self.freq_rate,
                                  real_time_scale=self.real_time_scale, precision=self.precision,
                                  max_episode_steps=self.max_episode_steps, device=self.device_index,
                                  env_index_offset=self._env_index_offset, init_noise=self._init_noise,
                                  integrator=self.ENGINE_INTEGRATOR or "euler", obs_noise=self._obs_noise,
                                  noise_layout=self._noise_layout, env_params=self._engine_env_params, solver=self._solver,
                                  ode_method=self.ODE_METHOD, rollout_chunk_steps=self._rollout_chunk_steps)
        return self._engine

    def _host_init_state(self, batch_size) -> np.ndarray:
        """[B, state_dim] float64 initial states drawn the way the reference draws them."""
        raise NotImplementedError

    def _state_to_obs_np(self, state: np.ndarray) -> np.ndarray:
        return state.copy()

    def _next_device_seed(self):
        """Key of the device reset generator for the episodes that follow this reset().  The device draws the
        initial state of every auto-reset episode from Philox(key, env, episode), so the key must follow the
        seed the caller passed (gym: reset(seed=) re-seeds everything random in the env) and must differ
        between successive un-seeded resets, or every reset() would replay the same post-reset episodes.
        reset(seed=s) -> s; the k-th un-seeded reset after it -> splitmix64(s + k).  np_random is NOT consumed:
        its stream stays draw-for-draw the reference's (base_control.py:38-47)."""
        if self._dev_seed_base is None:
            self._dev_seed_base = int(np.random.SeedSequence().entropy) & (2**64 - 1)
        k, self._dev_seed_count = self._dev_seed_count, self._dev_seed_count + 1
        if k == 0:
            return self._dev_seed_base
        z = (self._dev_seed_base + k * 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        return z ^ (z >> 31)

    # -- reset / step -------------------------------------------------------------------------------
    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None):
        if seed is not None:
            self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
            self._dev_seed_base, self._dev_seed_count = int(seed) & (2**64 - 1), 0
        dev_seed = self._next_device_seed()
        if options and options.get("device_rng"):
            # perf path: counter-based generator on the device, no host round trip
            self.engine.reset(dev_seed)
            self.state = True
            return (self._obs_single() if self.num_envs == 1 else self.engine.get_obs()), {}
        init = self._host_init_state(self.num_envs)
        self.engine.set_seed(dev_seed)
        self.engine.set_state(init)
        self.state = True
        if self.num_envs == 1:
            return self._state_to_obs_np(init)[0], {}
        return self.engine.get_obs(), {}

    def _obs_single(self):
        return self.engine.get_obs()[0].cpu().numpy()

    def _check_single_action(self, action):
        raise NotImplementedError

    def step(self, action):
        assert self.state is not None, "Call reset before using step method."  # base_control.py:67
        if self.num_envs == 1 and not _is_tensor(action):
            # the gym single-env call: one emei_step_host (engine.py:step_host); ~11 us for CartPole, of which ~3 are this method
            act = self._check_single_action(action)
            obs64, obs32, rew, done = (self._engine or self.engine).step_host(act, self.auto_reset)
            d = int(done[0])
            # after an auto-reset the handle already holds the next episode's state: report the step's own obs
            obs = obs64[0].copy() if not (d and self.auto_reset) else obs32[0].astype(np.float64)
            return obs, np.float64(rew[0]), np.bool_(d & 1), bool(d & 2), {}
        import torch

        eng = self.engine
        a = action if isinstance(action, torch.Tensor) else torch.as_tensor(action)
        a = a.to(eng.device)
        if eng.act_dim == 0 and a.dtype not in (torch.uint8, torch.int32, torch.int64):
            a = a.to(torch.int64)
        if eng.act_dim > 0:
            a = a.to(torch.float32)
        obs, rew, done = eng.step(a.contiguous(), auto_reset=self.auto_reset)
        return obs, rew, (done & 1).bool(), (done & 2).bool(), {}

    def rollout(self, actions, auto_reset=None):
        """T fused steps in one launch: actions [T, N(, act_dim)] -> obs [T,N,obs_dim] f32, reward [T,N] f32,
        terminal [T,N] bool, truncated [T,N] bool."""
        assert self.state is not None, "Call reset before using step method."
        obs, rew, done = self.engine.rollout(actions, auto_reset=self.auto_reset if auto_reset is None else auto_reset)
        return obs, rew, (done & 1).bool(), (done & 2).bool()

    # -- freeze / unfreeze: device-to-device snapshot of the state SoA -------------------------------
    def freeze(self) -> None:
        self.engine.freeze()  # state, counters and the device reset key (emei_freeze)
        self._frozen_seed = (self._dev_seed_base, self._dev_seed_count)  # ... and the host side of that key
        self.frozen = True

    def unfreeze(self) -> None:
        self.engine.unfreeze()
        self._dev_seed_base, self._dev_seed_count = self._frozen_seed
        self.frozen = False

    # -- batched functions of the EmeiEnv surface ------------------------------------------------------
    def _to_dev(self, x, dtype=None):
        import torch

        if x is None:
            return None
        t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
        t = t.to(self.engine.device)
        return t if dtype is None else t.to(dtype)

    def _like(self, out, ref, np_dtype):
        """[B] device tensor -> [B,1] in the caller's array type (the reference returns [B,1] arrays)."""
        import torch

        out = out.reshape(-1, 1)
        if isinstance(ref, torch.Tensor):
            return out
        return out.cpu().numpy().astype(np_dtype)

    def _rows_to_dev(self, x, like=None):
        """caller's rows -> device tensor: float64 (NumPy's default, what the reference computes on) stays float64, everything
        else becomes float32; `like` forces the dtype of the observation rows on pre_obs / action"""
        import torch

        if x is None:
            return None
        t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
        want = like.dtype if like is not None else (torch.float64 if t.dtype == torch.float64 else torch.float32)
        return t.to(device=self.engine.device, dtype=want)

    # whole-batch control cost of half_cheetah.py:61 / hopper.py:98 (np.sum without an axis): opt-in per call or per env
    reference_batch_semantics = False

    def get_batch_reward(self, obs, pre_obs=None, action=None, state=None, pre_state=None, reference_batch_semantics=None):
        from .. import engine as E

        o = self._rows_to_dev(obs)
        whole = self.reference_batch_semantics if reference_batch_semantics is None else bool(reference_batch_semantics)
        r = E.batch_reward(self.ENGINE_NAME, o, self._rows_to_dev(pre_obs, o), self._rows_to_dev(action, o),
                           self.real_time_scale, self.freq_rate, self._engine_env_params, batch_ctrl_cost=whole)
        return self._like(r, obs, np.float64)

    def get_batch_terminal(self, obs, pre_obs=None, action=None, state=None, pre_state=None):
        from .. import engine as E

        t = E.batch_terminal(self.ENGINE_NAME, self._rows_to_dev(obs), self._engine_env_params)
        return self._like(t, obs, np.bool_)

    def get_batch_next_obs(self, obs, pre_obs=None, action=None, state=None, pre_state=None):
        """core.py:190-193 asserts `frozen` and is abstract in the reference (no env implements it);
        here it is one kernel step from the given observations, without touching the env's own state."""
        import torch

        from .. import engine as E

        assert self.frozen
        o = self._rows_to_dev(obs)
        a = self._to_dev(action)
        if self.engine.act_dim == 0:
            a = a.reshape(-1).to(torch.int64)
        else:
            a = a.to(torch.float32).reshape(o.shape[0], -1)
        nxt = E.batch_next_obs(self.ENGINE_NAME, o, a.contiguous(), self.real_time_scale, self.freq_rate, self.precision,
                               self.ENGINE_INTEGRATOR or "euler", self.ODE_METHOD)
        if isinstance(obs, torch.Tensor):
            return nxt
        return nxt.cpu().numpy().astype(np.float64)

    def get_batch_init_state(self, batch_size):
        return self._host_init_state(batch_size)


def joint_sigmas(params, nq):
    """init_noise_params / obs_noise_params -> (pos sigma [nq], vel sigma [nq]) per (1-dof) joint:
    float -> the same sigma everywhere; (pos, vel) tuple -> per half; {joint: (pos, vel)} -> the listed
    joints, zero elsewhere (mujoco_env.py:218-227)."""
    if isinstance(params, dict):
        pos, vel = np.zeros(nq), np.zeros(nq)
        for j, pv in params.items():
            if not 0 <= int(j) < nq:
                continue  # `if jnt_id in noise_params` never matches it
            pos[int(j)], vel[int(j)] = float(pv[0]), float(pv[1])
        return pos, vel
    if isinstance(params, (tuple, list)):
        if len(params) != 2:
            raise ValueError(f"noise params {params!r}: expected a float, a (pos, vel) pair or a dict")
        return np.full(nq, float(params[0])), np.full(nq, float(params[1]))
    return np.full(nq, float(params)), np.full(nq, float(params))


JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3  # mjtJoint


def _quat_mul(a, b):
    """Hamilton product of scalar-LAST quaternions (x, y, z, w)"""
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def check_noise_joints(jnt_type):
    """`additive_gaussian_noise` (mujoco_env.py:229-237) with a free joint: the routine slices ROWS of the [B, nq] arrays
    (`origin_pos[cur + 3 : cur + 7]`), so `Rotation.from_quat` never sees a quaternion and raises ValueError for every batch
    size (tests/golden/freejoint_golden.npz: fj_noise_B1_raises, fj_noise_B4_raises) — the same error here, with the reason."""
    if any(t == JNT_FREE for t in jnt_type):
        raise ValueError("additive_gaussian_noise: a free joint's quaternion is sliced by rows (mujoco_env.py:235): the reference "
                         "raises ValueError for every batch size")
    if any(t == JNT_BALL for t in jnt_type):
        raise NotImplementedError


def euler_position_rule(old_pos, old_vel, jnt_type, dt):
    """`EmeiMujocoEnv.get_euler_pos` (mujoco_env.py:169-195) for ANY joint list: q += dt * v for slide / hinge joints (what the
    kernels do on the device: every env of the engine has only those), NotImplementedError for a ball joint (:185-186), and for
    a free joint (:176-184) exactly what the reference executes — position += dt * linear velocity; the four quaternion entries
    of qpos (MuJoCo stores w first) are handed to SciPy's Rotation AS IF scalar-last, read out as extrinsic z-y-x Euler angles
    in DEGREES, the angular velocity * dt (radians) is added to those degrees, and the sum is re-composed as extrinsic x-y-z
    angles into the new quaternion.  Restated with plain NumPy (no SciPy needed); pinned by tests/golden/freejoint_golden.npz."""
    old_pos, old_vel = np.asarray(old_pos, dtype=np.float64), np.asarray(old_vel, dtype=np.float64)
    new_pos = old_pos.copy()
    ip = iv = 0
    for t in jnt_type:
        if t == JNT_FREE:
            new_pos[ip:ip + 3] += old_vel[iv:iv + 3] * dt
            x, y, z, w = old_pos[ip + 3:ip + 7] / np.linalg.norm(old_pos[ip + 3:ip + 7])  # Rotation.from_quat normalises
            # rotation matrix entries needed for R = Rx(c) Ry(b) Rz(a)  (as_euler("zyx"): about fixed z, then y, then x)
            r00, r01, r02 = 1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)
            r12, r22 = 2 * (y * z - x * w), 1 - 2 * (x * x + y * y)
            ang = np.degrees([np.arctan2(-r01, r00), np.arcsin(np.clip(r02, -1.0, 1.0)), np.arctan2(-r12, r22)])
            ang = ang + old_vel[iv + 3:iv + 6] * dt
            h = np.radians(ang) / 2  # from_euler("xyz"): about fixed x by ang[0], then y by ang[1], then z by ang[2]
            qx = np.array([np.sin(h[0]), 0.0, 0.0, np.cos(h[0])])
            qy = np.array([0.0, np.sin(h[1]), 0.0, np.cos(h[1])])
            qz = np.array([0.0, 0.0, np.sin(h[2]), np.cos(h[2])])
            new_pos[ip + 3:ip + 7] = _quat_mul(qz, _quat_mul(qy, qx))
            ip, iv = ip + 7, iv + 6
        elif t == JNT_BALL:
            raise NotImplementedError
        else:
            new_pos[ip] += old_vel[iv] * dt
            ip, iv = ip + 1, iv + 1
    return new_pos


class ParityUnpinnedWarning(UserWarning):
    """The dynamics of a MuJoCo-backed env are a restatement of MuJoCo's published algorithms, not libmujoco itself."""


class MujocoHipEnv(HipEnv):
    """The part of EmeiMujocoEnv (mujoco_env.py:23-62,130-155,197-249) that is first-party Python:
    constructor kwargs, init state + Gaussian init noise, obs = concat(qpos, qvel).  The MuJoCo
    arithmetic itself is the engine's closed-form body model (parity unpinned)."""

    _warned = set()

    NQ = None  # number of (slide/hinge) joints = len(qpos) = len(qvel)
    INIT_QPOS = None  # non-zero entries of init_qpos (the Hopper's rootz ref); None = zeros

    def __init__(self, freq_rate, real_time_scale, integrator="euler", init_noise_params=0.0, obs_noise_params=0.0,
                 **kwargs):
        if integrator not in ("euler", "semi_implicit_euler", "rk4"):
            raise NotImplementedError(f"integrator {integrator!r}")  # mujoco_env.py:78-79
        self.ENGINE_INTEGRATOR = integrator
        if type(self).__name__ not in MujocoHipEnv._warned:  # once per env class
            import warnings

            MujocoHipEnv._warned.add(type(self).__name__)
            warnings.warn(
                f"{type(self).__name__}: the reference steps this env with the third-party `mujoco` package (mujoco_env.py:93), "
                "which is not available to this build.  The engine restates MuJoCo's published pipeline (inertia from geoms, "
                "soft constraints with solref / solimp, pyramidal friction cones, converged Newton solve, implicit joint damping); "
                "rewards, terminals, observation layout, integrator switch and noise routines are pinned to the reference, the "
                "next-state values are NOT pinned against libmujoco (DESIGN.md, section 5).", ParityUnpinnedWarning, stacklevel=3)
        self.init_noise_params, self.obs_noise_params = init_noise_params, obs_noise_params
        super().__init__(freq_rate=freq_rate, real_time_scale=real_time_scale, integrator=integrator,
                         init_noise=np.concatenate(joint_sigmas(init_noise_params, self.NQ)).tolist(),
                         obs_noise=np.concatenate(joint_sigmas(obs_noise_params, self.NQ)).tolist(), **kwargs)
        self.init_qpos = np.zeros(self.NQ) if self.INIT_QPOS is None else np.asarray(self.INIT_QPOS, dtype=np.float64)
        self.init_qvel = np.zeros(self.NQ)

    def _host_init_state(self, batch_size):
        """mujoco_env.py:137-140,197-249.  For batch_size == 1 the reference's row slicing adds ONE
        sigma*N(0,1) draw (global numpy stream) to every qpos entry and a second one to every qvel
        entry — with the sigmas of joint 0 — and consumes two more draws per further joint (added to
        empty slices); that is reproduced.  For batch_size > 1 the reference raises ValueError
        (non-broadcastable); the vectorised form draws per-coordinate i.i.d. noise from the env's
        seeded generator instead."""
        sp, sv = joint_sigmas(self.init_noise_params, self.NQ)
        nq = self.NQ
        if batch_size == 1:
            e = [np.random.randn(1, 1) for _ in range(2 * nq)]  # pos j0, vel j0, then the dropped draws
            return np.concatenate([self.init_qpos[None, :] + e[0] * sp[0], self.init_qvel[None, :] + e[1] * sv[0]], axis=1)
        base = np.concatenate([np.tile(self.init_qpos, (batch_size, 1)), np.tile(self.init_qvel, (batch_size, 1))], axis=1)
        return base + self.np_random.standard_normal((batch_size, 2 * nq)) * np.concatenate([sp, sv])

    def get_batch_init_state(self, batch_size):
        s = self._host_init_state(batch_size)
        return s[:, :self.NQ], s[:, self.NQ:]  # (pos, vel), mujoco_env.py:137-140

    JNT_TYPE = None  # model.jnt_type; None = 1-dof joints only (every env of this package)

    def get_euler_pos(self, old_pos, old_vel):
        """mujoco_env.py:169-195 on the host, for callers that use the position rule by itself (the kernels apply it per substep)"""
        jt = self.JNT_TYPE if self.JNT_TYPE is not None else [JNT_SLIDE] * self.NQ  # slide and hinge joints share the rule
        return euler_position_rule(old_pos, old_vel, jt, self.real_time_scale)

    def transform_state_to_obs(self, batch_state):
        pos, vel = batch_state
        return np.concatenate([pos, vel], axis=1)  # mujoco_env.py:142-144

    def _check_single_action(self, action):
        a = np.asarray(action, dtype=np.float32)
        if a.shape != self.action_space.shape:  # gym MujocoEnv.do_simulation
            raise ValueError(f"Action dimension mismatch. Expected {self.action_space.shape}, found {a.shape}")
        return a
