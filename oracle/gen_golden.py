#!/usr/bin/env python3
"""Generate the golden parity fixtures under tests/golden/ from the *real* reference.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container, where the read-only
reference checkout lives at /root/reference; the GPU box never sees the reference and
only consumes the small ``.npz`` files this script writes.

How the reference is imported: ``import emei`` pulls third-party
packages that are absent offline (gym, h5py, pygame, mujoco).  The arithmetic on the
classic-control path (``emei/envs/classic_control/{base_control,cartpole}.py``,
``emei/core.py``) and the reward/terminal/euler-position/noise/graph helpers of the
MuJoCo-backed classes is entirely first-party numpy/math, so we register inert
``sys.modules`` stand-ins for the *plumbing* of those packages (a ``gym.Env`` that seeds
``np_random`` exactly like gym 0.26 = ``Generator(PCG64(SeedSequence(seed)))``,
``spaces.Discrete/Box``, no-op ``register``) and import the unmodified reference files
from where they lie.  Nothing from the reference is copied; only inputs and the outputs
the reference computed are stored.

MuJoCo's ``mj_step`` itself is NOT available (no libmujoco in the image) so next-state
vectors of the MuJoCo-backed bodies are not generated here: parity of those dynamics is
"unpinned" (see DESIGN.md).  What *is* generated for them are the first-party pieces:
reward/terminal functions, the forward-Euler position rule, the angle wrap, the
init-noise routine and the transition-graph closure, called as unbound methods on a
minimal stand-in ``self``.

Usage:  python oracle/gen_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import math
import os
import sys
import types
from types import SimpleNamespace as NS

import numpy as np

sys.dont_write_bytecode = True


# --------------------------------------------------------------------------- stubs
def _install_plumbing_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Env:
        """gym 0.26 ``Env``: only the seeding behaviour of ``reset`` matters here."""

        _np_random = None
        metadata = {}

        @property
        def np_random(self):
            if self._np_random is None:
                self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence()))
            return self._np_random

        @np_random.setter
        def np_random(self, v):
            self._np_random = v

        def reset(self, *, seed=None, options=None):
            if seed is not None:
                self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

    class Space:
        def __init__(self, shape=(), dtype=None):
            self.shape, self.dtype = tuple(shape), dtype

    class Discrete(Space):
        def __init__(self, n):
            super().__init__((), np.int64)
            self.n = int(n)

        def contains(self, x):
            if isinstance(x, int):
                v = x
            elif isinstance(x, (np.generic, np.ndarray)) and np.issubdtype(np.asarray(x).dtype, np.integer) and np.asarray(x).shape == ():
                v = int(x)
            else:
                return False
            return 0 <= v < self.n

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            low = np.asarray(low, dtype=dtype)
            high = np.asarray(high, dtype=dtype)
            if shape is None:
                shape = low.shape
            super().__init__(shape, dtype)
            self.low = np.broadcast_to(low, shape)
            self.high = np.broadcast_to(high, shape)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    class EzPickle:
        def __init__(self, *a, **k):
            pass

    class MujocoEnv(Env):
        def __init__(self, *a, **k):
            raise RuntimeError("MuJoCo is not available offline; construct nothing MuJoCo-backed")

    class DependencyNotInstalled(Exception):
        pass

    gym = mod("gym", Env=Env)
    gym.spaces = mod("gym.spaces", Space=Space, Discrete=Discrete, Box=Box)
    gym.logger = mod("gym.logger", warn=lambda *a, **k: None)
    gym.error = mod("gym.error", DependencyNotInstalled=DependencyNotInstalled)
    gym.utils = mod("gym.utils", EzPickle=EzPickle)
    gym.envs = mod("gym.envs")
    gym.envs.registration = mod(
        "gym.envs.registration",
        registry={},
        register=lambda **k: None,
        make=lambda *a, **k: None,
        spec=lambda *a, **k: None,
        load_env_plugins=lambda *a, **k: None,
    )
    gym.envs.mujoco = mod("gym.envs.mujoco")
    gym.envs.mujoco.mujoco_env = mod("gym.envs.mujoco.mujoco_env", MujocoEnv=MujocoEnv)
    gym.wrappers = mod("gym.wrappers")
    mod("h5py", Dataset=type("Dataset", (), {}), File=None)
    pg = mod("pygame")
    pg.gfxdraw = mod("pygame.gfxdraw")
    mod("mujoco")


def import_reference(ref_root):
    _install_plumbing_stubs()
    sys.path.insert(0, ref_root)
    import emei  # noqa: F401  (the unmodified reference package)
    from emei.envs.classic_control import cartpole as cp
    from emei.envs.mujoco import inverted_pendulum as ip
    from emei.envs.mujoco import half_cheetah as hc
    from emei.envs.mujoco import mujoco_env as me
    from emei.envs.mujoco import inverted_double_pendulum as idp
    from emei.envs.mujoco import hopper as hp
    from emei import core

    return NS(cp=cp, ip=ip, hc=hc, me=me, core=core, idp=idp, hp=hp)


# --------------------------------------------------------------------------- inputs
def wide_states(rng, n, swingup):
    """States covering the regimes the envs visit plus threshold/NaN/Inf rows."""
    s = np.empty((n, 4))
    s[:, 0] = rng.uniform(-6.0, 6.0, n)
    s[:, 1] = rng.normal(0.0, 3.0, n)
    s[:, 2] = rng.uniform(-4 * math.pi, 4 * math.pi, n) if swingup else rng.uniform(-0.45, 0.45, n)
    s[:, 3] = rng.normal(0.0, 6.0, n)
    k = n // 8
    # tight initial-state-like rows
    s[:k] = rng.uniform(-0.05, 0.05, (k, 4))
    if swingup:
        s[:k, 2] += math.pi
    # rows that land near the terminal thresholds after one step
    thr = 5.0 if swingup else 2.4
    s[k : 2 * k, 0] = np.sign(rng.standard_normal(k)) * (thr + rng.uniform(-0.2, 0.2, k))
    if not swingup:
        th = 12 * 2 * math.pi / 360
        s[2 * k : 3 * k, 2] = np.sign(rng.standard_normal(k)) * (th + rng.uniform(-0.05, 0.05, k))
        s[2 * k : 3 * k, 0] = rng.uniform(-1.0, 1.0, k)
    # large spinning angles (SwingUp spins for hundreds of radians)
    if swingup:
        s[3 * k : 4 * k, 2] = rng.uniform(-600.0, 600.0, k)
        s[3 * k : 4 * k, 3] = rng.normal(0.0, 15.0, k)
    # non-finite rows
    s[-1] = [np.nan, 0.1, 0.2, 0.3]
    s[-2] = [0.1, np.inf, 0.2, 0.3]
    s[-3] = [0.1, 0.2, np.nan, 0.3]
    s[-4] = [-np.inf, 0.2, 0.1, 0.3]
    s[-5] = [0.1, 0.2, 0.3, -np.inf]
    return s


def gen_cartpole(ref, out):
    envs = {"swingup": ref.cp.CartPoleSwingUpEnv, "balancing": ref.cp.CartPoleBalancingEnv}
    data = {}
    rng = np.random.default_rng(20240)

    # (A) one-step pairs over a wide state distribution -------------------------------
    n = 1024
    for name, cls in envs.items():
        s0 = wide_states(rng, n, name == "swingup")
        act = rng.integers(2, size=n)
        data[f"onestep_{name}_state"] = s0
        data[f"onestep_{name}_action"] = act.astype(np.int64)
        for fr, dt in ((1, 0.02), (4, 0.02), (2, 0.01)):
            env = cls(freq_rate=fr, real_time_scale=dt)
            env.reset(seed=0)
            nxt = np.empty((n, 4))
            rew = np.empty(n)
            term = np.empty(n, dtype=bool)
            raised = np.zeros(n, dtype=bool)
            with np.errstate(all="ignore"):
                for i in range(n):
                    env.state = s0[i].copy()
                    try:
                        o, r, t, tr, info = env.step(int(act[i]))
                    except (ValueError, OverflowError):
                        # math.cos(+-inf) raises "math domain error" (cartpole.py:52): the reference
                        # has no result for this row; recorded so the parity tests skip it.
                        raised[i] = True
                        nxt[i], rew[i], term[i] = np.nan, np.nan, True
                        continue
                    assert tr is False and info == {}
                    nxt[i], rew[i], term[i] = o, r, t
            tag = f"onestep_{name}_fr{fr}_dt{dt}"
            data[tag + "_raised"] = raised
            data[tag + "_next"] = nxt
            data[tag + "_reward"] = rew
            data[tag + "_terminal"] = term

    # (B) seeded 1000-step open-loop trajectories (no reset after terminal: the reference
    #     keeps integrating, base_control.py:61-83) ---------------------------------------
    T = 1000
    for name, cls in envs.items():
        for fr in (1, 4):
            for seed in range(4):
                env = cls(freq_rate=fr, real_time_scale=0.02)
                o0, info = env.reset(seed=seed)
                acts = np.random.default_rng(1000 + seed).integers(2, size=T)
                traj = np.empty((T + 1, 4))
                rew = np.empty(T)
                term = np.empty(T, dtype=bool)
                traj[0] = o0
                with np.errstate(all="ignore"):
                    for t in range(T):
                        o, r, d, _, _ = env.step(int(acts[t]))
                        traj[t + 1], rew[t], term[t] = o, r, d
                tag = f"traj_{name}_fr{fr}_seed{seed}"
                data[tag + "_actions"] = acts.astype(np.int64)
                data[tag + "_states"] = traj
                data[tag + "_reward"] = rew
                data[tag + "_terminal"] = term

    # (C) reset / batch init states -----------------------------------------------------
    for name, cls in envs.items():
        rs = []
        for seed in range(16):
            env = cls()
            o, _ = env.reset(seed=seed)
            rs.append(o)
        data[f"reset_{name}_seeds0_15"] = np.asarray(rs)
        env = cls()
        env.reset(seed=7)
        data[f"batchinit_{name}_seed7_B8"] = env.get_batch_init_state(8)
        env.reset(seed=7)
        data[f"batchinitobs_{name}_seed7_B8"] = env.get_batch_init_obs(8)

    # (D) stand-alone batched reward / terminal ----------------------------------------
    B = 512
    for name, cls in envs.items():
        env = cls()
        obs = wide_states(rng, B, name == "swingup")
        thr = 5.0 if name == "swingup" else 2.4
        # exact-threshold rows: strict '<' must classify them terminal
        obs[10, 0] = thr
        obs[11, 0] = -thr
        obs[12, 0] = np.nextafter(thr, 0.0)
        obs[13, 0] = np.nextafter(thr, 10.0)
        if name == "balancing":
            th = env.theta_threshold_radians
            obs[10:14, 0] = 0.0
            obs[14, 2], obs[14, 0] = th, 0.0
            obs[15, 2], obs[15, 0] = -th, 0.0
            obs[16, 2], obs[16, 0] = np.nextafter(th, 0.0), 0.0
            obs[17, 0], obs[17, 2] = 2.4, 0.0
            obs[18, 0], obs[18, 2] = np.nextafter(2.4, 0.0), 0.0
        with np.errstate(all="ignore"):
            data[f"batch_{name}_obs"] = obs
            data[f"batch_{name}_reward"] = np.asarray(env.get_batch_reward(obs), dtype=np.float64)
            data[f"batch_{name}_terminal"] = env.get_batch_terminal(obs)
        data[f"const_{name}_x_threshold"] = np.float64(env.x_threshold)
        data[f"const_{name}_theta_threshold"] = np.float64(env.theta_threshold_radians)

    np.savez_compressed(os.path.join(out, "cartpole_golden.npz"), **data)
    return data


def gen_cartpole_rk4(ref, out):
    """The classic-control RK4 branch (base_control.py:165-170).  `step()` never reaches it (it calls ODE_approximation without
    `method`, :73), so it is called directly, exactly as a user of the function would: ODE_approximation(env._dsdt, s_aug, dt,
    steps, method="rk4").  Rows: one-step pairs over the wide state distribution of gen_cartpole, and seeded 1000-step open-loop
    trajectories that follow step()'s own sequence (:61-83) with only the `method` argument added; reward / terminal are the
    envs' own get_batch_reward / get_batch_terminal on the new state."""
    from emei.envs.classic_control.base_control import ODE_approximation

    envs = {"swingup": ref.cp.CartPoleSwingUpEnv, "balancing": ref.cp.CartPoleBalancingEnv}
    data = {}
    rng = np.random.default_rng(20245)

    def rk4_step(env, action):
        action = np.asarray(action)
        pre_obs = env.state.copy()
        s_aug = np.append(env.state, env._extract_action(action))
        env.state = ODE_approximation(env._dsdt, s_aug, env.real_time_scale, env.freq_rate, method="rk4")[: len(env.state)]
        obs = env.state.copy()
        r = env.get_batch_reward(obs[None], pre_obs[None], action[None])[0, 0]
        t = env.get_batch_terminal(obs[None], pre_obs[None], action[None])[0, 0]
        return obs, r, t

    n = 1024
    for name, cls in envs.items():
        s0 = wide_states(rng, n, name == "swingup")
        act = rng.integers(2, size=n)
        data[f"onestep_{name}_state"] = s0
        data[f"onestep_{name}_action"] = act.astype(np.int64)
        for fr, dt in ((1, 0.02), (4, 0.02), (2, 0.01)):
            env = cls(freq_rate=fr, real_time_scale=dt)
            env.reset(seed=0)
            nxt = np.empty((n, 4))
            rew = np.empty(n)
            term = np.empty(n, dtype=bool)
            raised = np.zeros(n, dtype=bool)
            with np.errstate(all="ignore"):
                for i in range(n):
                    env.state = s0[i].copy()
                    try:
                        o, r, t = rk4_step(env, int(act[i]))
                    except (ValueError, OverflowError):  # math.cos(+-inf) / float ** 2 overflow (cartpole.py:51-53): no result
                        raised[i] = True
                        nxt[i], rew[i], term[i] = np.nan, np.nan, True
                        continue
                    nxt[i], rew[i], term[i] = o, r, t
            tag = f"onestep_{name}_fr{fr}_dt{dt}"
            data[tag + "_raised"] = raised
            data[tag + "_next"] = nxt
            data[tag + "_reward"] = rew
            data[tag + "_terminal"] = term

    T = 1000
    for name, cls in envs.items():
        for fr in (1, 4):
            for seed in range(4):
                env = cls(freq_rate=fr, real_time_scale=0.02)
                o0, _ = env.reset(seed=seed)
                acts = np.random.default_rng(2000 + seed).integers(2, size=T)
                traj = np.empty((T + 1, 4))
                rew = np.empty(T)
                term = np.empty(T, dtype=bool)
                traj[0] = o0
                with np.errstate(all="ignore"):
                    for t in range(T):
                        traj[t + 1], rew[t], term[t] = rk4_step(env, int(acts[t]))
                tag = f"traj_{name}_fr{fr}_seed{seed}"
                data[tag + "_actions"] = acts.astype(np.int64)
                data[tag + "_states"] = traj
                data[tag + "_reward"] = rew
                data[tag + "_terminal"] = term
    np.savez_compressed(os.path.join(out, "cartpole_rk4_golden.npz"), **data)
    return data


def gen_mujoco_firstparty(ref, out):
    """First-party helpers of the MuJoCo-backed classes, called unbound on a stand-in self."""
    data = {}
    rng = np.random.default_rng(777)
    ip = ref.ip
    jnt_range = np.array([[-2.0, 2.0], [-math.pi / 2, math.pi / 2]])  # assets/inverted_pendulum.xml:14,17
    fake = NS(model=NS(jnt_range=jnt_range))
    B = 512
    obs = np.empty((B, 4))
    obs[:, 0] = rng.uniform(-2.5, 2.5, B)
    obs[:, 1] = rng.uniform(-math.pi, math.pi, B)
    obs[:, 2] = rng.normal(0, 3, B)
    obs[:, 3] = rng.normal(0, 6, B)
    obs[0, 0], obs[1, 0] = 2.0, -2.0
    obs[2, 0], obs[3, 0] = np.nextafter(2.0, 0.0), np.nextafter(-2.0, 0.0)
    obs[4, 1] = math.acos(0.9)
    obs[5, 1] = math.pi / 2
    obs[6, 1] = -math.pi / 2
    obs[7] = [0.0, 0.0, np.nan, 0.0]
    obs[8] = [0.0, 0.0, 0.0, np.inf]
    obs[9] = [np.nan, 0.0, 0.0, 0.0]
    obs[10] = [0.0, np.inf, 0.0, 0.0]
    obs[11] = [0.0, 0.0, 0.0, 0.0]
    data["ip_obs"] = obs
    with np.errstate(all="ignore"):
        for nm, cls in (
            ("rebound_balancing", ip.ReboundInvertedPendulumBalancingEnv),
            ("boundary_balancing", ip.BoundaryInvertedPendulumBalancingEnv),
            ("rebound_swingup", ip.ReboundInvertedPendulumSwingUpEnv),
            ("boundary_swingup", ip.BoundaryInvertedPendulumSwingUpEnv),
        ):
            data[f"ip_{nm}_reward"] = np.asarray(cls.get_batch_reward(fake, obs), dtype=np.float64)
            data[f"ip_{nm}_terminal"] = cls.get_batch_terminal(fake, obs)

    # angle wrap of current_obs (inverted_pendulum.py:45-49)
    sv = np.column_stack(
        [rng.uniform(-2, 2, 256), rng.uniform(-40, 40, 256), rng.normal(0, 3, 256), rng.normal(0, 6, 256)]
    )
    sv[0, 1], sv[1, 1], sv[2, 1], sv[3, 1] = math.pi, -math.pi, 3 * math.pi, 0.0
    wrapped = np.empty_like(sv)
    for i in range(len(sv)):
        f = NS(state_vector=lambda i=i: sv[i])
        wrapped[i] = ip.BaseInvertedPendulumEnv.current_obs.fget(f)
    data["ip_wrap_in"] = sv
    data["ip_wrap_out"] = wrapped

    # transition graph + closure (core.py:142-161; graph inverted_pendulum.py:39-41)
    g = np.array([[0, 0, 0, 0], [0, 0, 1, 1], [1, 0, 0, 0], [0, 1, 1, 1], [0, 0, 1, 1]])
    f = NS(_transition_graph=g, observation_space=NS(shape=(4,)), action_space=NS(shape=(1,)))
    for rt in (1, 2, 3, 5):
        data[f"ip_graph_repeat{rt}"] = np.asarray(ref.core.EmeiEnv.get_transition_graph(f, rt))

    # forward-Euler position rule (mujoco_env.py:169-195), slide=2 / hinge=3 joints
    for nm, nq in (("ip", 2), ("cheetah", 9)):
        jt = [2, 3] if nq == 2 else [2, 2, 3, 3, 3, 3, 3, 3, 3]
        for dt in (0.02, 0.002):
            f = NS(model=NS(jnt_type=jt), real_time_scale=dt)
            qp = rng.normal(0, 1, (64, nq))
            qv = rng.normal(0, 5, (64, nq))
            newp = np.stack([ref.me.EmeiMujocoEnv.get_euler_pos(f, qp[i], qv[i]) for i in range(64)])
            data[f"euler_{nm}_dt{dt}_qpos"] = qp
            data[f"euler_{nm}_dt{dt}_qvel"] = qv
            data[f"euler_{nm}_dt{dt}_newpos"] = newp

    # init-noise routine for B=1 (mujoco_env.py:197-249); global np.random, legacy MT19937
    for nm, nq, sig in (("ip", 2, 5e-3), ("cheetah", 9, 0.1)):
        jt = [2, 3] if nq == 2 else [2, 2, 3, 3, 3, 3, 3, 3, 3]
        f = NS(model=NS(jnt_type=jt))
        outs_p, outs_v = [], []
        for seed in range(8):
            np.random.seed(seed)
            p, v = ref.me.EmeiMujocoEnv.additive_gaussian_noise(f, np.zeros((1, nq)), np.zeros((1, nq)), sig)
            outs_p.append(p[0])
            outs_v.append(v[0])
        data[f"noise_{nm}_pos_seeds0_7"] = np.asarray(outs_p)
        data[f"noise_{nm}_vel_seeds0_7"] = np.asarray(outs_v)

    # HalfCheetah reward / terminal (half_cheetah.py:59-67); dt = timestep*frame_skip (gym MujocoEnv.dt)
    hc = ref.hc
    f = NS(_forward_reward_weight=1.0, _ctrl_cost_weight=0.1, dt=0.002 * 4)
    Bc = 256
    o = rng.normal(0, 1, (Bc, 18))
    po = o + rng.normal(0, 0.01, (Bc, 18))
    a = rng.uniform(-1, 1, (Bc, 6))
    o[3, 5] = np.nan
    o[4, 17] = np.inf
    # step() semantics: B = 1 per call (the batch form sums the control cost over the whole batch)
    r1 = np.stack([hc.HalfCheetahRunningEnv.get_batch_reward(f, o[i : i + 1], po[i : i + 1], a[i : i + 1])[0, 0] for i in range(Bc)])
    with np.errstate(all="ignore"):
        data["cheetah_obs"] = o
        data["cheetah_pre_obs"] = po
        data["cheetah_action"] = a
        data["cheetah_reward_B1"] = r1
        data["cheetah_reward_batchquirk"] = hc.HalfCheetahRunningEnv.get_batch_reward(f, o, po, a)
        data["cheetah_terminal"] = hc.HalfCheetahRunningEnv.get_batch_terminal(f, o)
        fw = NS(_forward_reward_weight=2.5, _ctrl_cost_weight=0.03, dt=0.002 * 4)  # non-default weights (half_cheetah.py:23-24)
        data["cheetah_reward_B1_w2p5_c0p03"] = np.stack([hc.HalfCheetahRunningEnv.get_batch_reward(fw, o[i : i + 1], po[i : i + 1], a[i : i + 1])[0, 0] for i in range(Bc)])

    # env_params_name string KAT (core.py:56-58; test/test_core.py:9-14)
    names = []
    for params in ({"a": 3, "b": 5, "d": 0.33, "c": "c"}, dict(freq_rate=1, real_time_scale=0.02), dict(freq_rate=4, real_time_scale=0.002, integrator="euler")):
        names.append(ref.core.OfflineEnv.env_params_name.fget(NS(env_params=params)))
    data["env_params_names"] = np.asarray(names)

    np.savez_compressed(os.path.join(out, "mujoco_firstparty_golden.npz"), **data)
    return data


def gen_dpend_firstparty(ref, out):
    """InvertedDoublePendulum: the four reward/terminal functions and the (quirky) observation wrap,
    called unbound on a stand-in self (slider range from assets/inverted_double_pendulum.xml:31)."""
    data = {}
    rng = np.random.default_rng(4242)
    idp = ref.idp
    fake = NS(model=NS(jnt_range=np.array([[-3.0, 3.0], [0.0, 0.0], [0.0, 0.0]])))
    B = 512
    obs = np.empty((B, 6))
    obs[:, 0] = rng.uniform(-3.5, 3.5, B)
    obs[:, 1:3] = rng.uniform(-math.pi, math.pi, (B, 2))
    obs[:, 3:] = rng.normal(0, 4, (B, 3))
    obs[: B // 4, 1:3] = rng.normal(0, 0.3, (B // 4, 2))  # near upright: y close to the 1.5 / 0 thresholds
    obs[0, 0], obs[1, 0] = 3.0, -3.0
    obs[2, 0], obs[3, 0] = np.nextafter(3.0, 0.0), np.nextafter(-3.0, 0.0)
    obs[4] = [0.0, np.nan, 0.0, 0.0, 0.0, 0.0]
    obs[5] = [0.0, 0.0, 0.0, 0.0, np.inf, 0.0]
    obs[6] = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    data["dp_obs"] = obs
    with np.errstate(all="ignore"):
        for nm, cls in (
            ("rebound_balancing", idp.ReboundInvertedDoublePendulumBalancingEnv),
            ("boundary_balancing", idp.BoundaryInvertedDoublePendulumBalancingEnv),
            ("rebound_swingup", idp.ReboundInvertedDoublePendulumSwingUpEnv),
            ("boundary_swingup", idp.BoundaryInvertedDoublePendulumSwingUpEnv),
        ):
            data[f"dp_{nm}_reward"] = np.asarray(cls.get_batch_reward(fake, obs), dtype=np.float64)
            data[f"dp_{nm}_terminal"] = cls.get_batch_terminal(fake, obs)
    sv = np.column_stack([rng.uniform(-3, 3, 256), rng.uniform(-30, 30, 256), rng.uniform(-30, 30, 256), rng.normal(0, 3, (256, 3))])
    sv[0, 1:3] = [0.0, math.pi]
    wrapped = np.empty_like(sv)
    for i in range(len(sv)):
        wrapped[i] = idp.BaseInvertedDoublePendulumEnv.current_obs.fget(NS(state_vector=lambda i=i: sv[i]))
    data["dp_wrap_in"] = sv
    data["dp_wrap_out"] = wrapped
    np.savez_compressed(os.path.join(out, "dpend_firstparty_golden.npz"), **data)
    return data


def gen_hopper_firstparty(ref, out):
    """Hopper: is_healthy / reward / terminal (hopper.py:79-106) called unbound on a stand-in self with
    the constructor defaults (:25-31), and the tuple / dict forms of additive_gaussian_noise
    (mujoco_env.py:218-227) for B = 1 on the 6-joint chain (test/test_envs/test_mujoco/test_hopper.py:38-53)."""
    data = {}
    rng = np.random.default_rng(9091)
    hp = ref.hp
    cls = hp.HopperRunningEnv
    f = NS(_forward_reward_weight=1.0, _ctrl_cost_weight=1e-3, _healthy_reward=1.0, _terminate_when_unhealthy=True,
           _healthy_state_range=(-100.0, 100.0), _healthy_z_range=(0.7, float("inf")), _healthy_angle_range=(-0.2, 0.2),
           dt=0.002 * 4)
    f.is_healthy = lambda o: cls.is_healthy(f, o)
    B = 256
    o = rng.normal(0, 1, (B, 12))
    o[:, 1] = rng.uniform(0.3, 1.6, B)          # z around the 0.7 threshold
    o[:, 2] = rng.uniform(-0.5, 0.5, B)         # angle around +-0.2 (never applied, see hopper.py:91)
    o[: B // 8, 6:] = rng.normal(0, 80, (B // 8, 6))  # some velocities beyond +-100
    o[0, 1], o[1, 1] = 0.7, np.nextafter(0.7, 1.0)
    o[2, 5], o[3, 5] = 100.0, np.nextafter(100.0, 0.0)
    o[4, 7] = np.nan
    o[5, 1] = np.inf
    po = o + rng.normal(0, 0.01, (B, 12))
    a = rng.uniform(-1, 1, (B, 3))
    with np.errstate(all="ignore"):
        data["hopper_obs"], data["hopper_pre_obs"], data["hopper_action"] = o, po, a
        data["hopper_is_healthy"] = np.asarray(cls.is_healthy(f, o.copy()))
        # step() semantics: B = 1 per call (np.sum(np.square(action)) has no axis, hopper.py:98)
        data["hopper_reward_B1"] = np.stack([cls.get_batch_reward(f, o[i : i + 1].copy(), po[i : i + 1], a[i : i + 1])[0, 0] for i in range(B)])
        data["hopper_terminal"] = np.asarray(cls.get_batch_terminal(f, o.copy()))
        # what get_batch_reward EXECUTES for B > 1: the control cost of the whole batch in every row (hopper.py:98)
        data["hopper_reward_batchquirk"] = np.asarray(cls.get_batch_reward(f, o.copy(), po, a))
        # the reference's own test inputs (test_hopper.py:9-13)
        data["hopper_is_healthy_ones"] = np.asarray(cls.is_healthy(f, np.ones([128, 12])))
        data["hopper_is_healthy_101"] = np.asarray(cls.is_healthy(f, np.ones([128, 12]) * 101))
    # the same functions with NON-default constructor parameters (hopper.py:25-30): terminate_when_unhealthy = False is
    # what makes the env terminate; custom weights, healthy reward and ranges
    f2 = NS(_forward_reward_weight=2.0, _ctrl_cost_weight=5e-3, _healthy_reward=0.5, _terminate_when_unhealthy=False,
            _healthy_state_range=(-50.0, 60.0), _healthy_z_range=(0.8, 1.5), _healthy_angle_range=(-0.2, 0.2), dt=0.002 * 4)
    f2.is_healthy = lambda o_: cls.is_healthy(f2, o_)
    with np.errstate(all="ignore"):
        data["hopper_custom_params"] = np.array([2.0, 5e-3, 0.5, 0.0, -50.0, 60.0, 0.8, 1.5])
        data["hopper_custom_is_healthy"] = np.asarray(cls.is_healthy(f2, o.copy()))
        data["hopper_custom_reward_B1"] = np.stack([cls.get_batch_reward(f2, o[i : i + 1].copy(), po[i : i + 1], a[i : i + 1])[0, 0] for i in range(B)])
        data["hopper_custom_terminal"] = np.asarray(cls.get_batch_terminal(f2, o.copy()))
    # additive_gaussian_noise for B = 1: float, tuple and dict parameters
    fm = NS(model=NS(jnt_type=[2, 2, 3, 3, 3, 3]))
    q0 = np.array([[0.0, 1.25, 0.0, 0.0, 0.0, 0.0]])
    for nm, prm in (("float", 5e-3), ("tuple", (0.01, 0.03)), ("dict0", {0: (0.1, 0.2)}), ("dict2", {2: (0.1, 0.2)})):
        ps, vs = [], []
        for seed in range(4):
            np.random.seed(seed)
            p, v = ref.me.EmeiMujocoEnv.additive_gaussian_noise(fm, q0.copy(), np.zeros((1, 6)), prm)
            ps.append(p[0]), vs.append(v[0])
        data[f"hopper_noise_{nm}_pos"], data[f"hopper_noise_{nm}_vel"] = np.asarray(ps), np.asarray(vs)
    np.savez_compressed(os.path.join(out, "hopper_firstparty_golden.npz"), **data)
    return data


def gen_lagrange(ref_root, out):
    """Equations of motion of the cart + n-pole chain from the reference's own derivation tool
    (emei/envs/classic_control/auxiliary/lagrange_eqs.py:12-69, SymPy Lagrangian; uniform rods of half-length
    l_i, inertia m l^2 / 3 about the centre, pole i+1 hinged at the tip of pole i, angles from the upright,
    relative joint angles): accelerations for random parameters / states / forces, n = 1 and n = 2.  They pin
    the STRUCTURE of the oracle's InvertedPendulum / InvertedDoublePendulum dynamics (evaluated with rod
    parameters instead of the xml's capsules) to first-party reference code."""
    import importlib.util
    import types

    import sympy as sp

    ipy, disp = types.ModuleType("IPython"), types.ModuleType("IPython.display")
    disp.display, disp.Latex = (lambda *a, **k: None), object
    sys.modules.setdefault("IPython", ipy)
    sys.modules.setdefault("IPython.display", disp)
    path = os.path.join(ref_root, "emei", "envs", "classic_control", "auxiliary", "lagrange_eqs.py")
    spec = importlib.util.spec_from_file_location("emei_lagrange_eqs", path)
    L = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(L)  # the unmodified reference file
    data = {}
    rng = np.random.default_rng(2024)
    for n in (1, 2):
        eqs, dyn = L.cartpole(n)
        t = sp.Symbol("t")
        q = sp.symbols(f"q0:{n + 1}")
        v = sp.symbols(f"v0:{n + 1}")
        a = sp.symbols(f"a0:{n + 1}")
        rep = []
        for k, f in enumerate(dyn):
            rep.append((sp.diff(f, t, 2), a[k]))
        for k, f in enumerate(dyn):
            rep.append((sp.diff(f, t), v[k]))
        for k, f in enumerate(dyn):
            rep.append((f, q[k]))
        exprs = [e.subs(rep) for e in eqs]
        A, b = sp.linear_eq_to_matrix(exprs, list(a))
        params = [sp.Symbol("g", real=True), sp.Symbol("F", real=True), sp.Symbol("M", real=True)] \
            + list(sp.symbols(f"m:{n}", real=True)) + list(sp.symbols(f"l:{n}", real=True))
        fn = sp.lambdify(params + list(q) + list(v), (A, b), "numpy")
        B = 96
        g = np.full(B, 9.81)
        F = rng.uniform(-300, 300, B)
        M = rng.uniform(1.0, 12.0, B)
        m0 = rng.uniform(0.5, 6.0, B)
        l0 = rng.uniform(0.1, 0.5, B)
        ms, ls = [m0] * n, [l0] * n  # identical poles (the xml's two poles are identical, :35-38)
        qs = np.column_stack([rng.uniform(-2, 2, B)] + [rng.uniform(-np.pi, np.pi, B) for _ in range(n)])
        vs = np.column_stack([rng.normal(0, 2, B)] + [rng.normal(0, 5, B) for _ in range(n)])
        acc = np.empty((B, n + 1))
        for i in range(B):
            Ai, bi = fn(g[i], F[i], M[i], *[x[i] for x in ms], *[x[i] for x in ls], *qs[i], *vs[i])
            acc[i] = np.linalg.solve(np.asarray(Ai, float), np.asarray(bi, float).reshape(-1))
        data[f"n{n}_g"], data[f"n{n}_F"], data[f"n{n}_M"], data[f"n{n}_m"], data[f"n{n}_l"] = g, F, M, m0, l0
        data[f"n{n}_q"], data[f"n{n}_v"], data[f"n{n}_acc"] = qs, vs, acc
    np.savez_compressed(os.path.join(out, "lagrange_golden.npz"), **data)
    return data


# --------------------------------------------------------------------------- model constants (XML data)
def gen_model_constants(ref_root, out):
    """The reference's only data for the MuJoCo-backed bodies: emei/envs/mujoco/assets/{inverted_pendulum,
    inverted_double_pendulum,half_cheetah,hopper}.xml, parsed with xml.etree into flat numeric arrays (defaults of
    <default> resolved, nothing derived).  tests/test_model_constants.py derives masses / inertias / link-frame geometry
    from these arrays with MuJoCo's documented compiler rules and compares the oracle's tables and the kernels'
    compile-time tables with them, so that a mistyped constant on either side is found (missing #4).
    Entries the XML does not state are NaN (the consumer applies MuJoCo's documented default)."""
    import xml.etree.ElementTree as ET

    def nums(txt, n=None):
        v = [float(x) for x in txt.split()] if txt is not None else []
        if n is not None:
            v = (v + [np.nan] * n)[:n]
        return v

    data = {}
    for tag, fname in (("ip", "inverted_pendulum.xml"), ("dp", "inverted_double_pendulum.xml"), ("ch", "half_cheetah.xml"),
                       ("hp", "hopper.xml")):
        root = ET.parse(os.path.join(ref_root, "emei", "envs", "mujoco", "assets", fname)).getroot()
        comp, opt = root.find("compiler"), root.find("option")
        dflt = root.find("default")
        dj = dict(dflt.find("joint").attrib) if dflt is not None and dflt.find("joint") is not None else {}
        dg = dict(dflt.find("geom").attrib) if dflt is not None and dflt.find("geom") is not None else {}
        dm = dict(dflt.find("motor").attrib) if dflt is not None and dflt.find("motor") is not None else {}
        data[f"{tag}_angle_degree"] = np.array(comp.get("angle", "degree") == "degree")  # MuJoCo's default unit is degree
        data[f"{tag}_coordinate_global"] = np.array(comp.get("coordinate", "local") == "global")
        data[f"{tag}_settotalmass"] = np.array(float(comp.get("settotalmass", "nan")))
        data[f"{tag}_gravity"] = np.array(nums(opt.get("gravity"), 3))
        data[f"{tag}_timestep"] = np.array(float(opt.get("timestep", "nan")))
        bodies, joints, geoms = [], [], []

        def walk(el, parent):
            for child in el:
                if child.tag == "body":
                    bodies.append((child.get("name"), parent, nums(child.get("pos"), 3), nums(child.get("quat"), 4)))
                    walk(child, len(bodies) - 1)
                elif child.tag == "joint":
                    a = dict(dj)
                    a.update(child.attrib)
                    joints.append((a.get("name"), parent, a))
                elif child.tag == "geom":
                    a = dict(dg)
                    a.update(child.attrib)
                    geoms.append((a.get("name"), parent, a))

        walk(root.find("worldbody"), -1)
        data[f"{tag}_body_names"] = np.array([b[0] for b in bodies])
        data[f"{tag}_body_parent"] = np.array([b[1] for b in bodies])
        data[f"{tag}_body_pos"] = np.array([b[2] for b in bodies])
        data[f"{tag}_body_quat"] = np.array([b[3] for b in bodies])
        data[f"{tag}_joint_names"] = np.array([j[0] for j in joints])
        data[f"{tag}_joint_body"] = np.array([j[1] for j in joints])
        data[f"{tag}_joint_is_hinge"] = np.array([j[2].get("type", "hinge") == "hinge" for j in joints])
        data[f"{tag}_joint_axis"] = np.array([nums(j[2].get("axis"), 3) for j in joints])
        data[f"{tag}_joint_pos"] = np.array([nums(j[2].get("pos"), 3) for j in joints])
        data[f"{tag}_joint_limited"] = np.array([j[2].get("limited", "false") == "true" for j in joints])
        data[f"{tag}_joint_range"] = np.array([nums(j[2].get("range"), 2) for j in joints])
        for key in ("stiffness", "damping", "armature", "margin", "ref"):
            data[f"{tag}_joint_{key}"] = np.array([float(j[2].get(key, "nan")) for j in joints])
        data[f"{tag}_joint_solreflimit"] = np.array([nums(j[2].get("solreflimit"), 2) for j in joints])
        data[f"{tag}_joint_solimplimit"] = np.array([nums(j[2].get("solimplimit"), 3) for j in joints])
        data[f"{tag}_geom_names"] = np.array([str(g[0]) for g in geoms])
        data[f"{tag}_geom_body"] = np.array([g[1] for g in geoms])
        data[f"{tag}_geom_is_capsule"] = np.array([g[2].get("type", "sphere") == "capsule" for g in geoms])
        data[f"{tag}_geom_size"] = np.array([nums(g[2].get("size"), 3) for g in geoms])
        data[f"{tag}_geom_fromto"] = np.array([nums(g[2].get("fromto"), 6) for g in geoms])
        data[f"{tag}_geom_pos"] = np.array([nums(g[2].get("pos"), 3) for g in geoms])
        data[f"{tag}_geom_axisangle"] = np.array([nums(g[2].get("axisangle"), 4) for g in geoms])
        data[f"{tag}_geom_quat"] = np.array([nums(g[2].get("quat"), 4) for g in geoms])
        data[f"{tag}_geom_friction"] = np.array([nums(g[2].get("friction"), 3) for g in geoms])
        data[f"{tag}_geom_solref"] = np.array([nums(g[2].get("solref"), 2) for g in geoms])
        data[f"{tag}_geom_solimp"] = np.array([nums(g[2].get("solimp"), 3) for g in geoms])
        data[f"{tag}_geom_margin"] = np.array([float(g[2].get("margin", "nan")) for g in geoms])
        data[f"{tag}_geom_condim"] = np.array([float(g[2].get("condim", "nan")) for g in geoms])
        data[f"{tag}_geom_contype"] = np.array([float(g[2].get("contype", "nan")) for g in geoms])
        jn = [j[0] for j in joints]
        acts = []
        for m_ in root.find("actuator"):
            a = dict(dm)
            a.update(m_.attrib)
            acts.append((jn.index(a["joint"]), float(a.get("gear", "1")), nums(a.get("ctrlrange"), 2), a.get("ctrllimited", "false") == "true"))
        data[f"{tag}_act_joint"] = np.array([a[0] for a in acts])
        data[f"{tag}_act_gear"] = np.array([a[1] for a in acts])
        data[f"{tag}_act_ctrlrange"] = np.array([a[2] for a in acts])
        data[f"{tag}_act_ctrllimited"] = np.array([a[3] for a in acts])
    np.savez_compressed(os.path.join(out, "model_constants_golden.npz"), **data)
    return data


def gen_freejoint(ref, out):
    """The free-joint branch of the Euler position rule (mujoco_env.py:176-184: the qpos quaternion goes through SciPy's
    Rotation as if it were scalar-last, `as_euler("zyx", degrees=True)`, velocity * dt is added to the DEGREES, and
    `from_euler("xyz", ...)` builds the new quaternion) on models with a free joint in front of / between 1-dof joints; and what
    the noise routine (:229-237) does with a free joint: it slices ROWS of the [B, nq] arrays, so `from_quat` always raises."""
    data = {}
    rng = np.random.default_rng(4242)
    for nm, jt in (("free", [0]), ("free_hinge2", [0, 3, 3]), ("slide_free_hinge", [2, 0, 3])):
        nq = sum(7 if t == 0 else 1 for t in jt)
        nv = sum(6 if t == 0 else 1 for t in jt)
        for dt in (0.02, 0.002):
            f = NS(model=NS(jnt_type=jt), real_time_scale=dt)
            qp = rng.normal(0, 1, (64, nq))
            k = 0
            for t in jt:  # unit quaternions where a free joint keeps one (the routine normalises anyway: rows 48.. stay unnormalised)
                if t == 0:
                    qp[:48, k + 3 : k + 7] /= np.linalg.norm(qp[:48, k + 3 : k + 7], axis=1, keepdims=True)
                    k += 7
                else:
                    k += 1
            qv = rng.normal(0, 5, (64, nv))
            data[f"fj_{nm}_dt{dt}_qpos"], data[f"fj_{nm}_dt{dt}_qvel"] = qp, qv
            data[f"fj_{nm}_dt{dt}_newpos"] = np.stack([ref.me.EmeiMujocoEnv.get_euler_pos(f, qp[i], qv[i]) for i in range(64)])
    for nm, jt in (("ball", [1]), ("hinge_ball", [3, 1])):  # ball joints: NotImplementedError (:185-186)
        try:
            ref.me.EmeiMujocoEnv.get_euler_pos(NS(model=NS(jnt_type=jt), real_time_scale=0.02), np.zeros(5), np.zeros(4))
            data[f"fj_{nm}_raises"] = np.asarray("")
        except Exception as e:  # noqa: BLE001
            data[f"fj_{nm}_raises"] = np.asarray(type(e).__name__)
    for B in (1, 4):
        try:
            ref.me.EmeiMujocoEnv.additive_gaussian_noise(NS(model=NS(jnt_type=[0])), np.tile([0, 0, 0, 1.0, 0, 0, 0], (B, 1)), np.zeros((B, 6)), 0.1)
            data[f"fj_noise_B{B}_raises"] = np.asarray("")
        except Exception as e:  # noqa: BLE001
            data[f"fj_noise_B{B}_raises"] = np.asarray(type(e).__name__)
    np.savez_compressed(os.path.join(out, "freejoint_golden.npz"), **data)
    return data


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    ref = import_reference(a.ref)
    c = gen_cartpole(ref, a.out)
    rk = gen_cartpole_rk4(ref, a.out)
    m = gen_mujoco_firstparty(ref, a.out)
    d = gen_dpend_firstparty(ref, a.out)
    h = gen_hopper_firstparty(ref, a.out)
    lg = gen_lagrange(a.ref, a.out)
    mc = gen_model_constants(a.ref, a.out)
    fj = gen_freejoint(ref, a.out)
    env = ref.cp.CartPoleSwingUpEnv()
    o, _ = env.reset(seed=0)
    print("reset(seed=0):", o)
    for act in (0, 1, 1):
        print(env.step(act)[:3])
    print("cartpole keys:", len(c), " cartpole-rk4 keys:", len(rk), " mujoco-firstparty keys:", len(m), " dpend-firstparty keys:", len(d), " hopper-firstparty keys:", len(h), " lagrange keys:", len(lg), " model-constant keys:", len(mc), " free-joint keys:", len(fj))


if __name__ == "__main__":
    main()
