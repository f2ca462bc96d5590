"""ctypes front-end of the C oracle plus a NumPy restatement of the CartPole step.

TEST INFRASTRUCTURE ONLY — never imported by emei_amd/.  Parity status: CartPole pinned by
tests/golden/cartpole_golden.npz; MuJoCo-backed dynamics unpinned (see emei_oracle.c header).
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libemei_oracle.so")
_lib = None

CARTPOLE_VARIANTS = {"swingup": 0, "balancing": 1}
IP_VARIANTS = {"rebound_balancing": 0, "boundary_balancing": 1, "rebound_swingup": 2, "boundary_swingup": 3}


def build(force=False):
    """Compile the oracle with gcc (no GPU needed)."""
    srcs = [os.path.join(_HERE, f) for f in ("emei_oracle.c", "planar_oracle.c", "dpend_oracle.c", "integrators.h")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.emei_oracle_ip_model_size.restype = C.c_int
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


# --------------------------------------------------------------------------- CartPole (C)
ODE_METHODS = {"euler": 0, "rk4": 1}  # ODE_approximation(method=) of base_control.py:133-173


def cartpole_step(variant, state, action, freq_rate=1, dt=0.02, method="euler"):
    """state [n,4] float64 (copied), action [n] int -> (next_state, reward, terminal).  method: the `method` argument of
    ODE_approximation ("euler" is what step() runs, base_control.py:73; "rk4" the branch of :165-170)."""
    v = CARTPOLE_VARIANTS[variant]
    st = np.array(state, dtype=np.float64, order="C", copy=True).reshape(-1, 4)
    n = st.shape[0]
    act = np.ascontiguousarray(action, dtype=np.int32).reshape(n)
    rew = np.empty(n, np.float64)
    term = np.empty(n, np.uint8)
    lib().emei_oracle_cartpole_step_method(
        C.c_int(v), C.c_int64(n), C.c_int(int(freq_rate)), C.c_double(float(dt)), C.c_int(ODE_METHODS[method]),
        _p(st, C.c_double), _p(act, C.c_int32), _p(rew, C.c_double), _p(term, C.c_uint8))
    return st, rew, term.astype(bool)


def cartpole_rollout(variant, state0, actions, freq_rate=1, dt=0.02, method="euler"):
    """Open-loop rollout without reset (base_control.py:61-83 never resets).
    actions [T,n] -> states [T+1,n,4], reward [T,n], terminal [T,n]."""
    st = np.array(state0, dtype=np.float64).reshape(-1, 4)
    T = len(actions)
    states = np.empty((T + 1,) + st.shape)
    rew = np.empty((T, st.shape[0]))
    term = np.empty((T, st.shape[0]), bool)
    states[0] = st
    for t in range(T):
        st, rew[t], term[t] = cartpole_step(variant, st, actions[t], freq_rate, dt, method)
        states[t + 1] = st
    return states, rew, term


def cartpole_step_f32(variant, soa, action, freq_rate=1, dt=0.02):
    """float32 SoA port (the arithmetic of the HIP kernel, libm sinf/cosf). soa: 4 arrays, updated in place."""
    v = CARTPOLE_VARIANTS[variant]
    n = soa[0].shape[0]
    act = np.ascontiguousarray(action, dtype=np.int32).reshape(n)
    rew = np.empty(n, np.float32)
    term = np.empty(n, np.uint8)
    lib().emei_oracle_cartpole_step_f32(
        C.c_int(v), C.c_int64(n), C.c_int(int(freq_rate)), C.c_float(float(dt)),
        *[_p(a, C.c_float) for a in soa], _p(act, C.c_int32), _p(rew, C.c_float), _p(term, C.c_uint8))
    return rew, term.astype(bool)


def cartpole_reward(variant, obs):
    obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 4)
    out = np.empty(obs.shape[0])
    lib().emei_oracle_cartpole_reward(C.c_int(CARTPOLE_VARIANTS[variant]), C.c_int64(len(out)), _p(obs, C.c_double), _p(out, C.c_double))
    return out


def cartpole_terminal(variant, obs):
    obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 4)
    out = np.empty(obs.shape[0], np.uint8)
    lib().emei_oracle_cartpole_terminal(C.c_int(CARTPOLE_VARIANTS[variant]), C.c_int64(len(out)), _p(obs, C.c_double), _p(out, C.c_uint8))
    return out.astype(bool)


def cartpole_init_state_host(variant, seed, batch_size):
    """Reference reset distribution on the host: gym 0.26 np_random = Generator(PCG64(SeedSequence(seed)))
    (base_control.py:38-47) -> uniform(-0.05, 0.05, (B,4)), SwingUp adds pi to theta (cartpole.py:131-132,153-156)."""
    s = np.random.default_rng(seed).uniform(low=-0.05, high=0.05, size=(batch_size, 4))
    if variant == "swingup":
        s[:, 2] += np.pi
    return s


def cartpole_rollout_autoreset(variant, state, actions, seed, env_ids=None, max_steps=0, freq_rate=1, dt=0.02, steps=None, episode=None,
                               want=("obs", "reward", "done"), reuse=None, method="euler"):
    """T fused steps with device-style auto-reset, in C over all envs (the CPU twin of emei_rollout with EMEI_FLAG_AUTO_RESET).
    state [n,4] float64 (copied), actions [T,n] uint8 -> dict(obs [T,n,4] f32, reward [T,n] f32, done [T,n] u8, state [n,4] f64,
    steps [n] i32, episode [n] u32); `want` selects which per-step outputs are produced; `reuse` = the dict of a previous call
    of the same shape, whose output buffers are written again (a timing loop does not page in fresh memory per call)."""
    st = np.array(state, dtype=np.float64, order="C", copy=True).reshape(-1, 4)
    n = st.shape[0]
    act = np.ascontiguousarray(actions, dtype=np.uint8).reshape(-1, n)
    T = act.shape[0]
    sc = np.zeros(n, np.int32) if steps is None else np.array(steps, np.int32, copy=True)
    ep = np.zeros(n, np.uint32) if episode is None else np.array(episode, np.uint32, copy=True)
    ids = None if env_ids is None else np.ascontiguousarray(env_ids, np.int64)
    if reuse is not None and reuse.get("obs") is not None and reuse["obs"].shape == (T, n, 4):
        obs, rew, dn = reuse["obs"], reuse["reward"], reuse["done"]  # output buffers of a previous call (already paged in)
    else:
        obs = np.empty((T, n, 4), np.float32) if "obs" in want else None
        rew = np.empty((T, n), np.float32) if "reward" in want else None
        dn = np.empty((T, n), np.uint8) if "done" in want else None
    lib().emei_oracle_cartpole_rollout_autoreset_method(
        C.c_int(CARTPOLE_VARIANTS[variant]), C.c_int64(n), C.c_int(T), C.c_int(int(freq_rate)), C.c_double(float(dt)), C.c_int(ODE_METHODS[method]),
        C.c_int(int(max_steps)),
        C.c_uint64(int(seed)), _p(ids, C.c_int64) if ids is not None else None, _p(st, C.c_double), _p(sc, C.c_int32), _p(ep, C.c_uint32),
        _p(act, C.c_uint8), _p(obs, C.c_float) if obs is not None else None, _p(rew, C.c_float) if rew is not None else None,
        _p(dn, C.c_uint8) if dn is not None else None)
    return dict(obs=obs, reward=rew, done=dn, state=st, steps=sc, episode=ep)


def philox(seed, env, episode, block=0):
    out = (C.c_uint32 * 4)()
    lib().emei_oracle_philox(C.c_uint64(seed), C.c_uint64(env), C.c_uint32(episode), C.c_uint32(block), out)
    return np.array(out[:], dtype=np.uint32)


def cartpole_init_f32(variant, seed, env, episode):
    out = (C.c_float * 4)()
    lib().emei_oracle_cartpole_init_f32(C.c_int(CARTPOLE_VARIANTS[variant]), C.c_uint64(seed), C.c_uint64(env), C.c_uint32(episode), out)
    return np.array(out[:], dtype=np.float32)


def ip_init_f32(seed, env, episode, sigma):
    out = (C.c_float * 4)()
    lib().emei_oracle_ip_init_f32(C.c_uint64(seed), C.c_uint64(env), C.c_uint32(episode), C.c_float(sigma), out)
    return np.array(out[:], dtype=np.float32)


def cartpole_reference_style_loop(variant, n_steps=1000, seed=0, freq_rate=1, dt=0.02):
    """ONE env stepped the way the reference steps it (base_control.py:61-83 -> ODE_approximation :162-164 -> _dsdt,
    cartpole.py:48-60): a CPython call per step, scalar math.sin / math.cos, a float32 `np.array` for the derivative,
    `np.append` for the augmented state, the batched reward / terminal functions on a [1, 4] row.  Written for bench.py's
    `cpu_baseline.reference_style` ("reference-style single-env throughput"); same values as cartpole_step
    (tests/test_oracle_golden.py).  -> (final state [4], rewards [n_steps], terminals [n_steps])."""
    rng = np.random.default_rng(seed)
    state = rng.uniform(low=-0.05, high=0.05, size=(1, 4))[0]
    if variant == "swingup":
        state[2] += np.pi
    actions = np.random.default_rng(1).integers(2, size=n_steps)
    gravity, mass_pole, length, force_mag = 9.8, 0.1, 0.5, 10.0
    total_mass = mass_pole + 1.0
    x_thr, th_thr = (5, None) if variant == "swingup" else (2.4, 12 * 2 * math.pi / 360)

    def dsdt(s_aug):
        x, x_dot, theta, theta_dot, force = s_aug
        pml = mass_pole * length
        cos_theta, sin_theta = math.cos(theta), math.sin(theta)
        temp = (force + pml * theta_dot**2 * sin_theta) / total_mass
        theta_acc = (gravity * sin_theta - cos_theta * temp) / (length * (4.0 / 3.0 - mass_pole * cos_theta**2 / total_mass))
        x_acc = temp - pml * theta_acc * cos_theta / total_mass
        return np.array([x_dot, x_acc, theta_dot, theta_acc, 0], dtype=np.float32)

    def batch_reward(obs):
        return (np.cos(obs[:, 2:3]) + 1) / 2 if variant == "swingup" else np.ones([obs.shape[0], 1])

    def batch_terminal(obs):
        if variant == "swingup":
            return ~(np.abs(obs[:, 0:1]) < x_thr)
        return ~np.logical_and(np.abs(obs[:, 2:3]) < th_thr, np.abs(obs[:, 0:1]) < x_thr)

    rew, term = np.empty(n_steps), np.empty(n_steps, bool)
    for t in range(n_steps):
        action = np.asarray(int(actions[t]))
        s_aug = np.append(state, force_mag if action == 1 else -force_mag)
        y = s_aug.copy()
        for _ in range(freq_rate):
            y += dsdt(y) * dt
        state = y[:4]
        obs = state.copy()
        rew[t] = batch_reward(obs[None])[0, 0]
        term[t] = batch_terminal(obs[None])[0, 0]
    return state, rew, term


def set_threads(n):
    """OpenMP threads of the oracle library's parallel loops from now on (libgomp reads OMP_NUM_THREADS only when it loads)."""
    lib()
    C.CDLL("libgomp.so.1").omp_set_num_threads(C.c_int(int(n)))


# --------------------------------------------------------------------------- CartPole (NumPy)
def cartpole_step_numpy(variant, state, action, freq_rate=1, dt=0.02):
    """Vectorised NumPy restatement of base_control.py:61-83 + cartpole.py:48-60: float64 state,
    derivative rounded to float32, float32 product with float32(dt), float64 accumulate."""
    y = np.array(state, dtype=np.float64).reshape(-1, 4).copy()
    force = np.where(np.asarray(action).reshape(-1) == 1, 10.0, -10.0)
    total_mass, pml = 0.1 + 1.0, 0.1 * 0.5
    dt32 = np.float32(dt)
    with np.errstate(all="ignore"):
        for _ in range(freq_rate):
            x_dot, theta, theta_dot = y[:, 1], y[:, 2], y[:, 3]
            # math.sin/cos of the reference are libm; np.sin/cos may differ in the last ulp, which the
            # float32 rounding of the derivative absorbs except at rare rounding boundaries.
            c, s = np.cos(theta), np.sin(theta)
            temp = (force + pml * theta_dot**2 * s) / total_mass
            theta_acc = (9.8 * s - c * temp) / (0.5 * (4.0 / 3.0 - 0.1 * c**2 / total_mass))
            x_acc = temp - pml * theta_acc * c / total_mass
            d = np.stack([x_dot, x_acc, theta_dot, theta_acc], axis=1).astype(np.float32)
            y += (d * dt32).astype(np.float64)
        if variant == "swingup":
            rew = (np.cos(y[:, 2]) + 1) / 2
            term = ~(np.abs(y[:, 0]) < 5)
        else:
            rew = np.ones(len(y))
            term = ~((np.abs(y[:, 2]) < 12 * 2 * math.pi / 360) & (np.abs(y[:, 0]) < 2.4))
    return y, rew, term


# --------------------------------------------------------------------------- integrator / noise options
INTEGRATORS = {"euler": 0, "semi_implicit_euler": 1, "rk4": 2}


class Opts(C.Structure):
    """oracle_opts_t (integrators.h)."""

    _fields_ = [("integrator", C.c_int32), ("shared", C.c_int32), ("obs_pos", C.c_float), ("obs_vel", C.c_float),
                ("seed", C.c_uint64), ("env_offset", C.c_uint64), ("episode", C.c_uint32), ("step_index", C.c_uint32),
                ("solver", C.c_int32), ("reserved", C.c_int32)]


SOLVERS = {"newton": 0, "sweep1": 1}


def opts(integrator="euler", obs_noise=0.0, shared=False, seed=0, env_offset=0, episode=0, step_index=0, solver="newton"):
    """Options of the *_step functions: integrator (mujoco_env.py:70-79) and per-substep observation noise
    (:98-104) drawn from the device's counter-based stream at (seed, env, episode, step_index)."""
    p, v = (obs_noise if isinstance(obs_noise, (tuple, list)) else (obs_noise, obs_noise))
    return Opts(INTEGRATORS[integrator], int(bool(shared)), float(p), float(v), int(seed), int(env_offset), int(episode),
                int(step_index), SOLVERS[solver], 0)


def _o(o):
    return C.byref(o) if o is not None else None


def body_init(seed, env, episode, nv, sigma_pos, sigma_vel=None, shared=False):
    """Device reset of a MuJoCo-backed body: zeros + Gaussian noise -> float64 [2*nv] = (qpos, qvel)."""
    out = np.empty(2 * nv)
    lib().emei_oracle_body_init(C.c_uint64(int(seed)), C.c_uint64(int(env)), C.c_uint32(int(episode)), C.c_int(nv),
                                C.c_float(sigma_pos), C.c_float(sigma_pos if sigma_vel is None else sigma_vel),
                                C.c_int(int(bool(shared))), _p(out, C.c_double))
    return out


# --------------------------------------------------------------------------- InvertedPendulum (C)
class IPModel(C.Structure):
    _fields_ = [(k, C.c_double) for k in (
        "mc", "mp", "r", "Icom", "phi0", "g", "gear", "ctrl_lo", "ctrl_hi", "x_lo", "x_hi",
        "invweight_slider", "invweight_hinge", "th_lo", "th_hi", "timeconst", "dampratio", "dmin", "dmax", "width")]


def ip_model():
    m = IPModel()
    assert C.sizeof(m) == lib().emei_oracle_ip_model_size()
    lib().emei_oracle_ip_model(C.byref(m))
    return m


def ip_step(variant, state, action, freq_rate=1, dt=0.02, opt=None):
    """state [n,4]=(x, theta_unwrapped, v, omega) float64 -> (next_state, obs(wrapped), reward, terminal)."""
    st = np.array(state, dtype=np.float64, order="C", copy=True).reshape(-1, 4)
    n = st.shape[0]
    act = np.ascontiguousarray(action, dtype=np.float64).reshape(n)
    obs = np.empty((n, 4))
    rew = np.empty(n)
    term = np.empty(n, np.uint8)
    lib().emei_oracle_ip_step_ex(C.c_int(IP_VARIANTS[variant]), C.c_int64(n), C.c_int(int(freq_rate)), C.c_double(float(dt)),
                                 _p(st, C.c_double), _p(act, C.c_double), _p(obs, C.c_double), _p(rew, C.c_double),
                                 _p(term, C.c_uint8), _o(opt))
    return st, obs, rew, term.astype(bool)


def ip_accel(variant, q, v, ctrl, dt=0.02):
    """(x'', theta'') of the oracle's InvertedPendulum at one state, limits included."""
    q = np.ascontiguousarray(q, np.float64).reshape(2)
    v = np.ascontiguousarray(v, np.float64).reshape(2)
    acc = np.empty(2)
    lib().emei_oracle_ip_accel(C.c_int(IP_VARIANTS[variant]), C.c_double(dt), _p(q, C.c_double), _p(v, C.c_double), C.c_double(ctrl),
                               _p(acc, C.c_double))
    return acc


def ip_accel_custom(mc, mp, r, Icom, g, q, v, force):
    """(x'', theta'') of the oracle's cart + pole equations with caller-supplied parameters (no gear, no limit)."""
    q = np.ascontiguousarray(q, np.float64).reshape(2)
    v = np.ascontiguousarray(v, np.float64).reshape(2)
    acc = np.empty(2)
    lib().emei_oracle_ip_accel_custom(C.c_double(mc), C.c_double(mp), C.c_double(r), C.c_double(Icom), C.c_double(g),
                                      _p(q, C.c_double), _p(v, C.c_double), C.c_double(force), _p(acc, C.c_double))
    return acc


def ip_reward(variant, obs):
    obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 4)
    out = np.empty(obs.shape[0])
    lib().emei_oracle_ip_reward(C.c_int(IP_VARIANTS[variant]), C.c_int64(len(out)), _p(obs, C.c_double), _p(out, C.c_double))
    return out


def ip_terminal(variant, obs):
    obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 4)
    out = np.empty(obs.shape[0], np.uint8)
    lib().emei_oracle_ip_terminal(C.c_int(IP_VARIANTS[variant]), C.c_int64(len(out)), _p(obs, C.c_double), _p(out, C.c_uint8))
    return out.astype(bool)


def ip_wrap(theta):
    th = np.ascontiguousarray(theta, np.float64).reshape(-1)
    out = np.empty_like(th)
    lib().emei_oracle_ip_wrap(C.c_int64(len(th)), _p(th, C.c_double), _p(out, C.c_double))
    return out


# --------------------------------------------------------------------------- HalfCheetah-style body (C)
ENV_PARAM_ORDER = ("forward_reward_weight", "ctrl_cost_weight", "healthy_reward", "terminate_when_unhealthy",
                   "healthy_state_lo", "healthy_state_hi", "healthy_z_lo", "healthy_z_hi")
CHEETAH_DEFAULTS = dict(forward_reward_weight=1.0, ctrl_cost_weight=0.1)
HOPPER_DEFAULTS = dict(forward_reward_weight=1.0, ctrl_cost_weight=1e-3, healthy_reward=1.0, terminate_when_unhealthy=1.0,
                       healthy_state_lo=-100.0, healthy_state_hi=100.0, healthy_z_lo=0.7, healthy_z_hi=float("inf"))


def _params(defaults, overrides):
    """Full parameter vector (order of emei_hip.h's enum emei_env_param) from defaults + {name: value} overrides."""
    d = dict(defaults)
    d.update(overrides or {})
    return np.array([float(d.get(k, 0.0)) for k in ENV_PARAM_ORDER])


def cheetah_step(state, action, freq_rate=4, dt=0.002, opt=None, params=None):
    """state [n,18] = (qpos, qvel) float64 (copied), action [n,6] -> (next_state, reward, terminal)."""
    st = np.array(state, dtype=np.float64, order="C", copy=True).reshape(-1, 18)
    n = st.shape[0]
    act = np.ascontiguousarray(action, dtype=np.float64).reshape(n, 6)
    rew = np.empty(n)
    term = np.empty(n, np.uint8)
    P = _params(CHEETAH_DEFAULTS, params)
    lib().cheetah_oracle_step_p(C.c_int64(n), C.c_int(int(freq_rate)), C.c_double(float(dt)), _p(st, C.c_double),
                                _p(act, C.c_double), _p(rew, C.c_double), _p(term, C.c_uint8), _o(opt), _p(P, C.c_double))
    return st, rew, term.astype(bool)


def cheetah_reward(obs, pre_obs, act, dt_env, params=None):
    obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 18)
    pre = np.ascontiguousarray(pre_obs, np.float64).reshape(-1, 18)
    a = np.ascontiguousarray(act, np.float64).reshape(-1, 6)
    out = np.empty(len(obs))
    P = _params(CHEETAH_DEFAULTS, params)
    lib().cheetah_oracle_reward_p(C.c_int64(len(obs)), _p(obs, C.c_double), _p(pre, C.c_double), _p(a, C.c_double),
                                  C.c_double(dt_env), _p(P, C.c_double), _p(out, C.c_double))
    return out


def cheetah_terminal(obs):
    obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 18)
    out = np.empty(len(obs), np.uint8)
    lib().cheetah_oracle_terminal(C.c_int64(len(obs)), _p(obs, C.c_double), _p(out, C.c_uint8))
    return out.astype(bool)


def cheetah_inertia(q, v):
    """(M [9,9], bias [9], energy) at one configuration: diagnostics for the tests."""
    q = np.ascontiguousarray(q, np.float64).reshape(9)
    v = np.ascontiguousarray(v, np.float64).reshape(9)
    M = np.empty((9, 9))
    b = np.empty(9)
    e = C.c_double()
    lib().cheetah_oracle_inertia(_p(q, C.c_double), _p(v, C.c_double), _p(M, C.c_double), _p(b, C.c_double), C.byref(e))
    return M, b, e.value


def planar_inertia(body, q, v):
    """(M [nv,nv], bias [nv], energy) of the planar-tree oracle; body "cheetah" (nv 9) or "hopper" (nv 6)."""
    nv = {"cheetah": 9, "hopper": 6}[body]
    q = np.ascontiguousarray(q, np.float64).reshape(nv)
    v = np.ascontiguousarray(v, np.float64).reshape(nv)
    M = np.empty((nv, nv))
    b = np.empty(nv)
    e = C.c_double()
    lib().planar_oracle_inertia(C.c_int(0 if body == "cheetah" else 1), _p(q, C.c_double), _p(v, C.c_double),
                                _p(M, C.c_double), _p(b, C.c_double), C.byref(e))
    return M, b, e.value


def planar_solve(body, q, v, ctrl, dt=0.002, hd=0.0):
    """One forward-dynamics evaluation of the planar-tree oracle with the Newton solver (MuJoCo's formulation) and
    with the one-sweep solver -> dict(acc_newton, acc_sweep1, nrows, iters, resid)."""
    nv = 9 if body == "cheetah" else 6
    q, v, ctrl = (np.ascontiguousarray(x, np.float64) for x in (q, v, ctrl))
    an, a1 = np.empty(nv), np.empty(nv)
    nr, it, res = C.c_int(), C.c_int(), C.c_double()
    lib().planar_oracle_solve(C.c_int(0 if body == "cheetah" else 1), C.c_double(dt), C.c_double(hd), _p(q, C.c_double),
                              _p(v, C.c_double), _p(ctrl, C.c_double), _p(an, C.c_double), _p(a1, C.c_double), C.byref(nr),
                              C.byref(it), C.byref(res))
    return dict(acc_newton=an, acc_sweep1=a1, nrows=nr.value, iters=it.value, resid=res.value)


def planar_solve_unit(body, q, v, ctrl, dt=0.002, warm=None, max_iter=24):
    """The HIP kernels' iteration restated (unit Newton steps, |g|_inf <= 1e-11 |f|_inf, optional warm start) on one state
    -> dict(a, passes, nrows); passes == max_iter + 1 means the cap was hit."""
    nv = 9 if body == "cheetah" else 6
    q, v, ctrl = (np.ascontiguousarray(x, np.float64) for x in (q, v, ctrl))
    a = np.empty(nv)
    w = None if warm is None else np.ascontiguousarray(warm, np.float64)
    ps, nr = C.c_int(), C.c_int()
    lib().planar_oracle_solve_unit(C.c_int(0 if body == "cheetah" else 1), C.c_double(dt), _p(q, C.c_double), _p(v, C.c_double),
                                   _p(ctrl, C.c_double), _p(w, C.c_double) if w is not None else None, C.c_int(max_iter),
                                   _p(a, C.c_double), C.byref(ps), C.byref(nr))
    return dict(a=a, passes=ps.value, nrows=nr.value)


def planar_count_rows(body, state, dt=0.002):
    """Scalar constraint rows of the solver at each state [n, 2 nv] (-1 for a non-finite state)."""
    nv = 9 if body == "cheetah" else 6
    st = np.ascontiguousarray(state, np.float64).reshape(-1, 2 * nv)
    out = np.empty(len(st), np.int32)
    lib().planar_oracle_count_rows(C.c_int(0 if body == "cheetah" else 1), C.c_int64(len(st)), C.c_double(dt), _p(st, C.c_double),
                                   _p(out, C.c_int32))
    return out


def body_rollout(kind, variant, state, actions, freq_rate, dt, opt=None, reuse=None):
    """T env-steps of a MuJoCo-backed body in one C call (no Python between steps, no reset): kind "ip" | "dp" | "cheetah" |
    "hopper"; state [n, dim] float64 (copied), actions float32 [T, n(, nu)] -> dict(state, obs f32 [T,n,dim], reward f32 [T,n],
    done u8 [T,n]).  Step for step the arithmetic of ip_step / dpend_step / cheetah_step / hopper_step (tests/test_oracle_golden.py);
    `reuse` = the dict of a previous call of the same shape (its output buffers are written again)."""
    dim = {"ip": 4, "dp": 6, "cheetah": 18, "hopper": 12}[kind]
    st = np.array(state, dtype=np.float64, order="C", copy=True).reshape(-1, dim)
    n = st.shape[0]
    act = np.ascontiguousarray(actions, dtype=np.float32)
    T = act.shape[0]
    if reuse is not None and reuse["obs"].shape == (T, n, dim):
        obs, rew, dn = reuse["obs"], reuse["reward"], reuse["done"]
    else:
        obs, rew, dn = np.empty((T, n, dim), np.float32), np.empty((T, n), np.float32), np.empty((T, n), np.uint8)
    args = (C.c_int64(n), C.c_int(T), C.c_int(int(freq_rate)), C.c_double(float(dt)), _p(st, C.c_double), _p(act, C.c_float),
            _p(obs, C.c_float), _p(rew, C.c_float), _p(dn, C.c_uint8), _o(opt))
    if kind == "ip":
        lib().emei_oracle_ip_rollout(C.c_int(IP_VARIANTS[variant]), *args)
    elif kind == "dp":
        lib().dpend_oracle_rollout(C.c_int(DP_VARIANTS[variant]), *args)
    else:
        lib().planar_oracle_rollout(C.c_int(0 if kind == "cheetah" else 1), *args, None)
    return dict(state=st, obs=obs, reward=rew, done=dn)


def xml_constants(model):
    """The oracle's model constants of "ip" | "dp" | "cheetah" | "hopper" in the layout of emei_model_constants."""
    out = np.full(256, np.nan)
    if model == "ip":
        n = lib().emei_oracle_ip_xml_constants(_p(out, C.c_double))
    elif model == "dp":
        n = lib().dpend_oracle_xml_constants(_p(out, C.c_double))
    else:
        n = lib().planar_oracle_xml_constants(C.c_int(0 if model == "cheetah" else 1), _p(out, C.c_double))
    return out[:n].copy()


def planar_row_mask(body, state):
    """Bit mask of the constraint rows present at each state [n, 2 nv]: limits of the actuated joints in the low bits, then two
    bits per capsule (its end spheres) in XML geom order, then one bit per colliding capsule pair (hopper: 3)."""
    nv = 9 if body == "cheetah" else 6
    st = np.ascontiguousarray(state, np.float64).reshape(-1, 2 * nv)
    out = np.empty(len(st), np.uint32)
    lib().planar_oracle_row_mask(C.c_int(0 if body == "cheetah" else 1), C.c_int64(len(st)), _p(st, C.c_double), _p(out, C.c_uint32))
    return out


def planar_pairs(body, q):
    """Colliding capsule pairs at configuration q -> rows {g1, g2, touching, dist, normal x, normal z} (hopper: torso-leg,
    torso-foot, thigh-foot; the cheetah has none: half_cheetah.xml:39 conaffinity 0)."""
    nv = 9 if body == "cheetah" else 6
    q = np.ascontiguousarray(q, np.float64).reshape(nv)
    out = np.zeros((28, 7))
    lib().planar_oracle_pairs.restype = C.c_int
    n = lib().planar_oracle_pairs(C.c_int(0 if body == "cheetah" else 1), _p(q, C.c_double), _p(out, C.c_double))
    return out[:n, :6].copy()


def planar_invweights(body):
    """(dof_invweight0 [nv], body_invweight0 [nb]) of the planar-tree oracle at qpos0 (mj_setConst)."""
    nv, nb = (9, 7) if body == "cheetah" else (6, 4)
    d, b = np.empty(nv), np.empty(nb)
    lib().planar_oracle_invweights(C.c_int(0 if body == "cheetah" else 1), _p(d, C.c_double), _p(b, C.c_double))
    return d, b


def dpend_invweight():
    """dof_invweight0 of the InvertedDoublePendulum's slider at qpos0 (dpend_oracle.c:dpend_oracle_model)."""
    lib().dpend_oracle_invweight.restype = C.c_double
    return float(lib().dpend_oracle_invweight())


def planar_rows(body, q, v, dt=0.002, cap=80):
    """(J [nr, nv], aref [nr], D [nr]) of the Newton solver's scalar constraint rows at one state (build_rows' order)."""
    nv = 9 if body == "cheetah" else 6
    q, v = np.ascontiguousarray(q, np.float64), np.ascontiguousarray(v, np.float64)
    J, aref, D = np.zeros((cap, nv)), np.zeros(cap), np.zeros(cap)
    lib().planar_oracle_rows.restype = C.c_int
    nr = lib().planar_oracle_rows(C.c_int(0 if body == "cheetah" else 1), C.c_double(dt), _p(q, C.c_double), _p(v, C.c_double),
                                  C.c_int(cap), _p(J, C.c_double), _p(aref, C.c_double), _p(D, C.c_double))
    assert nr <= cap
    return J[:nr], aref[:nr], D[:nr]


def planar_geometry(body, q):
    """(body masses [nb], world centres of the capsule end spheres [ng,2,2]) at configuration q."""
    nb, ng, nv = {"cheetah": (7, 8, 9), "hopper": (4, 4, 6)}[body]
    q = np.ascontiguousarray(q, np.float64).reshape(nv)
    mass = np.empty(nb)
    ends = np.empty((ng, 2, 2))
    lib().planar_oracle_geometry(C.c_int(0 if body == "cheetah" else 1), _p(q, C.c_double), _p(mass, C.c_double), _p(ends, C.c_double),
                                 None, None, None)
    return mass, ends


def planar_bodies(body, q):
    """(mass [nb], com [nb,2] world, absolute angle [nb], inertia about the com [nb]) of every body at q."""
    nb, ng, nv = {"cheetah": (7, 8, 9), "hopper": (4, 4, 6)}[body]
    q = np.ascontiguousarray(q, np.float64).reshape(nv)
    mass, ends, com, phi, inertia = np.empty(nb), np.empty((ng, 2, 2)), np.empty((nb, 2)), np.empty(nb), np.empty(nb)
    lib().planar_oracle_geometry(C.c_int(0 if body == "cheetah" else 1), _p(q, C.c_double), _p(mass, C.c_double), _p(ends, C.c_double),
                                 _p(com, C.c_double), _p(phi, C.c_double), _p(inertia, C.c_double))
    return mass, com, phi, inertia


# --------------------------------------------------------------------------- Hopper (C)
def hopper_step(state, action, freq_rate=4, dt=0.002, opt=None, params=None):
    """state [n,12] = (qpos, qvel) float64 (copied), action [n,3] -> (next_state, reward, terminal).
    The reference's default integrator for this env is "rk4" (hopper.py:22): pass opt=opts("rk4")."""
    st = np.array(state, dtype=np.float64, order="C", copy=True).reshape(-1, 12)
    n = st.shape[0]
    act = np.ascontiguousarray(action, dtype=np.float64).reshape(n, 3)
    rew = np.empty(n)
    term = np.empty(n, np.uint8)
    P = _params(HOPPER_DEFAULTS, params)
    lib().hopper_oracle_step_p(C.c_int64(n), C.c_int(int(freq_rate)), C.c_double(float(dt)), _p(st, C.c_double),
                               _p(act, C.c_double), _p(rew, C.c_double), _p(term, C.c_uint8), _o(opt), _p(P, C.c_double))
    return st, rew, term.astype(bool)


def hopper_reward(obs, pre_obs, act, dt_env, params=None):
    obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 12)
    pre = np.ascontiguousarray(pre_obs, np.float64).reshape(-1, 12)
    a = np.ascontiguousarray(act, np.float64).reshape(-1, 3)
    out = np.empty(len(obs))
    P = _params(HOPPER_DEFAULTS, params)
    lib().hopper_oracle_reward_p(C.c_int64(len(obs)), _p(obs, C.c_double), _p(pre, C.c_double), _p(a, C.c_double),
                                 C.c_double(dt_env), _p(P, C.c_double), _p(out, C.c_double))
    return out


def hopper_is_healthy(obs, params=None):
    return hopper_healthy_terminal(obs, params)[0]


def hopper_healthy_terminal(obs, params=None):
    """(is_healthy, terminal) of hopper.py:79-93,104-106 as executed, for the given constructor parameters."""
    obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 12)
    healthy, term = np.empty(len(obs), np.uint8), np.empty(len(obs), np.uint8)
    P = _params(HOPPER_DEFAULTS, params)
    lib().hopper_oracle_is_healthy_p(C.c_int64(len(obs)), _p(obs, C.c_double), _p(P, C.c_double), _p(healthy, C.c_uint8),
                                     _p(term, C.c_uint8))
    return healthy.astype(bool), term.astype(bool)


# --------------------------------------------------------------------------- InvertedDoublePendulum (C)
DP_VARIANTS = {"rebound_balancing": 0, "boundary_balancing": 1, "rebound_swingup": 2, "boundary_swingup": 3}


def dpend_step(variant, state, action, freq_rate=1, dt=0.02, opt=None):
    """state [n,6] = (x, th1, th2, v, w1, w2) float64 -> (next_state, obs (quirk-wrapped), reward, terminal)."""
    st = np.array(state, dtype=np.float64, order="C", copy=True).reshape(-1, 6)
    n = st.shape[0]
    act = np.ascontiguousarray(action, dtype=np.float64).reshape(n)
    obs = np.empty((n, 6))
    rew = np.empty(n)
    term = np.empty(n, np.uint8)
    lib().dpend_oracle_step_ex(C.c_int(DP_VARIANTS[variant]), C.c_int64(n), C.c_int(int(freq_rate)), C.c_double(float(dt)),
                               _p(st, C.c_double), _p(act, C.c_double), _p(obs, C.c_double), _p(rew, C.c_double),
                               _p(term, C.c_uint8), _o(opt))
    return st, obs, rew, term.astype(bool)


def dpend_accel_custom(mc, mp, Ip, lc, L1, g, q, v, gen_force):
    """(x'', theta1'', theta2'') of the oracle's cart + two-pole equations with caller-supplied parameters and
    generalized forces (on x, theta1, theta2)."""
    q = np.ascontiguousarray(q, np.float64).reshape(3)
    v = np.ascontiguousarray(v, np.float64).reshape(3)
    f = np.ascontiguousarray(gen_force, np.float64).reshape(3)
    acc = np.empty(3)
    lib().dpend_oracle_accel_custom(C.c_double(mc), C.c_double(mp), C.c_double(Ip), C.c_double(lc), C.c_double(L1),
                                    C.c_double(g), _p(q, C.c_double), _p(v, C.c_double), _p(f, C.c_double), _p(acc, C.c_double))
    return acc


def dpend_reward_terminal(variant, obs):
    obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 6)
    rew = np.empty(len(obs))
    term = np.empty(len(obs), np.uint8)
    lib().dpend_oracle_reward_terminal(C.c_int(DP_VARIANTS[variant]), C.c_int64(len(obs)), _p(obs, C.c_double), _p(rew, C.c_double), _p(term, C.c_uint8))
    return rew, term.astype(bool)


def dpend_wrap(theta):
    th = np.ascontiguousarray(theta, np.float64).reshape(-1)
    out = np.empty_like(th)
    lib().dpend_oracle_wrap(C.c_int64(len(th)), _p(th, C.c_double), _p(out, C.c_double))
    return out


def dpend_energy(variant, state):
    s = np.ascontiguousarray(state, np.float64).reshape(6)
    f = lib().dpend_oracle_energy
    f.restype = C.c_double
    return f(C.c_int(DP_VARIANTS[variant]), _p(s, C.c_double))
