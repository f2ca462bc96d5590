"""Batched env engine: a thin, torch-tensor-facing wrapper of the C ABI handle.

PyTorch is plumbing here (device memory + streams); every computation happens in libemei_hip.so.
"""
import ctypes as C

import torch

from . import _lib as L

_ACT_DTYPES = {torch.uint8: L.ACT_U8, torch.int32: L.ACT_I32, torch.int64: L.ACT_I64, torch.float32: L.ACT_F32}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream on the current device (the raw getter avoids building a
    Stream object on every call: 0.3 us instead of 3 us on the per-step path)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _stream_handle(device_index):
    """the same as an integer for a pre-bound ctypes call (emei_step_host)"""
    if _raw_stream is not None:
        return _raw_stream(device_index)
    return torch.cuda.current_stream(device_index).cuda_stream


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def env_dims(env_name):
    od, ad, sd = C.c_int(), C.c_int(), C.c_int()
    L.check(L.lib().emei_env_dims(L.ENV_IDS[env_name], C.byref(od), C.byref(ad), C.byref(sd)))
    return od.value, ad.value, sd.value


def _on_device(method):
    """Run an Engine method with the engine's device current (the C ABI refuses a handle used under another
    current device); a plain call when it already is (the per-step path pays one integer compare)."""
    import functools

    @functools.wraps(method)
    def wrapper(self, *a, **k):
        if torch.cuda.current_device() == self.device.index:
            return method(self, *a, **k)
        with torch.cuda.device(self.device):
            return method(self, *a, **k)

    return wrapper


def _sigmas(x, state_dim):
    """float -> the same sigma everywhere; (pos, vel) -> per half; a sequence of state_dim values as given."""
    half = state_dim // 2
    if isinstance(x, (int, float)):
        return [float(x)] * state_dim
    x = [float(v) for v in x]
    if len(x) == 2 and state_dim != 2:
        return [x[0]] * half + [x[1]] * half
    if len(x) != state_dim:
        raise ValueError(f"noise sigmas {x!r}: expected a float, a (pos, vel) pair or {state_dim} values")
    return x


class Engine:
    """N independent env instances of one kind on one GPU (this rank's shard)."""

    def __init__(self, env_name, n_envs, freq_rate=1, real_time_scale=0.02, precision="ref", max_episode_steps=0,
                 device=None, seed=0, env_index_offset=0, init_noise=0.0, integrator="euler", obs_noise=0.0,
                 noise_layout="iid", env_params=None, solver="newton", ode_method="euler", rollout_chunk_steps=0):
        """init_noise / obs_noise: one sigma, a (qpos sigma, qvel sigma) pair, or one sigma per state coordinate
        (qpos entries then qvel entries): the reduced forms of mujoco_env.py:218-227.
        env_params: {name: value} overriding reward / health constructor defaults (names: _lib.ENV_PARAMS).
        solver: constraint solver of HalfCheetah / Hopper: "newton" (MuJoCo's formulation, converged; the default) or
        "sweep1" (round 1's single Gauss-Seidel sweep).
        ode_method: CartPole only — the `method` of ODE_approximation: "euler" is what the reference's step() always runs
        (base_control.py:73), "rk4" its other branch (:165-170), an explicit opt-in.
        rollout_chunk_steps: rollouts of the one-wave-per-SIMD bodies as (64 envs) x (k steps) work items: 0 = automatic,
        -1 = off, k > 0 = k steps per item (emei_hip.h: emei_config.rollout_chunk_steps); results do not depend on it."""
        if env_name not in L.ENV_IDS:
            raise ValueError(f"unknown env {env_name!r}; known: {sorted(L.ENV_IDS)}")
        if integrator not in L.INTEGRATORS:
            raise NotImplementedError(f"integrator {integrator!r}")  # mujoco_env.py:78-79
        if not torch.cuda.is_available():
            raise L.EmeiHipError("emei_amd needs a HIP device (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.env_name, self.n_envs = env_name, int(n_envs)
        self.freq_rate, self.real_time_scale = int(freq_rate), float(real_time_scale)
        self.precision = {"ref": L.PRECISION_REF, "f32": L.PRECISION_F32}[precision]
        self.obs_dim, self.act_dim, self.state_dim = env_dims(env_name)
        self.integrator = integrator
        if solver not in L.SOLVERS:
            raise ValueError(f"solver {solver!r}; known: {sorted(L.SOLVERS)}")
        self.solver = solver
        if ode_method not in L.ODE_METHODS:
            raise NotImplementedError(f"approximation method `{ode_method}` is not suppoerted yet.")  # base_control.py:171-172
        self.ode_method = ode_method
        sig = C.c_float * L.MAX_STATE_DIM
        cfg = L.EmeiConfig(C.sizeof(L.EmeiConfig), L.ENV_IDS[env_name], self.n_envs, self.freq_rate, self.precision,
                           self.real_time_scale, int(max_episode_steps), self.device.index, int(seed),
                           int(env_index_offset), 0.0, L.INTEGRATORS[integrator],
                           {"iid": L.NOISE_IID, "shared": L.NOISE_SHARED}[noise_layout],
                           sig(*_sigmas(init_noise, self.state_dim)), sig(*_sigmas(obs_noise, self.state_dim)),
                           *((lambda m, a: (m, L.SOLVERS[solver], a))(*L.pack_env_params(env_params))),
                           L.ODE_METHODS[ode_method], int(rollout_chunk_steps))
        self._h = C.c_void_p()
        self._host_io = None
        with torch.cuda.device(self.device):
            L.check(L.lib().emei_create(C.byref(cfg), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            L.lib().emei_destroy(self._h)
            self._h = None
        # the pre-bound emei_step_host call holds the raw handle: it must not outlive it (a step_host() after
        # close() passed a dangling emei_env* into the library)
        self._host_io = None
        self._host_bufs = None

    def _live(self):
        if not self._h:
            raise L.EmeiHipError("this Engine has been closed")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state ---------------------------------------------------------------------------------
    @_on_device
    def reset(self, seed=0):
        L.check(L.lib().emei_reset(self._h, int(seed) & (2**64 - 1), _stream()))

    def set_seed(self, seed):
        """Re-key the device reset generator (auto-reset episodes) without touching the state."""
        L.check(L.lib().emei_set_seed(self._h, int(seed) & (2**64 - 1)))

    @_on_device
    def solver_cap_hits(self):
        """Newton solves of this engine's rollouts that ended at the iteration cap without converging (must stay 0)."""
        out = torch.zeros(1, dtype=torch.int64, device=self.device)
        L.check(L.lib().emei_get_solver_cap_hits(self._h, _ptr(out), _stream()))
        return int(out.item())

    @_on_device
    def rollout_faults(self):
        """Work items of this engine's chunked body rollouts that gave up waiting for their predecessor (must stay 0)."""
        out = torch.zeros(1, dtype=torch.int64, device=self.device)
        L.check(L.lib().emei_get_rollout_faults(self._h, _ptr(out), _stream()))
        return int(out.item())

    def set_obs_peers(self, peers, row_envs, col_offset, max_steps=None):
        """emei_set_obs_peers: from now on every rollout ALSO stores each step's observation row into the gathered buffers `peers`
        (device pointers as ints, or float32 tensors [rows, row_envs, obs_dim]; at most _lib.MAX_OBS_PEERS), this engine's envs at
        columns [col_offset, col_offset + n_envs) — the multi-GPU observation return by peer writes (sharding.PeerWriteExchange).
        An empty list switches it off.  Rollouts the staged CartPole kernel cannot serve then raise instead of skipping the peers.
        max_steps: rows of every buffer (taken from the tensors when they are tensors): a longer rollout is refused."""
        self._live()
        ptrs = [int(p.data_ptr()) if torch.is_tensor(p) else int(p) for p in peers]
        if max_steps is None and ptrs:
            if not all(torch.is_tensor(p) for p in peers):
                raise ValueError("set_obs_peers: max_steps is needed with raw pointers")
            for p in peers:
                if p.dtype != torch.float32 or not p.is_contiguous() or p.dim() != 3 or p.shape[1] != row_envs or p.shape[2] != self.obs_dim:
                    raise ValueError(f"set_obs_peers: a peer buffer is float32 [rows, {row_envs}, {self.obs_dim}], contiguous; got {p.dtype} {tuple(p.shape)}")
            max_steps = min(int(p.shape[0]) for p in peers)
        arr = (C.c_void_p * max(len(ptrs), 1))(*ptrs)
        L.check(L.lib().emei_set_obs_peers(self._h, len(ptrs), arr, int(row_envs), int(col_offset), int(max_steps or 0)))
        self._peer_refs = [p for p in peers if torch.is_tensor(p)]  # keep local targets alive while they are written to

    def last_kernel(self):
        """enum emei_kernel_id of the kernel the last step / rollout launched (_lib.KERNEL_NAMES)."""
        self._live()
        return int(L.lib().emei_last_rollout_kernel(self._h))

    @_on_device
    def set_state(self, state, reset_counters=True):
        st = torch.as_tensor(state, dtype=torch.float64, device=self.device).contiguous()
        if tuple(st.shape) != (self.n_envs, self.state_dim):
            raise ValueError(f"state shape {tuple(st.shape)} != {(self.n_envs, self.state_dim)}")
        L.check(L.lib().emei_set_state(self._h, _ptr(st), int(bool(reset_counters)), _stream()))
        torch.cuda.current_stream().synchronize()  # `st` may be a temporary

    @_on_device
    def get_state(self):
        out = torch.empty((self.n_envs, self.state_dim), dtype=torch.float64, device=self.device)
        L.check(L.lib().emei_get_state(self._h, _ptr(out), _stream()))
        return out

    @_on_device
    def get_obs(self):
        out = torch.empty((self.n_envs, self.obs_dim), dtype=torch.float64, device=self.device)
        L.check(L.lib().emei_get_obs(self._h, _ptr(out), _stream()))
        return out

    @_on_device
    def freeze(self):
        L.check(L.lib().emei_freeze(self._h, _stream()))

    @_on_device
    def unfreeze(self):
        L.check(L.lib().emei_unfreeze(self._h, _stream()))

    # -- hot path ------------------------------------------------------------------------------
    def _check_actions(self, actions, lead):
        if not isinstance(actions, torch.Tensor) or actions.device != self.device:
            raise ValueError("actions must be a tensor on the engine's device")
        if actions.dtype not in _ACT_DTYPES:
            raise ValueError(f"unsupported action dtype {actions.dtype}")
        want = lead + ((self.n_envs,) if self.act_dim <= 1 else (self.n_envs, self.act_dim))
        shape = tuple(actions.shape)
        if shape != want and not (self.act_dim == 1 and shape == lead + (self.n_envs, 1)):
            raise ValueError(f"actions shape {shape} != {want}")
        if not actions.is_contiguous():
            raise ValueError("actions must be contiguous")
        return _ACT_DTYPES[actions.dtype]

    def _check_outputs(self, out, lead):
        """Caller-supplied output buffers go to the ABI as raw pointers (it has no size arguments): a wrong shape, dtype,
        device or stride would be a silent out-of-bounds device write.  Checked on EVERY call and nothing is retained: a
        tensor can be resize_()d / set_() between calls, and holding callers' buffers alive would pin their device memory."""
        if not isinstance(out, (tuple, list)) or len(out) != 3:
            raise ValueError("out must be (obs, reward, done)")
        n = self.n_envs
        for t, shape, dtype, name in ((out[0], lead + (n, self.obs_dim), torch.float32, "obs"),
                                      (out[1], lead + (n,), torch.float32, "reward"), (out[2], lead + (n,), torch.uint8, "done")):
            if not isinstance(t, torch.Tensor) or t.device != self.device:
                raise ValueError(f"out: {name} must be a tensor on {self.device}")
            if t.shape != shape or t.dtype != dtype:
                raise ValueError(f"out: {name} is {tuple(t.shape)} {t.dtype}, expected {shape} {dtype}")
            if not t.is_contiguous():
                raise ValueError(f"out: {name} must be contiguous")
        return out

    @_on_device
    def step(self, actions, auto_reset=False, out=None):
        """-> obs [N,obs_dim] f32, reward [N] f32, done [N] u8 (bit0 terminal, bit1 truncated)."""
        dt = self._check_actions(actions, ())
        obs, rew, done = self._check_outputs(out, ()) if out is not None else self.alloc_outputs(None)
        L.check(L.lib().emei_step(self._h, _ptr(actions), dt, _ptr(obs), _ptr(rew), _ptr(done),
                                  L.FLAG_AUTO_RESET if auto_reset else 0, _stream()))
        return obs, rew, done

    def step_host(self, action, auto_reset=False):
        """The gym-style single-env call (`env.step(a)` of base_control.py:61-83 with host values in and out) through
        emei_step_host: actions and results live in pinned host memory that the kernels address directly; for one env of the
        4-state family a step is ONE launch whose last store is a completion word the library polls (no copy, no stream
        synchronisation), otherwise step + emei_get_obs + one synchronisation.
        action: array-like [N(,act_dim)] -> (obs float64 [N,obs_dim] of the post-step state, obs float32
        [N,obs_dim] as emitted by the step (pre auto-reset), reward float32 [N], done uint8 [N]) as NumPy
        views of the pinned buffers (valid until the next call)."""
        io = self._host_io
        if io is None:
            self._live()
            io = self._host_io = self._make_host_io()
        act_np, obs64_np, obs32_np, rew_np, done_np, call, flag_auto = io
        act_np[...] = action
        if torch.cuda.current_device() != self.device.index:
            with torch.cuda.device(self.device):
                rc = call(flag_auto if auto_reset else 0, _stream_handle(self.device.index))
        else:
            rc = call(flag_auto if auto_reset else 0, _stream_handle(self.device.index))
        if rc:
            L.check(rc)
        return obs64_np, obs32_np, rew_np, done_np

    def _make_host_io(self):
        import functools

        pin = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True)
        ashape = (self.n_envs,) if self.act_dim <= 1 else (self.n_envs, self.act_dim)
        act = pin(ashape, torch.int64 if self.act_dim == 0 else torch.float32)
        bufs = (act, pin((self.n_envs, self.obs_dim), torch.float64), pin((self.n_envs, self.obs_dim), torch.float32),
                pin((self.n_envs,), torch.float32), pin((self.n_envs,), torch.uint8))
        self._host_bufs = bufs  # keeps the pinned allocations alive
        # the pointers never change: bind them once (a ctypes call costs per converted argument)
        call = functools.partial(L.lib().emei_step_host, self._h, _ptr(act), _ACT_DTYPES[act.dtype], _ptr(bufs[1]), _ptr(bufs[2]),
                                 _ptr(bufs[3]), _ptr(bufs[4]))
        return tuple(b.numpy() for b in bufs) + (call, L.FLAG_AUTO_RESET)

    @_on_device
    def rollout(self, actions, auto_reset=False, out=None):
        """actions [T,N(,act_dim)] -> obs [T,N,obs_dim] f32, reward [T,N] f32, done [T,N] u8; one launch."""
        T = int(actions.shape[0])
        dt = self._check_actions(actions, (T,))
        obs, rew, done = self._check_outputs(out, (T,)) if out is not None else self.alloc_outputs(T)
        L.check(L.lib().emei_rollout(self._h, T, _ptr(actions), dt, _ptr(obs), _ptr(rew), _ptr(done),
                                     L.FLAG_AUTO_RESET if auto_reset else 0, _stream()))
        return obs, rew, done

    @_on_device
    def capture_step_graph(self, actions, auto_reset=False):
        """Capture one emei_step launch per row of `actions` [K, N(,act_dim)] into a hipGraph.

        For callers that need a launch per env-step (e.g. a policy evaluated between steps is captured
        alongside) the graph removes the per-launch host cost (6.7 -> 3.7 us per step at 65 536 envs).
        The ABI's launch functions neither allocate nor synchronise, so they are capturable as they
        are.  Returns (graph, obs [K,N,obs_dim], reward [K,N], done [K,N]); `actions` and the outputs are
        the graph's static buffers: refill `actions` in place, then `graph.replay()`."""
        K = int(actions.shape[0])
        dt = self._check_actions(actions, (K,))
        obs, rew, done = self.alloc_outputs(K)
        flags = L.FLAG_AUTO_RESET if auto_reset else 0
        lib = L.lib()

        def launch_all():
            st = _stream()
            for k in range(K):
                L.check(lib.emei_step(self._h, _ptr(actions[k]), dt, _ptr(obs[k]), _ptr(rew[k]), _ptr(done[k]), flags, st))

        if not self.get_state().isfinite().any():  # touches the state: also asserts reset() happened
            raise L.EmeiHipError("capture_step_graph: state has no finite entry")
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            launch_all()
        torch.cuda.current_stream().wait_stream(side)
        return graph, obs, rew, done

    def alloc_outputs(self, n_steps=None):
        lead = () if n_steps is None else (int(n_steps),)
        return (torch.empty(lead + (self.n_envs, self.obs_dim), dtype=torch.float32, device=self.device),
                torch.empty(lead + (self.n_envs,), dtype=torch.float32, device=self.device),
                torch.empty(lead + (self.n_envs,), dtype=torch.uint8, device=self.device))

    @_on_device
    def get_counters(self):
        """(steps since reset int32 [N], episode index int64 [N])."""
        steps = torch.empty(self.n_envs, dtype=torch.int32, device=self.device)
        epi = torch.empty(self.n_envs, dtype=torch.int32, device=self.device)  # uint32 payload
        L.check(L.lib().emei_get_counters(self._h, _ptr(steps), _ptr(epi), _stream()))
        return steps, epi.to(torch.int64) & 0xFFFFFFFF

    @_on_device
    def episode_init_obs(self, env_index, episode):
        """Initial observation of the device reset for (env, episode) pairs -> [count, obs_dim] float32."""
        env_index = env_index.to(device=self.device, dtype=torch.int64).contiguous()
        epi = (episode.to(device=self.device, dtype=torch.int64) & 0xFFFFFFFF).to(torch.int32).contiguous()  # uint32 payload
        out = torch.empty((env_index.numel(), self.obs_dim), dtype=torch.float32, device=self.device)
        if env_index.numel():
            L.check(L.lib().emei_episode_init_obs(self._h, env_index.numel(), _ptr(env_index), _ptr(epi), _ptr(out), _stream()))
        return out

    @_on_device
    def compact_done(self):
        """Sorted indices of the envs that were done at the last step (int32 tensor)."""
        idx = torch.empty(self.n_envs, dtype=torch.int32, device=self.device)
        cnt = torch.zeros(1, dtype=torch.int32, device=self.device)
        L.check(L.lib().emei_compact_done(self._h, _ptr(idx), _ptr(cnt), _stream()))
        return idx[: int(cnt.item())]


# -- stateless batched functions -----------------------------------------------------------------
def _rows(t, device, dtype=None):
    """caller's array -> contiguous device tensor; float64 stays float64 (the reference evaluates these functions on float64
    arrays: cartpole.py:124-129,145-151; half_cheetah.py:59-67), anything else becomes float32"""
    if t is None:
        return None
    t = torch.as_tensor(t, device=device)
    if dtype is None:
        dtype = torch.float64 if t.dtype == torch.float64 else torch.float32
    return t.to(dtype).contiguous()


def _io(t):
    return L.IO_F64 if t.dtype == torch.float64 else L.IO_F32


def batch_reward(env_name, obs, pre_obs=None, action=None, real_time_scale=0.02, freq_rate=1, env_params=None,
                 batch_ctrl_cost=False):
    """get_batch_reward on [B, obs_dim] rows -> [B] in the rows' dtype (float64 in -> float64 out, not narrowed).
    batch_ctrl_cost: HalfCheetah / Hopper only — the control cost summed over the WHOLE batch, which is what
    half_cheetah.py:61 / hopper.py:98 execute for B > 1 (np.sum without an axis)."""
    obs = _rows(obs, obs.device if isinstance(obs, torch.Tensor) else "cuda")
    pre_obs, action = _rows(pre_obs, obs.device, obs.dtype), _rows(action, obs.device, obs.dtype)
    out = torch.empty(obs.shape[0], dtype=obs.dtype, device=obs.device)
    mask, arr = L.pack_env_params(env_params)
    with torch.cuda.device(obs.device):
        L.check(L.lib().emei_reward_io(L.ENV_IDS[env_name], obs.shape[0], _io(obs), _ptr(obs), _ptr(pre_obs), _ptr(action),
                                       float(real_time_scale), int(freq_rate), mask, C.cast(arr, C.c_void_p),
                                       L.REWARD_BATCH_CTRL_COST if batch_ctrl_cost else 0, _ptr(out), _stream()))
    return out


def batch_terminal(env_name, obs, env_params=None):
    obs = _rows(obs, obs.device if isinstance(obs, torch.Tensor) else "cuda")
    out = torch.empty(obs.shape[0], dtype=torch.uint8, device=obs.device)
    mask, arr = L.pack_env_params(env_params)
    with torch.cuda.device(obs.device):
        L.check(L.lib().emei_terminal_io(L.ENV_IDS[env_name], obs.shape[0], _io(obs), _ptr(obs), mask, C.cast(arr, C.c_void_p),
                                         _ptr(out), _stream()))
    return out.bool()


def batch_next_obs(env_name, obs, actions, real_time_scale=0.02, freq_rate=1, precision="ref", integrator="euler", ode_method="euler"):
    """ode_method: classic control only (ODE_approximation's `method`, base_control.py:133-173)"""
    obs = _rows(obs, obs.device if isinstance(obs, torch.Tensor) else "cuda")
    actions = actions.to(obs.device).contiguous()
    out = torch.empty_like(obs)
    with torch.cuda.device(obs.device):
        L.check(L.lib().emei_next_obs_io(L.ENV_IDS[env_name], obs.shape[0], _io(obs), _ptr(obs), _ptr(actions),
                                         _ACT_DTYPES[actions.dtype], float(real_time_scale), int(freq_rate),
                                         {"ref": 0, "f32": 1}[precision],
                                         L.INTEGRATORS[integrator] | (L.NEXT_OBS_ODE_RK4 if L.ODE_METHODS[ode_method] else 0), _ptr(out), _stream()))
    return out
