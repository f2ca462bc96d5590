"""Multi-GPU sharding of env instances: one process per GPU, contiguous shard per rank.

Env instances are fully independent (no cross-env term in cartpole.py:48-60 or
mujoco_env.py:86-109), so the data path has no collective: rank r owns global envs
[r*n, (r+1)*n), with its own state SoA and a device RNG keyed by the GLOBAL env index, which makes
results independent of the number of ranks.  The only exchange is the batched observation return:
an all-gather of the [n, obs_dim] float32 observations (RCCL over xGMI on GPUs; gloo in CPU tests).
"""
import numpy as np
import torch


def shard_bounds(n_global, rank, world):
    """Contiguous block partition; the first (n_global % world) ranks get one extra env."""
    base, rem = divmod(int(n_global), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_obs(local_obs, group=None, out=None):
    """All-gather equal-sized [n, d] observation shards into [world*n, d] (rank-major = global env order)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return local_obs
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * local_obs.shape[0],) + tuple(local_obs.shape[1:]), dtype=local_obs.dtype,
                          device=local_obs.device)
    if dist.get_backend(group) == "gloo" and local_obs.is_cuda:
        # rehearsal path only (gloo has no device all-gather): stage through the host
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, local_obs.cpu().contiguous(), group=group)
        out.copy_(host)
        return out
    dist.all_gather_into_tensor(out, local_obs.contiguous(), group=group)
    return out


def synthetic_init_state(env, n_global, lo, hi, seed=0):
    """Initial states of SURVEY.md 8d: the reference's reset distribution drawn on the host for the
    GLOBAL env array (row-major like np_random.uniform(size=(B,4)), cartpole.py:153-156), then sliced."""
    rng = np.random.default_rng(seed)
    if env in ("CartPoleSwingUp", "CartPoleBalancing"):
        s = rng.uniform(low=-0.05, high=0.05, size=(n_global, 4))
        if env == "CartPoleSwingUp":
            s[:, 2] += np.pi
        return s[lo:hi]
    if "InvertedDoublePendulum" in env:
        return (rng.standard_normal((n_global, 6)) * 5e-3)[lo:hi]
    if "InvertedPendulum" in env:
        return (rng.standard_normal((n_global, 4)) * 5e-3)[lo:hi]  # mujoco_env.py:31,137-140
    if env in ("HalfCheetahRunning", "HopperRunning"):
        return None  # device reset (init_qpos + sigma*N(0,1)), body_kernels.h:body_init
    raise ValueError(env)


def synthetic_actions(env, horizon, n, rank, world, device, act_dim):
    """Random actions uploaded once: default_rng(1) on one GPU, default_rng([1, rank]) when sharded."""
    rng = np.random.default_rng(1 if world == 1 else [1, rank])
    if env.startswith("CartPole"):
        return torch.as_tensor(rng.integers(2, size=(horizon, n), dtype=np.uint8), device=device)
    lim = 3.0 if ("InvertedPendulum" in env and "Double" not in env) else 1.0
    shape = (horizon, n) if act_dim <= 1 else (horizon, n, act_dim)
    return torch.as_tensor(rng.uniform(-lim, lim, size=shape).astype(np.float32), device=device)


class ShardedRollout:
    """This rank's shard of a fused rollout + the all-gather of the batched observation return."""

    MAX_EPISODE_STEPS = {"CartPoleSwingUp": 1000, "CartPoleBalancing": 500}  # register_env.py:14-23

    def __init__(self, env, envs_per_rank, horizon, freq_rate=1, real_time_scale=0.02, precision="ref", rank=0,
                 world=1, device=0, seed=0, init_noise=None, integrator="euler"):
        from .engine import Engine

        self.env, self.n, self.horizon, self.rank, self.world = env, int(envs_per_rank), int(horizon), rank, world
        self.lo, self.hi = rank * self.n, (rank + 1) * self.n
        if init_noise is None:
            init_noise = 0.1 if env == "HalfCheetahRunning" else 5e-3
        self.engine = Engine(env, self.n, freq_rate=freq_rate, real_time_scale=real_time_scale, precision=precision,
                             max_episode_steps=self.MAX_EPISODE_STEPS.get(env, 1000), device=device, seed=seed,
                             env_index_offset=self.lo, init_noise=init_noise, integrator=integrator)
        self.device = self.engine.device
        self.obs_dim, self.act_dim = self.engine.obs_dim, self.engine.act_dim
        self.seed = seed
        self.actions = self.out = self.gathered = None
        self._events = []
        self.kernel_name = ("body_rollout_kernel" if (env in ("HalfCheetahRunning", "HopperRunning") or "Double" in env or integrator != "euler")
                            else "pend_rollout_staged_kernel")

    @property
    def action_bytes(self):
        return self.actions.element_size() * max(self.act_dim, 1)

    @property
    def action_dtype_name(self):
        return str(self.actions.dtype).replace("torch.", "")

    def make_synthetic_inputs(self):
        s0 = synthetic_init_state(self.env, self.world * self.n, self.lo, self.hi, self.seed)
        if s0 is None:
            self.engine.reset(self.seed)
        else:
            self.engine.set_state(s0)
        self.actions = synthetic_actions(self.env, self.horizon, self.n, self.rank, self.world, self.device, self.act_dim)
        self.out = self.engine.alloc_outputs(self.horizon)
        if self.world > 1:
            # The observation return overlaps the NEXT pass's rollout: the shard is copied to one of two
            # staging buffers on the launch stream and all-gathered from there on a dedicated stream (xGMI
            # transfers run under the kernel); events fence each buffer before it is reused two passes later.
            shape = (self.world * self.n, self.obs_dim)
            self.gathered = [torch.empty(shape, dtype=torch.float32, device=self.device) for _ in range(2)]
            self._stage = [torch.empty((self.n, self.obs_dim), dtype=torch.float32, device=self.device) for _ in range(2)]
            self._comm = torch.cuda.Stream(device=self.device)
            self._comm_done = [None, None]
            self._pass = 0
        torch.cuda.synchronize()

    def run_pass(self, record=False):
        """One rollout launch over the whole horizon, then the batched observation return."""
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        obs, rew, done = self.engine.rollout(self.actions, auto_reset=True, out=self.out)
        if record:
            e1.record()
            self._events.append((e0, e1))
        final_obs = obs[-1]
        if self.world > 1:
            k = self._pass & 1
            self._pass += 1
            main = torch.cuda.current_stream()
            if self._comm_done[k] is not None:
                main.wait_event(self._comm_done[k])  # the gather that last read this staging buffer has finished
            self._stage[k].copy_(final_obs)
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(self._comm):
                self._comm.wait_event(ready)
                final_obs = allgather_obs(self._stage[k], out=self.gathered[k])
                done = torch.cuda.Event()
                done.record(self._comm)
                self._comm_done[k] = done
        return final_obs

    def wait_gathers(self):
        """Make the launch stream wait for every outstanding observation all-gather."""
        if self.world > 1:
            for ev in self._comm_done:
                if ev is not None:
                    torch.cuda.current_stream().wait_event(ev)

    def mean_kernel_ms(self):
        torch.cuda.synchronize()
        ts = [a.elapsed_time(b) for a, b in self._events]
        return float(np.mean(ts)) if ts else float("nan")

    def timed_launches_ms(self, k):
        """Mean duration of k back-to-back rollout launches bracketed by ONE pair of HIP events on the
        launch stream (no per-launch marker packets inside the timed span)."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            self.engine.rollout(self.actions, auto_reset=True, out=self.out)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / k

    def time_per_step_api(self, n_steps=200):
        """One launch per env-step (emei_step), the gym-style API: launch-bound by construction."""
        import time

        out = self.engine.alloc_outputs(None)
        for t in range(10):
            self.engine.step(self.actions[t % self.horizon], auto_reset=True, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(n_steps):
            self.engine.step(self.actions[t % self.horizon], auto_reset=True, out=out)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        return {"env_steps_per_s": self.n * n_steps / el, "us_per_launch": el / n_steps * 1e6}
