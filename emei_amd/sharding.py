"""Multi-GPU sharding of env instances: one process per GPU, contiguous shard per rank.

Env instances are fully independent (no cross-env term in cartpole.py:48-60 or
mujoco_env.py:86-109), so the data path has no collective: rank r owns global envs
[r*n, (r+1)*n), with its own state SoA and a device RNG keyed by the GLOBAL env index, which makes
results independent of the number of ranks.  The only exchange is the batched observation return:
an all-gather of float32 observation blocks (RCCL over xGMI on GPUs; gloo in CPU tests).
"""
import numpy as np
import torch


def shard_bounds(n_global, rank, world):
    """Contiguous block partition; the first (n_global % world) ranks get one extra env."""
    base, rem = divmod(int(n_global), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


ALGOS = ("collective", "direct")
EXCHANGES = ALGOS + ("peer_write",)  # ShardedRollout(exchange_algo=...): the two all-gather forms, or no collective at all (PeerWriteExchange)


def _allgather_direct(local_obs, flat, group):
    """The all-gather as point-to-point transfers: this rank's block goes to EVERY peer and every peer's block comes straight
    into its place of the receive buffer, all 2 (world - 1) transfers posted as one batch (one RCCL group: they run
    concurrently).  On MI355X the 8 GPUs of a node are a full xGMI mesh — 7 links per GPU, each to one peer — so every transfer
    of the batch has a link of its own and nothing is forwarded (a ring all-gather forwards every block world - 2 times over
    one link per hop); results are identical to the collective (the copies are bit-exact either way)."""
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    rows = local_obs.shape[0]
    src = local_obs.contiguous()
    flat[rank * rows:(rank + 1) * rows].copy_(src)
    ops = []
    for k in range(1, world):  # peers in rotating order: at step k every rank sends to rank + k and receives from rank - k
        to, frm = (rank + k) % world, (rank - k) % world
        ops.append(dist.P2POp(dist.isend, src, dist.get_global_rank(group, to) if group is not None else to, group))
        ops.append(dist.P2POp(dist.irecv, flat[frm * rows:(frm + 1) * rows], dist.get_global_rank(group, frm) if group is not None else frm, group))
    for w in dist.batch_isend_irecv(ops) if ops else []:
        w.wait()


def allgather_obs(local_obs, group=None, out=None, algo="collective"):
    """All-gather equal-sized observation shards ([n, d] or [rows, n, d]) rank-major: `out` is [world * n, d], or
    [world, rows, n, d] when given (rank r's block at out[r]; for [n, d] shards rank-major IS global env order).
    algo: "collective" = the backend's all-gather (RCCL picks ring / tree itself), "direct" = 1-hop transfers to and from
    every peer at once (_allgather_direct: the shape of the xGMI mesh)."""
    import torch.distributed as dist

    if algo not in ALGOS:
        raise ValueError(f"algo={algo!r}; known: {ALGOS}")
    if not (dist.is_available() and dist.is_initialized()):
        return local_obs
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * local_obs.shape[0],) + tuple(local_obs.shape[1:]), dtype=local_obs.dtype,
                          device=local_obs.device)
    # the collective sees the receive buffer in its concatenated form ([world * rows, ...]: same memory as the
    # rank-major [world, rows, ...]); gloo accepts only that form
    flat = out.view((world * local_obs.shape[0],) + tuple(local_obs.shape[1:]))
    if dist.get_backend(group) == "gloo" and local_obs.is_cuda:
        # rehearsal path only (gloo has no device all-gather): stage through the host
        host = torch.empty(flat.shape, dtype=out.dtype)
        if algo == "direct":
            _allgather_direct(local_obs.cpu(), host, group)
        else:
            dist.all_gather_into_tensor(host, local_obs.cpu().contiguous(), group=group)
        flat.copy_(host)
        return out
    if algo == "direct":
        _allgather_direct(local_obs, flat, group)
    else:
        dist.all_gather_into_tensor(flat, local_obs.contiguous(), group=group)
    return out


class ObsExchange:
    """Double-buffered all-gather of observation blocks, overlapped with the producer's next launches.

    A block ([rows, n, obs_dim], written on the launch stream into output slot `slot`) is all-gathered on a
    dedicated stream into one of two rank-major receive buffers ([world, rows, n, obs_dim]).  Four hazards are
    ordered with events (device tensors; on CPU tensors — the gloo tests — every call is synchronous):
      ready        launch stream -> comm stream : the producer has written the block
      done[slot]   comm stream -> launch stream : the gather has READ the block; `fence(slot)` makes the launch
                                                  stream wait for it before the slot is overwritten
      filled[b]    comm stream -> consumer      : the gather has WRITTEN receive buffer b; `acquire(buf)` / `reading(buf)`
                                                  make the consumer's stream wait for it
      released[b]  consumer -> comm stream      : the consumer has finished with buffer b; the NEXT gather into b (two
                                                  collectives later: the buffers alternate) waits for it.  A caller that reads every
                                                  observation (zoo/util.py:54-59 is the reference's pattern) on a stream of its
                                                  own is therefore never overwritten; a buffer nobody acquired is free at once."""

    def __init__(self, world, rows, n, obs_dim, n_slots, device, algo="collective"):
        if algo not in ALGOS:
            raise ValueError(f"algo={algo!r}; known: {ALGOS}")
        self.algo = algo
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.gathered = [torch.empty((world, rows, n, obs_dim), dtype=torch.float32, device=self.device) for _ in range(2)]
        self._comm = torch.cuda.Stream(device=self.device) if self.cuda else None
        self._done = [None] * n_slots
        self._filled = [None, None]
        self._released = [None, None]
        self.collectives = 0

    def fence(self, slot):
        if self._done[slot] is not None:
            torch.cuda.current_stream().wait_event(self._done[slot])

    def exchange(self, slot, block, group=None):
        b = self.collectives & 1
        buf = self.gathered[b]
        self.collectives += 1
        if not self.cuda:
            allgather_obs(block, group=group, out=buf, algo=self.algo)
            return buf
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        with torch.cuda.stream(self._comm):
            self._comm.wait_event(ready)
            if self._released[b] is not None:  # a consumer still reads what this buffer held
                self._comm.wait_event(self._released[b])
                self._released[b] = None
            allgather_obs(block, group=group, out=buf, algo=self.algo)
            done = torch.cuda.Event()
            done.record(self._comm)
        self._done[slot] = done
        self._filled[b] = done
        return buf

    def _index(self, buf):
        for b in (0, 1):
            if buf is self.gathered[b]:
                return b
        raise ValueError("not a receive buffer of this exchange")

    def acquire(self, buf):
        """The current stream waits until the gather that fills `buf` (a tensor `exchange` / `last` returned) is complete."""
        b = self._index(buf)
        if self.cuda and self._filled[b] is not None:
            torch.cuda.current_stream().wait_event(self._filled[b])
        return buf

    def release(self, buf):
        """The current stream is done reading `buf`: the next gather into it may start once the stream gets here."""
        b = self._index(buf)
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._released[b] = ev

    def reading(self, buf):
        """`with xchg.reading(buf): consume(buf)` on the consumer's stream = acquire, then release on exit."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            self.acquire(buf)
            try:
                yield buf
            finally:
                self.release(buf)

        return cm()

    def last(self, back=0):
        """receive buffer of the collective `back` before the most recent one (0 or 1)"""
        return self.gathered[(self.collectives - 1 - back) & 1]

    def wait_all(self):
        for ev in self._done:
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)


class _DeviceArray:
    """A raw device pointer as torch.as_tensor accepts it (the CUDA array interface, which the ROCm build of torch honours)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


class PeerWriteExchange:
    """The batched observation return WITHOUT a collective ("hipIpc peer writes from the step kernel's epilogue"; C-ABI:
    emei_set_obs_peers + emei_peer_buffer_*).  Every rank owns two gathered buffers [rows, world * n, obs_dim] float32 (step-major,
    global env order) in memory its peers map through hipIpc; the rollout kernel of every rank stores each observation row into the
    current buffer of EVERY rank — its own included — at its columns, while it computes.  Nothing is re-read from HBM and no
    all-gather follows the launch; 7 x 16 B per env-step leave the GPU over xGMI either way, so the launch runs at the links' rate.

    Ordering (host side, no device-side flags): `begin(c)` selects buffer c & 1 and points the engine at it; after the launch
    `complete()` waits for this rank's launch (an event), for this rank's consumer of the OTHER buffer's previous content (its
    release event), and then joins a barrier of the group.  Past the barrier every rank's rows of chunk c have landed in every
    buffer, and every rank's consumer is done with what chunk c + 1 will overwrite.  The barrier costs a host round trip per
    chunk (~0.1 ms) against milliseconds of link time per chunk.

    How long a block stays: a rank's buffers are written by its PEERS, which may be one launch ahead of it.  The block `complete()`
    returns is whole and stays until this rank's NEXT `complete()` call; a reader that needs it longer reads it under `reading(buf)`
    (that next `complete()` then waits for the reader before it joins the barrier that lets the peers overwrite the buffer).  Unlike
    ObsExchange's receive buffers, the block BEFORE the last one may already be receiving the peers' next chunk.

    The receive buffers are handed out rank-major like ObsExchange's ([world, rows, n, obs_dim], here a strided VIEW of the
    step-major storage; `step_major[b]` is the storage itself, the shape a vectorised consumer wants)."""

    def __init__(self, engine, world, rank, rows, n, obs_dim, group=None):
        import ctypes as C

        import torch.distributed as dist

        from . import _lib as L

        if obs_dim != 4:
            raise NotImplementedError("peer writes are built for the 4-float observation rows of the CartPole family")
        if world > L.MAX_OBS_PEERS:
            raise ValueError(f"world={world} > {L.MAX_OBS_PEERS} gathered buffers per launch")
        self.engine, self.world, self.rank, self.rows, self.n, self.group = engine, world, rank, rows, n, group
        self.device = engine.device
        self.dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.cuda = True
        self._L = L
        nbytes = rows * world * n * obs_dim * 4
        self._own, handles = [], []
        for _ in range(2):
            ptr, h = C.c_void_p(), (C.c_ubyte * 64)()
            L.check(L.lib().emei_peer_buffer_create(self.dev_index, nbytes, C.byref(ptr), h))
            self._own.append(ptr.value)
            handles.append(bytes(h))
        self.distributed = world > 1 and dist.is_available() and dist.is_initialized()
        if world > 1 and not self.distributed:
            raise RuntimeError("PeerWriteExchange with world > 1 needs an initialised torch.distributed group (handles and barriers travel over it)")
        self._opened = []
        ptrs = [[None] * world, [None] * world]
        if self.distributed:
            all_handles = [None] * world
            dist.all_gather_object(all_handles, handles, group=group)
            for r in range(world):
                for b in range(2):
                    if r == rank:
                        ptrs[b][r] = self._own[b]
                    else:
                        p = C.c_void_p()
                        hb = (C.c_ubyte * 64).from_buffer_copy(all_handles[r][b])
                        L.check(L.lib().emei_peer_buffer_open(self.dev_index, hb, C.byref(p)))
                        self._opened.append(p.value)
                        ptrs[b][r] = p.value
        else:
            ptrs = [[self._own[0]], [self._own[1]]]
        self._ptrs = ptrs
        with torch.cuda.device(self.device):
            self.step_major = [torch.as_tensor(_DeviceArray(self._own[b], (rows, world * n, obs_dim)), device=self.device) for b in range(2)]
        assert all(t.data_ptr() == self._own[b] for b, t in enumerate(self.step_major)), "torch copied the buffer instead of viewing it"
        for t in self.step_major:
            t.zero_()
        torch.cuda.synchronize(self.device)
        self.gathered = [t.view(rows, world, n, obs_dim).permute(1, 0, 2, 3) for t in self.step_major]
        self._released = [None, None]
        self._launched = None
        self._current = None
        self.collectives = 0  # exchanges completed (no collective runs; the name is ObsExchange's)
        if self.distributed:
            dist.barrier(group=group)  # every rank has mapped every buffer before anyone writes

    # ObsExchange's surface, so that ShardedRollout and its consumers treat both alike
    def fence(self, slot):
        pass

    def wait_all(self):
        pass

    def begin(self, chunk_index):
        """Before the rollout launch of a chunk: the launch will write buffer `chunk_index & 1` of every rank."""
        b = self.collectives & 1
        self._current = b
        self.engine.set_obs_peers(self._ptrs[b], self.world * self.n, self.rank * self.n, max_steps=self.rows)
        return b

    def complete(self):
        """After the launch: returns the receive buffer once EVERY rank's rows are in it (see the class comment)."""
        import torch.distributed as dist

        b = self._current
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        ev.synchronize()
        other = self._released[b ^ 1]
        if other is not None:  # this rank's consumer of the buffer the NEXT chunk overwrites
            other.synchronize()
            self._released[b ^ 1] = None
        if self.distributed:
            dist.barrier(group=self.group)
        self.collectives += 1
        self._current = None
        return self.gathered[b]

    def _index(self, buf):
        for b in (0, 1):
            if buf is self.gathered[b] or buf is self.step_major[b]:
                return b
        raise ValueError("not a receive buffer of this exchange")

    def acquire(self, buf):
        self._index(buf)  # complete() returned it: already whole
        return buf

    def release(self, buf):
        b = self._index(buf)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._released[b] = ev

    def reading(self, buf):
        import contextlib

        @contextlib.contextmanager
        def cm():
            self.acquire(buf)
            try:
                yield buf
            finally:
                self.release(buf)

        return cm()

    def last(self, back=0):
        return self.gathered[(self.collectives - 1 - back) & 1]

    def close(self):
        """Unmap the peers' buffers, then (after a barrier: nobody writes or maps them any more) free this rank's own."""
        import torch.distributed as dist

        if self._own is None:
            return
        torch.cuda.synchronize(self.device)
        try:
            self.engine.set_obs_peers([], 0, 0)
        except Exception:
            pass  # the engine may already be closed
        for ev in self._released:
            if ev is not None:
                ev.synchronize()
        lib = self._L.lib()
        for p in self._opened:
            self._L.check(lib.emei_peer_buffer_close(self.dev_index, p))
        self._opened = []
        if self.distributed:
            dist.barrier(group=self.group)
        self.gathered = self.step_major = None
        for p in self._own:
            self._L.check(lib.emei_peer_buffer_destroy(self.dev_index, p))
        self._own = None


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
    """This rank's shard of a fused rollout + the all-gather of the batched observation return.

    gather modes (what `run_pass` exchanges; the reference's callers read the observation of EVERY step,
    zoo/util.py:54-59, so the whole [T, n, obs_dim] return is what a sharded caller needs on every rank):
      "final"      only the last [n, obs_dim] of a pass (one small collective per horizon)
      "per_chunk"  the horizon runs as T / chunk launches of `chunk` steps; each chunk's whole
                   [chunk, n, obs_dim] observation block is all-gathered (rank-major: [world, chunk, n, obs_dim])
                   on a dedicated stream while the next chunk's rollout runs
      "per_step"   per_chunk with chunk = 1: one emei_step launch + one [n, obs_dim] all-gather per env-step
    With one rank nothing is exchanged (unless `force_exchange`: a 1-rank group still runs the collective path,
    for tests); "per_chunk" / "per_step" still launch per chunk."""

    MAX_EPISODE_STEPS = {"CartPoleSwingUp": 1000, "CartPoleBalancing": 500}  # register_env.py:14-23

    def __init__(self, env, envs_per_rank, horizon, freq_rate=1, real_time_scale=0.02, precision="ref", rank=0,
                 world=1, device=0, seed=0, init_noise=None, integrator="euler", gather="final", chunk=None,
                 force_exchange=False, solver="newton", exchange_algo="collective", rollout_chunk_steps=0):
        from .engine import Engine

        self.env, self.n, self.horizon, self.rank, self.world = env, int(envs_per_rank), int(horizon), rank, world
        self.lo, self.hi = rank * self.n, (rank + 1) * self.n
        if gather not in ("final", "per_chunk", "per_step"):
            raise ValueError(f"gather={gather!r}")
        self.gather = gather
        self.chunk = self.horizon if gather == "final" else (1 if gather == "per_step" else int(chunk or 125))
        if self.chunk < 1 or self.horizon % self.chunk:
            raise ValueError(f"chunk {self.chunk} must divide the horizon {self.horizon}")
        self.n_chunks = self.horizon // self.chunk
        self.exchanging = world > 1 or bool(force_exchange)
        if exchange_algo not in EXCHANGES:
            raise ValueError(f"exchange_algo={exchange_algo!r}; known: {EXCHANGES}")
        self.exchange_algo = exchange_algo
        self.peer_write = self.exchanging and exchange_algo == "peer_write"
        if self.peer_write and gather != "per_chunk":
            raise ValueError("exchange_algo='peer_write' returns every step's observation: gather='per_chunk' (chunks of at least 16 steps)")
        if init_noise is None:
            init_noise = 0.1 if env == "HalfCheetahRunning" else 5e-3
        self.engine = Engine(env, self.n, freq_rate=freq_rate, real_time_scale=real_time_scale, precision=precision,
                             max_episode_steps=self.MAX_EPISODE_STEPS.get(env, 1000), device=device, seed=seed,
                             env_index_offset=self.lo, init_noise=init_noise, integrator=integrator, solver=solver,
                             rollout_chunk_steps=rollout_chunk_steps)
        self.device = self.engine.device
        self.obs_dim, self.act_dim = self.engine.obs_dim, self.engine.act_dim
        self.seed = seed
        self.actions = self.out = self.gathered = self.xchg = None
        self._events = []

    @property
    def kernel_name(self):
        """the rollout kernel the last launch selected (emei_last_rollout_kernel)"""
        from . import _lib as L

        return L.KERNEL_NAMES[self.engine.last_kernel()]

    @property
    def action_bytes(self):
        return self.actions.element_size() * max(self.act_dim, 1)

    @property
    def action_dtype_name(self):
        return str(self.actions.dtype).replace("torch.", "")

    @property
    def gathered_bytes_per_pass(self):
        """bytes every rank RECEIVES from its peers per pass (the xGMI inbound volume of the observation return)"""
        rows = self.n if self.gather == "final" else self.n * self.horizon
        return (self.world - 1) * rows * self.obs_dim * 4

    @property
    def collectives(self):
        return self.xchg.collectives if self.xchg is not None else 0

    def make_synthetic_inputs(self):
        s0 = synthetic_init_state(self.env, self.world * self.n, self.lo, self.hi, self.seed)
        if s0 is None:
            self.engine.reset(self.seed)
        else:
            self.engine.set_state(s0)
        self.actions = synthetic_actions(self.env, self.horizon, self.n, self.rank, self.world, self.device, self.act_dim)
        self.out = self.engine.alloc_outputs(self.horizon)
        K = self.chunk
        # per-chunk views of the whole-horizon buffers (contiguous: the leading dimension is the step)
        self._act_chunks = [self.actions[c * K:(c + 1) * K] for c in range(self.n_chunks)]
        self._out_chunks = [tuple(o[c * K:(c + 1) * K] for o in self.out) for c in range(self.n_chunks)]
        if self.exchanging:
            # a chunk's observation block is gathered straight from the rollout's output slice (slot = chunk index);
            # "final" copies the last row to a one-row staging buffer first (slot 0)
            rows = 1 if self.gather == "final" else K
            if self.peer_write:
                self.xchg = PeerWriteExchange(self.engine, self.world, self.rank, rows, self.n, self.obs_dim)
            else:
                self.xchg = ObsExchange(self.world, rows, self.n, self.obs_dim, self.n_chunks, self.device, algo=self.exchange_algo)
            self.gathered = self.xchg.gathered
            self._stage = torch.empty((1, self.n, self.obs_dim), dtype=torch.float32, device=self.device)
        torch.cuda.synchronize()

    def run_pass(self, record=False, on_gathered=None):
        """The whole horizon (one launch, or one per chunk) + the batched observation return of the gather mode.
        -> the last receive buffer ([world, rows, n, obs_dim], valid once `wait_gathers` has run) or, without an
        exchange, the local observations of the last chunk.
        on_gathered(chunk, buf): called right after the all-gather of a chunk has been ENQUEUED, with the receive buffer it fills;
        a consumer reads it under `self.xchg.reading(buf)` on a stream of its own (the buffer is then not handed to a later
        gather before the consumer is done — ObsExchange)."""
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        last = None
        for c in range(self.n_chunks):
            if self.exchanging:
                self.xchg.fence(c)  # the gather that read this slice during the previous pass is over before it is rewritten
            if self.peer_write:
                self.xchg.begin(c)  # the launch below writes its observation rows into every rank's current buffer
            obs, rew, done = self.engine.rollout(self._act_chunks[c], auto_reset=True, out=self._out_chunks[c])
            last = obs
            if self.peer_write:
                last = self.xchg.complete()
                if on_gathered is not None:
                    on_gathered(c, last)
            elif self.exchanging and self.gather != "final":
                last = self.xchg.exchange(c, obs)
                if on_gathered is not None:
                    on_gathered(c, last)
        if record:
            e1.record()
            self._events.append((e0, e1))
        if self.exchanging and self.gather == "final":
            self._stage.copy_(last[-1:])  # one slot: fence(0) above also covers the staging buffer
            last = self.xchg.exchange(0, self._stage)
            if on_gathered is not None:
                on_gathered(0, last)
        return last

    def close(self):
        """Release the exchange's device memory (peer-mapped buffers need an orderly teardown on every rank) and the engine."""
        if self.peer_write and self.xchg is not None:
            self.xchg.close()
        self.xchg = None
        self.engine.close()

    def wait_gathers(self):
        """Make the launch stream wait for every outstanding observation all-gather."""
        if self.exchanging:
            self.xchg.wait_all()

    def mean_kernel_ms(self):
        torch.cuda.synchronize()
        ts = [a.elapsed_time(b) for a, b in self._events]
        return float(np.mean(ts)) if ts else float("nan")

    def timed_launches_ms(self, k):
        """Mean duration of one rollout launch: k passes of back-to-back launches (no collective) bracketed by ONE
        pair of HIP events on the launch stream (no per-launch marker packets inside the timed span)."""
        if self.peer_write:
            self.engine.set_obs_peers([], 0, 0)  # the kernel alone: no peer stores (begin() points the engine at them again)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            for c in range(self.n_chunks):
                self.engine.rollout(self._act_chunks[c], auto_reset=True, out=self._out_chunks[c])
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (k * self.n_chunks)

    def time_per_step_api(self, n_steps=200):
        """One launch per env-step (emei_step), the gym-style API: launch-bound by construction."""
        import time

        out = self.engine.alloc_outputs(None)
        if self.peer_write:
            self.engine.set_obs_peers([], 0, 0)
        for t in range(10):
            self.engine.step(self.actions[t % self.horizon], auto_reset=True, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(n_steps):
            self.engine.step(self.actions[t % self.horizon], auto_reset=True, out=out)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        return {"env_steps_per_s": self.n * n_steps / el, "us_per_launch": el / n_steps * 1e6}
