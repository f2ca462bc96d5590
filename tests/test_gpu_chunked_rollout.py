"""Chunked body rollouts (emei_config.rollout_chunk_steps; body_kernels.h:WorkQueue): a launch cut into (64 envs) x (k steps)
work items handed out by an atomic ticket must give the SAME BITS as the one-piece launch — outputs, final state, counters,
done masks — for every body that runs one-wave blocks, ragged shards, auto-reset across item boundaries, RK4, and when the
launch is replayed from a hipGraph.  (mujoco_env.py:157-167 is what every item runs, step for step.)"""
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _engine(*a, **k):
    from emei_amd.engine import Engine

    return Engine(*a, **k)


def _acts(eng, T, scale=1.0):
    shape = (T, eng.n_envs) if eng.act_dim <= 1 else (T, eng.n_envs, eng.act_dim)
    return ((torch.rand(shape, device=eng.device) * 2 - 1) * scale).float()


CASES = [
    # env, kwargs, action scale
    ("HalfCheetahRunning", dict(freq_rate=2, real_time_scale=0.002, init_noise=0.1, max_episode_steps=9), 1.2),
    ("HalfCheetahRunning", dict(freq_rate=1, real_time_scale=0.002, init_noise=0.1, max_episode_steps=9, integrator="rk4"), 1.0),
    ("HopperRunning", dict(freq_rate=2, real_time_scale=0.002, init_noise=5e-3, max_episode_steps=9, integrator="euler"), 1.0),
    ("HopperRunning", dict(freq_rate=1, real_time_scale=0.002, init_noise=5e-3, max_episode_steps=9, integrator="rk4"), 1.0),
    ("BoundaryInvertedDoublePendulumSwingUp", dict(freq_rate=4, real_time_scale=0.02, init_noise=5e-3, max_episode_steps=9), 3.0),
    ("BoundaryInvertedPendulumBalancing", dict(freq_rate=4, real_time_scale=0.02, init_noise=5e-3, max_episode_steps=9, integrator="rk4"), 3.0),
]


@pytest.mark.parametrize("env,kw,scale", CASES, ids=[f"{c[0]}-{c[1].get('integrator', 'euler')}" for c in CASES])
@pytest.mark.parametrize("N", [64 * 37 + 17, 64, 1])
@pytest.mark.parametrize("chunk", [1, 5, -102, -301])  # fixed lengths; guided schedules (a quarter / half of what remains per item, at least 1 / 3)
def test_chunked_rollout_is_bit_identical(env, kw, scale, N, chunk):
    from emei_amd import _lib

    T = 23  # not a multiple of either chunk length: a short last item
    torch.manual_seed(N + abs(chunk))
    a = _engine(env, N, seed=3, rollout_chunk_steps=-1, **kw)
    b = _engine(env, N, seed=3, rollout_chunk_steps=chunk, **kw)
    a.reset(3)
    b.reset(3)
    acts = _acts(a, T, scale)
    oa, ra, da = a.rollout(acts, auto_reset=True)
    ob, rb, db = b.rollout(acts, auto_reset=True)
    assert a.last_kernel() in (_lib.KERNEL_BODY, _lib.KERNEL_BODY_RK4)
    assert b.last_kernel() in (_lib.KERNEL_BODY_CHUNKED, _lib.KERNEL_BODY_RK4_CHUNKED)
    assert b.rollout_faults() == 0 and a.rollout_faults() == 0
    assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db)
    assert torch.equal(a.get_state(), b.get_state())
    assert all(torch.equal(x, y) for x, y in zip(a.get_counters(), b.get_counters()))
    assert torch.equal(a.compact_done(), b.compact_done())
    assert bool((da != 0).any())  # TimeLimit 9 < T (or an earlier terminal): resets happened inside items, episode counters crossed item boundaries
    # a second launch on the same handle: the ticket and progress words start from zero again
    acts2 = _acts(a, 11, scale)
    o2a, _, d2a = a.rollout(acts2, auto_reset=True)
    o2b, _, d2b = b.rollout(acts2, auto_reset=True)
    assert torch.equal(o2a, o2b) and torch.equal(d2a, d2b) and b.rollout_faults() == 0
    assert a.solver_cap_hits() == 0 and b.solver_cap_hits() == 0


def test_chunk_policy():
    """-1 = off; k >= n_steps = one piece; 0 = automatic: only when the shard has more waves than the device holds at once
    (a small shard has nothing to balance)."""
    from emei_amd import _lib

    kw = dict(freq_rate=1, real_time_scale=0.002, init_noise=0.1, seed=1)
    for chunk, T, want in ((-1, 20, _lib.KERNEL_BODY), (20, 20, _lib.KERNEL_BODY), (25, 20, _lib.KERNEL_BODY), (19, 20, _lib.KERNEL_BODY_CHUNKED),
                           (0, 20, _lib.KERNEL_BODY)):
        e = _engine("HalfCheetahRunning", 256, rollout_chunk_steps=chunk, **kw)
        e.reset(1)
        e.rollout(_acts(e, T))
        assert e.last_kernel() == want, (chunk, T)
        e.close()
    # the step path never chunks
    e = _engine("HalfCheetahRunning", 256, rollout_chunk_steps=1, **kw)
    e.reset(1)
    e.step(_acts(e, 1)[0])
    assert e.last_kernel() == _lib.KERNEL_BODY
    # the staged 4-state family has no work queue: the field is accepted and unused
    c = _engine("CartPoleSwingUp", 256, rollout_chunk_steps=4)
    c.reset(1)
    c.rollout(torch.randint(0, 2, (32, 256), device=c.device, dtype=torch.uint8))
    assert c.last_kernel() == _lib.KERNEL_PEND_STAGED_FREQ1 and c.rollout_faults() == 0


def test_automatic_policy_chunks_a_shard_larger_than_the_device():
    """BASELINE configs[3]'s shape (131 072 envs = 2048 waves on 1024 SIMDs) takes the chunked path by itself and gives the one-piece
    launch's bits (8 steps here: the full horizon is tests/test_gpu_bench_shape.py's)."""
    from emei_amd import _lib

    N, T = 131072, 8
    kw = dict(freq_rate=4, real_time_scale=0.002, init_noise=0.1, seed=2, max_episode_steps=1000)
    a = _engine("HalfCheetahRunning", N, rollout_chunk_steps=-1, **kw)
    b = _engine("HalfCheetahRunning", N, **kw)
    a.reset(2)
    b.reset(2)
    acts = _acts(a, T)
    oa, ra, da = a.rollout(acts, auto_reset=True)
    ob, rb, db = b.rollout(acts, auto_reset=True)
    assert a.last_kernel() == _lib.KERNEL_BODY and b.last_kernel() == _lib.KERNEL_BODY_CHUNKED
    assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(a.get_state(), b.get_state())
    assert b.rollout_faults() == 0


def test_chunked_rollout_in_a_hip_graph():
    """The launch path stays capturable (one memset node + one kernel node) and a replay re-arms the queue."""
    N, T = 64 * 20, 12
    kw = dict(freq_rate=2, real_time_scale=0.002, init_noise=0.1, seed=4, rollout_chunk_steps=3)
    ref = _engine("HalfCheetahRunning", N, **dict(kw, rollout_chunk_steps=-1))
    e = _engine("HalfCheetahRunning", N, **kw)
    ref.reset(4)
    e.reset(4)
    acts = _acts(e, T)
    out = e.alloc_outputs(T)
    side = torch.cuda.Stream(device=e.device)
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        e.rollout(acts, out=out)
    torch.cuda.current_stream().wait_stream(side)
    e.reset(4)  # capture does not execute; start over from the reset state
    for rep in range(3):
        g.replay()
        o, r, d = ref.rollout(acts)
        torch.cuda.synchronize()
        assert torch.equal(out[0], o) and torch.equal(out[1], r) and torch.equal(out[2], d), rep
        acts.copy_(_acts(e, T))  # refill the graph's static input for the next replay
    assert e.rollout_faults() == 0
