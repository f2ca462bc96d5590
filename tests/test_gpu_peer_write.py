"""The multi-GPU observation return by PEER WRITES (emei_set_obs_peers / emei_peer_buffer_*; sharding.PeerWriteExchange): the staged
rollout kernel stores every observation row into up to 8 gathered buffers while it computes.  One process, one GPU here — the "peers"
are further buffers of this process, which is everything the kernel and the C-ABI can tell apart; two processes mapping each other's
buffers through hipIpc are tests/test_gpu_sharded_rollout.py::test_peer_write_exchange."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _engine(*a, **k):
    from emei_amd.engine import Engine

    return Engine(*a, **k)


@pytest.mark.parametrize("env,kw", [("CartPoleSwingUp", {}), ("CartPoleBalancing", {"freq_rate": 2}), ("CartPoleSwingUp", {"ode_method": "rk4"}),
                                    ("CartPoleBalancing", {"precision": "f32"})])
@pytest.mark.parametrize("n,T", [(1024, 37), (131072, 32)])  # 37: a tail of 5 steps behind two tiles; 131 072: split launches, XCD-contiguous map
@pytest.mark.parametrize("adtype", ["uint8", "int64"])
def test_rollout_with_peers_writes_every_row_everywhere_and_changes_nothing_else(env, kw, n, T, adtype):
    from emei_amd import _lib

    if n > 1024 and (adtype != "uint8" or kw):
        pytest.skip("the large shape once per env")
    a, b = _engine(env, n, max_episode_steps=20, seed=3, **kw), _engine(env, n, max_episode_steps=20, seed=3, **kw)
    a.reset(3), b.reset(3)
    acts = torch.randint(0, 2, (T, n), device=a.device, dtype=getattr(torch, adtype))
    world, rank = 3, 1
    peers = [torch.full((T, world * n, 4), -7.0, device=a.device) for _ in range(3)]
    b.set_obs_peers(peers, world * n, rank * n)
    ref = a.rollout(acts, auto_reset=True)
    got = b.rollout(acts, auto_reset=True)
    assert b.last_kernel() in (_lib.KERNEL_PEND_STAGED_PEERS_FREQ1, _lib.KERNEL_PEND_STAGED_PEERS)
    assert a.last_kernel() in (_lib.KERNEL_PEND_STAGED_FREQ1, _lib.KERNEL_PEND_STAGED)
    for x, y in zip(ref, got):
        assert torch.equal(x, y)
    assert torch.equal(a.get_state(), b.get_state()) and all(torch.equal(x, y) for x, y in zip(a.get_counters(), b.get_counters()))
    for p in peers:
        assert torch.equal(p[:, rank * n:(rank + 1) * n], ref[0])  # this shard's columns: every step's row
        assert bool((p[:, :rank * n] == -7.0).all()) and bool((p[:, (rank + 1) * n:] == -7.0).all())  # nobody else's
    # switched off again: the plain kernel, the buffers untouched
    for p in peers:
        p.fill_(-7.0)
    b.set_obs_peers([], 0, 0)
    b.rollout(acts, auto_reset=True)
    assert b.last_kernel() in (_lib.KERNEL_PEND_STAGED_FREQ1, _lib.KERNEL_PEND_STAGED)
    assert all(bool((p == -7.0).all()) for p in peers)


def test_eight_peers_and_the_limits():
    n, T = 256, 16
    e = _engine("CartPoleSwingUp", n)
    e.reset(0)
    acts = torch.randint(0, 2, (T, n), device=e.device, dtype=torch.uint8)
    peers = [torch.zeros((T, n, 4), device=e.device) for _ in range(8)]
    e.set_obs_peers(peers, n, 0)
    obs, _, _ = e.rollout(acts)
    assert all(torch.equal(p, obs) for p in peers)
    with pytest.raises(ValueError, match="n_peers"):
        e.set_obs_peers(peers + [torch.zeros((T, n, 4), device=e.device)], n, 0)
    with pytest.raises(ValueError, match="do not fit"):
        e.set_obs_peers(peers[:1], n, 64)
    with pytest.raises(ValueError, match="aligned"):
        e.set_obs_peers([peers[0].data_ptr() + 4], n, 0, max_steps=T)
    with pytest.raises(ValueError, match="max_steps"):
        e.set_obs_peers([peers[0].data_ptr()], n, 0)
    with pytest.raises(ValueError, match="float32"):
        e.set_obs_peers([torch.zeros((T, n, 4), device=e.device, dtype=torch.float64)], n, 0)
    e.set_obs_peers(peers[:2], n, 0)
    with pytest.raises(ValueError, match="hold 16 rows"):  # a longer rollout would write past the buffers: refused, nothing launched
        e.rollout(torch.randint(0, 2, (T + 16, n), device=e.device, dtype=torch.uint8))


def test_paths_without_peer_stores_refuse_instead_of_skipping():
    """nothing is skipped silently: a rollout the staged CartPole kernel cannot serve raises while peers are set"""
    n = 256
    e = _engine("CartPoleSwingUp", n)
    e.reset(0)
    peer = torch.zeros((64, n, 4), device=e.device)
    e.set_obs_peers([peer], n, 0)
    with pytest.raises(NotImplementedError, match="peers"):  # a horizon below one staged tile: the generic kernel
        e.rollout(torch.randint(0, 2, (8, n), device=e.device, dtype=torch.uint8))
    with pytest.raises(NotImplementedError, match="peers"):  # emei_step is a rollout of one step
        e.step(torch.randint(0, 2, (n,), device=e.device, dtype=torch.uint8))
    e.set_obs_peers([], 0, 0)
    e.step(torch.randint(0, 2, (n,), device=e.device, dtype=torch.uint8))
    r = _engine("CartPoleSwingUp", 1000)  # a ragged shard
    r.reset(0)
    r.set_obs_peers([torch.zeros((16, 1000, 4), device=r.device)], 1000, 0)
    with pytest.raises(NotImplementedError, match="multiple of 64"):
        r.rollout(torch.randint(0, 2, (16, 1000), device=r.device, dtype=torch.uint8))
    for name in ("BoundaryInvertedPendulumSwingUp", "HalfCheetahRunning"):
        o = _engine(name, 64)
        with pytest.raises(NotImplementedError, match="CartPole"):
            o.set_obs_peers([torch.zeros((16, 64, o.obs_dim), device=o.device)], 64, 0)


def test_peer_buffer_create_view_destroy():
    """emei_peer_buffer_create: hipMalloc + an ipc handle; torch views the raw pointer (no copy); opening one's OWN handle is not a thing
    (the owner uses the pointer), null / zero arguments are refused."""
    from emei_amd import _lib as L
    from emei_amd.sharding import _DeviceArray

    dev = torch.cuda.current_device()
    ptr, h = C.c_void_p(), (C.c_ubyte * 64)()
    L.check(L.lib().emei_peer_buffer_create(dev, 4 * 1024 * 4, C.byref(ptr), h))
    assert ptr.value and ptr.value % 256 == 0 and any(bytes(h))
    t = torch.as_tensor(_DeviceArray(ptr.value, (4, 256, 4)), device=f"cuda:{dev}")
    assert t.data_ptr() == ptr.value and t.dtype == torch.float32
    t.fill_(2.5)
    torch.cuda.synchronize()
    assert float(t.sum().item()) == 2.5 * 4096
    del t
    L.check(L.lib().emei_peer_buffer_destroy(dev, ptr))
    with pytest.raises(ValueError):
        L.check(L.lib().emei_peer_buffer_create(dev, 0, C.byref(ptr), h))
    with pytest.raises(ValueError):
        L.check(L.lib().emei_peer_buffer_destroy(dev, None))


@pytest.mark.parametrize("slow", [False, True])
def test_sharded_rollout_by_peer_writes_in_one_rank(slow):
    """ShardedRollout(exchange_algo="peer_write") with a one-rank job (force_exchange): the gathered buffers hold every chunk's block,
    double-buffered; a slow consumer on its own stream reads every one of them intact (release events honoured by complete())."""
    from emei_amd.sharding import ShardedRollout

    n, T, K = 4096, 128, 16
    sr = ShardedRollout("CartPoleSwingUp", n, T, gather="per_chunk", chunk=K, force_exchange=True, exchange_algo="peer_write")
    ref = ShardedRollout("CartPoleSwingUp", n, T, gather="per_chunk", chunk=K, force_exchange=False)
    sr.make_synthetic_inputs(), ref.make_synthetic_inputs()
    stream = torch.cuda.Stream()
    spin = torch.ones((512, 512), device=sr.device)
    seen = []

    def consumer(c, buf):
        with torch.cuda.stream(stream):
            with sr.xchg.reading(buf):
                x = spin
                for _ in range(20):
                    x = (x @ spin) * 1e-3
                seen.append((c, buf.clone(), float(x[0, 0].item() * 0)))

    for p in range(2):
        last = sr.run_pass(on_gathered=consumer if slow else None)
        ref.run_pass()
        torch.cuda.synchronize()
        want = ref.out[0]
        assert torch.equal(sr.out[0], want)
        assert tuple(last.shape) == (1, K, n, 4) and torch.equal(last[0], want[-K:])
        assert torch.equal(sr.xchg.last(1)[0], want[-2 * K:-K])
        assert torch.equal(sr.xchg.step_major[(sr.collectives - 1) & 1], want[-K:])  # one rank: step-major == the block itself
        if slow:
            stream.synchronize()
            assert len(seen) == (p + 1) * T // K
            for c, copy, _ in seen[p * T // K:]:
                assert torch.equal(copy[0], want[c * K:(c + 1) * K]), f"chunk {c} was overwritten under its reader"
    assert sr.collectives == 2 * T // K and "peers" in sr.kernel_name
    assert sr.timed_launches_ms(2) > 0 and "peers" not in sr.kernel_name  # the kernel-only timing runs without the peer stores
    with pytest.raises(ValueError, match="per_chunk"):
        ShardedRollout("CartPoleSwingUp", n, T, gather="final", force_exchange=True, exchange_algo="peer_write")
    sr.close(), ref.close()
