"""The N>1 path on CPU: world_size-2 and -4 gloo processes exercising the shard partition and the all-gather
of the batched observation return (RCCL on GPUs; the 8-rank run itself is the driver's)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_per_rank, q, algo="collective"):
    sys.path.insert(0, ROOT)
    from emei_amd.sharding import allgather_obs, shard_bounds, synthetic_init_state

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_global = world * n_per_rank
        lo, hi = shard_bounds(n_global, rank, world)
        assert (lo, hi) == (rank * n_per_rank, (rank + 1) * n_per_rank)
        # every rank draws the GLOBAL init array and keeps its slice: results independent of world size
        full = synthetic_init_state("CartPoleSwingUp", n_global, 0, n_global, seed=0)
        mine = synthetic_init_state("CartPoleSwingUp", n_global, lo, hi, seed=0)
        assert np.array_equal(full[lo:hi], mine)
        local = torch.as_tensor(mine, dtype=torch.float32)
        for _ in range(3):  # repeated collectives, as in a stepping loop
            gathered = allgather_obs(local, algo=algo)
        ok = torch.equal(gathered, torch.as_tensor(full, dtype=torch.float32))
        if algo == "direct":  # the point-to-point form fills the same buffer as the collective, bit for bit
            ok &= torch.equal(gathered, allgather_obs(local, algo="collective"))
        q.put((rank, bool(ok), tuple(gathered.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world,algo", [(2, "collective"), (4, "collective"), (2, "direct"), (4, "direct"), (3, "direct"),
                                        (8, "collective"), (8, "direct")])  # 8 = the ranks of BASELINE configs[4]
def test_allgather_obs(world, algo):
    n = 96
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q, algo)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert res == [(r, True, (world * n, 4)) for r in range(world)]


def _xchg_worker(rank, world, port, q, algo="collective"):
    """ObsExchange (the buffer rotation under ShardedRollout.run_pass) on CPU tensors: per-chunk blocks of a
    [T, n, d] return over several passes; every receive buffer must hold every rank's block of that collective."""
    sys.path.insert(0, ROOT)
    from emei_amd.sharding import ObsExchange

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        T, K, n, d = 12, 4, 10, 4
        x = ObsExchange(world, K, n, d, T // K, "cpu", algo=algo)

        def block(r, p, c):  # what rank r produces for chunk c of pass p
            return (torch.arange(K * n * d, dtype=torch.float32).reshape(K, n, d) + 1000.0 * r + 100.0 * p + 10.0 * c)

        ok = True
        for p in range(3):
            for c in range(T // K):
                x.fence(c)
                buf = x.exchange(c, block(rank, p, c))
                ok &= buf is x.last(0) and all(torch.equal(buf[r], block(r, p, c)) for r in range(world))
                if x.collectives >= 2:  # the previous collective's buffer is still intact (double buffering)
                    pp, pc = (p, c - 1) if c > 0 else (p - 1, T // K - 1)
                    ok &= all(torch.equal(x.last(1)[r], block(r, pp, pc)) for r in range(world))
        x.wait_all()
        q.put((rank, bool(ok), x.collectives))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world,algo", [(2, "collective"), (4, "collective"), (4, "direct"), (8, "collective"), (8, "direct")])
def test_obs_exchange_chunks(world, algo):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_xchg_worker, args=(r, world, port, q, algo)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert res == [(r, True, 9) for r in range(world)]


def test_shard_bounds_cover_and_balance():
    from emei_amd.sharding import shard_bounds

    for n, w in ((1048576, 8), (65536, 1), (10, 3), (7, 8)):
        spans = [shard_bounds(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
