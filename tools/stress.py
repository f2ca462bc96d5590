#!/usr/bin/env python3
"""Randomised determinism soak (run on the GPU box): for random (env, precision, integrator, noise, n, T) a
fused rollout with auto-reset must equal the same rollout cut into random chunks and equal a second identical
run, bit for bit (obs, reward, done, final state, counters).  Catches races / uninitialised reads that the
fixed-shape parity tests could miss.  Usage: python tools/stress.py [seconds] [big]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from emei_amd import _lib as L  # noqa: E402
from emei_amd.engine import Engine  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
big = len(sys.argv) > 2 and sys.argv[2] == "big"  # full-size shards and longer horizons (fewer cases)
rng = np.random.default_rng(12345)
names = sorted(L.ENV_IDS)
t0, cases = time.time(), 0
last_print = t0
while time.time() - t0 < budget:
    name = names[rng.integers(len(names))]
    classic = name.startswith("CartPole")
    n = int(rng.choice([65536, 100000, 131072] if big else [1, 63, 64, 65, 200, 256, 1000, 4096, 5000]))
    T = int(rng.choice([33, 100, 250] if big else [1, 3, 16, 17, 40, 64]))
    kw = dict(freq_rate=int(rng.integers(1, 4)), precision=str(rng.choice(["ref", "f32"])), seed=int(rng.integers(1 << 30)),
              max_episode_steps=int(rng.choice([0, 5, 20])), env_index_offset=int(rng.integers(0, 1 << 20)))
    kw["real_time_scale"] = 0.02 if (classic or "Pendulum" in name) else 0.002
    if not classic:
        kw["integrator"] = str(rng.choice(["euler", "semi_implicit_euler", "rk4"]))
        kw["init_noise"] = float(rng.choice([0.0, 5e-3, 0.1]))
        kw["obs_noise"] = float(rng.choice([0.0, 0.0, 1e-3]))
        kw["noise_layout"] = str(rng.choice(["iid", "shared"]))
    # round 5: the second engine runs its body rollouts as chunked launches (persistent workers on work items) with a random schedule,
    # the first as one piece, the third decides for itself: all must agree bit for bit (ignored by the 4-state staged kernels);
    # CartPole also takes a random ODE_approximation method
    if classic:
        kw["ode_method"] = str(rng.choice(["euler", "rk4"]))
    chunk_b = int(rng.choice([1, 2, 5, 7, -101, -201, -302, -403]))
    engs = [Engine(name, n, rollout_chunk_steps=c, **kw) for c in (-1, chunk_b, 0)]
    for e in engs:
        e.reset(kw["seed"])
    # round 5: the CartPole family's second engine also writes its observation rows into a random number of gathered buffers
    # (emei_set_obs_peers) when the staged kernel serves the shape: same bits, and the rows land at their columns
    peers = None
    if classic and n % 64 == 0 and T >= 16 and rng.integers(2):
        col, total = int(rng.integers(0, 3)) * n, 3 * n
        peers = [torch.full((T, total, 4), float("nan"), device="cuda") for _ in range(int(rng.integers(1, 9)))]
        engs[1].set_obs_peers(peers, total, col)
    if engs[0].act_dim == 0:
        acts = torch.randint(0, 2, (T, n), device="cuda", dtype=[torch.uint8, torch.int32, torch.int64][rng.integers(3)])
    else:
        shape = (T, n) if engs[0].act_dim == 1 else (T, n, engs[0].act_dim)
        acts = (torch.rand(shape, device="cuda") * 2.4 - 1.2).float()
    ref = engs[0].rollout(acts, auto_reset=True)
    again = engs[1].rollout(acts, auto_reset=True)
    if peers is not None:
        for pb in peers:
            assert bool((pb[:, col:col + n] == ref[0]).all()), ("peer rows", name, n, T, kw)
            assert bool(pb[:, :col].isnan().all()) and bool(pb[:, col + n:].isnan().all()), ("peer columns", name, n, T, kw)
        engs[1].set_obs_peers([], 0, 0)
    cuts = sorted(set([0, T] + [int(c) for c in rng.integers(0, T + 1, 2)]))
    parts = [engs[2].rollout(acts[a:b].contiguous(), auto_reset=True) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    for k in range(3):
        cat = torch.cat([p[k] for p in parts])
        same = lambda x, y: bool(((x == y) | (x.float().isnan() & y.float().isnan())).all())
        assert same(ref[k], again[k]), ("one piece != work items", name, n, T, kw, chunk_b, k)
        assert same(ref[k], cat), ("chunking changes the result", name, n, T, kw, k)
    for e in engs[1:]:
        a, b = engs[0].get_state(), e.get_state()
        assert bool(((a == b) | (a.isnan() & b.isnan())).all()), ("state", name, n, T, kw)
        assert all(torch.equal(x, y) for x, y in zip(engs[0].get_counters(), e.get_counters())), ("counters", name, n, T, kw)
    for e in engs:
        e.close()
    cases += 1
    if time.time() - last_print > 60:  # a long silent GPU run is taken for a hang by the job runner
        last_print = time.time()
        print(f"... {cases} cases, {time.time() - t0:.0f} s", flush=True)
torch.cuda.synchronize()
print(f"stress ok: {cases} random cases in {time.time() - t0:.0f} s")
