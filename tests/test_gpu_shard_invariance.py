"""What the multi-GPU partition rests on: an env's trajectory is a function of its GLOBAL index,
the seed and its actions only — not of the shard it lives in, its lane, its wave-mates or the number of ranks.

One handle steps all N envs; three handles step the contiguous shards [0, 64), [64, 200), [200, N) with
`env_index_offset` = the shard start (what `ShardedRollout` passes per rank).  Device reset (Philox keyed by the global
env index), T steps with auto-reset so that episodes end and re-draw inside the window, then every output and the final
state are compared BIT FOR BIT.  The shard sizes mix the staged kernel (multiples of 64) with the generic one, and put
envs that share a wave in the whole run into different waves of the shards: in the Newton solver a lane leaves the
iteration on its own gradient (cheetah_model.h), so not even the constraint solve may see its neighbours.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

CASES = [
    # env, kwargs, horizon
    ("CartPoleSwingUp", dict(freq_rate=1, real_time_scale=0.02, max_episode_steps=30), 80),
    ("CartPoleBalancing", dict(freq_rate=1, real_time_scale=0.02, max_episode_steps=500), 120),
    ("BoundaryInvertedPendulumSwingUp", dict(freq_rate=4, real_time_scale=0.02, max_episode_steps=25, init_noise=5e-3), 60),
    ("ReboundInvertedDoublePendulumBalancing", dict(freq_rate=2, real_time_scale=0.02, max_episode_steps=25, init_noise=5e-3), 60),
    ("HalfCheetahRunning", dict(freq_rate=4, real_time_scale=0.002, max_episode_steps=40, init_noise=0.1, solver="newton"), 90),
    ("HalfCheetahRunning", dict(freq_rate=2, real_time_scale=0.002, max_episode_steps=40, init_noise=0.1, solver="sweep1", integrator="rk4"), 60),
    ("HopperRunning", dict(freq_rate=2, real_time_scale=0.002, max_episode_steps=40, init_noise=0.1, solver="newton", integrator="rk4"), 90),
    ("HopperRunning", dict(freq_rate=4, real_time_scale=0.002, max_episode_steps=40, init_noise=0.1, solver="newton", obs_noise=1e-3), 60),
]


def _actions(eng, T, n, rng):
    if eng.act_dim == 0:
        return torch.as_tensor(rng.integers(2, size=(T, n), dtype=np.uint8), device=eng.device)
    lim = 3.0 if eng.act_dim == 1 else 1.0
    shape = (T, n) if eng.act_dim == 1 else (T, n, eng.act_dim)
    return torch.as_tensor(rng.uniform(-lim, lim, size=shape).astype(np.float32), device=eng.device)


@pytest.mark.parametrize("case", range(len(CASES)), ids=[f"{c[0]}-{c[1].get('solver', '')}{c[1].get('integrator', '')}" for c in CASES])
def test_trajectories_do_not_depend_on_the_shard(case):
    from emei_amd.engine import Engine

    name, kw, T = CASES[case]
    N, seed = 328, 77
    bounds = [(0, 64), (64, 200), (200, N)]
    whole = Engine(name, N, seed=seed, **kw)
    whole.reset(seed)
    acts = _actions(whole, T, N, np.random.default_rng(5))
    obs, rew, done = (x.cpu().numpy() for x in whole.rollout(acts, auto_reset=True))
    final = whole.get_state().cpu().numpy()
    steps, epi = (x.cpu().numpy() for x in whole.get_counters())
    assert (done != 0).any() and epi.max() >= 1, "the window must contain resets"
    for lo, hi in bounds:
        part = Engine(name, hi - lo, seed=seed, env_index_offset=lo, **kw)
        part.reset(seed)
        o, r, d = (x.cpu().numpy() for x in part.rollout(acts[:, lo:hi].contiguous(), auto_reset=True))
        for got, want, what in ((o, obs[:, lo:hi], "obs"), (r, rew[:, lo:hi], "reward"), (d, done[:, lo:hi], "done"),
                                (part.get_state().cpu().numpy(), final[lo:hi], "final state")):
            same = (got == want) | (np.isnan(got.astype(np.float64)) & np.isnan(want.astype(np.float64)))
            assert same.all(), (name, (lo, hi), what, int((~same).sum()), np.argwhere(~same)[:3].tolist())
        s, e = (x.cpu().numpy() for x in part.get_counters())
        assert np.array_equal(s, steps[lo:hi]) and np.array_equal(e, epi[lo:hi]), (name, (lo, hi), "counters")
