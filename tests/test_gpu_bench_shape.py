"""The BENCHMARKED kernel at the BENCHMARKED shapes, directly against the reference's golden trajectories
and the reset-injecting oracle ("What's weak" #1).

bench.py times `emei_rollout` of 1000 steps with uint8 actions and device auto-reset on 65 536 envs
(BASELINE configs[1]; 131 072 per GPU for configs[4]); that call is served by
`pend_rollout_staged_kernel<CartPole<0,double>, uint8_t, true>`.  Every test here asserts through
`emei_last_rollout_kernel` that the staged kernel is the one that ran.  Tolerance (BASELINE.json
north_star): 1e-5 relative on the float32 trajectories (|ref| < 1 compared absolutely), masks bit-exact.
"""
import numpy as np
import pytest

from conftest import oracle_autoreset_rollout, rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

RTOL = 1e-5
ENV = {"swingup": "CartPoleSwingUp", "balancing": "CartPoleBalancing"}


def _engine(*a, **k):
    from emei_amd.engine import Engine

    return Engine(*a, **k)


@pytest.mark.parametrize("name", ["swingup", "balancing"])
@pytest.mark.parametrize("fr", [1, 4])
def test_staged_kernel_reproduces_golden_trajectories(cartpole_golden, name, fr):
    """N = 64 (one whole wave), T = 1000, uint8 actions through emei_rollout: the four golden trajectories of
    (variant, freq_rate) tiled x16 over the lanes.  fr = 1 -> the FREQ1 instantiation bench.py times, fr = 4 ->
    the substep-loop instantiation.  All 1000 open-loop steps of the reference (base_control.py:61-83 never
    resets), terminal masks bit for bit."""
    from emei_amd import _lib as L

    g = cartpole_golden
    tags = [f"traj_{name}_fr{fr}_seed{s}" for s in range(4)]
    lane_tag = [tags[l % 4] for l in range(64)]
    s0 = np.stack([g[t + "_states"][0] for t in lane_tag])
    acts = np.stack([g[t + "_actions"] for t in lane_tag], axis=1).astype(np.uint8)  # [1000, 64]
    assert acts.shape == (1000, 64)
    eng = _engine(ENV[name], 64, freq_rate=fr, real_time_scale=0.02, precision="ref")
    eng.set_state(s0)
    obs, rew, done = eng.rollout(torch.as_tensor(acts, device=eng.device))
    assert eng.last_kernel() == (L.KERNEL_PEND_STAGED_FREQ1 if fr == 1 else L.KERNEL_PEND_STAGED)
    obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
    for l, t in enumerate(lane_tag):
        assert rel_err(obs[:, l], g[t + "_states"][1:]) <= RTOL, (l, t)
        assert rel_err(rew[:, l], g[t + "_reward"]) <= RTOL, (l, t)
        assert np.array_equal(done[:, l] & 1, g[t + "_terminal"].astype(np.uint8)), (l, t)
    assert not (done & 2).any()  # no TimeLimit on the handle: never truncated (base_control.py:80)
    # the same 64 trajectories through the generic kernel (ragged n = 63 + 1): bit-identical outputs
    eng2 = _engine(ENV[name], 63, freq_rate=fr, real_time_scale=0.02, precision="ref")
    eng2.set_state(s0[:63])
    o2, r2, d2 = eng2.rollout(torch.as_tensor(np.ascontiguousarray(acts[:, :63]), device=eng.device))
    assert eng2.last_kernel() == L.KERNEL_PEND_GENERIC_FULL
    assert np.array_equal(o2.cpu().numpy(), obs[:, :63]) and np.array_equal(d2.cpu().numpy(), done[:, :63])


def _sample_envs(n, k, rng):
    """k env indices covering the first and last wave, the wave boundaries of the first and last block, and a
    random spread over the rest."""
    fixed = np.concatenate([np.arange(64), np.arange(n - 64, n), np.arange(192, 320), np.arange(n - 320, n - 192)])
    fixed = np.unique(fixed[(fixed >= 0) & (fixed < n)])
    rest = rng.choice(n, size=max(k - len(fixed), 0), replace=False)
    return np.unique(np.concatenate([fixed, rest]))


def _bench_pass_vs_oracle(n, rank=0, world=1, k=512):
    """bench.py's exact call (ShardedRollout: host-drawn init state, default_rng actions, seed 0, TimeLimit 1000,
    one fused 1000-step launch with device auto-reset) against the oracle on k sampled envs, all 1000 steps."""
    from emei_amd import _lib as L
    from emei_amd.sharding import ShardedRollout

    T = 1000
    sr = ShardedRollout("CartPoleSwingUp", n, T, freq_rate=1, real_time_scale=0.02, precision="ref", rank=rank,
                        world=world, device=torch.cuda.current_device(), seed=0)
    sr.make_synthetic_inputs()
    assert sr.actions.dtype == torch.uint8 and tuple(sr.actions.shape) == (T, n)
    idx = _sample_envs(n, k, np.random.default_rng(n + rank))
    tidx = torch.as_tensor(idx, device=sr.device)
    s0 = sr.engine.get_state()[tidx].cpu().numpy()
    obs, rew, done = sr.engine.rollout(sr.actions, auto_reset=True, out=sr.out)  # == ShardedRollout.run_pass's launch
    torch.cuda.synchronize()
    assert sr.engine.last_kernel() == L.KERNEL_PEND_STAGED_FREQ1
    acts = sr.actions[:, tidx].cpu().numpy()
    o_obs, o_rew, o_done, o_st = oracle_autoreset_rollout("swingup", s0, acts, 0, sr.lo + idx, 1000)
    g_obs, g_rew, g_done = obs[:, tidx].cpu().numpy(), rew[:, tidx].cpu().numpy(), done[:, tidx].cpu().numpy()
    assert np.array_equal(g_done, o_done)
    assert (o_done[:-1] & 1).any() and (o_done[-1] & 2).any()  # terminals + resets inside, truncation at step 1000
    assert rel_err(g_obs, o_obs) <= RTOL
    assert rel_err(g_rew, o_rew) <= RTOL
    assert rel_err(sr.engine.get_state()[tidx].cpu().numpy(), o_st, floor=1e-30) <= 1e-9  # post-reset state: exact draw
    return sr, obs, rew, done


def _properties(obs, rew, done, eng):
    """size-independent properties of a CartPoleSwingUp rollout (checked on the device)."""
    assert bool(torch.isfinite(obs).all())
    assert float(rew.min()) >= 0.0 and float(rew.max()) <= 1.0
    x = obs[..., 0].abs()
    clear = (x - 5.0).abs() > 1e-4
    assert torch.equal(((done & 1) != 0)[clear], (x >= 5.0)[clear])  # cartpole.py:145-147
    assert bool(((done[-1] & 2) != 0).any()) and int((done > 3).sum()) == 0
    assert eng.compact_done().numel() == int((done[-1] != 0).sum())


def test_bench_shape_65536x1000_vs_oracle():
    """BASELINE configs[1] exactly as bench.py runs it."""
    sr, obs, rew, done = _bench_pass_vs_oracle(65536)
    _properties(obs, rew, done, sr.engine)


def test_every_env_of_config2_against_the_c_twin():
    """BASELINE configs[1], EVERY env and EVERY step (65 536 x 1000 = 65.5 M env-steps, not a sample): bench.py's exact call
    against the oracle's fused C rollout with the same reset injection (oracle.cartpole_rollout_autoreset: bit-identical to the
    per-step oracle, tests/test_oracle_golden.py).  A trajectory can leave the oracle's only where a float64 derivative
    lands within ~1e-16 of a float32 rounding boundary (cartpole.py:60 rounds it to float32; the kernel's sin / cos differ from
    libm's in the last bit): expected 0.5 such events in 2.6e8 roundings, after which THAT env's chaotic trajectory drifts —
    so at most two envs may differ, everything else must agree: done masks bit for bit, observations and rewards to 1e-5."""
    from emei_amd import _lib as L
    from emei_amd.sharding import ShardedRollout
    from oracle import oracle as O

    N, T = 65536, 1000
    sr = ShardedRollout("CartPoleSwingUp", N, T, freq_rate=1, real_time_scale=0.02, precision="ref", device=torch.cuda.current_device(), seed=0)
    sr.make_synthetic_inputs()
    s0 = sr.engine.get_state().cpu().numpy()
    obs, rew, done = sr.engine.rollout(sr.actions, auto_reset=True, out=sr.out)
    torch.cuda.synchronize()
    assert sr.engine.last_kernel() == L.KERNEL_PEND_STAGED_FREQ1
    o = O.cartpole_rollout_autoreset("swingup", s0, sr.actions.cpu().numpy(), 0, None, 1000)
    g_done = done.cpu().numpy()
    bad = (g_done != o["done"]).any(axis=0)
    g_obs, g_rew = obs.cpu().numpy(), rew.cpu().numpy()
    with np.errstate(all="ignore"):
        e_obs = (np.abs(g_obs.astype(np.float64) - o["obs"]) / np.maximum(np.abs(o["obs"]), 1e-3)).max(axis=(0, 2))
        e_rew = (np.abs(g_rew.astype(np.float64) - o["reward"]) / np.maximum(np.abs(o["reward"]), 1e-3)).max(axis=0)
    bad |= ~(e_obs <= RTOL) | ~(e_rew <= RTOL)
    assert int(bad.sum()) <= 2, (int(bad.sum()), np.nonzero(bad)[0][:8], float(np.nanmax(e_obs)), float(np.nanmax(e_rew)))
    ok = ~bad
    st = sr.engine.get_state().cpu().numpy()
    assert rel_err(st[ok], o["state"][ok], floor=1e-3) <= RTOL
    steps, epi = sr.engine.get_counters()
    assert np.array_equal(steps.cpu().numpy()[ok], o["steps"][ok]) and np.array_equal(epi.cpu().numpy()[ok], o["episode"][ok].astype(np.int64))
    assert int((o["done"] & 1).astype(bool).sum()) > N  # episodes end and restart inside the horizon: the reset path is exercised everywhere


def test_bench_shape_131072x1000_vs_oracle():
    """configs[4]'s per-GPU shard (1 048 576 / 8), as rank 5 of 8: global env offset 5 * 131 072."""
    sr, obs, rew, done = _bench_pass_vs_oracle(131072, rank=5, world=8)
    assert sr.lo == 5 * 131072
    _properties(obs, rew, done, sr.engine)


def test_config5_whole_1048576_on_one_gpu():
    """All of configs[4] (1 048 576 envs x 1000 steps) on one GPU: sampled envs against the oracle, properties,
    and shard-invariance — rank 7's shard run on its own (131 072 envs, global offset) reproduces columns
    [7 * 131 072, 8 * 131 072) of the whole bit for bit."""
    n, T, h = 1048576, 1000, 131072
    sr, obs, rew, done = _bench_pass_vs_oracle(n, k=256)
    _properties(obs, rew, done, sr.engine)
    lo = 7 * h
    keep_obs, keep_done, keep_rew = obs[:, lo:].clone(), done[:, lo:].clone(), rew[:, lo:].clone()
    acts = sr.actions[:, lo:].contiguous()
    from emei_amd.sharding import synthetic_init_state

    s0 = synthetic_init_state("CartPoleSwingUp", n, lo, n, 0)
    del sr, obs, rew, done
    torch.cuda.empty_cache()
    e = _engine("CartPoleSwingUp", h, max_episode_steps=1000, seed=0, env_index_offset=lo)
    e.set_state(s0)
    o, r, d = e.rollout(acts, auto_reset=True)
    assert torch.equal(o, keep_obs) and torch.equal(d, keep_done) and torch.equal(r, keep_rew)
