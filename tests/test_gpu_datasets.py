"""Offline-dataset writer: the rows of a GPU rollout are true transitions in the
schema of zoo/util.py:62-67, including the observation that follows a device reset."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import emei_amd  # noqa: E402
from emei_amd import datasets  # noqa: E402


def test_cartpole_dataset_rows_are_transitions(tmp_path):
    from oracle import oracle as O

    N, T = 256, 400
    env = emei_amd.make("CartPoleBalancing-v0", num_envs=N)  # short episodes: many resets and a TimeLimit of 500
    data, info = datasets.collect(env, T, seed=3)
    d = {k: v.cpu().numpy() for k, v in data.items()}
    assert set(d) == set(datasets.DATASET_KEYS) and all(len(v) == N * T for v in d.values())
    assert info["total_episode_num"] == int(d["dones"].sum()) > N and info["total_sample_num"] == N * T
    obs, nxt, act, done = (d[k].reshape(N, T, -1) for k in ("observations", "next_observations", "actions", "dones"))
    done = done[..., 0] != 0
    # (1) inside an episode the chain is continuous
    cont = ~done[:, :-1]
    assert np.array_equal(obs[:, 1:][cont], nxt[:, :-1][cont])
    # (2) after a reset the observation is the device reset state of (env, episode) — the Philox spec
    e, t = np.nonzero(done[:, :-1])
    epi = np.cumsum(done, axis=1)
    for k in range(0, len(e), max(1, len(e) // 64)):
        want = O.cartpole_init_f32("balancing", 3, int(e[k]), int(epi[e[k], t[k]]))
        assert np.array_equal(obs[e[k], t[k] + 1], want)
    assert np.array_equal(obs[:, 0], np.stack([O.cartpole_init_f32("balancing", 3, i, 0) for i in range(N)]))
    # (3) every row is one reference step: (obs, a) -> next_obs, reward, done (oracle on the float32 observation)
    rows = np.random.default_rng(0).choice(N * T, 2048, replace=False)
    o_nxt, o_rew, o_term = O.cartpole_step("balancing", d["observations"][rows].astype(np.float64), d["actions"][rows, 0].astype(np.int32))
    # the oracle restarts from the float32 observation the dataset stores, the kernel stepped from its float64 state: the input
    # itself differs by float32 rounding of O(1) coordinates (6e-8 absolute), which every output coordinate inherits
    assert rel_err(d["next_observations"][rows], o_nxt, floor=1.0) <= 1e-5
    assert rel_err(d["rewards"][rows], o_rew) <= 1e-5
    clear = (np.abs(np.abs(o_nxt[:, 0]) - 2.4) > 1e-5) & (np.abs(np.abs(o_nxt[:, 2]) - 12 * 2 * np.pi / 360) > 1e-6)
    tm = d["timeouts"][rows] != 0
    assert np.array_equal((d["dones"][rows] != 0)[clear & ~tm], o_term[clear & ~tm])
    # (4) round trip through the on-disk form with the reference's key check (core.py:118-126)
    p = tmp_path / "CartPoleBalancing-random.npz"
    datasets.save_npz(data, p, info)
    back = datasets.load_npz(p)
    assert np.array_equal(back["observations"], d["observations"]) and (tmp_path / "CartPoleBalancing-random.npz.info.json").exists()
    # ... and through the reference's own container (zoo/util.py:108-111 save_as_h5 / core.py:61-81 load_h5_data), bit for bit
    h5 = tmp_path / "CartPoleBalancing-random.h5"
    datasets.save_h5(data, h5, info)
    back = datasets.load_h5(h5)
    assert set(back) == set(datasets.DATASET_KEYS) and all(np.array_equal(back[k], d[k]) and back[k].dtype == d[k].dtype for k in d)


def test_timeouts_and_continuous_actions():
    env = emei_amd.make("ReboundInvertedPendulumSwingUp-v0", num_envs=64, max_episode_steps=50)  # never terminates: only TimeLimit
    data, info = datasets.collect(env, 120, seed=1)
    done = data["dones"].reshape(64, 120).cpu().numpy()
    tout = data["timeouts"].reshape(64, 120).cpu().numpy()
    assert np.array_equal(done, tout) and np.array_equal(np.nonzero(done[0])[0], [49, 99])
    assert data["actions"].shape == (64 * 120, 1) and float(data["actions"].abs().max()) <= 3.0
    assert info["avg_length"] == pytest.approx(120 / 2)


def test_cheetah_dataset_shapes():
    env = emei_amd.make("HalfCheetahRunning-v0", num_envs=64)
    data, info = datasets.collect(env, 20, seed=0)
    assert data["observations"].shape == (1280, 18) and data["actions"].shape == (1280, 6) and info["total_episode_num"] == 0
    o, n = data["observations"].reshape(64, 20, 18), data["next_observations"].reshape(64, 20, 18)
    assert torch.equal(o[:, 1:], n[:, :-1]) and bool(torch.isfinite(n).all())


def test_collect_with_a_policy_between_steps():
    """zoo/util.py:54-59: actions come from agent.predict(obs) each step; a bang-bang 'policy' here."""
    import emei_amd
    from emei_amd import datasets

    env = emei_amd.make("CartPoleBalancing-v0", num_envs=128, max_episode_steps=30, auto_reset=True)
    seen = []

    def policy(obs):
        seen.append(obs.clone())
        return (obs[:, 2] > 0).to(torch.int64)  # push towards the side the pole leans to

    d, info = datasets.collect(env, 40, policy=policy, seed=3)
    assert d["observations"].shape == (128 * 40, 4) and d["actions"].shape == (128 * 40, 1) and len(seen) == 40
    o = d["observations"].reshape(128, 40, 4)
    nxt = d["next_observations"].reshape(128, 40, 4)
    dn = d["dones"].reshape(128, 40) != 0
    # rows are true transitions: observations[t+1] == next_observations[t] unless step t ended an episode,
    # where the policy saw (and the dataset holds) the initial observation of the next episode
    cont = ~dn[:, :-1]
    assert torch.equal(o[:, 1:][cont], nxt[:, :-1][cont])
    assert bool((o[:, 1:][~cont].abs() <= 0.05 + 1e-6).all()) and bool(dn.any())  # cartpole.py:131-132: U(-0.05, 0.05)
    for t in (0, 7, 39):
        assert torch.equal(seen[t], o[:, t])
    # the bang-bang controller keeps the pole up longer than random actions do
    rnd, info_r = datasets.collect(emei_amd.make("CartPoleBalancing-v0", num_envs=128, max_episode_steps=30, auto_reset=True), 40, seed=3)
    assert info["avg_length"] > info_r["avg_length"]
