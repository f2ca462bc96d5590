"""Property tests (hypothesis) of the host-side logic around the hot path."""
import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

import emei_amd
from emei_amd.engine import _sigmas
from emei_amd.envs.base import joint_sigmas
from emei_amd.sharding import shard_bounds


@given(st.integers(1, 10_000_000), st.integers(1, 64))
def test_shard_bounds_partition_the_env_range(n, world):
    lo_hi = [shard_bounds(n, r, world) for r in range(world)]
    assert lo_hi[0][0] == 0 and lo_hi[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(lo_hi[:-1], lo_hi[1:]))           # contiguous, no gaps, rank-major
    sizes = [hi - lo for lo, hi in lo_hi]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)  # balanced, extras on the first ranks


@given(st.floats(0, 10), st.floats(0, 10), st.integers(1, 9))
def test_noise_parameter_forms_reduce_to_per_coordinate_sigmas(p, v, nq):
    assert _sigmas(p, 2 * nq) == [p] * (2 * nq)
    if nq > 1:
        assert _sigmas((p, v), 2 * nq) == [p] * nq + [v] * nq
    sp, sv = joint_sigmas((p, v), nq)
    assert np.array_equal(sp, np.full(nq, p)) and np.array_equal(sv, np.full(nq, v))
    sp, sv = joint_sigmas({nq - 1: (p, v), nq + 5: (1.0, 1.0)}, nq)  # joints outside the model never match (mujoco_env.py:219)
    assert sp[nq - 1] == p and sv[nq - 1] == v and sp[: nq - 1].sum() == 0 and sv[: nq - 1].sum() == 0


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 4), st.integers(1, 3), st.integers(1, 5), st.integers(0, 2**31 - 1))
def test_transition_graph_closure_matches_the_matrix_power_definition(n_obs, n_act, repeat, seed):
    """core.py:142-161: extend the [(n_obs+n_act), n_obs] graph by zero columns for the action rows, sum its
    powers 1..repeat, threshold at > 0, keep the first n_obs columns."""
    rng = np.random.default_rng(seed)
    g = (rng.random((n_obs + n_act, n_obs)) < 0.35).astype(int)
    env = emei_amd.core.EmeiEnv.__new__(emei_amd.core.EmeiEnv)
    env._transition_graph = g
    env.observation_space = emei_amd.spaces.Box(-1, 1, shape=(n_obs,), dtype=np.float64)
    env.action_space = emei_amd.spaces.Box(-1, 1, shape=(n_act,), dtype=np.float32)
    got = env.get_transition_graph(repeat)
    full = np.zeros((n_obs + n_act, n_obs + n_act))
    full[:, :n_obs] = g
    acc, power = np.zeros_like(full), np.eye(n_obs + n_act)
    for _ in range(repeat):
        power = power @ full
        acc += power
    want = (acc[:, :n_obs] > 0).astype(int) if repeat > 1 else g
    assert np.array_equal(got, want)
