"""InvertedDoublePendulum oracle: first-party pieces vs the golden vectors from the reference, and
consistency of the restated dynamics (parity with libmujoco is unpinned, see oracle/dpend_oracle.c)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_err
from oracle import oracle as O


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "dpend_firstparty_golden.npz"))


@pytest.mark.parametrize("variant", sorted(O.DP_VARIANTS))
def test_reward_terminal_vs_golden(g, variant):
    with np.errstate(all="ignore"):
        r, t = O.dpend_reward_terminal(variant, g["dp_obs"])
    assert rel_err(r, g[f"dp_{variant}_reward"][:, 0], floor=1e-300) <= 1e-12
    assert np.array_equal(t, g[f"dp_{variant}_terminal"][:, 0])


def test_quirky_wrap_vs_golden(g):
    """inverted_double_pendulum.py:59: (theta + pi) % 2 * pi - pi — precedence bug reproduced bit for bit."""
    for col in (1, 2):
        assert np.array_equal(O.dpend_wrap(g["dp_wrap_in"][:, col]), g["dp_wrap_out"][:, col])
    assert O.dpend_wrap([0.1])[0] == pytest.approx(0.7589857, abs=1e-6)  # not 0.1: the wrap scrambles the angle


def test_dynamics_conserve_energy_in_the_small_step_limit():
    s0 = np.array([[0.1, 0.5, -0.3, 0.2, -1.0, 2.0]])
    drift = []
    for dt in (1e-3, 1e-4):
        st = s0.copy()
        e0 = O.dpend_energy("rebound_swingup", st[0])
        for _ in range(int(0.2 / dt)):
            st, _, _, _ = O.dpend_step("rebound_swingup", st, [0.0], 1, dt)
        drift.append(O.dpend_energy("rebound_swingup", st[0]) - e0)
    assert drift[0] / drift[1] == pytest.approx(10.0, rel=0.05) and abs(drift[1]) < 5e-3  # explicit Euler: O(dt)


def test_euler_position_rule_and_rail(g):
    rng = np.random.default_rng(2)
    st = np.column_stack([rng.uniform(-2, 2, 32), rng.normal(0, 1, (32, 2)), rng.normal(0, 2, (32, 3))])
    nxt, obs, _, _ = O.dpend_step("boundary_balancing", st, rng.uniform(-1, 1, 32), 1, 0.02)
    assert rel_err(nxt[:, :3], st[:, :3] + 0.02 * st[:, 3:], floor=1e-30) <= 1e-15  # mujoco_env.py:189-191
    assert np.array_equal(obs[:, 1], O.dpend_wrap(nxt[:, 1]))
    s = np.zeros((1, 6))
    for _ in range(300):  # full push against the +3 rail: the soft limit holds the cart
        s, _, _, _ = O.dpend_step("rebound_swingup", s, [1.0])
    assert 2.9 < s[0, 0] < 3.3 and np.isfinite(s).all()
