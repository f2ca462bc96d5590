"""The HalfCheetah-style oracle: first-party pieces against the golden vectors, and internal consistency
of the restated dynamics (parity with libmujoco is unpinned, see oracle/planar_oracle.c)."""
import numpy as np

from conftest import rel_err
from oracle import oracle as O

STIFF = np.array([0, 0, 0, 240, 180, 120, 180, 120, 60.0])


def test_reward_terminal_vs_golden(mujoco_golden):
    g = mujoco_golden
    o, po, a = g["cheetah_obs"], g["cheetah_pre_obs"], g["cheetah_action"]
    with np.errstate(all="ignore"):
        r = O.cheetah_reward(o, po, a, 0.008)
    assert rel_err(r, g["cheetah_reward_B1"], floor=1e-300) <= 1e-12
    assert np.array_equal(O.cheetah_terminal(o)[:, None], g["cheetah_terminal"])
    # the reference's batch form differs (np.sum over the whole batch, half_cheetah.py:61): documented deviation
    assert not np.allclose(g["cheetah_reward_batchquirk"][:, 0], g["cheetah_reward_B1"], equal_nan=True)


def test_inertia_matrix_and_bias_satisfy_lagrange():
    rng = np.random.default_rng(0)
    for _ in range(4):
        q, v = rng.normal(0, 0.4, 9), rng.normal(0, 2.0, 9)
        M, b, _ = O.cheetah_inertia(q, v)
        assert np.abs(M - M.T).max() < 1e-12 and np.linalg.eigvalsh(M).min() > 0
        assert M[0, 0] == 14.0 and M[1, 1] == 14.0 and abs(M[0, 1]) < 1e-15  # settotalmass = 14
        h = 1e-6
        dM, dU = np.zeros((9, 9, 9)), np.zeros(9)
        for k in range(9):
            qp, qm = q.copy(), q.copy()
            qp[k] += h
            qm[k] -= h
            Mp, _, Ep = O.cheetah_inertia(qp, np.zeros(9))
            Mm, _, Em = O.cheetah_inertia(qm, np.zeros(9))
            dM[:, :, k] = (Mp - Mm) / (2 * h)
            dU[k] = (Ep - Em) / (2 * h) - STIFF[k] * q[k]
        c = np.einsum("ijk,j,k->i", dM, v, v) - 0.5 * np.einsum("jki,j,k->i", dM, v, v) + dU
        assert np.abs(c - b).max() <= 1e-6 * max(1.0, np.abs(b).max())


def test_euler_position_rule_and_rest_pose():
    rng = np.random.default_rng(1)
    st = np.concatenate([rng.normal(0, 0.1, (16, 9)) + [0, 0.5, 0, 0, 0, 0, 0, 0, 0], rng.normal(0, 1, (16, 9))], axis=1)
    nxt, _, _ = O.cheetah_step(st, np.zeros((16, 6)), 1, 0.002)
    assert rel_err(nxt[:, :9], st[:, :9] + 0.002 * st[:, 9:], floor=1e-30) <= 1e-15  # mujoco_env.py:189-191
    s = np.zeros((1, 18))
    for _ in range(750):
        s, r, t = O.cheetah_step(s, np.zeros((1, 6)))
    assert -0.2 < s[0, 1] < -0.05 and np.abs(s[0, 9:]).max() < 0.05 and not t[0]  # settles on its feet
    m = 0
    for _ in range(200):  # full forward drive moves the body and keeps it finite
        s, r, t = O.cheetah_step(s, np.array([[1, -1, 1, -1, 1, -1.0]]))
        m = max(m, abs(s[0, 9]))
    assert np.isfinite(s).all() and m > 0.1
