"""Equations of motion of the cart + n-pole chain: the oracle's restatements against accelerations generated
from the reference's own SymPy derivation (emei/envs/classic_control/auxiliary/lagrange_eqs.py:12-69, imported
unmodified by oracle/gen_golden.py:gen_lagrange).  The derivation models uniform rods (half-length l, inertia
m l^2/3 about the centre, next pole hinged at 2 l); the oracle's dynamics are evaluated with exactly those
parameters instead of the xml's capsules, so the STRUCTURE of the InvertedPendulum / InvertedDoublePendulum /
classic CartPole equations is pinned to first-party reference code (the MuJoCo-specific parts — capsule
inertias, soft limits, integrators — stay unpinned)."""
import numpy as np

from conftest import rel_err
from oracle import oracle as O


def test_inverted_pendulum_equations_vs_reference_lagrangian(lagrange_golden):
    g = lagrange_golden
    for i in range(len(g["n1_F"])):
        m, l = g["n1_m"][i], g["n1_l"][i]
        acc = O.ip_accel_custom(g["n1_M"][i], m, l, m * l * l / 3, g["n1_g"][i], g["n1_q"][i], g["n1_v"][i], g["n1_F"][i])
        assert rel_err(acc, g["n1_acc"][i]) <= 1e-11, i


def test_double_pendulum_equations_vs_reference_lagrangian(lagrange_golden):
    """For n = 2 the reference's derivation has a defect: the potential energy of pole i is taken as
    m g l_i cos(angle_i) (lagrange_eqs.py:44) instead of m g * (height of its centre), i.e. the height of the
    hinge of the second pole (2 l_0 cos theta_0) is missing.  Every other term — inertia matrix, Coriolis /
    centripetal terms, the first pole's gravity, the force — must agree; the missing generalized force
    -d/dtheta_0 [m g 2 l_0 cos theta_0] = m g 2 l_0 sin theta_0 is applied to the oracle with the opposite
    sign to reproduce the reference's numbers, and the oracle itself is checked to contain it."""
    g = lagrange_golden
    worst_plain = 0.0
    for i in range(len(g["n2_F"])):
        m, l, grav = g["n2_m"][i], g["n2_l"][i], g["n2_g"][i]
        q, v, F = g["n2_q"][i], g["n2_v"][i], g["n2_F"][i]
        args = (g["n2_M"][i], m, m * l * l / 3, l, 2 * l, grav, q, v)
        missing = m * grav * 2 * l * np.sin(q[1])
        acc = O.dpend_accel_custom(*args, [F, -missing, 0.0])
        assert rel_err(acc, g["n2_acc"][i]) <= 1e-10, i
        worst_plain = max(worst_plain, rel_err(O.dpend_accel_custom(*args, [F, 0.0, 0.0]), g["n2_acc"][i]))
    assert worst_plain > 1e-3  # the physically complete equations differ from the reference's n = 2 derivation


def test_classic_cartpole_dsdt_is_the_same_rod_model():
    """cartpole.py:48-60 (the Barto formula) is the n = 1 Lagrangian with M = 1.0, m = 0.1, l = 0.5, g = 9.8
    (cartpole.py:22-27): one oracle step moves the velocities by float32(acc) * float32(dt)."""
    rng = np.random.default_rng(3)
    st = np.column_stack([rng.uniform(-1, 1, 64), rng.normal(0, 2, 64), rng.uniform(-np.pi, np.pi, 64), rng.normal(0, 4, 64)])
    act = rng.integers(0, 2, 64)
    nxt, _, _ = O.cartpole_step("balancing", st, act, 1, 0.02)
    for i in range(64):
        F = 10.0 if act[i] == 1 else -10.0
        acc = O.ip_accel_custom(1.0, 0.1, 0.5, 0.1 * 0.25 / 3, 9.8, [st[i, 0], st[i, 2]], [st[i, 1], st[i, 3]], F)
        got = (nxt[i, [1, 3]] - st[i, [1, 3]]) / np.float64(np.float32(0.02))
        assert rel_err(got, acc, floor=1.0) <= 2e-6, i  # float32 rounding of the derivative (cartpole.py:60)
