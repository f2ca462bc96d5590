"""The InvertedPendulum's joint-limit rows in the oracle: the slider's range (every variant) and the hinge's range of +-90
degrees (inverted_pendulum.xml:17; Balancing variants only — SwingUp's _update_model frees the hinge, inverted_pendulum.py:
135-137).  With both rows present the oracle enumerates the active sets of the 2 x 2 complementarity problem; here its answer
is checked through the optimality conditions of MuJoCo's primal problem, restated independently in NumPy:
    a = argmin 1/2 (a - a0)' M (a - a0) + sum_i D_i / 2 min(0, J_i a - aref_i)^2."""
import numpy as np
import pytest

from oracle import oracle as O


def _rows(m, variant, q, v, dt):
    """[(J [2], aref, D)] of the rows that exist at (q, v): independent restatement of the limit rows"""
    tc = max(m.timeconst, 2 * dt)
    K, B = 1.0 / (m.dmax**2 * tc**2 * m.dampratio**2), 2.0 / (m.dmax * tc)
    rows = []
    for i, (lo, hi, invw) in enumerate(((m.x_lo, m.x_hi, m.invweight_slider), (m.th_lo, m.th_hi, m.invweight_hinge))):
        if i == 1 and "swingup" in variant:
            continue
        if q[i] < lo:
            dist, J = q[i] - lo, 1.0
        elif q[i] > hi:
            dist, J = hi - q[i], -1.0
        else:
            continue
        xx = abs(dist) / m.width
        y = 1.0 if xx >= 1 else (2 * xx * xx if xx <= 0.5 else 1 - 2 * (1 - xx) ** 2)
        imp = m.dmin + y * (m.dmax - m.dmin)
        Jv = np.zeros(2)
        Jv[i] = J
        rows.append((Jv, -B * J * v[i] - K * imp * dist, imp / ((1 - imp) * invw)))
    return rows


def _inertia(m, variant, theta):
    phi = theta + m.phi0 + (np.pi if "swingup" in variant else 0.0)
    M12 = m.mp * m.r * np.cos(phi)
    return np.array([[m.mc + m.mp, M12], [M12, m.Icom + m.mp * m.r**2]])


@pytest.mark.parametrize("variant", ["rebound_balancing", "boundary_balancing", "rebound_swingup", "boundary_swingup"])
def test_limit_rows_satisfy_the_optimality_conditions(variant):
    m = O.ip_model()
    rng = np.random.default_rng(3)
    counts = {0: 0, 1: 0, 2: 0}
    for k in range(4000):
        q = np.array([rng.choice([-1, 1]) * rng.uniform(1.9, 2.05), rng.choice([-1, 1]) * rng.uniform(1.4, 1.75)])
        v = rng.normal(0, 3, 2)
        u = rng.uniform(-3, 3)
        dt = rng.choice([0.02, 0.002])
        a = O.ip_accel(variant, q, v, u, dt)
        # the unconstrained acceleration: the same state far inside both ranges cannot be used (M depends on theta), so remove
        # the rows instead: a0 = a - M^-1 sum J' f with f from the rows' definition at the returned a
        rows = _rows(m, variant, q, v, dt)
        M = _inertia(m, variant, q[1])
        force = sum(D * max(0.0, aref - J @ a) * J for J, aref, D in rows) if rows else np.zeros(2)
        a0 = a - np.linalg.solve(M, force)
        # gradient of the primal cost at a with that a0 is zero by construction; the independent check is a0 itself:
        # it must be the smooth acceleration, which the SwingUp variant without a hinge row / an interior state reproduces
        s = np.sin(q[1] + m.phi0 + (np.pi if "swingup" in variant else 0.0))
        f1 = m.gear * np.clip(u, m.ctrl_lo, m.ctrl_hi) + m.mp * m.r * s * v[1] ** 2
        f2 = m.mp * m.g * m.r * s
        smooth = np.linalg.solve(M, np.array([f1, f2]))
        assert np.abs(a0 - smooth).max() <= 1e-9 * max(1.0, np.abs(smooth).max()), (k, q, rows)
        counts[sum(1 for J, aref, D in rows if aref - J @ a > 0)] += 1
    if "balancing" in variant:
        assert counts[2] > 100 and counts[1] > 100  # both rows pushing at once, and one of two
    else:
        assert counts[2] == 0  # SwingUp: the hinge is free


def test_the_hinge_stop_holds_the_fallen_pole_of_the_balancing_variants():
    """post-terminal behaviour (the reference keeps stepping, base_control.py:80 / mujoco_env.py:157-167 have no reset): the pole
    of a Balancing variant comes to rest against its +-90 degree stop; SwingUp's swings through"""
    s0 = np.array([[0.0, 1.2, 0.0, 0.0]])
    for variant, stops in (("rebound_balancing", True), ("boundary_balancing", True)):
        s = s0.copy()
        for _ in range(300):
            s, _, _, _ = O.ip_step(variant, s, np.zeros(1), 1, 0.02)
        assert (abs(s[0, 1]) < np.pi / 2 + 0.05) == stops and np.isfinite(s).all(), (variant, s)
    # the same fall with the hinge free (the SwingUp model hangs: start it near its own upright, theta = pi - 1.2 from hanging)
    s = np.array([[0.0, np.pi - 1.2, 0.0, 0.0]])
    peak = 0.0
    for _ in range(300):
        s, _, _, _ = O.ip_step("rebound_swingup", s, np.zeros(1), 1, 0.02)
        peak = max(peak, abs(s[0, 1]))
    assert peak > np.pi  # swings through the bottom and beyond any +-90 degree stop
