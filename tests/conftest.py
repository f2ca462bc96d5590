import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _seed_torch(request):
    """Every test starts from the same torch seed (host and device generators): tests that draw actions with torch.rand* are
    reproducible run to run (a rare draw once made a one-ulp property assertion fail on one box and pass on the next)."""
    try:
        import torch
    except ImportError:
        return
    torch.manual_seed(0x5EED)  # seeds the CUDA / HIP generators too (lazily, without initialising a device)


@pytest.fixture(scope="session")
def cartpole_golden():
    return np.load(os.path.join(GOLDEN, "cartpole_golden.npz"))


@pytest.fixture(scope="session")
def cartpole_rk4_golden():
    """ODE_approximation(..., method="rk4") of the reference (base_control.py:165-170), called directly (oracle/gen_golden.py)."""
    return np.load(os.path.join(GOLDEN, "cartpole_rk4_golden.npz"))


@pytest.fixture(scope="session")
def mujoco_golden():
    return np.load(os.path.join(GOLDEN, "mujoco_firstparty_golden.npz"))


@pytest.fixture(scope="session")
def hopper_golden():
    return np.load(os.path.join(GOLDEN, "hopper_firstparty_golden.npz"))


@pytest.fixture(scope="session")
def lagrange_golden():
    return np.load(os.path.join(GOLDEN, "lagrange_golden.npz"))


def rel_err(a, b, floor=1e-3):
    """max |a-b| / max(|b|, floor) ignoring rows where both are NaN.  The default floor keeps "1e-5 relative" relative down to
    |ref| = 1e-3 (rewards in [0, 1], x ~ 0.01); a test that compares quantities passing through zero with an error set by O(1)
    dynamics says so with an explicit floor."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    same_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    with np.errstate(all="ignore"):
        d = np.abs(a - b) / np.maximum(np.abs(b), floor)
    d = np.where(both_nan | same_inf, 0.0, d)
    d = np.where(np.isnan(d), np.inf, d)
    return float(d.max()) if d.size else 0.0


def oracle_autoreset_rollout(variant, s0, acts, seed, env_ids, max_steps, fr=1, dt=0.02):
    """Oracle restatement of a device rollout WITH auto-reset for the envs `env_ids` (GLOBAL indices = the
    counter word of the device reset generator): step with the float64 reference arithmetic
    (oracle/emei_oracle.c, pinned bit-exact to the reference by tests/golden); on done re-initialise from the
    Philox spec, i.e. identical reset injection.  s0 [k,4], acts [T,k] -> obs [T,k,4], rew [T,k], done [T,k], state."""
    from oracle import oracle as O

    T, N = acts.shape
    env_ids = np.asarray(env_ids, np.int64)
    st = np.array(s0, np.float64)
    steps = np.zeros(N, np.int64)
    episode = np.zeros(N, np.int64)
    obs = np.empty((T, N, 4))
    rew = np.empty((T, N))
    done = np.empty((T, N), np.uint8)
    for t in range(T):
        st, r, term = O.cartpole_step(variant, st, acts[t], fr, dt)
        steps += 1
        trunc = (steps >= max_steps) if max_steps > 0 else np.zeros(N, bool)
        d = term.astype(np.uint8) | (trunc.astype(np.uint8) << 1)
        obs[t], rew[t], done[t] = st, r, d
        for i in np.nonzero(d)[0]:
            episode[i] += 1
            steps[i] = 0
            st[i] = O.cartpole_init_f32(variant, seed, int(env_ids[i]), int(episode[i])).astype(np.float64)
    return obs, rew, done, st
