"""The at-scale parity sweeps (tests/host/staged_parity_sweep.py, tests/host/newton_parity_sweep.py) as tests, at a quarter of the
size of the recorded runs (profiles/r02_*_parity_sweep.txt): hundreds of thousands of random states per kernel instead of
the ~1000 of the per-kernel tests, same tolerances."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, n):
    p = subprocess.run([sys.executable, os.path.join("tests", "host", script), str(n)], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    return [l for l in p.stdout.splitlines() if "difference" in l]


def test_staged_kernels_against_the_oracle_at_scale():
    lines = _run("staged_parity_sweep.py", 262144)
    assert len(lines) == 6, lines
    for l in lines:
        assert "terminal mismatches 0" in l, l
        worst = float(re.search(r"difference ([0-9.e+-]+)", l).group(1))
        if l.startswith("CartPole"):
            assert worst <= 1e-5, l  # north_star's tolerance on the float32 trajectories (measured: one float32 ulp)
            assert float(re.search(r"reward ([0-9.e+-]+)", l).group(1)) <= 1e-6, l
        else:
            assert worst <= 1e-9 and " 0 above 1e-9" in l, l


def test_newton_solve_against_the_oracle_at_scale():
    lines = _run("newton_parity_sweep.py", 250000)
    assert len(lines) == 6, lines  # cheetah, hopper (Newton) x euler / rk4; double pendulum x euler / rk4
    for l in lines:
        assert " 0 above 1e-9" in l and "non-finite 0" in l, l
        if not l.startswith("dpend"):  # the unit-step Newton iteration never ran into its cap (emei_get_solver_cap_hits)
            assert l.rstrip().endswith("solves at the iteration cap 0"), l
