#!/usr/bin/env python3
"""One-off parity sweep of the Newton constraint solve at scale (run on the GPU box; the oracle runs on the host cores):
N random rough states per body — flight, deep contact, joints past their limits, velocities up to 5x the tests' — one
env-step on the GPU against the oracle (exact line search), float64.  Prints the worst scaled difference and how many
states exceed the tests' 1e-9.  Usage: python tests/host/newton_parity_sweep.py [n_states=200000]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from emei_amd.engine import Engine  # noqa: E402
from oracle import oracle as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
rng = np.random.default_rng(2024)
for body, name, nv, nu, step in (("cheetah", "HalfCheetahRunning", 9, 6, O.cheetah_step), ("hopper", "HopperRunning", 6, 3, O.hopper_step)):
    q = rng.normal(0, 0.25, (n, nv))
    q[:, 1] = rng.uniform(-0.45, 0.3, n) if body == "cheetah" else 1.25 + rng.uniform(-0.4, 0.1, n)
    q[: n // 4, 3:] = rng.uniform(-1.5, 1.5, (n // 4, nv - 3))
    if body == "hopper":  # round 4: a quarter with the leg folded towards its -150 degree limits, in the air and on the ground: the
        k = n // 4        # capsule-capsule rows (torso-leg, torso-foot, thigh-foot; hopper.xml:5)
        q[k:2 * k, 3] = rng.uniform(-2.9, -0.9, k)
        q[k:2 * k, 4] = rng.uniform(-2.9, -1.4, k)
        q[k:2 * k, 5] = rng.uniform(-0.9, 0.9, k)
        q[k:2 * k, 1] = rng.uniform(0.1, 2.5, k)
        q[k:2 * k, 2] = rng.normal(0, 1.2, k)
    v = rng.normal(0, 2.0, (n, nv)) * np.where(np.arange(n) % 3 == 0, 5.0, 1.0)[:, None]
    s0 = np.concatenate([q, v], axis=1)
    act = rng.uniform(-1.3, 1.3, (n, nu)).astype(np.float32)
    if body == "hopper":
        pm = (O.planar_row_mask("hopper", s0[n // 4: n // 2][:400000]) >> 11) & 7
        print(f"hopper: share of the folded quarter's states with a capsule-pair row {float((pm != 0).mean()):.3f} (torso-leg {float((pm & 1 != 0).mean()):.3f}, "
              f"torso-foot {float((pm & 2 != 0).mean()):.3f}, thigh-foot {float((pm & 4 != 0).mean()):.3f})", flush=True)
    for integ, fr in (("euler", 1), ("rk4", 2)):
        eng = Engine(name, n, freq_rate=fr, real_time_scale=0.002, precision="ref", integrator=integ, solver="newton")
        eng.set_state(s0)
        eng.step(torch.as_tensor(act, device=eng.device))
        got = eng.get_state().cpu().numpy()
        want = step(s0, act.astype(np.float64), fr, 0.002, O.opts(integ, solver="newton"))[0]
        err = np.abs(got - want).max(axis=1) / np.maximum(1.0, np.abs(want).max(axis=1))
        bad = int((err > 1e-9).sum())
        print(f"{body} {integ} freq_rate={fr}: {n} states, worst scaled difference {err.max():.2e}, {bad} above 1e-9, "
              f"non-finite {int((~np.isfinite(got)).any(axis=1).sum())}, solves at the iteration cap {eng.solver_cap_hits()}", flush=True)
        eng.close()

# the single-constraint body (InvertedDoublePendulum, body_rollout_kernel): states on and beyond the rail and its margin
for integ, fr in (("euler", 2), ("rk4", 1)):
    s0 = np.column_stack([rng.uniform(-3.2, 3.2, n), rng.uniform(-3.5, 3.5, (n, 2)), rng.normal(0, 2, (n, 3))])
    act = rng.uniform(-1.2, 1.2, n).astype(np.float32)
    eng = Engine("BoundaryInvertedDoublePendulumSwingUp", n, freq_rate=fr, real_time_scale=0.02, precision="ref", integrator=integ)
    eng.set_state(s0)
    eng.step(torch.as_tensor(act[:, None], device=eng.device))
    got = eng.get_state().cpu().numpy()
    want = O.dpend_step("boundary_swingup", s0, act.astype(np.float64), fr, 0.02, O.opts(integ))[0]
    err = np.abs(got - want).max(axis=1) / np.maximum(1.0, np.abs(want).max(axis=1))
    print(f"dpend {integ} freq_rate={fr}: {n} states, worst scaled difference {err.max():.2e}, {int((err > 1e-9).sum())} above 1e-9, "
          f"non-finite {int((~np.isfinite(got)).any(axis=1).sum())}", flush=True)
    eng.close()

