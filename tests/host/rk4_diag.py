#!/usr/bin/env python3
"""Diagnostic for the cheetah RK4 instantiation: one RK4 step of the test states of
tests/test_gpu_integrators.py against the oracle, with the error broken down per state coordinate, per lane
and by which constraints are active.  Run with EMEI_HIP_LIB=<variant .so> to compare builds."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_integrators import _case  # noqa: E402

from emei_amd.engine import Engine  # noqa: E402
from oracle import oracle as O  # noqa: E402

rng = np.random.default_rng(11)
n = 777
name, s0, act, ostep, dt = _case("cheetah", rng, n)
act32 = act.astype(np.float32)
names = ["x", "z", "ry", "bth", "bsh", "bft", "fth", "fsh", "fft"]
for integ in ("euler", "rk4"):
    for fr in (1,):
        eng = Engine(name, n, freq_rate=fr, real_time_scale=dt, integrator=integ)
        eng.set_state(s0)
        eng.step(torch.as_tensor(act32, device=eng.device))
        st = eng.get_state().cpu().numpy()
        o_st = ostep(s0, act32.astype(np.float64), fr, dt, O.opts(integ))[0]
        err = np.abs(st - o_st) / np.maximum(np.abs(o_st), 1.0)
        bad = err.max(axis=1) > 1e-9
        print(f"== {os.environ.get('EMEI_HIP_LIB', 'default')} {integ} fr={fr}: max err {err.max():.3e}; bad envs {bad.sum()} / {n}")
        print("   per coordinate max err: " + " ".join(f"{nm}:{e:.1e}" for nm, e in zip(names + ["v" + x for x in names], err.max(axis=0))))
        if bad.any():
            idx = np.nonzero(bad)[0]
            print("   bad env indices (first 40):", idx[:40].tolist())
            print("   bad lanes histogram (idx % 64):", np.bincount(idx % 64, minlength=64).tolist())
            print("   bad by wave (idx // 64):", np.bincount(idx // 64, minlength=(n + 63) // 64).tolist())
            # which constraints are active at s0: joint limits / low torso (contacts likely)
            lo = np.array([-0.52, -0.785, -0.4, -1.0, -1.2, -0.5]); hi = np.array([1.05, 0.785, 0.785, 0.7, 0.87, 0.5])
            viol = (s0[:, 3:9] < lo) | (s0[:, 3:9] > hi)
            print("   limit violated at s0 [bth bsh bft fth fsh fft], among bad:", viol[bad].sum(axis=0).tolist(), " among good:", viol[~bad].sum(axis=0).tolist())
            print("   z (torso height offset) among bad: mean %.3f min %.3f; among good: mean %.3f" % (s0[bad, 1].mean(), s0[bad, 1].min(), s0[~bad, 1].mean()))

# the stateless one-step kernel (body_next_obs_kernel<CheetahBody, true>): the same RK4 substep without the rollout scaffolding
from emei_amd import engine as E  # noqa: E402

o32 = torch.as_tensor(s0, dtype=torch.float32, device="cuda")
nxt = E.batch_next_obs(name, o32, torch.as_tensor(act32, device="cuda"), dt, 1, "ref", "rk4").cpu().numpy()
o_st = ostep(o32.double().cpu().numpy(), act32.astype(np.float64), 1, dt, O.opts("rk4"))[0]
err = np.abs(nxt - o_st) / np.maximum(np.abs(o_st), 1.0)
print(f"== next_obs rk4: max err {err.max():.3e}; bad (>1e-5) {(err.max(axis=1) > 1e-5).sum()} / {n}")
