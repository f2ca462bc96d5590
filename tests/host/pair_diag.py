"""Diagnostics (GPU box): Hopper one-step kernel vs oracle on folded-leg states, errors grouped by which capsule pairs touch."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from emei_amd.engine import Engine
from oracle import oracle as O

rng = np.random.default_rng(21)
n = 1536
q = np.concatenate([rng.normal(0, 0.3, (n, 1)), rng.uniform(0.1, 2.5, (n, 1)), rng.normal(0, 1.2, (n, 1)),
                    rng.uniform(-2.9, -0.9, (n, 1)), rng.uniform(-2.9, -1.4, (n, 1)), rng.uniform(-0.9, 0.9, (n, 1))], axis=1)
s0 = np.concatenate([q, rng.normal(0, 2.0, (n, 6))], axis=1)
mask = O.planar_row_mask("hopper", s0)
pairs = (mask >> 11) & 7
act = rng.uniform(-1.2, 1.2, (n, 3)).astype(np.float32)
for integ, solver, fr in (("euler", "sweep1", 1), ("euler", "newton", 1), ("rk4", "newton", 1), ("rk4", "newton", 4)):
    eng = Engine("HopperRunning", n, freq_rate=fr, real_time_scale=0.002, integrator=integ, solver=solver)
    eng.set_state(s0)
    eng.step(torch.as_tensor(act, device=eng.device))
    got = eng.get_state().cpu().numpy()
    want, _, _ = O.hopper_step(s0, act.astype(np.float64), fr, 0.002, O.opts(integ, solver=solver))
    err = (np.abs(got - want) / np.maximum(np.abs(want), 1.0)).max(axis=1)
    print(integ, solver, fr, "max err", err.max())
    for pv in range(8):
        sel = pairs == pv
        if sel.any():
            print(f"   pairs {pv:03b}: n {sel.sum():4d}  max err {err[sel].max():.3e}  bad {(err[sel] > 1e-9).sum()}")
    bad = np.nonzero(err > 1e-9)[0][:3]
    for i in bad:
        print("   bad state", i, "mask", bin(int(mask[i])), "q", np.round(s0[i, :6], 4), "pairs", O.planar_pairs("hopper", s0[i, :6])[:, 2:].round(5).tolist())
        print("      got", np.round(got[i, 6:], 5), "\n      want", np.round(want[i, 6:], 5))
