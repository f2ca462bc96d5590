#!/usr/bin/env python3
"""One-off parity sweep of the STAGED rollout kernels at scale (run on the GPU box): N random states, a short fused rollout
through emei_rollout (so that pend_rollout_staged_kernel runs — asserted), against the C oracle stepped the same way.
CartPole (the pinned oracle): 16 steps, uint8 actions, both variants, freq_rate 1 and 4: float32 observations to 1e-5,
terminal masks bit for bit.  InvertedPendulum (config 3's kernel): 8 steps x 4 substeps, states on and beyond the rail,
float64 final state to 1e-9.  Usage: python tests/host/staged_parity_sweep.py [n=1048576]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from emei_amd import _lib as L  # noqa: E402
from emei_amd.engine import Engine  # noqa: E402
from oracle import oracle as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
rng = np.random.default_rng(7)
for variant, name in (("swingup", "CartPoleSwingUp"), ("balancing", "CartPoleBalancing")):
    for fr in (1, 4):
        T = 16
        s0 = np.column_stack([rng.uniform(-5.2, 5.2, n), rng.normal(0, 2, n), rng.uniform(-4, 4, n), rng.normal(0, 3, n)])
        acts = rng.integers(2, size=(T, n), dtype=np.uint8)
        eng = Engine(name, n, freq_rate=fr, real_time_scale=0.02, precision="ref")
        eng.set_state(s0)
        obs, rew, done = eng.rollout(torch.as_tensor(acts, device=eng.device))
        assert eng.last_kernel() in (L.KERNEL_PEND_STAGED_FREQ1, L.KERNEL_PEND_STAGED)
        states, orew, oterm = O.cartpole_rollout(variant, s0, acts, fr, 0.02)
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        err = np.abs(obs - states[1:]) / np.maximum(np.abs(states[1:]), 1.0)
        rerr = np.abs(rew - orew).max()
        mism = int(((done & 1).astype(bool) != oterm).sum())
        print(f"{name} freq_rate={fr}: {n} envs x {T} steps, worst scaled obs difference {np.nanmax(err):.2e} (float32 outputs), reward {rerr:.2e}, "
              f"terminal mismatches {mism}", flush=True)
        eng.close()
for vname, name in (("boundary_swingup", "BoundaryInvertedPendulumSwingUp"), ("rebound_balancing", "ReboundInvertedPendulumBalancing")):
    T, fr = 8, 4
    s0 = np.column_stack([rng.uniform(-2.3, 2.3, n), rng.uniform(-3.5, 3.5, n), rng.normal(0, 2, n), rng.normal(0, 4, n)])
    acts = rng.uniform(-3.5, 3.5, (T, n)).astype(np.float32)
    eng = Engine(name, n, freq_rate=fr, real_time_scale=0.02, precision="ref")
    eng.set_state(s0)
    obs, rew, done = eng.rollout(torch.as_tensor(acts, device=eng.device))
    assert eng.last_kernel() in (L.KERNEL_PEND_STAGED_FREQ1, L.KERNEL_PEND_STAGED)
    st = s0
    term = np.zeros((T, n), bool)
    for t in range(T):
        st, o_obs, o_rew, term[t] = O.ip_step(vname, st, acts[t].astype(np.float64), fr, 0.02)
    got = eng.get_state().cpu().numpy()
    err = np.abs(got - st).max(axis=1) / np.maximum(1.0, np.abs(st).max(axis=1))
    mism = int(((done.cpu().numpy() & 1).astype(bool) != term).sum())
    print(f"{name}: {n} envs x {T} steps x {fr} substeps, worst scaled final-state difference {np.nanmax(err):.2e} (float64), "
          f"{int((err > 1e-9).sum())} above 1e-9, terminal mismatches {mism}", flush=True)
    eng.close()
