#!/usr/bin/env python3
"""Second diagnostic for the rolled cheetah RK4 build: does the error need divergence (constraints active in SOME
lanes)?  (a) one free-flight state replicated over all lanes, (b) random free-flight states (no limit violated, torso
high: no branch of accel() is taken by any lane), (c) one lane in deep contact among free-flight lanes."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from emei_amd.engine import Engine  # noqa: E402
from oracle import oracle as O  # noqa: E402

rng = np.random.default_rng(3)
n, dt = 128, 0.002
names = ["x", "z", "ry", "bth", "bsh", "bft", "fth", "fsh", "fft"]


def run(tag, s0, act):
    eng = Engine("HalfCheetahRunning", len(s0), freq_rate=1, real_time_scale=dt, integrator="rk4")
    eng.set_state(s0)
    eng.step(torch.as_tensor(act.astype(np.float32), device=eng.device))
    st = eng.get_state().cpu().numpy()
    o_st = O.cheetah_step(s0, act.astype(np.float32).astype(np.float64), 1, dt, O.opts("rk4"))[0]
    err = np.abs(st - o_st) / np.maximum(np.abs(o_st), 1.0)
    print(f"== {tag}: max err {err.max():.3e}, bad lanes {(err.max(axis=1) > 1e-9).sum()} / {len(s0)}")
    print("   per coordinate: " + " ".join(f"{nm}:{e:.1e}" for nm, e in zip(names + ["v" + x for x in names], err.max(axis=0))))
    k = int(err.max(axis=1).argmax())
    print(f"   worst lane {k}: got dv = {np.array2string((st[k, 9:] - s0[k, 9:]) / dt, precision=4)}")
    print(f"                  want dv = {np.array2string((o_st[k, 9:] - s0[k, 9:]) / dt, precision=4)}")


free = np.zeros((n, 18))
free[:, 1] = 0.8  # torso 1.5 m above the floor
free[:, 3:9] = rng.uniform(-0.3, 0.3, (n, 6))
free[:, 9:] = rng.normal(0, 1.0, (n, 9))
act = rng.uniform(-1, 1, (n, 6))
same = np.tile(free[:1], (n, 1))
run("(a) identical free-flight lanes", same, np.tile(act[:1], (n, 1)))
run("(b) random free-flight lanes", free, act)
mixed = free.copy()
mixed[5, 1] = -0.3  # one lane pressed into the floor
run("(c) one lane in contact", mixed, act)
lim = free.copy()
lim[7, 3] = 1.3  # one lane beyond the bthigh limit
run("(d) one lane beyond a joint limit", lim, act)
