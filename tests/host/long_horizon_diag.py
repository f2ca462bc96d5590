#!/usr/bin/env python3
"""Long-horizon diagnosis of the contact bodies: 4096 envs x 1000 steps of Hopper RK4 and HalfCheetah
Euler from the device reset, the oracle re-synchronised every 20 steps.  Prints per segment: the share of finite lanes, the
worst scaled difference to the oracle inside the segment, and the share of lanes with constraint rows at the segment's end
(kernel state / oracle state).  Run on the GPU box; EMEI_HIP_LIB selects the library (shipped or a variant build).

Test infrastructure: uses oracle/ as the checker."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from emei_amd.engine import Engine  # noqa: E402
from oracle import oracle as O  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
SEG = 20
CASES = [("HopperRunning", "rk4", "hopper", 3, 5e-3), ("HalfCheetahRunning", "euler", "cheetah", 6, 0.1)]


def scaled(a, b):
    with np.errstate(all="ignore"):
        d = np.abs(a - b) / np.maximum(np.abs(b), 1.0)
    return np.where(np.isnan(d), np.inf, d)


for env, integ, body, na, sigma in CASES:
    step = O.hopper_step if body == "hopper" else O.cheetah_step
    for solver in ("newton", "sweep1"):
        rng = np.random.default_rng(11)
        acts = rng.uniform(-1, 1, (T, N, na)).astype(np.float32)
        eng = Engine(env, N, freq_rate=4, real_time_scale=0.002, integrator=integ, solver=solver, init_noise=sigma, seed=7)
        eng.reset(7)
        dev = torch.as_tensor(acts, device=eng.device)
        print(f"== {env} {integ} {solver}: N={N} T={T}", flush=True)
        t_start = time.time()
        worst_all = 0.0
        for t0 in range(0, T, SEG):
            st = eng.get_state().cpu().numpy()
            obs, rew, done = eng.rollout(dev[t0 : t0 + SEG].contiguous())
            obs = obs.cpu().numpy()
            worst, first_bad = 0.0, None
            ost = st.copy()
            for t in range(SEG):
                ost, orew, _ = step(ost, acts[t0 + t].astype(np.float64), 4, 0.002, O.opts(integ, solver=solver))
                e = scaled(obs[t], ost).max(axis=1)
                if e.max() > worst:
                    worst = float(e.max())
                if first_bad is None and e.max() > 1e-5:
                    first_bad = (t, int(e.argmax()), float(e.max()))
            end = eng.get_state().cpu().numpy()
            fin = np.isfinite(end).all(axis=1)
            rows_k = O.planar_count_rows(body, end)
            rows_o = O.planar_count_rows(body, ost)
            worst_all = max(worst_all, worst)
            print(f"  steps {t0:4d}-{t0 + SEG:4d}: finite {fin.mean():.4f} (oracle {np.isfinite(ost).all(axis=1).mean():.4f})  worst {worst:.2e}"
                  f"  rows>0: kernel {(rows_k > 0).mean():.3f} oracle {(rows_o > 0).mean():.3f}  z median {np.median(end[:, 1]):.3f}"
                  + (f"  first>1e-5 at step {t0 + first_bad[0]} env {first_bad[1]} ({first_bad[2]:.2e})" if first_bad else ""), flush=True)
            if first_bad and worst > 1e-3:
                i = first_bad[1]
                print(f"     env {i}: segment start state {np.array2string(st[i], precision=6)}", flush=True)
        print(f"   worst over the horizon {worst_all:.2e}; {time.time() - t_start:.1f} s", flush=True)
        eng.close()
