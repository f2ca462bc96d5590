#!/usr/bin/env python3
"""What binning lanes by their constraint-row signature buys the Newton body kernels, measured with the EXISTING kernel:
after 300 steps the envs are re-ordered on the host by the row mask of their state (oracle's row builder: which limits are
violated, which capsule ends touch the floor), so that the waves of the next launch are as homogeneous as the signature
makes them; the launch of K steps is timed against the same envs in their original order.  K = 1 .. 32 shows how fast the
homogeneity decays (contacts change), i.e. how often a real implementation would have to re-bin.
(tools/rebin_bound.py measured the limit: identical lanes per wave.)  Lives under tests/host/ because it uses the oracle row builder."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from emei_amd.engine import Engine  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = [("HalfCheetahRunning", "euler", "cheetah", 6, 0.1), ("HopperRunning", "rk4", "hopper", 3, 5e-3), ("HopperRunning", "euler", "hopper", 3, 5e-3)]
N = 131072
for env, integ, body, na, sigma in CASES:
    eng = Engine(env, N, freq_rate=4, real_time_scale=0.002, integrator=integ, init_noise=sigma, seed=0)
    eng.reset(0)
    gen = torch.Generator(device=eng.device)
    gen.manual_seed(1)
    acts = (torch.rand((100, N, na), device=eng.device, generator=gen) * 2 - 1).float()
    out = eng.alloc_outputs(100)
    for _ in range(3):
        eng.rollout(acts, out=out)
    st = eng.get_state()
    mask = O.planar_row_mask(body, st.cpu().numpy())
    u, c = np.unique(mask, return_counts=True)
    top = np.argsort(-c)[:6]
    print(f"{env} {integ}: {len(u)} distinct row masks at step 300; the most frequent: " + ", ".join(f"{u[i]:#x} {c[i] / N:.3f}" for i in top), flush=True)
    order = torch.as_tensor(np.argsort(mask, kind="stable"), device=eng.device)
    for K in (1, 2, 4, 8, 16, 32):
        res = {}
        for mode in ("original order", "sorted by row mask"):
            s, a = (st, acts[:K]) if mode == "original order" else (st[order].contiguous(), acts[:K, order].contiguous())
            o = eng.alloc_outputs(K)
            ts = []
            for rep in range(5):
                eng.set_state(s, reset_counters=True)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                eng.rollout(a.contiguous(), out=o)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            res[mode] = sorted(ts)[2]
        print(f"   K = {K:2d} steps: original order {res['original order'] * 1e3 / K:7.1f} us/step, sorted by row mask {res['sorted by row mask'] * 1e3 / K:7.1f} us/step "
              f"({100 * (1 - res['sorted by row mask'] / res['original order']):.1f} % less)", flush=True)
    eng.close()
