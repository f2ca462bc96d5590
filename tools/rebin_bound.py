#!/usr/bin/env python3
"""Upper bound of what re-binning lanes into homogeneous waves could buy the Newton body kernels.

Config 4's cheetah executes 2.07 Newton passes per lane but 4.45 per wave (profiles/r03_newton_stats.txt): a wave iterates until
its slowest lane is done.  Perfect binning = every wave made of lanes that need the same passes and the same row blocks.  This
script MEASURES that limit without building the exchange: after 300 steps (bodies on the ground, the regime of the statistics)
the state of lane 0 of every wave is copied to its 63 wave-mates and the actions are made wave-uniform, so that wave passes =
lane passes and a wave enters the primal loop only if its one distinct env does; the same launches are timed with the
original (heterogeneous) lanes.  Same kernel, same launch shape, same mix of contact situations across waves."""
import sys

import torch

sys.path.insert(0, ".")
from emei_amd.engine import Engine  # noqa: E402

CASES = [("HalfCheetahRunning", "euler", 6, 0.1), ("HopperRunning", "rk4", 3, 5e-3), ("HopperRunning", "euler", 3, 5e-3)]
N, T = 131072, 100
for env, integ, na, sigma in CASES:
    eng = Engine(env, N, freq_rate=4, real_time_scale=0.002, integrator=integ, init_noise=sigma, seed=0)
    eng.reset(0)
    gen = torch.Generator(device=eng.device)
    gen.manual_seed(1)
    acts = (torch.rand((T, N, na), device=eng.device, generator=gen) * 2 - 1).float()
    out = eng.alloc_outputs(T)
    for _ in range(3):
        eng.rollout(acts, out=out)
    st = eng.get_state()
    res = {}
    for mode in ("heterogeneous", "wave-homogeneous"):
        if mode == "wave-homogeneous":
            s = st.view(N // 64, 64, -1)[:, :1].expand(-1, 64, -1).reshape(N, -1).contiguous()
            a = acts.view(T, N // 64, 64, na)[:, :, :1].expand(-1, -1, 64, -1).reshape(T, N, na).contiguous()
        else:
            s, a = st, acts
        ts = []
        for rep in range(5):
            eng.set_state(s, reset_counters=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.rollout(a, out=out)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[mode] = sorted(ts)[len(ts) // 2]
        assert bool(torch.isfinite(eng.get_state()).all())
    print(f"{env} {integ}: {N} envs x {T} steps from step 300: heterogeneous lanes {res['heterogeneous']:.2f} ms, wave-homogeneous lanes "
          f"{res['wave-homogeneous']:.2f} ms -> perfect re-binning could save at most {100 * (1 - res['wave-homogeneous'] / res['heterogeneous']):.1f} %"
          f"; Newton solves at the cap: {eng.solver_cap_hits()}", flush=True)
    eng.close()
