#!/usr/bin/env python3
"""ms per 100-step rollout of the multi-constraint bodies (131 072 envs, freq_rate 4) for both constraint solvers."""
import sys

import torch

sys.path.insert(0, ".")
from emei_amd.sharding import ShardedRollout  # noqa: E402

for solver in sys.argv[1:] or ["newton", "sweep1"]:
    for env, integ in (("HalfCheetahRunning", "euler"), ("HopperRunning", "rk4"), ("HopperRunning", "euler")):
        sr = ShardedRollout(env, 131072, 100, freq_rate=4, real_time_scale=0.002, integrator=integ, solver=solver)
        sr.make_synthetic_inputs()
        for _ in range(3):
            sr.run_pass()
        torch.cuda.synchronize()
        print(solver, env, integ, "ms per 100 steps: %.2f" % sr.timed_launches_ms(10), flush=True)
