#!/usr/bin/env python3
"""Every env id at a bench-like shape (ShardedRollout.run_pass, default integrator): a sweep for performance cliffs that the
four bench workloads do not visit (round 4: BoundaryInvertedPendulumBalancing ran the general two-row limit solve in most waves).
Run on the GPU box."""
import sys, torch
sys.path.insert(0, ".")
from emei_amd import _lib as L
from emei_amd.sharding import ShardedRollout
for env in sorted(L.ENV_IDS):
    classic = env.startswith("CartPole")
    heavy = env in ("HalfCheetahRunning", "HopperRunning")
    n = 65536 if classic else (131072 if heavy else 262144)
    T = 1000 if classic else (100 if heavy else 250)
    kw = dict(freq_rate=1 if classic else 4, real_time_scale=0.002 if heavy else 0.02)
    sr = ShardedRollout(env, n, T, **kw)
    sr.make_synthetic_inputs()
    for _ in range(2):
        sr.run_pass()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        sr.run_pass()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"{env:45s} n={n:7d} T={T:5d} {sr.kernel_name:40s} {ms:8.3f} ms per pass  {n*T/ms/1e6:9.2f}e9 env-steps/s", flush=True)
