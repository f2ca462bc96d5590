#!/usr/bin/env python3
"""The shader clock a rollout kernel really runs at (round 4).  Build and run on the GPU box:

    tools/build_variant.sh clk -DEMEI_CLOCK_PROBE && EMEI_HIP_LIB=$PWD/gpurun_abl_clk.so python tools/clock_probe.py

Every wave reads s_memtime (shader cycles) and s_memrealtime (100 MHz) at its start and at its end (emei_device.h:ClockProbe);
sum of cycles / sum of ticks x 100 MHz is the clock averaged over the waves' lifetimes.  Also prints the mean wave lifetime
against the launch duration (waves resident per SIMD)."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from emei_amd import _lib  # noqa: E402
from emei_amd.sharding import ShardedRollout  # noqa: E402

CASES = (("CartPoleSwingUp", 65536, 1000, 1, 0.02, "euler", "pend_tu_cp0_f64"), ("CartPoleBalancing", 65536, 500, 1, 0.02, "euler", "pend_tu_cp1_f64"),
         ("BoundaryInvertedPendulumSwingUp", 262144, 250, 4, 0.02, "euler", "pend_tu_ip3_f64"),
         ("BoundaryInvertedPendulumBalancing", 262144, 250, 4, 0.02, "euler", "pend_tu_ip1_f64"),
         ("BoundaryInvertedDoublePendulumSwingUp", 262144, 100, 4, 0.02, "euler", "body_tu_dp3_f64"),
         ("HalfCheetahRunning", 131072, 100, 4, 0.002, "euler", "body_tu_ch_f64"), ("HopperRunning", 131072, 100, 4, 0.002, "rk4", "body_tu_hp_f64"))
for env, n, T, fr, dt, integ, tu in CASES:
    kw = {} if env.startswith("CartPole") or tu.startswith("pend_tu_ip") else {"integrator": integ}
    sr = ShardedRollout(env, n, T, freq_rate=fr, real_time_scale=dt, **kw)
    sr.make_synthetic_inputs()
    # Round 5: warm up for ~0.3 s.  Round 4 ran three passes here and read 1.65 GHz for config 3 — the clock of a GPU still ramping
    # up from idle, not a power limit: after a bench-like settle phase the same kernel runs at 2.3 GHz (profiles/r05_clock_probe.txt).
    import time
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        for _ in range(10):
            sr.run_pass()
        torch.cuda.synchronize()
    fn = getattr(_lib.lib(), "emei_debug_stats_" + tu)
    out = (C.c_ulonglong * 32)()
    assert fn(out) == 0  # clear
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        sr.run_pass()
    e1.record()
    torch.cuda.synchronize()
    assert fn(out) == 0
    cyc, ticks, waves = int(out[28]), int(out[29]), int(out[30])
    ms = e0.elapsed_time(e1) / reps
    life_us = ticks / waves / 100.0
    print(f"{env} ({integ}): {waves // reps} waves per launch, shader clock {cyc / ticks * 0.1:.3f} GHz over the waves' lifetimes; mean wave lifetime "
          f"{life_us:.1f} us of a {ms * 1e3:.1f} us launch", flush=True)
