#!/usr/bin/env python3
"""SIMD-time utilisation of the one-wave-per-SIMD body rollouts, one-piece launch against chunked work items (round 5).

    tools/build_variant.sh clk -DEMEI_CLOCK_PROBE && EMEI_HIP_LIB=$PWD/gpurun_abl_clk.so python tools/tail_probe.py [chunk ...]

With the probe build every wave (one-piece launch) / every work item (chunked launch) adds its lifetime in ticks of the 100 MHz
counter (s_memrealtime) to a sum and the launch's first begin / last end are kept in the same ticks, so for ONE launch
    utilisation = sum of lifetimes / (SIMDs x (last end - first begin))
is a ratio of readings of one counter; the kernel time from a HIP-event pair around back-to-back launches is printed beside it
(and calibrates the counter: ticks per microsecond).  With the shipped library (no EMEI_HIP_LIB) only kernel times are printed.
Chunk lengths: -1 = one piece (round 4's launch), 0 = the automatic policy, k = k steps per item."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from emei_amd import _lib  # noqa: E402
from emei_amd.sharding import ShardedRollout  # noqa: E402

chunks = [int(a) for a in sys.argv[1:] if not a.startswith('--')] or [-1, 25, 10, 7, 5, 4, 2]
CASES = (("HalfCheetahRunning", 131072, 100, 4, 0.002, "euler", "body_tu_ch_f64"), ("HopperRunning", 131072, 100, 4, 0.002, "rk4", "body_tu_hp_f64"),
         ("HopperRunning", 131072, 100, 4, 0.002, "euler", "body_tu_hp_f64"))
simds = torch.cuda.get_device_properties(0).multi_processor_count * 4
M64 = (1 << 64) - 1
for env, n, T, fr, dt, integ, tu in CASES:
    for ch in chunks:
        sr = ShardedRollout(env, n, T, freq_rate=fr, real_time_scale=dt, integrator=integ, rollout_chunk_steps=ch)
        sr.make_synthetic_inputs()
        for _ in range(2):
            sr.run_pass()
        torch.cuda.synchronize()
        fn = getattr(_lib.lib(), "emei_debug_stats_" + tu, None)
        out = (C.c_ulonglong * 32)()
        reps = 4
        ms = sr.timed_launches_ms(reps)
        line = f"{env} ({integ}) chunk {ch:>3}: kernel {ms * 1e3:8.1f} us [{sr.kernel_name.replace('body_rollout_kernel', 'k')}]"
        if fn is not None:
            assert fn(out) == 0  # clear
            sr.run_pass()  # ONE launch
            torch.cuda.synchronize()
            assert fn(out) == 0
            wait, end, cyc, ticks, items, nbegin = int(out[26]), int(out[27]), int(out[28]), int(out[29]), int(out[30]), int(out[31])
            first = (~nbegin) & M64
            span = end - first
            workers, last_first = int(out[24]), int(out[25])
            line += (f"  items {items:6d}  mean lifetime {ticks / items / 100.0:8.1f} us  span {span / 100.0:8.1f} us  clock {cyc / ticks * 0.1:.3f} GHz"
                     f"  waiting for a predecessor {wait / ticks * 100:5.2f} %  SIMD-time utilisation {ticks / (simds * span) * 100:5.1f} %")
            line += f"  longest lifetime {int(out[23]) / 100.0:.1f} us"
            if int(out[21]):
                line += f"  mean worker exit {(end - int(out[22]) / int(out[21])) / 100.0:.1f} us before the last end"
            if "--corr" in sys.argv and int(out[0]):  # a -DEMEI_CLOCK_HIST_CORR build
                nn, sx, sy, sxy, sxx, syy = (float(int(out[k])) for k in range(6))
                cov, vx, vy = sxy / nn - sx * sy / nn**2, sxx / nn - (sx / nn) ** 2, syy / nn - (sy / nn) ** 2
                line += f"  ticks/step of an item vs the same wave's previous item: r = {cov / (vx * vy) ** 0.5:.3f} (mean {sy / nn:.0f}, sd {vy ** 0.5:.0f})"
            if workers:
                line += f"  workers with an item {workers}, the last began its first item {(last_first - first) / 100.0:.1f} us into the launch"
        print(line + f"  faults {sr.engine.rollout_faults()}", flush=True)
        sr.engine.close()
        del sr
