#!/usr/bin/env python3
"""Where the Newton solver's time goes (tools/build_variant.sh stats -DEMEI_NEWTON_STATS, then run this under
EMEI_HIP_LIB=$PWD/gpurun_abl_stats.so): per forward-dynamics evaluation of HalfCheetah (config 4's shape) and Hopper, the number
of Newton passes a lane needs, the number its wave executes, and the contact-row blocks a wave executes per pass
against the ones an average lane has."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from emei_amd import _lib  # noqa: E402
from emei_amd.sharding import ShardedRollout  # noqa: E402

NAMES = ["evals_lane", "evals_rows_lane", "passes_lane", "passes_wave", "contact_blocks_wave", "contact_blocks_lane",
         "limit_blocks_wave", "evals_wave"]
for env, integ, tu in (("HalfCheetahRunning", "euler", "body_tu_ch_f64"), ("HalfCheetahRunning", "rk4", "body_tu_ch_f64"),
                       ("HopperRunning", "rk4", "body_tu_hp_f64"), ("HopperRunning", "euler", "body_tu_hp_f64")):
    for freq, rts in ((4, 0.002), (1, 0.008)):
        sr = ShardedRollout(env, 131072, 100, freq_rate=freq, real_time_scale=rts, integrator=integ, solver="newton")
        sr.make_synthetic_inputs()
        for _ in range(3):
            sr.run_pass()
        torch.cuda.synchronize()
        fn = getattr(_lib.lib(), "emei_debug_stats_" + tu)
        out = (C.c_ulonglong * 32)()
        assert fn(out) == 0  # clear
        sr.run_pass()
        torch.cuda.synchronize()
        assert fn(out) == 0
        s = dict(zip(NAMES, [int(x) for x in out[:8]]))
        ew, el = max(s["evals_wave"], 1), max(s["evals_lane"], 1)
        hist = [int(x) for x in out[8:22]]
        dl, dw, pl, pw = (int(out[k]) for k in (22, 23, 24, 25))
        print(f"{env} {integ} freq_rate={freq} dt={rts}: {s}")
        if dl + pl:
            print(f"   constraint-space path: {dl / (dl + pl):.3f} of the lane evaluations with rows, entered by {dw / ew:.3f} of the wave evaluations; "
                  f"primal loop entered by {pw / ew:.3f}")
        nb = [int(out[k]) for k in range(26, 32)]
        if sum(nb):
            print("   row blocks per lane with rows (permille): " + " ".join(f"{k + 1}{'+' if k == 5 else ''}:{1000 * h / sum(nb):.1f}" for k, h in enumerate(nb)))
        print("   passes per lane evaluation, histogram (permille): " + " ".join(f"{k}:{1000 * h / max(sum(hist), 1):.1f}" for k, h in enumerate(hist) if h))
        print(f"   lanes with rows {s['evals_rows_lane'] / el:.3f}; passes per evaluation: lane {s['passes_lane'] / el:.2f}, wave {s['passes_wave'] / ew:.2f}; "
              f"contact blocks per pass: lane {s['contact_blocks_lane'] / max(s['passes_lane'], 1):.2f}, wave {s['contact_blocks_wave'] / max(s['passes_wave'], 1):.2f}; "
              f"limit blocks per wave pass {s['limit_blocks_wave'] / max(s['passes_wave'], 1):.2f}", flush=True)
