#!/usr/bin/env python3
"""The workload of tools/newton_stats.py (131 072 envs, passes of 100 steps continuing the same episodes) with the state
looked at after every pass: share of finite lanes, share of lanes with constraint rows (oracle's row builder on the
kernel's state), torso height — and, when the library is a -DEMEI_NEWTON_STATS build, the device counters of that pass.
Run once with the shipped library and once with EMEI_HIP_LIB=$PWD/gpurun_abl_stats.so.

Test infrastructure: uses oracle/ as the checker."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from emei_amd import _lib  # noqa: E402
from emei_amd.sharding import ShardedRollout  # noqa: E402
from oracle import oracle as O  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
PASSES = int(sys.argv[2]) if len(sys.argv) > 2 else 6
print("library:", _lib.LIB_PATH, flush=True)
for env, integ, tu, body in (("HopperRunning", "rk4", "body_tu_hp_f64", "hopper"), ("HopperRunning", "euler", "body_tu_hp_f64", "hopper"),
                             ("HalfCheetahRunning", "euler", "body_tu_ch_f64", "cheetah")):
    for freq, rts in ((4, 0.002), (1, 0.008)):
        sr = ShardedRollout(env, N, 100, freq_rate=freq, real_time_scale=rts, integrator=integ, solver="newton")
        sr.make_synthetic_inputs()
        fn = getattr(_lib.lib(), "emei_debug_stats_" + tu, None)
        out = (C.c_ulonglong * 32)()
        print(f"== {env} {integ} freq_rate={freq} dt={rts}", flush=True)
        for p in range(PASSES):
            if fn is not None:
                assert fn(out) == 0  # clear
            sr.run_pass()
            torch.cuda.synchronize()
            st = sr.engine.get_state().cpu().numpy()
            fin = np.isfinite(st).all(axis=1)
            rows = O.planar_count_rows(body, st, rts)
            big = (np.abs(np.nan_to_num(st, nan=0.0, posinf=0.0, neginf=0.0)) > 1e3).any(axis=1)
            line = (f"  pass {p} (steps {100 * p}-{100 * p + 100}): finite {fin.mean():.4f}  |x|>1e3 {big.mean():.4f}  rows>0 {(rows > 0).mean():.3f}"
                    f"  z median {np.nanmedian(st[:, 1]):.3f}")
            if fn is not None:
                assert fn(out) == 0
                el, er, pl, pw, ew = int(out[0]), int(out[1]), int(out[2]), int(out[3]), int(out[7])
                cap = int(out[8 + 13])
                line += f"  | counters: lanes with rows {er / max(el, 1):.3f}, passes/eval lane {pl / max(el, 1):.2f} wave {pw / max(ew, 1):.2f}, >=13 passes {cap}"
            print(line, flush=True)
        sr.engine.close()
