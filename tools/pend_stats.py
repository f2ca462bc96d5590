#!/usr/bin/env python3
"""Event counters of the staged pendulum kernels (tools/build_variant.sh stats -DEMEI_NEWTON_STATS, then
EMEI_HIP_LIB=$PWD/gpurun_abl_stats.so python tools/pend_stats.py): per wave, the fraction of env-steps in which some lane
resets, in which the spare initial states are redrawn, and of substeps that execute the slider-limit block."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from emei_amd import _lib  # noqa: E402
from emei_amd.sharding import ShardedRollout  # noqa: E402

for env, n, T, fr, tu in (("BoundaryInvertedPendulumSwingUp", 262144, 250, 4, "pend_tu_ip3_f64"), ("BoundaryInvertedPendulumBalancing", 262144, 250, 4, "pend_tu_ip1_f64"),
                          ("CartPoleBalancing", 65536, 500, 1, "pend_tu_cp1_f64"),
                          ("CartPoleSwingUp", 65536, 1000, 1, "pend_tu_cp0_f64")):
    sr = ShardedRollout(env, n, T, freq_rate=fr, real_time_scale=0.02)
    sr.make_synthetic_inputs()
    for _ in range(3):
        sr.run_pass()
    torch.cuda.synchronize()
    fn = getattr(_lib.lib(), "emei_debug_stats_" + tu)
    out = (C.c_ulonglong * 32)()
    assert fn(out) == 0
    sr.run_pass()
    torch.cuda.synchronize()
    assert fn(out) == 0
    steps, resets, refills, sub, lim, lanes = (int(out[k]) for k in range(16, 22))
    print(f"{env}: wave env-steps {steps}; some lane resets in {resets / max(steps, 1):.3f} of them; spares redrawn in {refills / max(steps, 1):.3f}"
          + (f"; substeps {sub}, limit block in {lim / max(sub, 1):.3f}, lanes beyond the rail {lanes / max(sub * 64, 1):.4f}" if sub else ""), flush=True)
