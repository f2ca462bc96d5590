#!/usr/bin/env python3
"""Where the Newton solver's time goes (tools/build_variant.sh stats -DEMEI_NEWTON_STATS, then run this under
EMEI_HIP_LIB=$PWD/gpurun_abl_stats.so): per forward-dynamics evaluation of HalfCheetah (config 4's shape) and Hopper, the number
of Newton passes a lane needs, the number its wave executes, and the contact-row blocks a wave executes per pass
against the ones an average lane has."""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

import torch

sys.path.insert(0, ".")
from emei_amd import _lib  # noqa: E402
from emei_amd.sharding import ShardedRollout  # noqa: E402

NAMES = ["evals_lane", "evals_rows_lane", "passes_lane", "passes_wave", "contact_blocks_wave", "contact_blocks_lane",
         "limit_blocks_wave", "evals_wave"]
CASES = [(env, integ, tu, freq, rts)
         for env, integ, tu in (("HalfCheetahRunning", "euler", "body_tu_ch_f64"), ("HalfCheetahRunning", "rk4", "body_tu_ch_f64"),
                                ("HopperRunning", "rk4", "body_tu_hp_f64"), ("HopperRunning", "euler", "body_tu_hp_f64"))
         for freq, rts in ((4, 0.002), (1, 0.008))]
DIGEST_ONLY = "--digest-only" in sys.argv


def digest(sr):
    st = sr.engine.get_state()
    assert bool(torch.isfinite(st).all()), "non-finite state: this build of the kernel is broken"
    return hashlib.sha256(st.cpu().numpy().tobytes()).hexdigest()


# Round 2 took its Hopper RK4 numbers from a statistics build that hipcc had miscompiled (every lane but one per wave went
# non-finite: profiles/r03_hopper_rk4_stats_diag.txt).  The counters are only reported when the instrumented library leaves
# the SAME bits in the state as the shipped one after the same passes (the counters do not touch the arithmetic).
shipped = None
if not DIGEST_ONLY and os.environ.get("EMEI_HIP_LIB"):
    env = {k: v for k, v in os.environ.items() if k != "EMEI_HIP_LIB"}
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--digest-only"], env=env, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    shipped = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])

digests = {}
for env, integ, tu, freq, rts in CASES:
    sr = ShardedRollout(env, 131072, 100, freq_rate=freq, real_time_scale=rts, integrator=integ, solver="newton")
    sr.make_synthetic_inputs()
    for _ in range(3):
        sr.run_pass()
    torch.cuda.synchronize()
    key = f"{env} {integ} freq_rate={freq} dt={rts}"
    if DIGEST_ONLY:
        sr.run_pass()
        torch.cuda.synchronize()
        digests[key] = digest(sr)
        continue
    fn = getattr(_lib.lib(), "emei_debug_stats_" + tu)
    out = (C.c_ulonglong * 32)()
    assert fn(out) == 0  # clear
    sr.run_pass()
    torch.cuda.synchronize()
    assert fn(out) == 0
    d = digest(sr)
    if shipped is not None:
        assert shipped[key] == d, f"{key}: the instrumented build's state differs from the shipped library's"
    s = dict(zip(NAMES, [int(x) for x in out[:8]]))
    ew, el = max(s["evals_wave"], 1), max(s["evals_lane"], 1)
    hist = [int(x) for x in out[8:22]]
    dl, dw, pl, pw = (int(out[k]) for k in (22, 23, 24, 25))
    print(f"{key}: {s}" + ("   [state bit-identical to the shipped library's]" if shipped is not None else ""))
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
if DIGEST_ONLY:
    print(json.dumps(digests))
