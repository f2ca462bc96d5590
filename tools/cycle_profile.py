#!/usr/bin/env python3
"""Where a wave of the cheetah rollout kernel spends its cycles (round 4).  Build and run on the GPU box:

    tools/build_variant.sh cyc -DEMEI_CYCLE_PROFILE && EMEI_HIP_LIB=$PWD/gpurun_abl_cyc.so python tools/cycle_profile.py

In that build every EMEI_MARK charges the shader-clock cycles since the wave's previous mark to the region that ends there
(emei_device.h:emei_cycle_mark): divergent regions, waits and spill traffic included.  COARSE: a mark costs ~500 cycles itself and
hipcc moves arithmetic across marks (the M factorisation lands in `nw_direct`), so the instrumented kernel is ~35 % slower and
only multi-thousand-cycle regions mean anything.  It must leave the same bits in the state as the shipped library (checked below
when EMEI_HIP_LIB points at the variant).  Build: the cheetah translation unit alone with -DEMEI_CYCLE_PROFILE (hipcc fails on other
units with the marks: "Operand has incorrect register class"), the other objects from the normal build."""
import ctypes as C
import hashlib
import os
import subprocess
import sys

import torch

sys.path.insert(0, ".")
from emei_amd import _lib  # noqa: E402
from emei_amd.sharding import ShardedRollout  # noqa: E402

REGIONS = ["entry", "nw_trig", "nw_forces", "nw_rows", "nw_direct", "nw_smooth0", "dual_fill", "dual_gram", "dual_loop", "dual_final",
           "nw_pass_base", "nw_limits", "nw_contacts", "nw_conv", "nw_step", "nw_final", "nw_euler", "nw_out", "step_io", "step_reset",
           "hp_pairs", "hp_verify", "tri_gram", "tri_loop", "tri_final"]
WHAT = {"entry": "(kernel prologue / between steps)", "nw_trig": "7 sincos, rotated link vectors", "nw_forces": "smooth forces",
        "nw_rows": "which rows exist (16 points, 6 limits)", "nw_direct": "free-flight solve", "nw_smooth0": "M = L D L', qacc_smooth",
        "dual_fill": "constraint slots: J, L^-1 J', LDS puts (union of the wave's blocks)", "dual_gram": "slots back, G = Y' D^-1 Y",
        "dual_loop": "constraint-space active-set passes", "dual_final": "a = a0 - L^-T D^-1 sum Y g",
        "nw_pass_base": "primal loop: gradient base", "nw_limits": "primal loop: limit rows", "nw_contacts": "primal loop: contact rows",
        "nw_conv": "primal loop: convergence test", "nw_step": "primal loop: factor H, Newton step", "nw_final": "warm store / cap report",
        "nw_euler": "Euler damping step: factor M + h B, solve", "nw_out": "back to joint coordinates + the integrator's update",
        "step_io": "outputs: reward / terminal / obs staging and stores", "step_reset": "auto-reset check (+ action staging of the next step)", "hp_pairs": "primal loop: capsule-pair rows (Hopper)",
        "hp_verify": "primal loop: re-evaluation of the rows' residuals after the step (RK4 kernels)",
        "tri_gram": "three-block lanes: G (6 x 6) from three slots", "tri_loop": "three-block lanes: active-set passes",
        "tri_final": "three-block lanes: a = a0 - L^-T D^-1 sum Y g"}
env, integ, tu = (sys.argv[1:4] + ["HalfCheetahRunning", "euler", "body_tu_ch_f64"][len(sys.argv) - 1:])[:3]


def run(read):
    sr = ShardedRollout(env, 131072, 100, freq_rate=4, real_time_scale=0.002, integrator=integ, solver="newton")
    sr.make_synthetic_inputs()
    for _ in range(3):
        sr.run_pass()
    torch.cuda.synchronize()
    out = None
    if read:
        fn = getattr(_lib.lib(), "emei_cycle_stats_" + tu)
        out = (C.c_ulonglong * 32)()
        assert fn(out) == 0  # clear
    sr.run_pass()
    torch.cuda.synchronize()
    if read:
        assert fn(out) == 0
    st = sr.engine.get_state()
    assert bool(torch.isfinite(st).all())
    return hashlib.sha256(st.cpu().numpy().tobytes()).hexdigest(), (list(out) if read else None)


if "--digest-only" in sys.argv:
    print(run(False)[0])
    sys.exit(0)
dig, cyc = run(True)
if os.environ.get("EMEI_HIP_LIB"):
    e = {k: v for k, v in os.environ.items() if k != "EMEI_HIP_LIB"}
    ref = subprocess.run([sys.executable, os.path.abspath(__file__), env, integ, tu, "--digest-only"], env=e, capture_output=True, text=True)
    assert ref.returncode == 0, ref.stderr[-1500:]
    same = ref.stdout.strip().splitlines()[-1] == dig
    print(f"# state after the same passes {'BIT-IDENTICAL to' if same else 'DIFFERS FROM'} the shipped library's")
tot = sum(cyc[:len(REGIONS)])
evals = 2048 * 100 * 4 * (4 if integ == "rk4" else 1)
print(f"# {env} {integ}: 131072 envs x 100 steps (4th launch); {tot / 2048 / 100:.0f} cycles per wave env-step, {tot / evals:.0f} per forward-dynamics evaluation")
for i, nm in enumerate(REGIONS):
    if cyc[i]:
        print(f"{nm:14s} {100.0 * cyc[i] / tot:5.1f} %  {cyc[i] / evals:7.0f} cycles per evaluation   {WHAT[nm]}")
