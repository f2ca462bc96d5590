#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/collect_profiles.sh (gpurun_out/prof_<tag>/) into the
tracked summaries under profiles/: <round>_<workload>_kernel_stats.csv, <round>_<workload>_pmc.txt and
profiles/traffic.json (HBM bytes per launch with the gfx950 FETCH_SIZE correction of
MI355X_MICROARCH.md: FETCH_SIZE counts 128-B requests of a wide coalesced stream as 64 B -> x2;
WRITE_SIZE is exact for 16 B/lane stores; both are in KiB)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mean_counters(d, pat):
    agg = collections.defaultdict(list)
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:  # the newest pass only (gpurun_out/ accumulates earlier collections)
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def pass_duration_us(d, pat):
    """mean duration of the matching kernel inside a counter pass (its own --kernel-trace): counter collection
    serialises and slows the dispatches, so cycles of a pass go with the durations of the SAME pass"""
    files = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    ts = []
    for f in files[-1:]:
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                ts.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return sum(ts) / len(ts) if ts else None


def main():
    raw, rnd, workload, pat = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
    n_envs, horizon, bytes_per_step, waves = int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    stats = sorted(glob.glob(os.path.join(raw, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
    shutil.copy(stats, os.path.join(ROOT, "profiles", f"{rnd}_{workload}_kernel_stats.csv"))
    krow = [r for r in csv.DictReader(open(stats)) if pat in r["Name"]][0]
    fetch = mean_counters(os.path.join(raw, "fetch"), pat).get("FETCH_SIZE")
    write = mean_counters(os.path.join(raw, "write"), pat).get("WRITE_SIZE")
    sq = mean_counters(os.path.join(raw, "sq"), pat)
    alg = bytes_per_step * n_envs * horizon
    traffic = (2 * fetch + write) * 1024
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    allj = json.load(open(tj)) if os.path.exists(tj) else {}
    allj[workload] = {
        "bytes_per_launch": traffic, "fetch_size_kib_raw": fetch, "write_size_kib_raw": write,
        "fetch_correction": "x2 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B; MI355X_MICROARCH.md, HBM section)",
        "algorithmic_bytes_per_launch": alg, "ratio_traffic_over_algorithmic": traffic / alg,
        "rocprof_kernel_avg_us": float(krow["AverageNs"]) / 1e3, "rocprof_kernel_min_us": float(krow["MinNs"]) / 1e3,
        "rocprof_kernel_calls": int(krow["Calls"]), "profile": f"profiles/{rnd}_{workload}_*",
        "valu_insts_per_launch": sq.get("SQ_INSTS_VALU"), "salu_insts_per_launch": sq.get("SQ_INSTS_SALU"),
        "sq_pass_kernel_us": pass_duration_us(os.path.join(raw, "sq"), pat),
    }
    json.dump(allj, open(tj, "w"), indent=1)
    with open(os.path.join(ROOT, "profiles", f"{rnd}_{workload}_pmc.txt"), "w") as fo:
        fo.write(f"rocprofv3 PMC summary ({rnd}), bench.py workload {workload}: {n_envs} envs x {horizon} steps per launch, REF (f64)\n")
        fo.write(f"kernel: {krow['Name'][:140]}\n")
        fo.write(f"kernel-trace --stats: {krow['Calls']} calls, avg {float(krow['AverageNs'])/1e3:.1f} us, min {float(krow['MinNs'])/1e3:.1f}, max {float(krow['MaxNs'])/1e3:.1f}\n\n")
        fo.write(f"pass --pmc FETCH_SIZE : {fetch:.1f} KiB raw -> x2 gfx950 correction = {2*fetch*1024/1e6:.1f} MB read per launch\n")
        fo.write(f"pass --pmc WRITE_SIZE : {write:.1f} KiB     = {write*1024/1e6:.1f} MB written per launch\n")
        fo.write(f"HBM traffic per launch: {traffic/1e6:.1f} MB (algorithmic {bytes_per_step} B x {n_envs*horizon/1e6:.3f}e6 env-steps = {alg/1e6:.1f} MB; ratio {traffic/alg:.3f})\n\n")
        fo.write("pass SQ counters (sum over all waves; WAVE_CYCLES / WAIT_* / ACTIVE_* are quad-cycles)\n")
        for k, v in sorted(sq.items()):
            fo.write(f"   {k:24s} {v:18.1f}\n")
        w = sq["SQ_WAVE_CYCLES"]
        per = waves * horizon
        fo.write(f"\n   per wave per env-step: VALU {sq['SQ_INSTS_VALU']/per:.1f}  SALU {sq['SQ_INSTS_SALU']/per:.1f}  LDS {sq['SQ_INSTS_LDS']/per:.2f} instructions; {4*w/per:.0f} cycles\n")
        sq_us = pass_duration_us(os.path.join(raw, "sq"), pat)
        if sq_us:
            fo.write(f"   kernel duration INSIDE the SQ pass: {sq_us:.1f} us (stats pass: {float(krow['AverageNs'])/1e3:.1f}); "
                     f"implied clock = 4 x SQ_WAVE_CYCLES / waves / duration = {4*w/waves/sq_us/1e3:.2f} GHz "
                     f"(only meaningful when every wave is resident for the whole launch)\n")
            fo.write(f"   VALU issue roofline: {sq['SQ_INSTS_VALU']:.4g} wave-instructions x 4 cycles / (1024 SIMDs x 2.4 GHz x {float(krow['AverageNs'])/1e3:.1f} us) = "
                     f"{sq['SQ_INSTS_VALU']*4/(1024*2.4e9*float(krow['AverageNs'])*1e-9):.1%}\n")
        fo.write(f"   wave time split: issuing {sq['SQ_ACTIVE_INST_ANY']/w:.1%}  parked at s_waitcnt {sq['SQ_WAIT_ANY']/w:.1%}  issue stall {sq['SQ_WAIT_INST_ANY']/w:.1%}\n")
    print(open(os.path.join(ROOT, "profiles", f"{rnd}_{workload}_pmc.txt")).read())


if __name__ == "__main__":
    main()
