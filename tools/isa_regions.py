#!/usr/bin/env python3
"""Static vector-instruction counts per source region of a body kernel.

    tools/isa_regions.py ch_f64 [kernel-name-substring]

Compiles emei_amd/csrc/body_tu.hip for the tag (as the Makefile does, plus -DEMEI_ISA_MARKS -S) into /tmp and, for
every kernel whose name contains the substring (default: rollout), prints the number of VALU / SALU / memory
instructions between consecutive `; EMEI_MARK <name>` comments in address order.  The marks are volatile asm comments:
hipcc may move arithmetic across them, so the split is approximate (good to ~10 instructions); the totals are exact.
Static counts, not executed ones: a region inside a loop or behind a branch counts once.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "emei_amd", "csrc")


def flags_for(tag):
    body = tag.split("_")[0]
    fam = "dp" if body.startswith("dp") else "ip" if body.startswith("ip") else body.rstrip("s") if body in ("chs", "hps") else body
    var = re.sub(r"^(ch|hp|dp|ip)", "", body).replace("s", "") or "0"
    real = "double" if tag.endswith("_f64") else "float"
    extra = ["-mllvm", "-disable-machine-licm"] if fam in ("ch", "dp") else []  # as emei_amd/csrc/Makefile: body_flags (ch, chs, dp)
    return extra + [f"-DEMEI_TU_NAME=body_tu_{tag}", f"-DEMEI_BODY_FAM_{body if body in ('chs', 'hps') else fam}", f"-DEMEI_BODY_VARIANT={var}",
            f"-DEMEI_TU_REAL={real}"]


def main():
    tag = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "rollout"
    extra = sys.argv[3:]
    out = f"/tmp/isa_regions_{tag}.s"
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fvisibility=hidden",
           "--cuda-device-only", "-S", "-DEMEI_ISA_MARKS", *flags_for(tag), *extra, os.path.join(CSRC, "body_tu.hip"), "-o", out]
    subprocess.run(cmd, check=True)
    kern, counts, order, region = None, None, None, None
    for line in open(out):
        t = line.strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            kern = m.group(1) if want in m.group(1) else None
            counts, order, region = collections.defaultdict(collections.Counter), ["<entry>"], "<entry>"
            continue
        if kern is None:
            continue
        if t.startswith("; EMEI_MARK"):
            region = t.split()[2]
            if region not in order:
                order.append(region)
            continue
        if t.startswith(".Lfunc_end") or t.startswith("s_endpgm"):
            if t.startswith("s_endpgm"):
                continue
            demangled = subprocess.run(["c++filt", kern], capture_output=True, text=True).stdout.strip()
            print(demangled[:150])
            tot = collections.Counter()
            for r in order:
                c = counts[r]
                tot.update(c)
                print(f"  {r:<22} valu {c['v']:>6}  salu {c['s']:>6}  mem {c['m']:>5}")
            print(f"  {'total':<22} valu {tot['v']:>6}  salu {tot['s']:>6}  mem {tot['m']:>5}")
            kern = None
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith(("v_",)):
            counts[region]["v"] += 1
        elif op.startswith(("s_",)):
            counts[region]["s"] += 1
        elif op.startswith(("global_", "scratch_", "buffer_", "ds_", "flat_")):
            counts[region]["m"] += 1


if __name__ == "__main__":
    main()
