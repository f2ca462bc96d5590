#!/usr/bin/env python3
"""Registers / scratch / LDS of every gfx950 kernel in the built library whose (demangled) name contains the given substrings:
    tools/kernel_resources.py rollout Cheetah"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("EMEI_HIP_LIB") or os.path.join(ROOT, "emei_amd", "libemei_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin/"
d = tempfile.mkdtemp()
shutil.copy(LIB, d)
subprocess.check_call([LLVM + "llvm-objdump", "--offloading", os.path.basename(LIB)], cwd=d, stdout=subprocess.DEVNULL)
rows = {}
for f in sorted(os.listdir(d)):
    if "gfx950" not in f:
        continue
    txt = subprocess.check_output([LLVM + "llvm-readelf", "--notes", os.path.join(d, f)], text=True)
    for blk in txt.split("- .agpr_count:")[1:]:
        blk = ".agpr_count:" + blk
        g = {k: v for k, v in re.findall(r"\.(\w+):\s+(\S+)", blk)}
        if "name" in g:
            rows[g["name"]] = g
names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.split("\n")
for mangled, name in zip(rows, names):
    if all(w in name for w in sys.argv[1:]):
        g = rows[mangled]
        print(f"vgpr {g.get('vgpr_count'):>4} agpr {g.get('agpr_count'):>4} sgpr {g.get('sgpr_count'):>4} scratch {g.get('private_segment_fixed_size'):>5} "
              f"lds {g.get('group_segment_fixed_size'):>6} spill(v/s) {g.get('vgpr_spill_count')}/{g.get('sgpr_spill_count')}  {name[:110]}")
shutil.rmtree(d)
