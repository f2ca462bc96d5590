#!/usr/bin/env python3
"""Basic-block listing of one kernel of an assembly file written by tools/isa_regions.py (/tmp/isa_regions_<tag>.s):
    tools/isa_blocks.py /tmp/isa_regions_ch_f64.s Lb0
per block: vector / total instruction counts, EMEI_MARK names seen in it, branch targets, BACK for a loop back-edge."""
import re
import sys

path, want = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and want in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = [], ["<entry>", 0, 0, [], []]
for l in lines[start + 1:end]:
    t = l.strip()
    mm = re.match(r"^(\.LBB\d+_\d+):", t)
    if mm:
        blocks.append(cur)
        cur = [mm.group(1), 0, 0, [], []]
        continue
    if t.startswith("; EMEI_MARK"):
        cur[3].append(t.split()[2])
        continue
    if not t or t.startswith((";", ".")):
        continue
    op = t.split()[0]
    cur[1] += op.startswith("v_")
    cur[2] += 1
    if op.startswith("s_cbranch") or op == "s_branch":
        cur[4].append(t.split()[1])
blocks.append(cur)
idx = {b[0]: i for i, b in enumerate(blocks)}
for i, b in enumerate(blocks):
    back = [t for t in b[4] if t in idx and idx[t] <= i]
    print(f"{i:4d} {b[0]:<12} valu {b[1]:5d} all {b[2]:5d}  {','.join(b[3]):<28} -> {','.join(b[4])} {'BACK' if back else ''}")
