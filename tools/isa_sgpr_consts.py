#!/usr/bin/env python3
"""Constant propagation of scalar registers (and SGPR spill lanes) over a hipcc `-save-temps` kernel: lists the
64-bit constants that f64 vector instructions read from SGPR pairs, decoded as doubles.  Comparing the list of two
builds of the same source finds mis-paired or clobbered literal constants (cheetah RK4 analysis).
usage: isa_sgpr_consts.py file.s <mangled kernel name>"""
import re
import struct
import sys

TOP, NAC = "T", "N"


def parse(path, kern):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur, order = {"entry": []}, "entry", ["entry"]
    for i in range(start + 1, end):
        l = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        if not l.startswith("\t") or l.startswith("\t.") or l.startswith("\t;"):
            continue
        body = l.split(";")[0].strip()
        if body:
            mn, _, rest = body.partition(" ")
            blocks[cur].append((i + 1, mn, [o.strip() for o in rest.split(",")] if rest else [], body))
    succ = {b: [] for b in blocks}
    for k, b in enumerate(order):
        ins = blocks[b]
        for (_, mn, ops, _) in ins:
            if mn.startswith("s_cbranch"):
                succ[b].append(ops[-1])
            elif mn == "s_branch":
                succ[b].append(ops[0])
        if not (ins and ins[-1][1] in ("s_branch", "s_endpgm", "s_setpc_b64")) and k + 1 < len(order):
            succ[b].append(order[k + 1])
    return blocks, order, succ


def sregs(op):
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", op)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", op)
    return [int(m.group(1))] if m else []


def imm32(op):
    try:
        return int(op, 0) & 0xFFFFFFFF
    except ValueError:
        return None


def imm64(op):
    try:
        v = int(op, 0)
        return v & 0xFFFFFFFFFFFFFFFF
    except ValueError:
        try:
            return struct.unpack("<Q", struct.pack("<d", float(op)))[0]
        except ValueError:
            return None


def meet(a, b):
    if a == TOP:
        return b
    if b == TOP:
        return a
    return a if a == b else NAC


def transfer(ins, st, sink=None):
    st = dict(st)
    for (ln, mn, ops, body) in ins:
        if sink is not None and re.match(r"v_(fma|fmac|mul|add|max|min|cmp\w*)_f64", mn):
            for o in ops[1:]:
                r = sregs(o.lstrip("-|").rstrip("|"))
                if len(r) == 2:
                    lo, hi = st.get(("s", r[0]), TOP), st.get(("s", r[1]), TOP)
                    sink.append((ln, body, lo, hi))
        if mn == "s_mov_b32" and sregs(ops[0]):
            v = imm32(ops[1])
            src = sregs(ops[1])
            st[("s", sregs(ops[0])[0])] = v if v is not None else (st.get(("s", src[0]), TOP) if src else NAC)
        elif mn == "s_mov_b64" and len(sregs(ops[0])) == 2:
            d = sregs(ops[0])
            v = imm64(ops[1])
            src = sregs(ops[1])
            if v is not None:
                st[("s", d[0])], st[("s", d[1])] = v & 0xFFFFFFFF, v >> 32
            elif len(src) == 2:
                st[("s", d[0])], st[("s", d[1])] = st.get(("s", src[0]), TOP), st.get(("s", src[1]), TOP)
            else:
                st[("s", d[0])] = st[("s", d[1])] = NAC
        elif mn == "v_writelane_b32" and ops[2].isdigit():
            s = sregs(ops[1])
            st[("L", ops[0], int(ops[2]))] = st.get(("s", s[0]), TOP) if s else NAC
        elif mn == "v_readlane_b32" and ops[2].isdigit():
            st[("s", sregs(ops[0])[0])] = st.get(("L", ops[1], int(ops[2])), TOP)
        else:
            # any other instruction: its scalar destinations become unknown (first operand; v_cmp / v_mad carry-out too)
            dests = ops[:1]
            if mn.startswith(("v_mad_u64", "v_mad_i64", "v_add_co", "v_sub_co", "v_addc", "v_subb", "v_div_scale")):
                dests = ops[:2]
            if mn.startswith(("s_cmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "global_store", "ds_write", "s_bitcmp")):
                dests = []
            for o in dests:
                for r in sregs(o):
                    st[("s", r)] = NAC
    return st


def main():
    blocks, order, succ = parse(sys.argv[1], sys.argv[2])
    inn = {b: None for b in blocks}
    inn["entry"] = {}
    out, work = {}, ["entry"]
    while work:
        b = work.pop()
        o = transfer(blocks[b], inn[b])
        if out.get(b) == o:
            continue
        out[b] = o
        for s in succ[b]:
            if s not in blocks:
                continue
            if inn[s] is None:
                new = dict(o)
            else:
                keys = set(inn[s]) | set(o)
                new = {k: meet(inn[s].get(k, TOP), o.get(k, TOP)) for k in keys}
            if new != inn[s]:
                inn[s] = new
                work.append(s)
    sink = []
    for b in order:
        if inn[b] is not None:
            transfer(blocks[b], inn[b], sink)
    vals = {}
    unknown = 0
    for ln, body, lo, hi in sink:
        if isinstance(lo, int) and isinstance(hi, int):
            d = struct.unpack("<d", struct.pack("<Q", (hi << 32) | lo))[0]
            vals.setdefault((hi << 32) | lo, [d, 0, ln])[1] += 1
        else:
            unknown += 1
    print(f"# {sys.argv[2]}: {len(vals)} distinct f64 constants read from SGPR pairs ({unknown} reads of non-constant / unknown pairs)")
    for bits, (d, cnt, ln) in sorted(vals.items(), key=lambda kv: kv[1][0]):
        print(f"{bits:016x} {d!r:>26} x{cnt} first at line {ln}")


if __name__ == "__main__":
    main()
