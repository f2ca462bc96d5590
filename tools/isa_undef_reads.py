#!/usr/bin/env python3
"""Must-define dataflow over a hipcc `-save-temps` .s kernel: reports vector / accumulator registers that an
instruction READS although no instruction has written them on SOME path from the kernel entry (EXEC-masked writes
count as writes).  A report is not proof of a bug — hipcc may legitimately read an undefined register whose value
it ignores — but a hoisted or mis-scheduled reload shows up here.  Used for the cheetah RK4 root-cause analysis
(DESIGN.md §6).   usage: isa_undef_reads.py file.s <mangled kernel name>"""
import re
import sys

NO_DEF = ("global_store", "ds_write", "scratch_store", "buffer_store", "flat_store", "s_cmp", "s_cbranch", "s_branch",
          "s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_setprio", "s_sleep", "s_bitcmp", "v_cmpx", "s_setreg",
          "global_atomic", "ds_add", "s_dcache", "buffer_wbl2", "buffer_inv", "s_sendmsg", "s_trap", "s_code_end", "v_nop", "s_setpc")
RMW = ("v_fmac", "v_mac", "v_writelane", "v_dot2c", "v_pk_fmac", "v_cndmask_b32_sdwa")


def regs(op):
    """-> set of ('v'|'a', n) named in one operand"""
    out = set()
    for k, a, b in re.findall(r"\b([va])\[(\d+):(\d+)\]", op):
        out.update((k, i) for i in range(int(a), int(b) + 1))
    for k, a in re.findall(r"\b([va])(\d+)\b", op):
        out.add((k, int(a)))
    return out


def main():
    path, kern = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = {"entry": []}, "entry"
    order = ["entry"]
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
        if not body:
            continue
        mn, _, rest = body.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        blocks[cur].append((i + 1, mn, ops, body))
    succ = {b: [] for b in blocks}
    for k, b in enumerate(order):
        ins = blocks[b]
        last = ins[-1] if ins else None
        fall = True
        for (_, mn, ops, _) in ins:
            if mn.startswith("s_cbranch"):
                succ[b].append(ops[-1])
            elif mn == "s_branch":
                succ[b].append(ops[0])
        if last and last[1] in ("s_branch", "s_endpgm", "s_setpc_b64"):
            fall = False
        if fall and k + 1 < len(order):
            succ[b].append(order[k + 1])

    def transfer(b, defined, report=None):
        d = set(defined)
        for (ln, mn, ops, body) in blocks[b]:
            nodef = mn.startswith(NO_DEF)
            uses = set()
            for j, o in enumerate(ops):
                if j == 0 and not nodef and not mn.startswith(RMW):
                    continue
                uses |= regs(o)
            if mn.startswith("v_mad_u64_u32") or mn.startswith("v_mad_i64_i32"):  # two destinations
                uses -= regs(ops[1]) if len(ops) > 1 else set()
            if report is not None:
                bad = uses - d
                if bad:
                    report.append((ln, body, sorted(bad)))
            if not nodef and ops:
                d |= regs(ops[0])
                if mn.startswith("v_mad_u64_u32") and len(ops) > 1:
                    d |= regs(ops[1])
        return d

    ALL = None
    inn = {b: ALL for b in blocks}
    inn["entry"] = {("v", 0), ("v", 1), ("v", 2)}  # work-item ids
    work = ["entry"]
    out = {}
    while work:
        b = work.pop()
        o = transfer(b, inn[b])
        if out.get(b) == o:
            continue
        out[b] = o
        for s in succ[b]:
            if s not in blocks:
                continue
            new = o if inn[s] is ALL else (inn[s] & o)
            if inn[s] is ALL or new != inn[s]:
                inn[s] = new
                work.append(s)
    rep = []
    for b in order:
        if inn[b] is not ALL:
            transfer(b, inn[b], rep)
    print(f"{kern}: {len(rep)} instruction(s) read a vector register that is not written on every path")
    for ln, body, bad in rep[:200]:
        print(f"  line {ln}: {body}    <- {' '.join(k + str(n) for k, n in bad)}")


if __name__ == "__main__":
    main()
