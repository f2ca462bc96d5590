#!/usr/bin/env python3
"""Static scan of a built libemei_hip.so (or a variant build) for the hipcc (ROCm 7.2) register-allocator bug that
DESIGN.md §6 documents: VGPR spill stores placed at the top of a control-flow JOIN block, in front of the
`s_or_b64 exec, exec, sX` that re-enables the lanes.  The stores then execute under the reduced EXEC of the branch that
was just left (EXEC = 0 when the branch was skipped, or the single lane of a wave-reduced atomic), write nothing for the
other lanes, and the later reload returns stale registers.

Two shapes have been seen:
  (a) round 1 / 2, the rolled cheetah RK4:      s_and_saveexec sX ; s_cbranch_execnz COLD ; JOIN: <spill stores> ; s_or_b64 exec, exec, sX
  (b) round 3, the -DEMEI_NEWTON_STATS Hopper RK4 (the statistics build: every lane but one per wave went non-finite):
      s_and_saveexec sX ; s_cbranch_execz JOIN ; <then block: global_atomic_add> ; JOIN: <spill stores> ; s_or_b64 exec, exec, sX

Round 4 added a second scan, lds_inflight_hazards: the destination of an LDS read touched before the read has retired
(an asm LDS read is invisible to hipcc's register allocator and wait-count insertion).

Usage: tools/isa_scan.py <library.so> [name filter]      exit status 1 if any kernel is flagged.
tests/test_isa_guards.py runs the same functions over the shipped library; tools/build_variant.sh over every variant build.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
_INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
VECTOR = ("v_", "ds_", "global_", "scratch_", "buffer_", "flat_")
SPILL_STORE = ("v_accvgpr_write_b32", "scratch_store")


def disassemble(lib, workdir):
    """{mangled name: [(addr, mnemonic, operands)]} over every gfx950 code object bundled in the library."""
    lib = shutil.copy(lib, workdir)
    subprocess.check_call([OBJDUMP, "--offloading", lib], stdout=subprocess.DEVNULL, cwd=workdir)
    funcs = {}
    for f in sorted(os.listdir(workdir)):
        if "gfx950" not in f:
            continue
        txt = subprocess.check_output([OBJDUMP, "-d", os.path.join(workdir, f)], text=True)
        cur = None
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = funcs.setdefault(m.group(1), [])
                continue
            m = _INS.match(line)
            if m and cur is not None:
                cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return funcs


def branch_target(a, m, o):
    if m.startswith("s_cbranch") or m == "s_branch":
        off = int(o.split()[0])
        off = off - 65536 if off >= 32768 else off
        return a + 4 + 4 * off
    return None


def _is_exec_restore(m, o):
    return m == "s_or_b64" and re.match(r"exec, exec, s\[\d+:\d+\]", o) is not None


def _spill_slot(m, o):
    """slot a spill store writes: ('a', n) for v_accvgpr_write_b32 aN, vM (a VGPR source: `aN, 0` is an initialisation, not
    a spill), ('s', offset text) for scratch_store_*; None otherwise"""
    if m == "v_accvgpr_write_b32":
        g = re.fullmatch(r"a(\d+), v\d+", o.strip())
        return ("a", int(g.group(1))) if g else None
    if m.startswith("scratch_store"):
        g = re.search(r"offset:(\d+)", o)
        return ("s", (o.split(",")[0].strip(), int(g.group(1)) if g else 0))
    return None


def exec_restore_hazards(ins, warnings=None):
    """[(addr of the restore, [offending instructions])] for both shapes of the module docstring.

    Shape (a): ANY vector instruction between a fall-through `s_cbranch_execnz` (fed by the saveexec whose mask the restore
    consumes) and the restore.
    Shape (b): a SPILL STORE (v_accvgpr_write_b32 aN, vM / scratch_store_*) between the start of a block some branch lands on
    and the restore at its top, into a slot that is stored NOWHERE ELSE in the function: every reload of that slot then reads
    what the reduced EXEC let through.  (A slot that also has an unmasked store elsewhere is hipcc re-storing a split live
    range whose value the slot already holds — seen in the shipped Hopper / cheetah kernels behind their cold trigonometry
    paths, verified harmless by the long-horizon parity tests; such sites go to `warnings` when a list is passed.)"""
    targets = {t for t in (branch_target(*i) for i in ins) if t is not None}
    out, cand = [], []
    for k, (a, m, o) in enumerate(ins):
        if not _is_exec_restore(m, o):
            continue
        saved = o.split(",")[2].strip()
        j, vec, hit_b = k - 1, [], None
        while j >= 0 and not ins[j][1].startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")) and "exec" not in ins[j][2].split(",")[0]:
            if ins[j][1].startswith(VECTOR) and ins[j][1] not in ("v_readlane_b32", "v_writelane_b32"):  # SGPR spill traffic ignores EXEC
                vec.append((ins[j][0], ins[j][1], ins[j][2]))
            if ins[j][0] in targets and hit_b is None:  # a join: everything collected so far sits between the label and the restore
                spills = [v for v in vec if _spill_slot(v[1], v[2]) is not None]
                if spills:
                    hit_b = spills[::-1]
            j -= 1
        if hit_b:
            cand.append((a, hit_b))
            continue
        if not vec or j < 1 or ins[j][1] != "s_cbranch_execnz":
            continue
        # shape (a): the branch is fed by the saveexec that produced the mask this restore consumes
        for q in range(j - 1, max(j - 4, -1), -1):
            if ins[q][1] == "s_and_saveexec_b64" and ins[q][2].split(",")[0].strip() == saved:
                out.append((hex(a), [(hex(x[0]), x[1], x[2]) for x in vec[::-1]]))
                break
    if cand:
        masked = {x[0] for _, sp in cand for x in sp}
        unmasked_slots = {_spill_slot(m, o) for a, m, o in ins if a not in masked and _spill_slot(m, o) is not None}
        for a, sp in cand:
            only_masked = [(hex(x[0]), x[1], x[2]) for x in sp if _spill_slot(x[1], x[2]) not in unmasked_slots]
            if only_masked:
                out.append((hex(a), only_masked))
            elif warnings is not None:
                warnings.append((hex(a), [(hex(x[0]), x[1], x[2]) for x in sp]))
    return out


_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_NO_VGPR_DEST = ("global_store", "scratch_store", "buffer_store", "flat_store", "ds_write", "ds_add", "ds_max", "ds_min", "v_cmp", "v_nop",
                 "v_readlane", "v_readfirstlane", "global_atomic", "ds_swizzle_nop", "v_cmpx")


def _vregs(text):
    out = set()
    for g in _VREG.finditer(text):
        if g.group(1) is not None:
            out.add(int(g.group(1)))
        else:
            out.update(range(int(g.group(2)), int(g.group(3)) + 1))
    return out


def _dest_and_sources(m, o):
    """(VGPRs an instruction writes, VGPRs it reads) — good enough for the LDS-in-flight scan below: the first operand is
    the destination unless the mnemonic stores / compares / reads a lane into a scalar; LDS-DMA loads have no VGPR destination."""
    if not m.startswith(VECTOR):
        return set(), set()
    ops = [x.strip() for x in o.split(",")]
    if m.startswith(_NO_VGPR_DEST) or "_lds_" in m:
        return set(), _vregs(o)
    return _vregs(ops[0]) if ops else set(), _vregs(", ".join(ops[1:]))


def lds_inflight_hazards(ins, max_walk=400):
    """[(addr of the LDS read, addr of the offender, mnemonic, 'write' | 'read')]: an instruction that touches the destination
    registers of a ds_read before a wait that is guaranteed to have retired it.

    hipcc keeps this invariant for its own LDS loads.  An `asm volatile("ds_read_b128 ...")` (emei_device.h:
    sincos_begin_ctx) is invisible to it: it takes the output for available at once and, when the value is dead on some
    path, for free — round 4: the cold large-angle branch got the in-flight read's registers as scratch, and the LDS then
    delivered into the middle of the angle reduction (CartPole rewards wrong by O(1), differently from run to run).
    A wait `lgkmcnt(N)` retires the read when N <= the number of LDS operations issued after it (LDS returns in order);
    scalar loads share the counter and return out of order, so a path with one in between only counts lgkmcnt(0)."""
    index = {a: k for k, (a, _, _) in enumerate(ins)}
    out = []
    for k, (a, m, o) in enumerate(ins):
        if not m.startswith("ds_read"):
            continue
        dest = _vregs(o.split(",")[0])
        stack, seen = [(k + 1, 0, False, 0)], set()
        while stack:
            j, younger, smem, walked = stack.pop()
            while j < len(ins) and walked < max_walk and j not in seen:
                seen.add(j)
                aj, mj, oj = ins[j]
                if mj == "s_waitcnt":
                    g = re.search(r"lgkmcnt\((\d+)\)", oj)
                    if g and (int(g.group(1)) == 0 or (not smem and int(g.group(1)) <= younger)):
                        break
                w, r = _dest_and_sources(mj, oj)
                if w & dest or r & dest:
                    out.append((hex(a), hex(aj), mj, "write" if w & dest else "read"))
                    break
                if mj.startswith("ds_"):
                    younger += 1
                if mj.startswith(("s_load", "s_buffer_load")):
                    smem = True
                if mj in ("s_endpgm", "s_setpc_b64"):
                    break
                t = branch_target(aj, mj, oj)
                if t is not None and t in index:
                    stack.append((index[t], younger, smem, walked))
                    if mj == "s_branch":
                        break
                j += 1
                walked += 1
    return out


def scan_library(lib, name_filter=""):
    """-> (functions scanned, {name: hazards}, {name: re-store sites that are not failures})"""
    with tempfile.TemporaryDirectory() as d:
        funcs = disassemble(lib, d)
    bad, warn = {}, {}
    for name, ins in funcs.items():
        if name_filter in name:
            w = []
            h = exec_restore_hazards(ins, w)
            h += [(rd, [(at, mn, kind + " of the destination of an LDS read still in flight")]) for rd, at, mn, kind in lds_inflight_hazards(ins)]
            if h:
                bad[name] = h
            if w:
                warn[name] = w
    return len(funcs), bad, warn


if __name__ == "__main__":
    n, bad, warn = scan_library(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
    for name, h in bad.items():
        print(f"HAZARD {name}: {len(h)} site(s): EXEC-masked instructions in front of an exec restore / LDS reads touched in flight")
        for addr, vec in h[:3]:
            print(f"   at {addr}: " + "; ".join(f"{x[0]} {x[1]} {x[2]}" for x in vec[:6]))
    for name, w in warn.items():
        print(f"note {name}: {len(w)} join(s) re-store a slot that also has an unmasked store ({w[0][0]}: {len(w[0][1])} stores)")
    print(f"{sys.argv[1]}: {n} functions scanned, {len(bad)} flagged, {len(warn)} with re-store notes")
    sys.exit(1 if bad else 0)
