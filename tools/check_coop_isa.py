#!/usr/bin/env python
"""Checks an invariant of k_coop the source cannot express: between an `ll_peek` (inline-asm global load) and the
`s_waitcnt` that follows, NOTHING may read the load's destination registers -- the compiler does not know the load is
still in flight, so a spill (scratch_store, or on gfx950 more likely a copy into an AGPR: v_accvgpr_write), a v_mov / phi
copy or an LDS store placed there would move a register whose data has not arrived.  Every instruction between a poll
load and its wait that names one of the load's destination VGPRs is a violation, and so is any scratch access there
(code under more register pressure uses ll_peek_wait*, loads + wait as one statement).  Compiles csrc/engine.hip to gfx950 assembly (device only, ~30 s) and inspects the main loop.

    python tools/check_coop_isa.py            # prints one line per instantiation, exit 1 on violation
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs_named(text):
    """the VGPR numbers an instruction's operands name: v7, v[12:15]"""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out |= set(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        out.add(int(a))
    return out


def poll_hazards(body):
    """Forward data flow over the loop's basic blocks (the poll loop is branchy: lanes reload only the entries they still
    miss, so text order is not execution order): `pending` = destination VGPRs of poll loads that may still be in flight
    at a point (union over the predecessors); a poll load adds its destination, `s_waitcnt vmcnt(0)` clears the set; any
    other instruction that names a pending register -- or touches scratch while something is pending -- is counted."""
    blocks, cur, name = {}, [], "entry"
    order = ["entry"]
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks[name] = cur
            name, cur = m.group(1), []
            order.append(name)
            continue
        c = l.split(";")[0].strip()
        if c and not c.startswith("."):
            cur.append(c)
    blocks[name] = cur
    succ = {}
    for k, nm in enumerate(order):
        out, fall = [], True
        for c in blocks[nm]:
            m = re.match(r"^s_cbranch\S*\s+(\.LBB\d+_\d+)", c)
            if m:
                out.append(m.group(1))
            m = re.match(r"^s_branch\s+(\.LBB\d+_\d+)", c)
            if m:
                out.append(m.group(1))
                fall = False
        if fall and k + 1 < len(order):
            out.append(order[k + 1])
        succ[nm] = [t for t in out if t in blocks]
    pend_in = {nm: set() for nm in order}
    flagged = set()

    def run(nm, pending, record):
        for i, c in enumerate(blocks[nm]):
            if "global_load_dwordx4" in c and "sc1" in c:
                m = re.search(r"global_load_dwordx4\s+v\[(\d+):(\d+)\]", c)
                if not m or (regs_named(c.split(",", 1)[1]) & pending):
                    if record:
                        flagged.add((nm, i))
                if m:
                    pending = pending | set(range(int(m.group(1)), int(m.group(2)) + 1))
            elif "s_waitcnt" in c and "vmcnt(0)" in c:
                pending = set()
            elif pending and ("scratch_" in c or regs_named(c) & pending):
                if record:
                    flagged.add((nm, i))
        return pending

    changed = True
    while changed:
        changed = False
        for nm in order:
            outp = run(nm, set(pend_in[nm]), False)
            for t in succ[nm]:
                if not outp <= pend_in[t]:
                    pend_in[t] |= outp
                    changed = True
    for nm in order:
        run(nm, set(pend_in[nm]), True)
    return len(flagged)


def loops(asm):
    out = []
    L = asm.split("\n")
    # the kernels of the launch-per-relaxation form with the test inside the grid, and the functions that hold the exchange
    # grid of the launches with testers: coop_grid_one (k_coop calls it), coop_grid_run (per node of the resident run)
    starts = [i for i, l in enumerate(L) if re.match(r"^_ZN\S*6k_coopILi\d+ELi\d+ELi\d+ELi\d+ELb0EEE\S*:", l)
              or re.match(r"^_ZN\S*13coop_grid_(run|one)ILi\d+ELi\d+ELi\d+ELi\d+EEE\S*:", l)]
    for s in starts:
        name = L[s].split(":")[0]
        e = next(i for i in range(s, len(L)) if L[i].startswith(".Lfunc_end"))
        F = L[s:e]
        hdrs = [i for i, l in enumerate(F) if "Loop Header: Depth=1" in l]
        fma = [i for i, l in enumerate(F) if "v_fmac_f64" in l or "v_fma_f64" in l]
        # the main loop: the depth-1 loop whose first 400 lines hold the most fp64 FMAs (the row sums)
        h = max(hdrs, key=lambda h: sum(1 for i in fma if h < i < h + 400))
        sleeps = [i for i in range(h, len(F)) if "s_sleep" in F[i]]
        # from the loop header to the end of the poll loop that follows the nap
        end = sleeps[1] + 60 if len(sleeps) > 1 else sleeps[0] + 300
        body = F[h:end]
        # THE invariant: between a poll load (global_load_dwordx4 ... sc1, inline asm) and the s_waitcnt vmcnt(0) that
        # follows it, no scratch access -- a spill there would store a register whose data has not arrived
        hazard = poll_hazards(body)
        # the function's own vector registers (".set <name>.num_vgpr, max(N, <callees>)" behind its end: N) and the loop's
        # INSTRUCTIONS (labels, directives and comments are not counted): a change that looks neutral in the source shows here
        # first -- r06: a constant read inside the loop instead of before it was +20 instructions and 0.1 us per iteration
        vg = -1
        for l in L[e:e + 40]:
            m = re.match(r"^\s*\.set\s+\S*" + re.escape(name.lstrip("_")) + r"\.num_vgpr,\s*(?:max\()?(\d+)", l) or \
                re.match(r"^\s*\.set\s+\S*\.num_vgpr,\s*(?:max\()?(\d+)", l)
            if m:
                vg = int(m.group(1))
                break
        instr = sum(1 for l in body if l.split(";")[0].strip() and not l.split(";")[0].strip().startswith(".")
                    and not l.split(";")[0].strip().endswith(":"))
        out.append((name, len(body), sum("scratch_" in l for l in body), sum("v_readlane" in l for l in body), hazard, vg, instr))
    return out


def main():
    src = os.path.join(ROOT, "miosqp_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "engine.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S",
                               "--cuda-device-only", "-w", "engine.hip", "-o", asm], cwd=src)
        res = loops(open(asm).read())
    bad = 0
    for name, lines, scratch, readlane, hazard, vg, instr in res:
        print("%s: exchange loop %d lines, scratch accesses %d, v_readlane %d, instructions between a poll and its wait that touch its registers (or scratch) %d, "
              "vector registers of the function %d, instructions in the loop %d" % (name, lines, scratch, readlane, hazard, vg, instr))
        bad += hazard
        # (scratch accesses elsewhere in the loop cost time, not correctness: reported, and kept at zero for the layout of the
        #  headline -- 3 columns per thread, testers -- by tests/test_abi.py)
    if not res:
        print("no k_coop instantiation found")
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
