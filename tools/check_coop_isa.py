#!/usr/bin/env python
"""Checks an invariant of k_coop the source cannot express: between an `ll_peek` (inline-asm global load) and the
`s_waitcnt` that follows, the compiler must not spill the destination registers -- it does not know the load is
still in flight, and a spill there stores garbage.  The exchange loop of every instantiation is therefore required to
contain NO scratch access at all (code under more register pressure uses ll_peek_wait*, loads + wait as one
statement).  Compiles csrc/engine.hip to gfx950 assembly (device only, ~30 s) and inspects the main loop.

    python tools/check_coop_isa.py            # prints one line per instantiation, exit 1 on violation
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def loops(asm):
    out = []
    L = asm.split("\n")
    starts = [i for i, l in enumerate(L) if re.match(r"^_ZN\S*6k_coopILi\d+ELi\d+ELi\d+ELi\d+ELb[01]EEE\S*:", l)]
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
        hazard, open_poll = 0, False
        for l in body:
            c = l.split(";")[0]
            if "global_load_dwordx4" in c and "sc1" in c:
                open_poll = True
            elif "s_waitcnt" in c and "vmcnt(0)" in c:
                open_poll = False
            elif open_poll and "scratch_" in c:
                hazard += 1
        out.append((name, len(body), sum("scratch_" in l for l in body), sum("v_readlane" in l for l in body), hazard))
    return out


def main():
    src = os.path.join(ROOT, "miosqp_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "engine.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S",
                               "--cuda-device-only", "-w", "engine.hip", "-o", asm], cwd=src)
        res = loops(open(asm).read())
    bad = 0
    for name, lines, scratch, readlane, hazard in res:
        print("%s: exchange loop %d lines, scratch accesses %d, v_readlane %d, scratch accesses between a poll and its wait %d"
              % (name, lines, scratch, readlane, hazard))
        bad += hazard
        # (scratch accesses elsewhere in the loop cost time, not correctness: reported, and kept at zero for the layout of the
        #  headline -- 3 columns per thread, testers -- by tests/test_abi.py)
    if not res:
        print("no k_coop instantiation found")
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
