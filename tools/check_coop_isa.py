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
        out.append((name, len(body), sum("scratch_" in l for l in body), sum("v_readlane" in l for l in body)))
    return out


def main():
    src = os.path.join(ROOT, "miosqp_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "engine.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S",
                               "--cuda-device-only", "-w", "engine.hip", "-o", asm], cwd=src)
        res = loops(open(asm).read())
    bad = 0
    for name, lines, scratch, readlane in res:
        print("%s: exchange loop %d lines, scratch accesses %d, v_readlane %d" % (name, lines, scratch, readlane))
        # the instantiations with testers (...Lb1E) poll with separate loads and wait: their loop must be spill-free;
        # the others (in-grid test) issue loads and wait as one asm statement, a spill there only costs time
        if "Lb1EEE" in name:
            bad += scratch
    if not res:
        print("no k_coop instantiation found")
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
