#!/usr/bin/env python
"""Writes miosqp_amd/coop_nap_table.txt: the calibrated poll delay of the cooperative solver for a grid of
problem sizes, measured on the GPU this runs on (run on the MI355X box WITHOUT the shipped table in the way: gpurun -- 'mv miosqp_amd/coop_nap_table.txt /tmp/; python
tools/make_nap_table.py gpurun_out/coop_nap_table.txt', then copy the file next to the library).  Values only affect speed."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out = sys.argv[1]
    cache = tempfile.mkdtemp()
    os.environ["MIOSQP_CACHE_DIR"] = cache
    import numpy as np
    from miosqp_amd import bnb, problems, qp, search
    # N = n + M with M = m + p; T = ceil(N / 8); cover T = 8 .. 256 in steps of <= 16
    for N in list(range(64, 1025, 128)) + list(range(1088, 2049, 128)) + [160, 1750, 2040]:
        n = max(8, N // 4)
        p = n // 2
        m = N - n - p
        pr = problems.random_miqp(n, m, p, density=0.5, seed=1)
        A, l, u = problems.extended(pr)
        g = qp.OSQP()
        g.setup(pr["P"], pr["q"], A, l, u, coop=1, resident=0, **problems.QP_SETTINGS)
        g.set_integer_rows(pr["i_idx"], m)  # (the exchange without the identity rows: its own grid, its own entry)
        fs = g.factor_stats()
        print("N %d T %d coop %s nap %d" % (N, (N + 7) // 8, fs["coop"], fs["coop_nap"]), flush=True)
        g.close()
        # the resident grid of the hosted search has an entry of its own (columns per thread + 1000), measured on the first
        # node of a search
        mdl = bnb.MIOSQP()
        mdl.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                  dict(problems.BNB_SETTINGS, device_tree=False), dict(problems.QP_SETTINGS, coop=1, resident=0))
        if mdl.work.solver.factor_stats()["coop"]:
            search.HostedSearch(mdl).step(2)
        mdl.work.solver.close()
    lines = sorted(set(open(os.path.join(cache, "coop_nap.txt")).read().splitlines()),
                   key=lambda s: [int(v) if v.isdigit() else v for v in s.split("|")])
    with open(out, "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
