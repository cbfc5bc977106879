#!/usr/bin/env python
"""The reference's random-MIQP benchmark grid on the MI355X engine.

Same problem grid, generator recipe, settings and CSV columns as
/root/reference/examples/random_miqp/run_example.py:155-216 (problem set 1: n in {10..150},
10 repeats, density 0.7, seed 0), minus the GUROBI columns (GUROBI / mathprogbasepy are not part
of this build).  Times are milliseconds like the reference's (`1e3 * run_time`);
`t_miosqp_osqp_avg` is the relaxation solver's share of the run time in percent
(run_example.py:142-143).

    python examples/random_miqp.py [--repeat 10] [--out results/random_miqp.csv]

The engine is the only relaxation solver this script knows.  (The same grid on the CPU restatement, for side-by-side
numbers, is run by tests/side_by_side.py, which hands `main` another backend module.)
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from miosqp_amd import bnb, problems  # noqa: E402

N_ARR = [10, 10, 50, 50, 100, 100, 150, 150]
M_ARR = [5, 100, 25, 200, 50, 200, 100, 300]
P_ARR = [2, 2, 5, 10, 2, 15, 5, 20]


def main(argv=None, backend=None):
    """backend: a module with the osqp surface (None: miosqp_amd.qp, the HIP engine)"""
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeat", type=int, default=10)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "results", "random_miqp.csv"))
    ap.add_argument("--concurrent", type=int, default=0,
                    help="K > 0: after the reference's one-at-a-time pass, the `repeat` instances of a shape are set up and "
                         "solved again on K host threads at once (every instance its own engine and stream: the MIQPs of a "
                         "shape have different P and A, hence different factors) and the extra column t_miosqp_conc_avg = "
                         "wall time of the pass / repeat is written: what an instance costs when the GPU (or, for the CPU "
                         "backend, the host's cores) is not left to one small problem at a time")
    ap.add_argument("--cold", action="store_true",
                    help="do not run the untimed warm-up instance first (the first engine of a process pays ~0.2 s of "
                         "one-time GPU context / code-object loading, which would land in the first grid row)")
    args = ap.parse_args(argv)
    if not args.cold:  # one-time process start-up (the CPU analogue is import time), outside every timed instance
        pr = problems.random_miqp(10, 5, 2, density=0.7, seed=12345)
        model = bnb.MIOSQP(backend=backend)
        model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                    dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
        model.solve()
    np.random.seed(args.seed)
    rows = []

    def one(pr):
        model = bnb.MIOSQP(backend=backend)
        model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                    dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
        return model.solve()

    for n, m, p in zip(N_ARR, M_ARR, P_ARR):
        t, share, iters, kept = [], [], [], []
        for _ in range(args.repeat):
            pr = problems.random_miqp(n, m, p, density=0.7, reseed=False)
            kept.append(pr)
            model = bnb.MIOSQP(backend=backend)
            model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                        dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
            res = model.solve()
            if res.status != bnb.MI_SOLVED:
                print("warning: n=%d m=%d p=%d ended with status %r" % (n, m, p, res.status))
            t.append(1e3 * res.run_time)
            share.append(100 * res.osqp_solve_time / res.run_time)
            iters.append(res.osqp_iter_avg)
        conc = float("nan")
        if args.concurrent > 0:
            import time
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(args.concurrent) as ex:
                list(ex.map(one, kept[:args.concurrent]))  # the threads' first calls (stream / buffer bundles)
                t0 = time.perf_counter()
                res_c = list(ex.map(one, kept))
                conc = 1e3 * (time.perf_counter() - t0) / len(kept)
        rows.append((n, m, p, np.mean(t), np.std(t), np.max(t), np.mean(share), np.mean(iters), conc))
        print("n=%4d m=%4d p=%3d  t_avg %9.2f ms  t_std %8.2f  t_max %9.2f  osqp share %5.1f %%  iters/node %6.1f  "
              "concurrent %8.3f ms" % rows[-1])
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        f.write("n,m,p,t_miosqp_avg,t_miosqp_std,t_miosqp_max,t_miosqp_osqp_avg,osqp_iter_avg" +
                (",t_miosqp_conc_avg" if args.concurrent > 0 else "") + "\n")
        for r in rows:
            f.write("%d,%d,%d,%.4f,%.4f,%.4f,%.2f,%.1f" % r[:8] + (",%.4f" % r[8] if args.concurrent > 0 else "") + "\n")
    print("wrote", args.out)


if __name__ == "__main__":
    main()
