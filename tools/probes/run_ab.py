"""A/B of the hosted search at config 2: cooperative grid resident over a search_run call (k_coop_run) against one
cooperative launch per node, for rho = 0.1 and rho chosen at set-up.  Prints nodes/s, ADMM iterations/s and the time per
node outside its iterations (wall time per node - iterations x back-to-back iteration time).

    python tools/probes/run_ab.py [nodes]         (MIOSQP_SEARCH_STAMPS=1 adds the device's per-node timeline on stderr)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from miosqp_amd import bnb, problems, search  # noqa: E402


def one(rho, run, nodes, seed=0, shape=None):
    os.environ["MIOSQP_COOP_RUN"] = "1" if run else "0"
    pr = problems.random_miqp(**(shape or problems.CONFIGS["cfg2"]), seed=seed)
    qs = dict(problems.QP_SETTINGS)
    if rho == "auto":
        qs["rho"] = "auto"
    mdl = bnb.MIOSQP()
    mdl.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(problems.BNB_SETTINGS), qs)
    eng = mdl.work.solver
    hs = search.HostedSearch(mdl)
    hs.step(30)
    it_us = eng.time_kernel(4, 2000)[0]
    eng.loop_stats(reset=True)
    n0, i0 = hs.nodes, hs.iters
    t0 = time.perf_counter()
    left = nodes
    while left > 0:
        before = hs.nodes
        alive = hs.step(left)
        left -= hs.nodes - before
        if alive == 0:
            break
    dt = time.perf_counter() - t0
    nn, ii = hs.nodes - n0, hs.iters - i0
    ms, _ = eng.loop_stats()
    ns = eng.node_stats()
    out = dict(rho=eng.rho() if hasattr(eng, "rho") else rho, resident=bool(run), nodes=nn, iters_per_node=round(ii / nn, 1),
               nodes_per_s=round(nn / dt, 1), iters_per_s=round(ii / dt, 1), usec_per_node=round(1e6 * dt / nn, 1),
               usec_iter_back_to_back=round(it_us, 3), usec_per_node_outside_iterations=round(1e6 * dt / nn - it_us * ii / nn, 1),
               launches=eng.loop_launches(), device_ms=round(ms, 2), wall_ms=round(1e3 * dt, 2),
               node_us_per_iter_min_med_max=[round(v, 3) for v in ns[:3]])
    eng.close()
    return out


if __name__ == "__main__":
    nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    for rho in (0.1, "auto"):
        for run in (1, 0, 1, 0):
            print(one(rho, run, nodes), flush=True)
