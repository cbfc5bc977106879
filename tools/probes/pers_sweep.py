"""Sweeps the tuning knobs of the persistent streaming solver (environment variables read at setup) on one problem:
microseconds per iteration of the single launch, no tests.  usage: python tools/probes/pers_sweep.py [n m p dens fold]"""
import os, sys, subprocess, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1:6] if len(sys.argv) >= 6 else ["500", "1000", "250", "0.7", "1"]
code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from miosqp_amd import qp, problems
n, m, p, dens, fold = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
pr = problems.random_miqp(n, m, p, density=dens, seed=0)
A, l, u = problems.extended(pr)
g = qp.OSQP(); g.setup(pr["P"], pr["q"], A, l, u, fold=fold, coop=0, resident=0, pers=1, **problems.QP_SETTINGS)
assert g.factor_stats()["pers"]
g.warm_start(x=np.zeros(n), y=np.zeros(A.shape[0]))
us = min(g.time_kernel(4, 1000)[0] for _ in range(3))
print("%%.3f us/it" %% us)
''' % ROOT
grids = os.environ.get("SWEEP_GRID", "256,128").split(",")
naps = os.environ.get("SWEEP_NAP", "0,4,8").split(",")
segs = os.environ.get("SWEEP_SEGS", "").split(";")
dbgs = os.environ.get("SWEEP_DBG", "0").split(",")
for G, nap, sg, dbg in itertools.product(grids, naps, segs, dbgs):
    env = dict(os.environ, MIOSQP_PERS_GRID=G, MIOSQP_PERS_NAP=nap, MIOSQP_PERS_DBG=dbg)
    if sg:
        env["MIOSQP_PERS_SEGS"] = sg
    out = subprocess.run([sys.executable, "-c", code] + args, env=env, capture_output=True, text=True)
    print("grid %4s nap %3s segs %-6s dbg %s: %s" % (G, nap, sg or "auto", dbg, (out.stdout.strip() or out.stderr.strip()[-300:])), flush=True)
