"""Convert the reference's hard-instance inputs to one small data fixture.

Run in the build container only:  python tests/golden/make_maxiter_fixture.py

/root/reference/max_iter_examples/*.pickle are 49 relaxations on which the reference authors'
OSQP hit max_iter (dump code at /root/reference/miosqp/solver.py:93-109, loader
/root/reference/extra/run_maxiter_problem.py).  They hold INPUTS only (P, q, A, l, u, i_idx,
settings) -- no solutions.  Stored here as plain arrays in tests/golden/maxiter_inputs.npz; the 49 nodes come from 7
distinct (P, A, q), each stored once (P_g<k>, A_g<k>, q_g<k>; group_<name> maps a node to it).
"""
import glob
import json
import os
import pickle

import numpy as np
import scipy.sparse as spa

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    d = {}
    names = []
    groups = {}   # (P, A, q) shared by several node instances -> stored once
    for path in sorted(glob.glob("/root/reference/max_iter_examples/*.pickle"),
                       key=lambda p: int(os.path.basename(p)[:-7])):
        k = os.path.basename(path)[:-7]
        with open(path, "rb") as f:
            p = pickle.load(f, encoding="latin1")
        P = spa.csc_matrix(p["P"]); A = spa.csc_matrix(p["A"])
        q = np.asarray(p["q"], dtype=float).ravel()
        key = P.toarray().tobytes() + A.toarray().tobytes() + q.tobytes()
        if key not in groups:
            g = str(len(groups))
            groups[key] = g
            d["P_g" + g] = P.toarray()
            d["A_g" + g] = A.toarray()
            d["q_g" + g] = q
        d["group_" + k] = int(groups[key])
        for v in ("l", "u"):
            d[v + "_" + k] = np.asarray(p[v], dtype=float).ravel()
        d["i_idx_" + k] = np.asarray(p["i_idx"], dtype=np.int64)
        d["settings_" + k] = json.dumps({a: (bool(b) if isinstance(b, (bool, np.bool_)) else b)
                                         for a, b in p["settings"].items()})
        names.append(k)
    d["names"] = np.array(names)
    out = os.path.join(HERE, "maxiter_inputs.npz")
    np.savez_compressed(out, **d)
    print(len(names), "instances ->", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
