"""Synthetic MIQP instances for tests and bench.py.

`random_miqp` follows the generator recipe of the reference's benchmark script
(/root/reference/examples/random_miqp/run_example.py:49,71-83): one `np.random.seed`, then per
instance, in this RNG order, i_idx, Pt, q, A, u, l.  `scipy.sparse.random`'s sampling is
scipy-version dependent, so instances are identified by `instance_digest` (a hash of the
generated arrays), not by the seed alone.
"""
import hashlib

import numpy as np
import scipy.sparse as spa

# BASELINE.json configs (n, m, p, density)
CONFIGS = {
    "cfg1": dict(n=50, m=100, p=10, density=0.7),
    "cfg2": dict(n=500, m=1000, p=250, density=0.7),
    "cfg5": dict(n=5000, m=10000, p=2500, density=0.01),
    # not a BASELINE config: config 5's shape scaled until the factor's dense tail (2 x 512 MB per iteration) no longer
    # fits the 256 MiB Infinity Cache -- the data point that shows HBM, not the cache, delivering the stream
    "cfg5x": dict(n=8000, m=16000, p=4000, density=0.01),
}

# settings dicts of the reference benchmark (run_example.py:98-116)
BNB_SETTINGS = {"eps_int_feas": 1e-03, "max_iter_bb": 1000, "tree_explor_rule": 1,
                "branching_rule": 0, "verbose": False, "print_interval": 1}
QP_SETTINGS = {"eps_abs": 1e-03, "eps_rel": 1e-03, "eps_prim_inf": 1e-04, "verbose": False}


def random_miqp(n, m, p, density=0.7, seed=0, reseed=True):
    """One random MIQP  min .5 x'Px + q'x  s.t. l <= Ax <= u, x[i_idx] in {0,1}."""
    if reseed:
        np.random.seed(seed)
    i_idx = np.random.choice(np.arange(0, n), p, replace=False)
    Pt = spa.random(n, n, density=density)
    P = spa.csc_matrix(Pt.dot(Pt.T))
    q = np.random.randn(n)
    A = spa.csc_matrix(spa.random(m, n, density=density))
    u = 2 + np.random.rand(m)
    l = -2 + np.random.rand(m)
    i_l = np.zeros(p)
    i_u = np.ones(p)
    return dict(P=P, q=q, A=A, l=l, u=u, i_idx=i_idx, i_l=i_l, i_u=i_u)


def instance_digest(prob):
    h = hashlib.sha256()
    for k in ("P", "A"):
        M = spa.csc_matrix(prob[k])
        M.sort_indices()
        for a in (M.indptr, M.indices, M.data):
            h.update(np.ascontiguousarray(a).tobytes())
    for k in ("q", "l", "u", "i_idx", "i_l", "i_u"):
        h.update(np.ascontiguousarray(prob[k]).tobytes())
    return h.hexdigest()[:16]


def extended(prob):
    """(A_ext, l_ext, u_ext): integer bounds appended as identity rows, the layout of
    /root/reference/miosqp/data.py:5-33 (rows m .. m+p-1 hold x[i_idx])."""
    A = spa.csc_matrix(prob["A"])
    n = A.shape[1]
    I = spa.identity(n, format="csc")[prob["i_idx"], :]
    A_ext = spa.vstack([A, I]).tocsc()
    return A_ext, np.append(prob["l"], prob["i_l"]), np.append(prob["u"], prob["i_u"])


def load_power_converter(path=None):
    """BASELINE configs[3]: the power-converter MPC sequence, horizon N=3 (n=18, 45 rows), as recorded from
    the reference's closed-loop simulation (/root/reference/examples/power_converter/power_converter.py:
    421-508, 589-675) by tests/golden/make_power_converter.py -- data only: the MIQP matrices, and per MPC
    step the vectors passed to MIOSQP.update_vectors, the warm start passed to set_x0 and what the
    reference's tree search returned."""
    import json
    import os
    if path is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                            "power_converter_N3.npz")
    z = np.load(path, allow_pickle=False)
    P = spa.csc_matrix((z["P_data"], z["P_indices"], z["P_indptr"]), shape=tuple(z["P_shape"]))
    A = spa.csc_matrix((z["A_data"], z["A_indices"], z["A_indptr"]), shape=tuple(z["A_shape"]))
    pc = dict(P=P, A=A, l=z["l"].copy(), i_idx=z["i_idx"].copy(), i_l=z["i_l"].copy(), i_u=z["i_u"].copy(),
              settings=json.loads(str(z["settings"])), qp_settings=json.loads(str(z["qp_settings"])),
              q=z["q"], u=z["u"], x0=z["x0"], x=z["x"], upper=z["upper"], status=[str(s) for s in z["status"]],
              nodes=z["nodes"], osqp_iter=z["osqp_iter"])
    for k in ("U", "Y_phase", "t", "init_periods", "sim_periods", "Nstpp", "freq", "fsw", "thd"):
        if k in z.files:  # the long fixture: closed-loop signals + the statistics the reference computed from them
            pc[k] = z[k] if z[k].ndim else z[k].item()
    return pc


def run_power_converter(pc, backend, steps=None, model=None):
    """Replays the MPC sequence exactly as the reference's compute_mpc_input drives MIOSQP
    (/root/reference/examples/power_converter/power_converter.py:467-476): setup once, then per step
    update_vectors(q, l, u) + set_x0(shifted previous solution) + solve.  Returns (per-step records, model)."""
    from miosqp_amd import bnb
    steps = len(pc["q"]) if steps is None else steps
    out = []
    l = pc["l"].copy()
    for k in range(steps):
        q, u = pc["q"][k].copy(), pc["u"][k].copy()
        if model is None:
            model = bnb.MIOSQP(backend=backend)
            model.setup(pc["P"], q, pc["A"], l, u, pc["i_idx"], pc["i_l"], pc["i_u"], pc["settings"],
                        pc["qp_settings"])
        else:
            model.update_vectors(q, l, u)
        model.set_x0(pc["x0"][k].copy())
        res = model.solve()
        out.append(dict(x=np.array(res.x, dtype=float), upper=res.upper_glob, status=res.status,
                        nodes=model.work.iter_num - 1, osqp_iter=model.work.osqp_iter, run_time=res.run_time,
                        osqp_solve_time=res.osqp_solve_time, osqp_iter_avg=res.osqp_iter_avg))
    return out, model
