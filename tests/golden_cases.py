"""Loader for tests/golden/bnb_*.npz (written by tests/golden/make_bnb_traces.py)."""
import glob
import json
import os

import numpy as np
import scipy.sparse as spa

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_names():
    return sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLDEN, "bnb_*.npz")))


def load_case(name):
    z = np.load(os.path.join(GOLDEN, "bnb_%s.npz" % name), allow_pickle=False)
    P = spa.csc_matrix((z["P_data"], z["P_indices"], z["P_indptr"]), shape=tuple(z["P_shape"]))
    A = spa.csc_matrix((z["A_data"], z["A_indices"], z["A_indptr"]), shape=tuple(z["A_shape"]))
    prob = dict(P=P, A=A, q=z["q"].copy(), l=z["l"].copy(), u=z["u"].copy(),
                i_idx=z["i_idx"].copy(), i_l=z["i_l"].copy(), i_u=z["i_u"].copy())
    c = dict(prob=prob, settings=json.loads(str(z["settings"])),
             qp_settings=json.loads(str(z["qp_settings"])),
             cols=json.loads(str(z["trace_cols"])),
             x0=z["x0"].copy() if bool(z["has_x0"]) else None, updates=[], solves=[])
    for k in range(int(z["n_updates"])):
        x0u = z["upd%d_x0" % k].copy() if bool(z["upd%d_has_x0" % k]) else None
        c["updates"].append((z["upd%d_q" % k].copy(), z["upd%d_l" % k].copy(),
                             z["upd%d_u" % k].copy(), x0u))
    for k in range(int(z["n_solves"])):
        c["solves"].append(dict(trace=z["s%d_trace" % k], x=z["s%d_x" % k],
                                upper_glob=float(z["s%d_upper_glob" % k]),
                                status=str(z["s%d_status" % k]),
                                osqp_iter=int(z["s%d_osqp_iter" % k]),
                                osqp_iter_avg=float(z["s%d_osqp_iter_avg" % k]),
                                iter_num=int(z["s%d_iter_num" % k])))
    return c


def run_case(case, backend):
    """Drive miosqp_amd's MIOSQP exactly as make_bnb_traces.py drove the reference."""
    from miosqp_amd import bnb
    prob = case["prob"]
    model = bnb.MIOSQP(backend=backend)
    model.setup(prob["P"], prob["q"], prob["A"], np.copy(prob["l"]), np.copy(prob["u"]),
                prob["i_idx"], prob["i_l"], prob["i_u"], case["settings"], case["qp_settings"])
    rows, out = [], []

    def obs(work, leaf):
        rows.append([work.iter_num, leaf.depth, leaf.status, leaf.num_iter, leaf.lower,
                     work.upper_glob, work.lower_glob, len(work.leaves),
                     -1 if leaf.constr_idx is None else leaf.constr_idx,
                     -1 if leaf.nextvar_idx is None else leaf.nextvar_idx,
                     -1 if leaf.intinf is None else leaf.intinf])

    def one():
        del rows[:]
        res = model.solve(observer=obs)
        out.append(dict(trace=np.array(rows, dtype=float).reshape(-1, len(case["cols"])),
                        x=np.array(res.x, dtype=float), upper_glob=res.upper_glob,
                        status=res.status, osqp_iter=model.work.osqp_iter,
                        osqp_iter_avg=res.osqp_iter_avg, iter_num=model.work.iter_num))

    if case["x0"] is not None:
        model.set_x0(np.copy(case["x0"]))
    one()
    for (q, l, u, x0u) in case["updates"]:
        model.update_vectors(q=q, l=l, u=u)
        if x0u is not None:
            model.set_x0(np.copy(x0u))
        one()
    return out


def load_maxiter():
    """The reference's 49 hard relaxations (inputs only): list of dicts."""
    z = np.load(os.path.join(GOLDEN, "maxiter_inputs.npz"), allow_pickle=False)
    out = []
    for k in z["names"]:
        k = str(k)
        g = str(int(z["group_" + k]))
        out.append(dict(name=k, P=spa.csc_matrix(z["P_g" + g]), A=spa.csc_matrix(z["A_g" + g]),
                        q=z["q_g" + g].copy(), l=z["l_" + k].copy(), u=z["u_" + k].copy(),
                        i_idx=z["i_idx_" + k].copy(),
                        settings=json.loads(str(z["settings_" + k]))))
    return out


def load_power_converter(name="power_converter_N3.npz"):
    from miosqp_amd import problems
    return problems.load_power_converter(os.path.join(GOLDEN, name))


def run_power_converter(pc, backend, steps=None):
    """Replays the MPC sequence exactly as the reference's compute_mpc_input drives MIOSQP
    (/root/reference/examples/power_converter/power_converter.py:467-476)."""
    from miosqp_amd import problems
    return problems.run_power_converter(pc, backend, steps)[0]
