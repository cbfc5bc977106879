"""Golden fixture for BASELINE config 4: power_converter MPC, horizon N=3 (n=18).

Run in the build container only:  python tests/golden/make_power_converter.py [--rho-auto]

Imports the REFERENCE's example package from /root/reference (with stand-in modules only for the
absent third-party imports it never uses on this path: `osqp` -> the CPU oracle with the osqp
surface, `mathprogbasepy` -> empty module) and replays the first steps of its closed-loop
simulation (/root/reference/examples/power_converter/power_converter.py:589-675 `simulate_cl`,
421-508 `compute_mpc_input`) with the parameters of run_example.py:25-77 (tail cost
delta_550.mat).  Stored in tests/golden/power_converter_N3.npz (data only):

  the MIQP matrices built by the reference (quadratic_program.py:11-136): P, A, l, i_idx, i_l, i_u
  per MPC step k: q_k, u_k (the vectors passed to MIOSQP.update_vectors), the shifted warm
  start passed to set_x0, and what the reference's tree search returned with the oracle
  underneath: x, upper_glob, status, number of nodes, total ADMM iterations.
"""
import json
import os
import sys
import types

import numpy as np
import scipy.sparse as spa

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402

AUTO = "--rho-auto" in sys.argv  # second fixture: rho chosen once at set-up, *_rhoauto.npz


class _AutoRho(oracle.OSQP):
    """the oracle with rho = "auto" whatever the caller passes: the reference's example hands only eps_* to osqp.setup"""

    def setup(self, *a, **kw):
        kw["rho"] = "auto"
        return oracle.OSQP.setup(self, *a, **kw)


shim = types.ModuleType("osqp")
shim.OSQP = _AutoRho if AUTO else oracle.OSQP
shim.constant = oracle.constant
sys.modules["osqp"] = shim
sys.modules["mathprogbasepy"] = types.ModuleType("mathprogbasepy")
import matplotlib  # noqa: E402
matplotlib.use("Agg")
sys.path.insert(0, "/root/reference")
import miosqp as ref  # noqa: E402
from examples.power_converter.power_converter import Model  # noqa: E402
from examples.power_converter.quadratic_program import MIQP  # noqa: E402

STEPS = 40
N = 3


def main():
    cwd = os.getcwd()
    os.chdir("/root/reference")  # the example loads its tail cost by relative path
    try:
        model = Model()
        model.set_params(25.0e-06, 50., 0.8e03, 0.8e03, 1.)
        model.set_time(0.0, 1, 2)
        model.set_initial_conditions()
        model.gen_dynamical_system(300, 5.5)
        model.gen_tail_cost(50, 0.95, name='delta_550.mat')
    finally:
        os.chdir(cwd)
    qp = MIQP(model.dyn_system, N, model.tail_cost)
    model.qp_matrices = qp
    settings = {'eps_int_feas': 1e-02, 'max_iter_bb': 2000, 'tree_explor_rule': 1, 'branching_rule': 0,
                'verbose': False, 'print_interval': 1}
    qp_settings = {'eps_abs': 1e-03, 'eps_rel': 1e-03, 'eps_prim_inf': 1e-04, 'verbose': False}
    nu = model.dyn_system.B.shape[1]
    x = np.asarray(model.init_conditions.x0, dtype=float).ravel()
    u_prev = np.zeros(nu * N)
    P = spa.csc_matrix(qp.P); A = spa.csc_matrix(qp.A)
    P.sort_indices(); A.sort_indices()
    l0 = np.array(qp.l, dtype=float).copy()
    solver = None
    rec = dict(q=[], u=[], x0=[], x=[], upper=[], status=[], nodes=[], osqp_iter=[])
    for k in range(STEPS):
        q = 2. * (qp.q_x.dot(x) + qp.q_u)
        qp.u[:6 * N] = qp.SA_tilde.dot(x)
        if solver is None:
            solver = ref.MIOSQP()
            solver.setup(qp.P, q, qp.A, qp.l, qp.u, qp.i_idx, qp.i_l, qp.i_u, settings, qp_settings)
        else:
            solver.update_vectors(q, qp.l, qp.u)
        rec["q"].append(np.array(q, dtype=float).ravel().copy())
        rec["u"].append(np.array(qp.u, dtype=float).ravel().copy())
        rec["x0"].append(u_prev.copy())
        solver.set_x0(u_prev)
        res = solver.solve()
        rec["x"].append(np.array(res.x, dtype=float).copy())
        rec["upper"].append(res.upper_glob)
        rec["status"].append(res.status)
        rec["nodes"].append(solver.work.iter_num - 1)
        rec["osqp_iter"].append(solver.work.osqp_iter)
        u_mpc = res.x
        x = np.asarray(model.dyn_system.A.dot(x) + model.dyn_system.B.dot(u_mpc[:6])).ravel()
        u_prev = np.append(u_mpc[nu:], u_mpc[-nu:])
    if AUTO:
        qp_settings = dict(qp_settings, rho="auto")
    out = os.path.join(HERE, "power_converter_N3_rhoauto.npz" if AUTO else "power_converter_N3.npz")
    np.savez_compressed(
        out, P_indptr=P.indptr, P_indices=P.indices, P_data=P.data, P_shape=P.shape,
        A_indptr=A.indptr, A_indices=A.indices, A_data=A.data, A_shape=A.shape, l=l0,
        i_idx=np.asarray(qp.i_idx), i_l=np.asarray(qp.i_l, dtype=float), i_u=np.asarray(qp.i_u, dtype=float),
        settings=json.dumps(settings), qp_settings=json.dumps(qp_settings),
        q=np.array(rec["q"]), u=np.array(rec["u"]), x0=np.array(rec["x0"]), x=np.array(rec["x"]),
        upper=np.array(rec["upper"]), status=np.array(rec["status"]), nodes=np.array(rec["nodes"]),
        osqp_iter=np.array(rec["osqp_iter"]))
    print("n=%d rows=%d steps=%d nodes/step: min %d mean %.1f max %d; statuses %s" % (
        A.shape[1], A.shape[0], STEPS, min(rec["nodes"]), np.mean(rec["nodes"]), max(rec["nodes"]),
        sorted(set(rec["status"]))))
    print("l has -inf:", np.isinf(l0).sum(), " P eig range:", np.linalg.eigvalsh(P.toarray())[[0, -1]])


if __name__ == "__main__":
    main()
