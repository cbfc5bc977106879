"""Golden fixture for the power-converter benchmark harness: one settling period + one
measured period of the reference's closed-loop simulation at horizon N=3.

Run in the build container only:  python tests/golden/make_power_converter_long.py [--rho-auto]

Imports the REFERENCE's example package from /root/reference (stand-in modules only for third-party imports
absent here that this path never exercises: `osqp` -> the CPU oracle with the osqp surface, `mathprogbasepy`,
`tqdm`) and replays its closed loop exactly as simulate_cl does (/root/reference/examples/power_converter/
power_converter.py:589-675): compute_mpc_input (421-508) -> simulate_one_step (510-517) -> shifted warm start.
Stored in tests/golden/power_converter_N3_long.npz (data only):

  the MIQP matrices; per step the vectors handed to MIOSQP (q, u, x0) and what came back (x, nodes, ADMM
  iterations); the applied inputs U, the phase currents Y_phase and the time axis the reference's statistics
  are computed from; and the statistics the REFERENCE's own code computes from them (get_statistics,
  549-587; utils.compute_on_transitions / get_thd, utils.py:80-215): switching frequency and THD.
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
tq = types.ModuleType("tqdm")
tq.tqdm = lambda it, *a, **k: it
sys.modules["tqdm"] = tq
import matplotlib  # noqa: E402
matplotlib.use("Agg")
sys.path.insert(0, "/root/reference")
import miosqp as ref  # noqa: E402
from examples.power_converter.power_converter import Model, SimulationResults  # noqa: E402
from examples.power_converter.quadratic_program import MIQP  # noqa: E402

N = 3


def main():
    cwd = os.getcwd()
    os.chdir("/root/reference")  # the example loads its tail cost by relative path
    try:
        model = Model()
        model.set_params(25.0e-06, 50., 0.8e03, 0.8e03, 1.)
        model.set_time(0.0, 1, 1)  # one settling period, one measured period (run_example.py uses 1 + 2)
        model.set_initial_conditions()
        model.gen_dynamical_system(300, 5.5)
        model.gen_tail_cost(50, 0.95, name='delta_550.mat')
    finally:
        os.chdir(cwd)
    qp = MIQP(model.dyn_system, N, model.tail_cost)
    model.qp_matrices = qp
    model.solver = None
    nx = model.dyn_system.A.shape[0]
    nu = model.dyn_system.B.shape[1]
    T_final, T_timing = model.time.T_final, model.time.T_timing
    X = np.zeros((nx, T_final + 1))
    U = np.zeros((nu, T_final))
    X[:, 0] = model.init_conditions.x0
    u_prev = np.zeros(nu * N)
    P = spa.csc_matrix(qp.P); A = spa.csc_matrix(qp.A)
    P.sort_indices(); A.sort_indices()
    l0 = np.array(qp.l, dtype=float).copy()
    rec = dict(q=[], u=[], x0=[], x=[], nodes=[], osqp_iter=[], upper=[])
    solve_times = np.zeros(T_timing)
    settings = qp_settings = None
    for i in range(T_final):
        rec["x0"].append(np.array(u_prev, dtype=float).copy())
        u0, obj, t_solve, u_full, _, _ = model.compute_mpc_input(X[:, i], u_prev, solver='miosqp')
        w = model.solver.work
        if settings is None:
            settings, qp_settings = dict(w.settings), dict(w.qp_settings)
        rec["q"].append(np.array(w.data.q, dtype=float).ravel().copy())
        rec["u"].append(np.array(qp.u, dtype=float).ravel().copy())
        rec["x"].append(np.array(u_full, dtype=float).copy())
        rec["nodes"].append(w.iter_num - 1)
        rec["osqp_iter"].append(w.osqp_iter)
        rec["upper"].append(obj)
        U[:, i] = u0
        if i >= model.time.init_periods * model.time.Nstpp:
            solve_times[i - model.time.init_periods * model.time.Nstpp] = t_solve
        X[:, i + 1], _ = model.simulate_one_step(X[:, i], U[:, i])
        u_prev = np.append(u_full[nu:], u_full[-nu:])
    Y_phase, Y_star_phase, T_e, T_e_des = model.compute_signals(X)
    stats = model.get_statistics(SimulationResults(X, U, Y_phase, Y_star_phase, T_e, T_e_des, solve_times))
    if AUTO:
        qp_settings = dict(qp_settings, rho="auto")
    out = os.path.join(HERE, "power_converter_N3_long_rhoauto.npz" if AUTO else "power_converter_N3_long.npz")
    np.savez_compressed(
        out, P_indptr=P.indptr, P_indices=P.indices, P_data=P.data, P_shape=P.shape,
        A_indptr=A.indptr, A_indices=A.indices, A_data=A.data, A_shape=A.shape, l=l0,
        i_idx=np.asarray(qp.i_idx), i_l=np.asarray(qp.i_l, dtype=float), i_u=np.asarray(qp.i_u, dtype=float),
        settings=json.dumps(settings), qp_settings=json.dumps(qp_settings),
        q=np.array(rec["q"]), u=np.array(rec["u"]), x0=np.array(rec["x0"]), x=np.array(rec["x"]),
        upper=np.array(rec["upper"]), status=np.array(["Solved"] * T_final),
        nodes=np.array(rec["nodes"]), osqp_iter=np.array(rec["osqp_iter"]),
        U=U, Y_phase=Y_phase, t=model.time.t, init_periods=model.time.init_periods, sim_periods=model.time.sim_periods,
        Nstpp=model.params.Nstpp, freq=model.params.freq, fsw=stats.fsw, thd=stats.thd)
    print("steps %d, nodes/step mean %.2f, fsw %.3f Hz, THD %.4f %%, file %.1f KB" % (
        T_final, np.mean(rec["nodes"]), stats.fsw, stats.thd, os.path.getsize(out) / 1024.0))


if __name__ == "__main__":
    main()
