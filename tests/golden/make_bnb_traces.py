"""Generate golden branch-and-bound traces by running the REFERENCE's own tree search.

Run in the build container only (needs /root/reference):

    python tests/golden/make_bnb_traces.py              # the frozen default rho = 0.1:   bnb_<case>.npz
    python tests/golden/make_bnb_traces.py --rho-auto   # rho chosen once at set-up:      bnb_rhoauto_<case>.npz

(second set, round 5: `rho="auto"` -- OSQP's own update rule applied once at set-up, then frozen, DESIGN.md sec. 1 -- is
what comes closest to the reference's actual settings, which leave rho to OSQP's adaptive default
(/root/reference/miosqp/workspace.py:67-68); the same cases searched by the reference's tree search with that setting in
the shim, so that the mode is pinned count for count like the default.)

The reference's B&B layer (/root/reference/miosqp/*.py) imports a module called `osqp`
(node.py:2, workspace.py:6) that is not installed here.  A shim module with the same surface
(`OSQP.setup/update/warm_start/solve`, `constant`) backed by the CPU oracle is placed in
sys.modules, the reference is imported from /root/reference, and `MIOSQP.setup/solve/
update_vectors/set_x0` are run on small instances.  Every node visit is recorded through a
wrapper around `Workspace.bound_and_branch`.  Outputs (data only -- inputs and expected
outputs; no reference source is stored):

    tests/golden/bnb_<case>.npz      instance arrays + settings + per-node trace + final result

What these fixtures pin: the host control flow in miosqp_amd/bnb.py (tree order, pruning,
incumbent updates, statistics, statuses) given identical relaxation results.  They do NOT pin
the relaxation arithmetic itself (see oracle/qp_oracle.c header: "parity unpinned").
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
from miosqp_amd import problems  # noqa: E402

shim = types.ModuleType("osqp")
shim.OSQP = oracle.OSQP
shim.constant = oracle.constant
sys.modules["osqp"] = shim
sys.path.insert(0, "/root/reference")
import miosqp as ref  # noqa: E402

TRACE_COLS = ("iter_num", "depth", "status", "num_iter", "lower", "upper_glob", "lower_glob",
              "n_leaves", "constr_idx", "nextvar_idx", "intinf")


def run_reference(prob, settings, qp_settings, x0=None, updates=()):
    """Returns (list of per-solve dicts).  `updates` is a sequence of (q, l, u, x0) re-solves."""
    model = ref.MIOSQP()
    model.setup(prob["P"], prob["q"], prob["A"], np.copy(prob["l"]), np.copy(prob["u"]),
                prob["i_idx"], prob["i_l"], prob["i_u"], settings, qp_settings)
    work = model.work
    rows = []
    orig = work.bound_and_branch

    def recording(leaf):
        orig(leaf)
        rows.append([work.iter_num, leaf.depth, leaf.status, leaf.num_iter, leaf.lower,
                     work.upper_glob, work.lower_glob, len(work.leaves),
                     -1 if leaf.constr_idx is None else leaf.constr_idx,
                     -1 if leaf.nextvar_idx is None else leaf.nextvar_idx,
                     -1 if leaf.intinf is None else leaf.intinf])

    work.bound_and_branch = recording
    out = []

    def one_solve():
        del rows[:]
        res = model.solve()
        out.append(dict(trace=np.array(rows, dtype=float).reshape(-1, len(TRACE_COLS)),
                        x=np.array(res.x, dtype=float), upper_glob=res.upper_glob,
                        status=res.status, osqp_iter=work.osqp_iter,
                        osqp_iter_avg=res.osqp_iter_avg, iter_num=work.iter_num))

    if x0 is not None:
        model.set_x0(x0)
    one_solve()
    for (q, l, u, x0u) in updates:
        model.update_vectors(q=q, l=l, u=u)
        if x0u is not None:
            model.set_x0(x0u)
        one_solve()
    return out


PREFIX = ""  # "rhoauto_" for the second fixture set


def save_case(name, prob, settings, qp_settings, solves, updates=(), x0=None):
    name = PREFIX + name
    P = spa.csc_matrix(prob["P"]); A = spa.csc_matrix(prob["A"])
    P.sort_indices(); A.sort_indices()
    d = dict(P_indptr=P.indptr, P_indices=P.indices, P_data=P.data, P_shape=P.shape,
             A_indptr=A.indptr, A_indices=A.indices, A_data=A.data, A_shape=A.shape,
             q=prob["q"], l=prob["l"], u=prob["u"], i_idx=np.asarray(prob["i_idx"]),
             i_l=prob["i_l"], i_u=prob["i_u"],
             settings=json.dumps(settings), qp_settings=json.dumps(qp_settings),
             trace_cols=json.dumps(TRACE_COLS), n_solves=len(solves),
             n_updates=len(updates), has_x0=x0 is not None)
    if x0 is not None:
        d["x0"] = x0
    for k, (q, l, u, x0u) in enumerate(updates):
        d["upd%d_q" % k] = q; d["upd%d_l" % k] = l; d["upd%d_u" % k] = u
        d["upd%d_has_x0" % k] = x0u is not None
        if x0u is not None:
            d["upd%d_x0" % k] = x0u
    for k, s in enumerate(solves):
        d["s%d_trace" % k] = s["trace"]; d["s%d_x" % k] = s["x"]
        d["s%d_upper_glob" % k] = s["upper_glob"]; d["s%d_status" % k] = s["status"]
        d["s%d_osqp_iter" % k] = s["osqp_iter"]; d["s%d_osqp_iter_avg" % k] = s["osqp_iter_avg"]
        d["s%d_iter_num" % k] = s["iter_num"]
    path = os.path.join(HERE, "bnb_%s.npz" % name)
    np.savez_compressed(path, **d)
    print("%-22s nodes=%s status=%s upper=%s" % (
        name, [len(s["trace"]) for s in solves], [s["status"] for s in solves],
        ["%.6g" % s["upper_glob"] for s in solves]))


def main():
    global PREFIX
    auto = "--rho-auto" in sys.argv
    PREFIX = "rhoauto_" if auto else ""
    qp = dict(problems.QP_SETTINGS)
    if auto:
        qp["rho"] = "auto"
    base = dict(problems.BNB_SETTINGS)
    cases = [
        ("n10m5p2_s0", dict(n=10, m=5, p=2), 0, {}, {}),
        ("n10m100p2_s1", dict(n=10, m=100, p=2), 1, {}, {}),
        ("n12m60p6_s2", dict(n=12, m=60, p=6), 2, {}, {}),
        ("n20m100p10_s3", dict(n=20, m=100, p=10), 3, {}, {}),
        ("n20m100p10_s3_dfs", dict(n=20, m=100, p=10), 3, {"tree_explor_rule": 0}, {}),
        ("n30m150p15_s4", dict(n=30, m=150, p=15), 4, {}, {"rho": 0.03}),
        ("n50m25p5_s5", dict(n=50, m=25, p=5), 5, {}, {}),
        ("cfg1_n50m100p10_s0", dict(n=50, m=100, p=10), 0, {}, {}),
        ("n20m100p10_s6_cap", dict(n=20, m=100, p=10), 6, {"max_iter_bb": 6}, {}),
    ]
    for name, dims, seed, so, qo in cases:
        prob = problems.random_miqp(density=0.7, seed=seed, **dims)
        st = dict(base); st.update(so)
        qs = dict(qp)
        if not auto:
            qs.update(qo)
        save_case(name, prob, st, qs, run_reference(prob, st, qs))

    # infeasible relaxation at the root (contradictory rows)
    prob = problems.random_miqp(n=10, m=20, p=3, density=0.7, seed=7)
    A = spa.csc_matrix(prob["A"]).tolil()
    A[1, :] = A[0, :]
    prob["A"] = A.tocsc()
    prob["l"][0], prob["u"][0] = 1.0, 2.0
    prob["l"][1], prob["u"][1] = -2.0, -1.0
    save_case("infeasible_n10", prob, base, qp, run_reference(prob, base, qp))

    # update_vectors + set_x0 re-solve path (solver.py:174-212), MPC-style sequence
    prob = problems.random_miqp(n=12, m=30, p=6, density=0.7, seed=8)
    rng = np.random.RandomState(80)
    first = run_reference(prob, base, qp)
    x_prev = first[0]["x"]
    updates = []
    for k in range(3):
        q = prob["q"] + 0.3 * rng.randn(12)
        l = prob["l"] - 0.1 * rng.rand(30)
        u = prob["u"] + 0.1 * rng.rand(30)
        updates.append((q, l, u, np.copy(x_prev) if k != 1 else None))
    save_case("mpc_n12m30p6_s8", prob, base, qp,
              run_reference(prob, base, qp, updates=updates), updates=updates)

    # set_x0 with an invalid guess, then with the true optimum
    prob = problems.random_miqp(n=10, m=40, p=4, density=0.7, seed=9)
    opt = run_reference(prob, base, qp)[0]["x"]
    bad = np.full(10, 0.5)
    save_case("x0_bad_n10", prob, base, qp, run_reference(prob, base, qp, x0=bad), x0=bad)
    save_case("x0_opt_n10", prob, base, qp, run_reference(prob, base, qp, x0=np.copy(opt)),
              x0=np.copy(opt))


if __name__ == "__main__":
    main()
