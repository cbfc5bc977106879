"""Parity of the HIP relaxation path against the CPU oracle, through the C ABI (needs an MI355X).

Tolerances (fp64, stated per the north star): scaled iterates after k raw ADMM iterations agree
to 1e-9 relative (inf-norm); a finished solve has the identical status and iteration count and
x, y within 1e-8 relative; node lower bounds within 1e-9 relative.  Both sides implement the
frozen spec of DESIGN.md; differences are summation order and FMA contraction only.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from golden_cases import case_names, load_case, load_maxiter, run_case
from miosqp_amd import problems
from qp_check import kkt_certificate, osqp_tolerances

pytestmark = pytest.mark.gpu

ITER_TOL = 1e-9
SOL_TOL = 1e-8


def rel(a, b):
    return np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))


def both(oracle_mod, pr, **kw):
    from miosqp_amd import qp
    A, l, u = problems.extended(pr)
    st = dict(problems.QP_SETTINGS)
    st.update(kw)
    g, o = qp.OSQP(), oracle_mod.OSQP()
    g.setup(pr["P"], pr["q"], A, l, u, **st)
    o.setup(pr["P"], pr["q"], A, l, u, **st)
    g.set_integer_rows(pr["i_idx"], pr["A"].shape[0])
    return g, o, A, l, u


@pytest.mark.parametrize("n,m,p,seed", [(10, 5, 2, 0), (50, 100, 10, 1), (130, 260, 65, 2), (64, 1, 3, 3)])
def test_iterates_match_oracle(oracle_mod, n, m, p, seed):
    pr = problems.random_miqp(n, m, p, seed=seed)
    g, o, A, l, u = both(oracle_mod, pr)
    rng = np.random.RandomState(seed)
    x0, y0 = rng.randn(n), rng.randn(A.shape[0])
    Dg, Eg, cg = g.scaling()
    Do, Eo, co = o.scaling()
    assert rel(Dg, Do) <= 1e-14 and rel(Eg, Eo) <= 1e-14 and abs(cg - co) <= 1e-14 * co
    for k in (1, 2, 10, 50, 200):
        g.warm_start(x=x0, y=y0)
        o.warm_start(x=x0, y=y0)
        xg, zg, yg = g.debug_iterate(k)
        o.iterate(k)
        xo, zo, yo = o.iterates()
        assert rel(xg, xo) <= ITER_TOL and rel(zg, zo) <= ITER_TOL and rel(yg, yo) <= ITER_TOL, k


@pytest.mark.parametrize("n,m,p,seed", [(10, 5, 2, 0), (12, 60, 6, 1), (50, 100, 10, 2), (130, 260, 65, 3),
                                        (200, 50, 100, 4)])
def test_solve_matches_oracle(oracle_mod, n, m, p, seed):
    pr = problems.random_miqp(n, m, p, seed=seed)
    g, o, A, l, u = both(oracle_mod, pr)
    M = A.shape[0]
    g.warm_start(x=np.zeros(n), y=np.zeros(M))
    o.warm_start(x=np.zeros(n), y=np.zeros(M))
    rg, ro = g.solve(), o.solve()
    assert rg.info.status_val == ro.info.status_val
    assert rg.info.iter == ro.info.iter
    assert rel(rg.x, ro.x) <= SOL_TOL and rel(rg.y, ro.y) <= SOL_TOL
    assert abs(rg.info.obj_val - ro.info.obj_val) <= 1e-9 * max(1, abs(ro.info.obj_val))
    assert abs(rg.info.pri_res - ro.info.pri_res) <= 1e-9 + 1e-6 * ro.info.pri_res
    assert abs(rg.info.dua_res - ro.info.dua_res) <= 1e-9 + 1e-6 * ro.info.dua_res
    # the reference's call order on a branched node: update -> warm_start(parent) -> solve
    l2, u2 = l.copy(), u.copy()
    u2[-1] = 0.0
    l2[-2] = 1.0
    for s in (g, o):
        s.update(l=l2, u=u2)
        s.warm_start(x=ro.x, y=ro.y)
    rg2, ro2 = g.solve(), o.solve()
    assert (rg2.info.status_val, rg2.info.iter) == (ro2.info.status_val, ro2.info.iter)
    assert rel(rg2.x, ro2.x) <= SOL_TOL and rel(rg2.y, ro2.y) <= SOL_TOL
    # fused node entry == the four calls + numpy epilogue of node.py:131-143
    r3 = g.solve_node(l2, u2, ro.x, ro.y)
    assert (r3.status_val, r3.iter) == (ro2.info.status_val, ro2.info.iter)
    xo = ro2.x.copy()
    ii, k = pr["i_idx"], len(pr["i_idx"])
    xo[ii] = np.minimum(np.maximum(xo[ii], l2[-k:]), u2[-k:])
    assert rel(r3.x, xo) <= SOL_TOL and rel(r3.y, ro2.y) <= SOL_TOL
    lo = 0.5 * xo.dot(pr["P"].dot(xo)) + pr["q"].dot(xo)
    assert abs(r3.lower - lo) <= 1e-9 * max(1.0, abs(lo))
    ig = r3.x[ii]
    assert np.all(ig >= l2[-k:]) and np.all(ig <= u2[-k:])


def test_rerun_is_bit_identical_and_order_independent(oracle_mod):
    pr = problems.random_miqp(50, 100, 10, seed=7)
    g, o, A, l, u = both(oracle_mod, pr)
    n, M = 50, A.shape[0]
    rng = np.random.RandomState(1)
    x0, y0 = rng.randn(n), rng.randn(M)
    a = g.solve_node(l, u, x0, y0)
    u2 = u.copy()
    u2[-3] = 0.0
    g.solve_node(l, u2, a.x, a.y)
    b = g.solve_node(l, u, x0, y0)
    assert a.iter == b.iter and a.lower == b.lower
    np.testing.assert_array_equal(a.x, b.x)
    np.testing.assert_array_equal(a.y, b.y)


def test_infeasibility_certificates(oracle_mod):
    import scipy.sparse as spa
    from miosqp_amd import qp
    # primal infeasible
    P = spa.csc_matrix(np.eye(2))
    A = spa.csc_matrix(np.array([[1.0, 1.0], [1.0, 1.0], [1.0, 0.0]]))
    l = np.array([1.0, -np.inf, -np.inf])
    u = np.array([np.inf, -1.0, np.inf])
    for (Pm, q, Am, lm, um, code) in (
            (P, np.zeros(2), A, l, u, -3),
            (spa.csc_matrix(np.diag([1.0, 0.0])), np.array([0.0, 1.0]),
             spa.csc_matrix(np.eye(2)), np.array([-1.0, -np.inf]), np.array([1.0, 5.0]), -4)):
        g, o = qp.OSQP(), oracle_mod.OSQP()
        g.setup(Pm, q, Am, lm, um)
        o.setup(Pm, q, Am, lm, um)
        z2, z3 = np.zeros(2), np.zeros(Am.shape[0])
        g.warm_start(x=z2, y=z3)
        o.warm_start(x=z2, y=z3)
        rg, ro = g.solve(), o.solve()
        assert rg.info.status_val == ro.info.status_val == code
        assert rg.info.iter == ro.info.iter
        np.testing.assert_allclose(rg.x, ro.x, rtol=1e-7, atol=1e-9, equal_nan=True)
        np.testing.assert_allclose(rg.y, ro.y, rtol=1e-7, atol=1e-9, equal_nan=True)


def test_bounds_validation():
    from miosqp_amd import qp
    pr = problems.random_miqp(10, 5, 2, seed=0)
    A, l, u = problems.extended(pr)
    g = qp.OSQP()
    bad = l.copy()
    bad[0] = u[0] + 1
    with pytest.raises(ValueError):
        g.setup(pr["P"], pr["q"], A, bad, u)
    g = qp.OSQP()
    g.setup(pr["P"], pr["q"], A, l, u)
    with pytest.raises(ValueError):
        g.update(l=bad, u=u)
    with pytest.raises(ValueError):
        g.update(l=l[:-1], u=u)


@pytest.mark.parametrize("name", case_names())
def test_bnb_traces_on_gpu(name):
    """The reference's recorded tree (tests/golden) replayed with the HIP engine underneath:
    identical discrete decisions, bounds within the relaxation tolerance."""
    from miosqp_amd import qp
    case = load_case(name)
    got = run_case(case, qp)
    cols = case["cols"]
    disc = [cols.index(c) for c in ("iter_num", "depth", "status", "num_iter", "n_leaves", "constr_idx",
                                    "nextvar_idx", "intinf")]
    cont = [cols.index(c) for c in ("lower", "upper_glob", "lower_glob")]
    assert len(got) == len(case["solves"])
    for g, e in zip(got, case["solves"]):
        assert g["status"] == e["status"]
        assert g["iter_num"] == e["iter_num"] and g["osqp_iter"] == e["osqp_iter"]
        assert g["trace"].shape == e["trace"].shape
        np.testing.assert_array_equal(g["trace"][:, disc], e["trace"][:, disc])
        np.testing.assert_allclose(g["trace"][:, cont], e["trace"][:, cont], rtol=1e-8, atol=1e-9)
        if e["status"] in ("Solved", "Max-iter feasible"):
            assert rel(g["x"], e["x"]) <= SOL_TOL
            assert abs(g["upper_glob"] - e["upper_glob"]) <= 1e-8 * max(1, abs(e["upper_glob"]))


def test_reference_hard_instances_on_gpu(oracle_mod):
    """The reference's 49 max-iter inputs: same status and iteration count as the oracle, incl. +-inf
    bounds and the older setting names they carry."""
    from miosqp_amd import qp
    cache = {}
    for inst in load_maxiter():
        key = (id(inst["P"]), str(sorted(inst["settings"].items())))
        st = dict(inst["settings"])
        g, o = qp.OSQP(), oracle_mod.OSQP()
        g.setup(inst["P"], inst["q"], inst["A"], inst["l"], inst["u"], **st)
        o.setup(inst["P"], inst["q"], inst["A"], inst["l"], inst["u"], **st)
        n, M = inst["A"].shape[1], inst["A"].shape[0]
        g.warm_start(x=np.zeros(n), y=np.zeros(M))
        o.warm_start(x=np.zeros(n), y=np.zeros(M))
        rg, ro = g.solve(), o.solve()
        assert rg.info.status_val == ro.info.status_val, inst["name"]
        assert rg.info.iter == ro.info.iter, inst["name"]
        if ro.info.status_val in (1, -2):
            assert rel(rg.x, ro.x) <= 1e-6 and rel(rg.y, ro.y) <= 1e-6, inst["name"]
        g.close()


def test_config2_full_size_root_and_children(oracle_mod):
    """BASELINE config 2 (n=500, m=1000, p=250): root relaxation and two branched children."""
    pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
    g, o, A, l, u = both(oracle_mod, pr)
    n, M = 500, A.shape[0]
    fs = g.factor_stats()
    assert fs["nnz_L"] == o.factor_nnz() == 475000
    z0, y0 = np.zeros(n), np.zeros(M)
    rg = g.solve_node(l, u, z0, y0)
    o.update(l=l, u=u)
    o.warm_start(x=z0, y=y0)
    ro = o.solve()
    assert (rg.status_val, rg.iter) == (ro.info.status_val, ro.info.iter)
    assert rel(rg.y, ro.y) <= SOL_TOL
    ii, p_int = pr["i_idx"], len(pr["i_idx"])
    cont = np.setdiff1d(np.arange(n), ii)
    xo = ro.x.copy()
    xo[ii] = np.minimum(np.maximum(xo[ii], l[-p_int:]), u[-p_int:])  # node.py:131-136
    assert rel(rg.x[cont], ro.x[cont]) <= SOL_TOL and rel(rg.x, xo) <= SOL_TOL
    c = kkt_certificate(pr["P"], pr["q"], A, l, u, ro.x, ro.y)
    zc = np.clip(A.dot(ro.x), l, u)
    ep, ed = osqp_tolerances(pr["P"], pr["q"], A, ro.x, ro.y, zc, 1e-3, 1e-3)
    assert c["pri"] <= ep and c["dua"] <= ed
    # branch on the most fractional integer like workspace.py:205-230
    xi = rg.x[ii]
    k = int(np.argmax(np.abs(xi - np.round(xi))))
    for side in (0, 1):
        l2, u2 = l.copy(), u.copy()
        if side == 0:
            u2[1000 + k] = np.floor(xi[k])
        else:
            l2[1000 + k] = np.ceil(xi[k])
        r2 = g.solve_node(l2, u2, rg.x, rg.y)
        o.update(l=l2, u=u2)
        o.warm_start(x=rg.x, y=rg.y)
        ro2 = o.solve()
        assert (r2.status_val, r2.iter) == (ro2.info.status_val, ro2.info.iter)
        assert rel(r2.y, ro2.y) <= SOL_TOL
        xo2 = ro2.x.copy()
        xo2[ii] = np.minimum(np.maximum(xo2[ii], l2[-p_int:]), u2[-p_int:])
        assert rel(r2.x[cont], ro2.x[cont]) <= SOL_TOL and rel(r2.x, xo2) <= SOL_TOL
        lo2 = 0.5 * xo2.dot(pr["P"].dot(xo2)) + pr["q"].dot(xo2)
        assert abs(r2.lower - lo2) <= 1e-9 * max(1.0, abs(lo2))


def _wave_of_nodes(oracle_mod, pr, count):
    """A wave of distinct, realistic nodes: breadth-first branching (floor / ceil on the three most
    fractional integers of every solved node) from the root, children warm-started from the parent."""
    import types
    A, l, u = problems.extended(pr)
    n, M, m = A.shape[1], A.shape[0], pr["A"].shape[0]
    o = oracle_mod.OSQP()
    o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
    nodes = [types.SimpleNamespace(l=l.copy(), u=u.copy(), x=np.zeros(n), y=np.zeros(M))]
    k = 0
    ii = pr["i_idx"]
    while len(nodes) < count and k < len(nodes):
        nd = nodes[k]
        k += 1
        o.update(l=nd.l, u=nd.u)
        o.warm_start(x=nd.x, y=nd.y)
        r = o.solve()
        if r.info.status_val != 1:
            continue
        frac = np.abs(r.x[ii] - np.round(r.x[ii]))
        for v in np.argsort(-frac)[:3]:
            if frac[v] < 1e-3:
                continue
            for side in (0, 1):
                l2, u2 = nd.l.copy(), nd.u.copy()
                if side == 0:
                    u2[m + v] = np.floor(r.x[ii[v]])
                else:
                    l2[m + v] = np.ceil(r.x[ii[v]])
                if np.all(l2 <= u2):
                    nodes.append(types.SimpleNamespace(l=l2, u=u2, x=r.x.copy(), y=r.y.copy()))
    return nodes[:count]


@pytest.mark.parametrize("n,m,p,seed,count,fold,cap", [(20, 40, 10, 1, 7, -1, 64), (50, 100, 25, 2, 70, 0, 64),
                                                        (50, 100, 25, 2, 70, 1, 64), (130, 260, 65, 3, 130, -1, 64),
                                                        (37, 3, 20, 4, 9, 1, 64), (33, 2, 16, 6, 9, 0, 64),
                                                        (130, 260, 65, 3, 200, -1, 256), (50, 100, 25, 2, 150, 0, 192)])
def test_batch_equals_node_by_node(oracle_mod, n, m, p, seed, count, fold, cap):
    """solve_batch on a wave of real B&B leaves == solve_node on each == the oracle."""
    from miosqp_amd import qp
    pr = problems.random_miqp(n, m, p, seed=seed)
    leaves = _wave_of_nodes(oracle_mod, pr, count)
    assert len(leaves) >= 2
    A, l, u = problems.extended(pr)
    g, o = qp.OSQP(), oracle_mod.OSQP()
    # cap 64 < count: the wave is cut into slices; cap >= 128: several tiles, compacted as columns finish
    g.setup(pr["P"], pr["q"], A, l, u, max_batch=cap, fold=fold, **problems.QP_SETTINGS)
    o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
    g.set_integer_rows(pr["i_idx"], m)
    g.set_root(l, u, 1e-3, 1e-3)  # node digest on: integrality, branching variable, rounding heuristic
    L = np.stack([lf.l for lf in leaves]); U = np.stack([lf.u for lf in leaves])
    X = np.stack([lf.x for lf in leaves]); Y = np.stack([lf.y for lf in leaves])
    rb = g.solve_batch(L, U, X, Y)
    k_int = len(pr["i_idx"])
    for k in range(len(leaves)):
        r1 = g.solve_node(L[k], U[k], X[k], Y[k])
        assert rb.status_val[k] == r1.status_val and rb.iter[k] == r1.iter, k
        o.update(l=L[k], u=U[k])
        o.warm_start(x=X[k], y=Y[k])
        ro = o.solve()
        assert (rb.status_val[k], rb.iter[k]) == (ro.info.status_val, ro.info.iter), k
        if ro.info.status_val in (1, -2):
            assert rel(rb.x[k], r1.x) <= SOL_TOL and rel(rb.y[k], r1.y) <= SOL_TOL
            xo = ro.x.copy()
            ii = pr["i_idx"]
            xo[ii] = np.minimum(np.maximum(xo[ii], L[k][-k_int:]), U[k][-k_int:])
            assert rel(rb.x[k], xo) <= SOL_TOL and rel(rb.y[k], ro.y) <= SOL_TOL
            lo = 0.5 * xo.dot(pr["P"].dot(xo)) + pr["q"].dot(xo)
            assert abs(rb.lower[k] - lo) <= 1e-9 * max(1.0, abs(lo))
            assert abs(rb.lower[k] - r1.lower) <= 1e-9 * max(1.0, abs(lo))
            # digest: batched == single node == numpy restatement of workspace.py:245-272, 232-243
            db, d1 = rb.digest[k], r1.digest
            xi_all = xo[ii]
            frac = np.abs(xi_all - np.round(xi_all))
            assert db.int_inf == d1.int_inf == int(np.sum(frac > 1e-3))
            if frac.max() > 1e-6:
                assert db.nextvar == d1.nextvar == int(np.argmax(frac))
            xr = xo.copy()
            xr[ii] = np.round(xo[ii])
            zz = A.dot(xr)
            margin = np.max(np.maximum(l - 1e-3 - zz, zz - u - 1e-3))
            if abs(margin) > 1e-7:
                assert db.heur_feasible == d1.heur_feasible == bool(margin <= 0)
            ho = 0.5 * xr.dot(pr["P"].dot(xr)) + pr["q"].dot(xr)
            assert abs(db.heur_obj - ho) <= 1e-8 * max(1.0, abs(ho)) and abs(d1.heur_obj - ho) <= 1e-8 * max(1.0, abs(ho))
        else:
            assert np.isnan(rb.lower[k]) and rb.digest[k] is None
    if cap >= 128 and len(leaves) > 128:
        assert g.compactions() >= 1  # columns finish at different tests: the wave was compacted
    # a second identical call is bit-identical
    rb2 = g.solve_batch(L, U, X, Y)
    np.testing.assert_array_equal(rb.x, rb2.x)
    np.testing.assert_array_equal(rb.iter, rb2.iter)


def _frontier(g, pr, l, u, width):
    """Level-by-level expansion of the tree from the root (every leaf of a level is solved node-at-a-time
    and branched floor / ceil on the digest's branching variable, children warm-started from the parent,
    workspace.py:157-203) until at least `width` leaves are open -- the kind of frontier bench.py's batched
    leg works on.  Nothing is pruned, so the wave mixes cheap and expensive, feasible and infeasible leaves."""
    import types
    m, ii = pr["A"].shape[0], pr["i_idx"]
    n, M = len(pr["q"]), len(l)
    nodes = [types.SimpleNamespace(l=l.copy(), u=u.copy(), x=np.zeros(n), y=np.zeros(M))]
    while 0 < len(nodes) < width:
        level, nodes = nodes, []
        for nd in level:
            r = g.solve_node(nd.l, nd.u, nd.x, nd.y)
            if r.status_val != 1 or r.digest is None or r.digest.int_inf == 0:
                continue
            v = r.digest.nextvar
            xv = r.x[ii[v]]
            for side in (0, 1):
                l2, u2 = nd.l.copy(), nd.u.copy()
                if side == 0:
                    u2[m + v] = np.floor(xv)
                else:
                    l2[m + v] = np.ceil(xv)
                nodes.append(types.SimpleNamespace(l=l2, u=u2, x=r.x, y=r.y))
    return nodes


def _check_wave(g, o, pr, A, l, u, leaves, sample, want_compaction):
    """solve_batch(wave) == solve_node on every leaf (status, iterations, x, y, lower, digest) and == the
    oracle on every `sample`-th leaf; returns the per-leaf solve_node results."""
    L = np.stack([lf.l for lf in leaves]); U = np.stack([lf.u for lf in leaves])
    X = np.stack([lf.x for lf in leaves]); Y = np.stack([lf.y for lf in leaves])
    ii, p_int = pr["i_idx"], len(pr["i_idx"])
    c0 = g.compactions()
    rb = g.solve_batch(L, U, X, Y)
    if want_compaction:
        assert g.compactions() > c0  # columns finish at different tests: the wave was compacted
    singles = []
    for k in range(len(leaves)):
        r1 = g.solve_node(L[k], U[k], X[k], Y[k])
        singles.append(r1)
        assert (rb.status_val[k], rb.iter[k]) == (r1.status_val, r1.iter), k
        if r1.status_val in (1, -2):
            assert rel(rb.x[k], r1.x) <= SOL_TOL and rel(rb.y[k], r1.y) <= SOL_TOL, k
            assert abs(rb.lower[k] - r1.lower) <= 1e-9 * max(1.0, abs(r1.lower)), k
            db, d1 = rb.digest[k], r1.digest
            assert db.int_inf == d1.int_inf, k
            frac = np.abs(r1.x[ii] - np.round(r1.x[ii]))
            srt = np.sort(frac)
            if srt[-1] - srt[-2] > 1e-7:  # a clear winner: the branching variable cannot depend on rounding
                assert db.nextvar == d1.nextvar == int(np.argmax(frac)), k
            if abs(d1.info_viol) > 1e-7:
                assert db.heur_feasible == d1.heur_feasible, k
            assert abs(db.heur_obj - d1.heur_obj) <= 1e-9 * max(1.0, abs(d1.heur_obj)), k
        else:
            assert np.isnan(rb.lower[k]) and rb.digest[k] is None
        if k % sample == 0:
            o.update(l=L[k], u=U[k])
            o.warm_start(x=X[k], y=Y[k])
            ro = o.solve()
            assert (rb.status_val[k], rb.iter[k]) == (ro.info.status_val, ro.info.iter), k
            if ro.info.status_val in (1, -2):
                xo = ro.x.copy()
                xo[ii] = np.minimum(np.maximum(xo[ii], L[k][-p_int:]), U[k][-p_int:])
                assert rel(rb.x[k], xo) <= SOL_TOL and rel(rb.y[k], ro.y) <= SOL_TOL, k
                lo = 0.5 * xo.dot(pr["P"].dot(xo)) + pr["q"].dot(xo)
                assert abs(rb.lower[k] - lo) <= 1e-9 * max(1.0, abs(lo)), k
    rb2 = g.solve_batch(L, U, X, Y)  # bit-identical rerun
    np.testing.assert_array_equal(rb.x, rb2.x)
    np.testing.assert_array_equal(rb.y, rb2.y)
    np.testing.assert_array_equal(rb.iter, rb2.iter)
    return singles


def test_config3_full_size_wave_of_256_leaves(oracle_mod):
    """BASELINE config 3: random_miqp n=500 m=1000 p=250, 256 open leaves in ONE batched device call
    (four 64-column tiles, compacted as columns finish), then their children -- more than 256 leaves, up
    to eight tiles -- on an engine with capacity 1024."""
    from miosqp_amd import qp
    pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
    A, l, u = problems.extended(pr)
    m = pr["A"].shape[0]
    g = qp.OSQP()
    g.setup(pr["P"], pr["q"], A, l, u, **dict(problems.QP_SETTINGS, max_batch=256))
    g.set_integer_rows(pr["i_idx"], m)
    g.set_root(l, u, 1e-3, 1e-3)
    leaves = _frontier(g, pr, l, u, 256)[:256]
    assert len(leaves) == 256
    o = oracle_mod.OSQP()
    o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
    singles = _check_wave(g, o, pr, A, l, u, leaves, 8, True)   # 32 leaves against the oracle
    iters = np.array([r.iter for r in singles])
    assert iters.min() < iters.max()  # a real wave: leaves need different numbers of tests
    # children of the wave (floor / ceil on the digest's branching variable, warm-started from the parent)
    import types
    kids = []
    for lf, r in zip(leaves, singles):
        if r.status_val != 1 or r.digest is None or r.digest.int_inf == 0:
            continue
        v = r.digest.nextvar
        xv = r.x[pr["i_idx"][v]]
        for side in (0, 1):
            l2, u2 = lf.l.copy(), lf.u.copy()
            if side == 0:
                u2[m + v] = np.floor(xv)
            else:
                l2[m + v] = np.ceil(xv)
            kids.append(types.SimpleNamespace(l=l2, u=u2, x=r.x, y=r.y))
    kids = kids[:448]
    assert len(kids) > 256
    g2 = qp.OSQP()
    g2.setup(pr["P"], pr["q"], A, l, u, **dict(problems.QP_SETTINGS, max_batch=1024))
    g2.set_integer_rows(pr["i_idx"], m)
    g2.set_root(l, u, 1e-3, 1e-3)
    _check_wave(g2, o, pr, A, l, u, kids, 28, True)             # 16 more against the oracle


def _cfg3_engine(pr, A, l, u, width, batch_pers):
    from miosqp_amd import qp
    g = qp.OSQP()
    g.setup(pr["P"], pr["q"], A, l, u, **dict(problems.QP_SETTINGS, max_batch=width, batch_pers=batch_pers))
    g.set_integer_rows(pr["i_idx"], pr["A"].shape[0])
    g.set_root(l, u, 1e-3, 1e-3)
    return g


def _same_wave(a, b):
    np.testing.assert_array_equal(a.status_val, b.status_val)
    np.testing.assert_array_equal(a.iter, b.iter)
    ok = a.status_val == 1
    assert ok.sum() > len(ok) // 2
    np.testing.assert_array_equal(a.x[ok], b.x[ok])  # bit for bit: same chunk-to-wave assignment, chains and reduction order
    np.testing.assert_array_equal(a.y[ok], b.y[ok])
    np.testing.assert_array_equal(a.lower[ok], b.lower[ok])


def test_batched_persistent_chunks_equal_the_launches(monkeypatch):
    """Config 3's wave with the lock-step iterations of a chunk as ONE persistent launch (kbp1: iterates in registers,
    256 columns; kbp: 512 columns, two tiles per group) against two launches per iteration: the same bits.  Then with a
    workgroup that never shows up (fault injection): the launch is called off within 100 ms, nothing was modified, the
    engine goes on with the launches and returns the same wave."""
    pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
    A, l, u = problems.extended(pr)
    g0 = _cfg3_engine(pr, A, l, u, 512, 0)
    assert not g0.factor_stats()["batch_pers"]
    leaves = _frontier(g0, pr, l, u, 256)[:256]
    L = np.stack([lf.l for lf in leaves]); U = np.stack([lf.u for lf in leaves])
    X = np.stack([lf.x for lf in leaves]); Y = np.stack([lf.y for lf in leaves])
    ref = g0.solve_batch(L, U, X, Y)
    g1 = _cfg3_engine(pr, A, l, u, 256, 1)
    _same_wave(ref, g1.solve_batch(L, U, X, Y))
    assert g1.batch_pers_fallbacks() == 0 and g1.factor_stats()["batch_pers"]
    # twice the columns: the general kernel (a group owns two tiles; the iterates live in memory)
    L2, U2, X2, Y2 = (np.concatenate([v, v[::-1]]) for v in (L, U, X, Y))
    ref2 = g0.solve_batch(L2, U2, X2, Y2)
    g2 = _cfg3_engine(pr, A, l, u, 512, 1)
    _same_wave(ref2, g2.solve_batch(L2, U2, X2, Y2))
    assert g2.batch_pers_fallbacks() == 0 and g2.factor_stats()["batch_pers"]
    # one workgroup missing (the flag is read when the batch arrays are made, at the first solve_batch)
    g3 = _cfg3_engine(pr, A, l, u, 256, 1)
    monkeypatch.setenv("MIOSQP_COOP_DBG", "64")
    r3 = g3.solve_batch(L, U, X, Y)
    monkeypatch.delenv("MIOSQP_COOP_DBG")
    _same_wave(ref, r3)
    assert g3.batch_pers_fallbacks() == 1 and not g3.factor_stats()["batch_pers"]


@pytest.mark.parametrize("n,m,p,dens,B,idrows", [(401, 590, 62, 0.15, 256, True), (300, 709, 105, 0.4, 320, False),
                                                   (100, 62, 42, 0.4, 256, True), (512, 1000, 200, 0.7, 192, True)])
def test_batched_persistent_chunks_over_shapes(monkeypatch, n, m, p, dens, B, idrows):
    """The persistent batched sweeps against the launches inside their limits (n <= 512, rows in the products <= 1024):
    odd sizes, with and without the identity-row shortcut, one and two column tiles per group, waves of random 0/1
    fixings -- the same bits (tools/probes/soak_kbp.py runs more of them)."""
    from miosqp_amd import qp
    if not idrows:
        monkeypatch.setenv("MIOSQP_NO_IDROWS", "1")
    pr = problems.random_miqp(n, m, p, density=dens, seed=n + m)
    A, l, u = problems.extended(pr)
    rng = np.random.RandomState(p)
    L = np.tile(l, (B, 1)); U = np.tile(u, (B, 1))
    for b in range(B):
        idx = rng.choice(p, size=min(p, 1 + b % 6), replace=False)
        val = rng.randint(0, 2, size=len(idx)).astype(float)
        L[b, m + idx] = val
        U[b, m + idx] = val
    X = np.zeros((B, n)); Y = np.zeros((B, m + p))
    out = []
    for bp in (0, 1):
        g = qp.OSQP()
        g.setup(pr["P"], pr["q"], A, l, u, **dict(problems.QP_SETTINGS, max_batch=max(B, 256), batch_pers=bp, max_iter=500))
        g.set_integer_rows(pr["i_idx"], m)
        g.set_root(l, u, 1e-3, 1e-3)
        out.append(g.solve_batch(L, U, X, Y))
        assert g.factor_stats()["batch_pers"] == bool(bp) and g.batch_pers_fallbacks() == 0
        g.close()
    a, b_ = out
    np.testing.assert_array_equal(a.status_val, b_.status_val)
    np.testing.assert_array_equal(a.iter, b_.iter)
    np.testing.assert_array_equal(a.x, b_.x)
    np.testing.assert_array_equal(a.y, b_.y)


def test_batched_search_finds_the_same_optimum():
    """Waves of 8 leaves through solve_batch reach the sequential search's optimum."""
    from miosqp_amd import bnb, dist
    pr = problems.random_miqp(30, 150, 15, seed=4)
    out = []
    for width in (1, 8):
        model = bnb.MIOSQP()
        qs = dict(problems.QP_SETTINGS, max_batch=64)
        model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                    dict(problems.BNB_SETTINGS), qs)
        s = dist.ShardedSearch(model)
        while model.work.leaves:
            s.step_batched(width)
        model.work.get_return_status()
        model.work.get_return_solution()
        out.append((model.work.status, model.work.upper_glob, model.work.x.copy(), s.nodes))
    assert out[0][0] == out[1][0] == bnb.MI_SOLVED
    assert abs(out[0][1] - out[1][1]) <= 1e-3 * max(1.0, abs(out[0][1]))
    ii = pr["i_idx"]
    np.testing.assert_array_equal(out[0][2][ii], out[1][2][ii])


# (fold, resident, coop, pers)
FORMS = [(0, 0, 0, 0), (1, 0, 0, 0), (1, 1, 0, 0), (1, 0, 1, 0), (0, 0, 0, 1), (1, 0, 0, 1), (0, 0, 0, 2)]


@pytest.mark.parametrize("fold,resident,coop,pers", FORMS)
def test_both_factor_forms_match_oracle(oracle_mod, fold, resident, coop, pers):
    """fold=0: factor form L (4 kernels per iteration); fold=1: product form L^-1 (2 kernels);
    resident=1: the whole solve in one LDS-resident workgroup; coop=1: the whole solve in one
    cooperative launch, explicit KKT inverse in registers, one exchange per iteration; pers=1: the whole solve
    in one persistent launch that streams the factor (either form) from memory in every iteration; pers=2: the same in
    factor form with the dense tail as the explicit inverse of the reduced Hessian (one dense phase instead of two)."""
    from miosqp_amd import qp
    pr = problems.random_miqp(60, 120, 30, seed=11)
    A, l, u = problems.extended(pr)
    g, o = qp.OSQP(), oracle_mod.OSQP()
    g.setup(pr["P"], pr["q"], A, l, u, fold=fold, resident=resident, coop=coop, pers=pers, **problems.QP_SETTINGS)
    o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
    fs = g.factor_stats()
    assert (fs["fold"], fs["resident"], fs["coop"], fs["pers"]) == (bool(fold), bool(resident), bool(coop), bool(pers))
    rng = np.random.RandomState(3)
    x0, y0 = rng.randn(60), rng.randn(A.shape[0])
    for k in (1, 3, 40):
        g.warm_start(x=x0, y=y0)
        o.warm_start(x=x0, y=y0)
        xg, zg, yg = g.debug_iterate(k)
        o.iterate(k)
        xo, zo, yo = o.iterates()
        assert rel(xg, xo) <= ITER_TOL and rel(zg, zo) <= ITER_TOL and rel(yg, yo) <= ITER_TOL, k
    g.warm_start(x=x0, y=y0)
    o.warm_start(x=x0, y=y0)
    rg, ro = g.solve(), o.solve()
    assert (rg.info.status_val, rg.info.iter) == (ro.info.status_val, ro.info.iter)
    assert rel(rg.x, ro.x) <= SOL_TOL and rel(rg.y, ro.y) <= SOL_TOL
    # a new linear cost keeps the factor (solver.py:183-185)
    q2 = pr["q"] + 0.5 * rng.randn(60)
    for s_ in (g, o):
        s_.update(q=q2)
        s_.warm_start(x=np.zeros(60), y=np.zeros(A.shape[0]))
    rg, ro = g.solve(), o.solve()
    assert (rg.info.status_val, rg.info.iter) == (ro.info.status_val, ro.info.iter)
    assert rel(rg.x, ro.x) <= SOL_TOL and rel(rg.y, ro.y) <= SOL_TOL


@pytest.mark.parametrize("n,m,p,seed", [(30, 150, 15, 4), (50, 100, 10, 0), (60, 20, 30, 9)])
def test_device_digest_equals_host_logic(n, m, p, seed):
    """The on-device branching epilogue (integrality test, branching variable, rounding heuristic:
    workspace.py:245-272, 321-324) takes the same decisions as the host numpy code."""
    from miosqp_amd import bnb
    pr = problems.random_miqp(n, m, p, seed=seed)
    out = []
    for dev in (False, True):
        st = dict(problems.BNB_SETTINGS, device_digest=dev)
        model = bnb.MIOSQP()
        model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], st,
                    dict(problems.QP_SETTINGS))
        rows = []
        res = model.solve(observer=lambda w, lf: rows.append(
            (lf.depth, lf.status, lf.num_iter, int(lf.intinf) if lf.intinf is not None else -1,
             -1 if lf.constr_idx is None else int(lf.constr_idx), len(w.leaves), w.upper_glob, lf.lower)))
        assert (rows[0][3] >= 0) and (model.work.leaves == [])
        out.append((res.status, res.upper_glob, np.array(res.x), rows))
    assert out[0][0] == out[1][0]
    assert len(out[0][3]) == len(out[1][3])
    for a, b in zip(out[0][3], out[1][3]):
        assert a[:6] == b[:6]
        assert abs(a[6] - b[6]) <= 1e-9 * max(1.0, abs(a[6])) or (np.isinf(a[6]) and np.isinf(b[6]))
    assert abs(out[0][1] - out[1][1]) <= 1e-9 * max(1.0, abs(out[0][1]))
    np.testing.assert_allclose(out[0][2], out[1][2], rtol=0, atol=1e-9)


def test_config5_full_size_properties():
    """BASELINE config 5 (n=5000, m=10000, p=2500, 1 % dense A; factor form, 13 M factor entries): the
    oracle would take minutes here, so the full size is checked through size-independent properties:
    the KKT certificate at the solver's own tolerances, bit-identical reruns, batch == node by node."""
    from miosqp_amd import qp
    pr = problems.random_miqp(**problems.CONFIGS["cfg5"], seed=0)
    A, l, u = problems.extended(pr)
    n, M, m = 5000, A.shape[0], 10000
    g = qp.OSQP()
    g.setup(pr["P"], pr["q"], A, l, u, max_batch=64, **problems.QP_SETTINGS)
    g.set_integer_rows(pr["i_idx"], m)
    fs = g.factor_stats()
    assert fs["nnz_L"] == A.nnz + n * (n - 1) // 2 and not fs["fold"]
    g.warm_start(x=np.zeros(n), y=np.zeros(M))
    r = g.solve()
    assert r.info.status_val == 1 and r.info.iter % 25 == 0
    c = kkt_certificate(pr["P"], pr["q"], A, l, u, r.x, r.y)
    zc = np.clip(A.dot(r.x), l, u)
    ep, ed = osqp_tolerances(pr["P"], pr["q"], A, r.x, r.y, zc, 1e-3, 1e-3)
    assert c["pri"] <= ep and c["dua"] <= ed and c["stray"] == 0.0
    assert abs(c["dua"] - r.info.dua_res) <= 1e-9 + 1e-6 * c["dua"]
    assert abs(c["obj"] - r.info.obj_val) <= 1e-9 * max(1.0, abs(c["obj"]))
    ii = pr["i_idx"]
    xi = r.x[ii]
    k = int(np.argmax(np.abs(xi - np.round(xi))))
    L2, U2 = np.stack([l, l]), np.stack([u, u])
    U2[0, m + k] = np.floor(xi[k])
    L2[1, m + k] = np.ceil(xi[k])
    X0, Y0 = np.stack([r.x, r.x]), np.stack([r.y, r.y])
    a = [g.solve_node(L2[i], U2[i], X0[i], Y0[i]) for i in (0, 1)]
    b = [g.solve_node(L2[i], U2[i], X0[i], Y0[i]) for i in (0, 1)]
    rb = g.solve_batch(L2, U2, X0, Y0)
    for i in (0, 1):
        assert a[i].iter == b[i].iter and a[i].lower == b[i].lower
        np.testing.assert_array_equal(a[i].x, b[i].x)
        assert (rb.status_val[i], rb.iter[i]) == (a[i].status_val, a[i].iter)
        assert rel(rb.x[i], a[i].x) <= SOL_TOL and rel(rb.y[i], a[i].y) <= SOL_TOL
        lo = 0.5 * a[i].x.dot(pr["P"].dot(a[i].x)) + pr["q"].dot(a[i].x)
        assert abs(a[i].lower - lo) <= 1e-9 * max(1.0, abs(lo))
        xi2 = a[i].x[ii]
        assert np.all(xi2 >= L2[i][m:] - 0) and np.all(xi2 <= U2[i][m:] + 0)


def test_config5_against_the_oracle_fixture():
    """BASELINE config 5 at full size against the CPU oracle: tests/golden/cfg5_root.npz holds what oracle/qp_oracle.c
    returned (build container, tests/golden/make_cfg5_fixture.py) for the root and its two children in the reference's
    call order (node.py:102-143): status, iteration count, x after the integer clamp, y, lower bound."""
    from miosqp_amd import qp
    from golden_cases import GOLDEN
    z = np.load(os.path.join(GOLDEN, "cfg5_root.npz"), allow_pickle=False)
    pr = problems.random_miqp(**problems.CONFIGS["cfg5"], seed=0)
    assert problems.instance_digest(pr) == str(z["digest"]), "scipy's sampling changed: regenerate the fixture"
    A, l, u = problems.extended(pr)
    n, M, m = 5000, A.shape[0], 10000
    g = qp.OSQP()
    g.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
    g.set_integer_rows(pr["i_idx"], m)
    k = int(z["branch_k"])
    lo, up = [l.copy(), l.copy()], [u.copy(), u.copy()]
    up[0][m + k] = float(z["branch_floor"])
    lo[1][m + k] = float(z["branch_ceil"])
    nodes = [("root", l, u, np.zeros(n), np.zeros(M)),
             ("child0", lo[0], up[0], z["root_x"], z["root_y"]), ("child1", lo[1], up[1], z["root_x"], z["root_y"])]
    for name, nl, nu, x0, y0 in nodes:
        r = g.solve_node(nl, nu, x0, y0)
        assert (r.status_val, r.iter) == (int(z[name + "_status"]), int(z[name + "_iter"])), name
        assert rel(r.x, z[name + "_x"]) <= SOL_TOL and rel(r.y, z[name + "_y"]) <= SOL_TOL, name
        want = float(z[name + "_lower"])
        assert abs(r.lower - want) <= 1e-9 * max(1.0, abs(want)), name
    # the child pair is the one the GPU's own root solution would branch on as well
    xi = g.solve_node(l, u, np.zeros(n), np.zeros(M)).x[pr["i_idx"]]
    assert int(np.argmax(np.abs(xi - np.round(xi)))) == k
    g.close()


def test_setup_rejects_bad_input():
    import scipy.sparse as spa
    from miosqp_amd import qp
    P = spa.csc_matrix(np.eye(3))
    with pytest.raises(RuntimeError, match="at least one constraint"):
        qp.OSQP().setup(P, np.zeros(3), spa.csc_matrix((0, 3)), np.zeros(0), np.zeros(0))
    with pytest.raises(RuntimeError, match="non-positive pivot|KKT"):
        qp.OSQP().setup(spa.csc_matrix(-50.0 * np.eye(3)), np.zeros(3), spa.csc_matrix(np.eye(3)), -np.ones(3),
                        np.ones(3), scaling=0)
    with pytest.raises(ValueError):
        qp.OSQP().setup(P, np.zeros(3), spa.csc_matrix(np.eye(3)), -np.ones(3), np.ones(3), adaptive_rho=True)


@pytest.mark.parametrize("fold,resident,coop,pers", FORMS)
@pytest.mark.parametrize("max_iter,check", [(60, 25), (50, 0), (30, 7), (25, 25), (3, 1)])
def test_iteration_limits_and_test_cadence(oracle_mod, fold, resident, coop, pers, max_iter, check):
    """MAX_ITER_REACHED, a tail chunk shorter than the cadence, cadence 1, and the test switched off:
    status, iteration count and iterates equal the oracle's in every engine form."""
    from miosqp_amd import qp
    pr = problems.random_miqp(40, 60, 10, seed=21)
    A, l, u = problems.extended(pr)
    kw = dict(problems.QP_SETTINGS, max_iter=max_iter, check_termination=check, eps_abs=1e-9, eps_rel=1e-9)
    g, o = qp.OSQP(), oracle_mod.OSQP()
    g.setup(pr["P"], pr["q"], A, l, u, fold=fold, resident=resident, coop=coop, pers=pers, **kw)
    o.setup(pr["P"], pr["q"], A, l, u, **kw)
    x0, y0 = np.full(40, 0.3), np.zeros(A.shape[0])
    g.warm_start(x=x0, y=y0)
    o.warm_start(x=x0, y=y0)
    rg, ro = g.solve(), o.solve()
    assert ro.info.status_val == -2 and ro.info.iter == max_iter
    assert (rg.info.status_val, rg.info.iter) == (ro.info.status_val, ro.info.iter)
    assert rel(rg.x, ro.x) <= SOL_TOL and rel(rg.y, ro.y) <= SOL_TOL
    # loose tolerances: converges at the first test that passes; same iteration as the oracle
    kw2 = dict(problems.QP_SETTINGS, max_iter=4000, check_termination=max(check, 1))
    g2, o2 = qp.OSQP(), oracle_mod.OSQP()
    g2.setup(pr["P"], pr["q"], A, l, u, fold=fold, resident=resident, coop=coop, **kw2)
    o2.setup(pr["P"], pr["q"], A, l, u, **kw2)
    g2.warm_start(x=x0, y=y0)
    o2.warm_start(x=x0, y=y0)
    rg2, ro2 = g2.solve(), o2.solve()
    assert ro2.info.status_val == 1
    assert (rg2.info.status_val, rg2.info.iter) == (ro2.info.status_val, ro2.info.iter)
    assert rel(rg2.x, ro2.x) <= SOL_TOL and rel(rg2.y, ro2.y) <= SOL_TOL


def test_cold_start_setting(oracle_mod):
    """warm_start=False: every solve restarts from zero (x1-x15 'cold start'), as the oracle does."""
    from miosqp_amd import qp
    pr = problems.random_miqp(30, 40, 8, seed=22)
    A, l, u = problems.extended(pr)
    kw = dict(problems.QP_SETTINGS, warm_start=False)
    g, o = qp.OSQP(), oracle_mod.OSQP()
    g.setup(pr["P"], pr["q"], A, l, u, **kw)
    o.setup(pr["P"], pr["q"], A, l, u, **kw)
    a, b = g.solve(), o.solve()
    a2, b2 = g.solve(), o.solve()
    assert (a.info.iter, a2.info.iter) == (b.info.iter, b2.info.iter) and a.info.iter == a2.info.iter
    assert rel(a2.x, b2.x) <= SOL_TOL


@pytest.mark.parametrize("n,m,p,seed", [(50, 100, 10, 1), (130, 260, 65, 2), (200, 50, 100, 3), (257, 40, 20, 4)])
def test_device_setup_matches_host_setup(oracle_mod, n, m, p, seed):
    """setup_on_device=1 (dense LDL^T + triangular inverse on the GPU) gives
    the same iterates as the host setup and as the oracle; block sizes around the 64-tile edges."""
    from miosqp_amd import qp
    pr = problems.random_miqp(n, m, p, seed=seed)
    A, l, u = problems.extended(pr)
    o = oracle_mod.OSQP()
    o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
    rng = np.random.RandomState(seed)
    x0, y0 = rng.randn(n), rng.randn(A.shape[0])
    outs = []
    for dev in (0, 1):
        g = qp.OSQP()
        g.setup(pr["P"], pr["q"], A, l, u, setup_on_device=dev, resident=0, **problems.QP_SETTINGS)
        assert g.factor_stats()["setup_on_device"] == bool(dev)
        g.warm_start(x=x0, y=y0)
        outs.append(g.debug_iterate(40))
    o.warm_start(x=x0, y=y0)
    o.iterate(40)
    ref = o.iterates()
    for got in outs:
        for a, b in zip(got, ref):
            assert rel(a, b) <= ITER_TOL
    for a, b in zip(outs[0], outs[1]):
        assert rel(a, b) <= 1e-11
    # the device stages in their shapes -- equilibration's maxima / products on the device or on the host, Schur
    # complement assembled on the device or uploaded from the host, Linv left on the device or sent down and up again
    # -- give bitwise the same scaling and factor: identical iterates
    import os
    variants = []
    for env in ({}, {"MIOSQP_SETUP_HOST_SCHUR": "1"}, {"MIOSQP_SETUP_HOST_RUIZ": "1"},
                {"MIOSQP_SETUP_ROUNDTRIP": "1", "MIOSQP_SETUP_HOST_SCHUR": "1", "MIOSQP_SETUP_HOST_RUIZ": "1"}):
        os.environ.update(env)
        try:
            g = qp.OSQP()
            g.setup(pr["P"], pr["q"], A, l, u, setup_on_device=1, resident=0, coop=0, fold=0, **problems.QP_SETTINGS)
            g.warm_start(x=x0, y=y0)
            variants.append(g.debug_iterate(40))
        finally:
            for k in env:
                del os.environ[k]
    for got in variants[1:]:
        for a, b in zip(variants[0], got):
            np.testing.assert_array_equal(a, b)
    for a, b in zip(variants[0], ref):
        assert rel(a, b) <= ITER_TOL


def _infeasible_variants(pr):
    """(P, q, A, l, u, expected status) built from a random instance: a duplicated constraint row with a
    disjoint interval (primal infeasible), a free variable with zero curvature and a linear cost
    (dual infeasible)."""
    import scipy.sparse as spa
    A, l, u = problems.extended(pr)
    A = spa.csc_matrix(A)
    n = A.shape[1]
    row0 = A.getrow(0)
    Ap = spa.vstack([A, row0]).tocsc()
    lp, up = np.append(l, u[0] + 1.0), np.append(u, u[0] + 2.0)
    yield pr["P"], pr["q"], Ap, lp, up, -3
    k = [i for i in range(n) if i not in set(pr["i_idx"])][0]
    keep = spa.diags([0.0 if i == k else 1.0 for i in range(n)])
    Pd = (keep @ spa.csc_matrix(pr["P"]) @ keep).tocsc()
    Ad = (A @ keep).tocsc()
    qd = np.array(pr["q"], dtype=float)
    qd[k] = 1.0
    yield Pd, qd, Ad, l, u, -4


@pytest.mark.parametrize("coop", [0, 1])
@pytest.mark.parametrize("n,m,p,seed", [(60, 120, 30, 5), (300, 600, 150, 6)])
def test_certificates_on_random_instances(oracle_mod, n, m, p, seed, coop):
    """Infeasibility detection (OSQP paper sec. 3.4) on instances large enough for every engine form;
    status, iteration and the normalised certificate equal the oracle's."""
    from miosqp_amd import qp
    pr = problems.random_miqp(n, m, p, seed=seed)
    for (P, q, A, l, u, code) in _infeasible_variants(pr):
        g, o = qp.OSQP(), oracle_mod.OSQP()
        g.setup(P, q, A, l, u, coop=coop, resident=0, **problems.QP_SETTINGS)
        o.setup(P, q, A, l, u, **problems.QP_SETTINGS)
        assert g.factor_stats()["coop"] == bool(coop)
        z0, w0 = np.zeros(A.shape[1]), np.zeros(A.shape[0])
        g.warm_start(x=z0, y=w0)
        o.warm_start(x=z0, y=w0)
        rg, ro = g.solve(), o.solve()
        assert ro.info.status_val == code
        assert (rg.info.status_val, rg.info.iter) == (ro.info.status_val, ro.info.iter)
        np.testing.assert_allclose(rg.x, ro.x, rtol=1e-6, atol=1e-8, equal_nan=True)
        np.testing.assert_allclose(rg.y, ro.y, rtol=1e-6, atol=1e-8, equal_nan=True)


@pytest.mark.parametrize("n,m,p,seed", [(100, 150, 40, 7), (300, 500, 150, 8), (500, 1000, 250, 0)])
def test_cooperative_solver_equals_two_kernel_form(oracle_mod, n, m, p, seed):
    """The cooperative launch (auto for 64 <= n + M <= 2048) against the product-form kernels on a chain
    of node relaxations: same status and iteration count, solutions within SOL_TOL, identical digests; and
    against the oracle on raw iterates.  Covers both register layouts (n + M <= 1024 and above) and many
    consecutive launches (the exchange tags run on across launches)."""
    from miosqp_amd import qp
    pr = problems.random_miqp(n, m, p, seed=seed)
    A, l, u = problems.extended(pr)
    M = A.shape[0]
    eng = []
    for coop in (1, 0):
        g = qp.OSQP()
        g.setup(pr["P"], pr["q"], A, l, u, coop=coop, resident=0, **problems.QP_SETTINGS)
        assert g.factor_stats()["coop"] == bool(coop)
        g.set_integer_rows(pr["i_idx"], pr["A"].shape[0])
        g.set_root(l, u, 1e-3, 1e-3)
        eng.append(g)
    o = oracle_mod.OSQP()
    o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
    rng = np.random.RandomState(seed)
    x0, y0 = rng.randn(n), rng.randn(M)
    for k in (1, 2, 26, 75):
        o.warm_start(x=x0, y=y0)
        o.iterate(k)
        xo, zo, yo = o.iterates()
        eng[0].warm_start(x=x0, y=y0)
        xg, zg, yg = eng[0].debug_iterate(k)
        assert rel(xg, xo) <= ITER_TOL and rel(zg, zo) <= ITER_TOL and rel(yg, yo) <= ITER_TOL, k
    x, y = np.zeros(n), np.zeros(M)
    lo, hi = l.copy(), u.copy()
    for depth in range(6):
        ra, rb = (g.solve_node(lo, hi, x, y) for g in eng)
        assert (ra.status_val, ra.iter) == (rb.status_val, rb.iter), depth
        if ra.status_val not in (1, -2):
            break
        assert rel(ra.x, rb.x) <= SOL_TOL and rel(ra.y, rb.y) <= SOL_TOL
        assert abs(ra.lower - rb.lower) <= 1e-9 * max(1.0, abs(rb.lower))
        assert (ra.digest.int_inf, ra.digest.nextvar) == (rb.digest.int_inf, rb.digest.nextvar)
        if ra.digest.int_inf == 0:
            break
        # branch down on the chosen variable, as Workspace.branch does (workspace.py:143-203)
        j = pr["A"].shape[0] + ra.digest.nextvar
        hi = hi.copy()
        hi[j] = np.floor(ra.x[pr["i_idx"][ra.digest.nextvar]])
        x, y = ra.x, ra.y


@pytest.mark.parametrize("n,m,p", [(20, 39, 4), (20, 40, 4), (21, 38, 6), (40, 80, 8), (60, 120, 12), (60, 121, 12),
                                    (300, 600, 124), (300, 600, 125), (600, 1100, 348),
                                    (600, 1100, 349), (700, 1000, 341), (1000, 40, 8), (30, 900, 20)])
def test_engine_form_boundaries(oracle_mod, n, m, p):
    """Sizes on the edges of the automatic choice (n+M <= 192: one workgroup -- its loop on the explicit inverse in
    registers, in 2 or 4 parts per row: 64 / 128 / 192 are edges of that layout; cooperative from 193;
    1024 / 1025: two register layouts; 2041…2048: the last full grid of 256 workgroups; 2049: back to the
    two-kernel form) and lopsided shapes (few constraints, few variables, odd widths)."""
    from miosqp_amd import qp
    pr = problems.random_miqp(n, m, p, seed=n + m)
    A, l, u = problems.extended(pr)
    N = n + A.shape[0]
    g, o = qp.OSQP(), oracle_mod.OSQP()
    g.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
    o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
    fs = g.factor_stats()
    assert fs["coop"] == (192 < N <= 2048) and fs["resident"] == (N <= 192), (N, fs)
    rng = np.random.RandomState(N)
    x0, y0 = 0.1 * rng.randn(n), 0.1 * rng.randn(A.shape[0])
    for k in (1, 27):
        g.warm_start(x=x0, y=y0)
        o.warm_start(x=x0, y=y0)
        xg, zg, yg = g.debug_iterate(k)
        o.iterate(k)
        xo, zo, yo = o.iterates()
        assert rel(xg, xo) <= ITER_TOL and rel(zg, zo) <= ITER_TOL and rel(yg, yo) <= ITER_TOL, (N, k)
    g.warm_start(x=x0, y=y0)
    o.warm_start(x=x0, y=y0)
    rg, ro = g.solve(), o.solve()
    assert (rg.info.status_val, rg.info.iter) == (ro.info.status_val, ro.info.iter), N
    assert rel(rg.x, ro.x) <= SOL_TOL and rel(rg.y, ro.y) <= SOL_TOL


@pytest.mark.parametrize("n,m,p,seed", [(60, 120, 30, 1), (300, 600, 124, 2), (300, 600, 125, 3), (500, 1000, 250, 0), (600, 1100, 290, 4)])
def test_termination_test_on_tester_workgroups_equals_the_test_inside_the_grid(n, m, p, seed, monkeypatch):
    """k_coop with its termination test on tester workgroups (decision picked up `lag` iterations later, the owners'
    iterates of the test iteration restored) against the same solver with the test inside the grid
    (MIOSQP_COOP_TESTERS=0): same status, same iteration count, bitwise the same x and y -- the iterates do not depend on
    where the norms are reduced -- for several lags, both register layouts (n+M <= 1024 and above), a node that hits
    the iteration limit between two tests, and an infeasible node."""
    from miosqp_amd import qp
    pr = problems.random_miqp(n, m, p, seed=seed)
    A, l, u = problems.extended(pr)
    monkeypatch.setenv("MIOSQP_COOP_TESTERS", "0")
    ref = qp.OSQP()
    ref.setup(pr["P"], pr["q"], A, l, u, coop=1, resident=0, **problems.QP_SETTINGS)
    monkeypatch.delenv("MIOSQP_COOP_TESTERS")
    rng = np.random.RandomState(seed)
    x0, y0 = 0.1 * rng.randn(n), 0.1 * rng.randn(A.shape[0])
    ref.set_integer_rows(pr["i_idx"], m)
    l2, u2 = l.copy(), u.copy()
    k = m + p // 2
    l2[k], u2[k] = 1.0, 1.0   # one binary fixed
    l3, u3 = l.copy(), u.copy()
    l3[:m] = u3[:m] = 5.0     # every general row pinned outside its reach: (very likely) primal infeasible
    cases = [(l, u), (l2, u2), (l3, u3)]
    want = [ref.solve_node(a, b, x0, y0) for a, b in cases]
    for lag in (3, 12, 20):
        monkeypatch.setenv("MIOSQP_COOP_LAG", str(lag))
        g = qp.OSQP()
        g.setup(pr["P"], pr["q"], A, l, u, coop=1, resident=0, **problems.QP_SETTINGS)
        g.set_integer_rows(pr["i_idx"], m)
        assert g.factor_stats()["coop"]
        for (a, b), w in zip(cases, want):
            r = g.solve_node(a, b, x0, y0)
            assert (r.status_val, r.iter) == (w.status_val, w.iter), (lag, w.status_val, w.iter)
            np.testing.assert_array_equal(r.x, w.x)
            np.testing.assert_array_equal(r.y, w.y)
    # the iteration limit falls between two tests: the last test is the final one, waited for at once
    monkeypatch.setenv("MIOSQP_COOP_LAG", "12")
    for max_iter in (60, 101, 110):
        out = []
        for testers in ("0", None):
            if testers is not None:
                monkeypatch.setenv("MIOSQP_COOP_TESTERS", testers)
            g = qp.OSQP()
            g.setup(pr["P"], pr["q"], A, l, u, coop=1, resident=0, **dict(problems.QP_SETTINGS, max_iter=max_iter))
            g.set_integer_rows(pr["i_idx"], m)
            out.append(g.solve_node(l, u, x0, y0))
            if testers is not None:
                monkeypatch.delenv("MIOSQP_COOP_TESTERS")
        assert (out[0].status_val, out[0].iter) == (out[1].status_val, out[1].iter)
        np.testing.assert_array_equal(out[0].x, out[1].x)


def test_randomised_sweep_over_shapes_forms_and_bounds(oracle_mod):
    """60 small random instances: random shape and density, random engine form, random warm start, bounds
    randomly tightened (some relaxations become infeasible): status, iteration count and solution equal
    the oracle's every time."""
    import scipy.sparse as spa
    from miosqp_amd import qp
    rng = np.random.RandomState(2024)
    seen = set()
    for trial in range(60):
        n = int(rng.randint(2, 45))
        m = int(rng.randint(1, 90))
        p = int(rng.randint(0, n + 1))
        dens = float(rng.choice([0.05, 0.2, 0.7, 1.0]))
        pr = problems.random_miqp(n, m, p, density=dens, seed=1000 + trial)
        A, l, u = problems.extended(pr)
        A = spa.csc_matrix(A)
        M = A.shape[0]
        form = [dict(fold=0, resident=0, coop=0), dict(fold=1, resident=0, coop=0), dict(fold=1, resident=1, coop=0),
                dict(coop=1, resident=0), dict()][int(rng.randint(0, 5))]
        l, u = l.copy(), u.copy()
        for j in rng.choice(M, size=min(M, 3), replace=False):  # tighten or cross a few intervals' centres
            c0 = 0.5 * (l[j] + u[j]) if np.isfinite(l[j]) and np.isfinite(u[j]) else 0.0
            w = float(rng.choice([0.0, 0.05, 1.0]))
            l[j], u[j] = c0 - w + rng.randn() * 0.5, c0 + w + rng.randn() * 0.5
            if l[j] > u[j]:
                l[j], u[j] = u[j], l[j]
        g, o = qp.OSQP(), oracle_mod.OSQP()
        g.setup(pr["P"], pr["q"], A, l, u, **form, **problems.QP_SETTINGS)
        o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
        x0, y0 = rng.randn(n) * rng.choice([0.0, 1.0]), rng.randn(M) * rng.choice([0.0, 1.0])
        g.warm_start(x=x0, y=y0)
        o.warm_start(x=x0, y=y0)
        rg, ro = g.solve(), o.solve()
        fs = g.factor_stats()
        seen.add((fs["fold"], fs["resident"], fs["coop"], ro.info.status_val))
        assert (rg.info.status_val, rg.info.iter) == (ro.info.status_val, ro.info.iter), (trial, n, m, p, dens, form)
        np.testing.assert_allclose(rg.x, ro.x, rtol=1e-6, atol=1e-7, equal_nan=True, err_msg=str((trial, form)))
        np.testing.assert_allclose(rg.y, ro.y, rtol=1e-6, atol=1e-7, equal_nan=True, err_msg=str((trial, form)))
    assert len({s[:3] for s in seen}) >= 4 and len({s[3] for s in seen}) >= 2, seen


def test_persistent_solver_random_sweep(oracle_mod):
    """80 random instances on the persistent streaming solver, both factor forms (and so all three kernels: lean product
    form, general product form for the larger ones, factor form): shapes from 4 variables up -- rows shorter than a wave,
    odd sizes, workgroups without a constraint row --, densities from 5 % to full, random warm starts, bounds randomly
    tightened or opened to infinity (infeasible relaxations among them): status, iteration count and solution equal
    the oracle's.  (The first version read its LDS operand out of range for segments shorter than 128 columns: any such
    slip shows up here as a NaN or a wrong count.)"""
    import scipy.sparse as spa
    from miosqp_amd import qp
    rng = np.random.RandomState(77)
    kinds = set()
    for trial in range(80):
        big = trial % 10 == 9
        n = int(rng.randint(150, 400)) if big else int(rng.randint(4, 70))
        m = int(rng.randint(100, 900)) if big else int(rng.randint(1, 120))
        p = int(rng.randint(0, n + 1))
        dens = float(rng.choice([0.05, 0.2, 0.7, 1.0]))
        pr = problems.random_miqp(n, m, p, density=dens, seed=5000 + trial)
        A, l, u = problems.extended(pr)
        A = spa.csc_matrix(A)
        M = A.shape[0]
        l, u = l.copy(), u.copy()
        for j in rng.choice(M, size=min(M, 4), replace=False):
            mode = int(rng.randint(0, 4))
            if mode == 0:
                u[j] = np.inf
            elif mode == 1:
                l[j] = -np.inf
            else:
                c0 = 0.5 * (l[j] + u[j]) if np.isfinite(l[j]) and np.isfinite(u[j]) else 0.0
                w = float(rng.choice([0.0, 0.05, 1.0]))
                l[j], u[j] = c0 - w + rng.randn() * 0.5, c0 + w + rng.randn() * 0.5
                if l[j] > u[j]:
                    l[j], u[j] = u[j], l[j]
        fold = int(rng.randint(0, 2))
        g, o = qp.OSQP(), oracle_mod.OSQP()
        g.setup(pr["P"], pr["q"], A, l, u, fold=fold, resident=0, coop=0, pers=1 + int(rng.randint(0, 2)), **problems.QP_SETTINGS)
        o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
        fs = g.factor_stats()
        assert fs["pers"] is True, (trial, n, m, p)
        x0, y0 = rng.randn(n) * rng.choice([0.0, 1.0]), rng.randn(M) * rng.choice([0.0, 1.0])
        g.warm_start(x=x0, y=y0)
        o.warm_start(x=x0, y=y0)
        rg, ro = g.solve(), o.solve()
        kinds.add((fold, ro.info.status_val))
        assert (rg.info.status_val, rg.info.iter) == (ro.info.status_val, ro.info.iter), (trial, n, m, p, dens, fold)
        np.testing.assert_allclose(rg.x, ro.x, rtol=1e-6, atol=1e-7, equal_nan=True, err_msg=str((trial, fold)))
        np.testing.assert_allclose(rg.y, ro.y, rtol=1e-6, atol=1e-7, equal_nan=True, err_msg=str((trial, fold)))
        g.close()
    assert len({k[0] for k in kinds}) == 2 and len({k[1] for k in kinds}) >= 2, kinds


def test_cooperative_solver_reports_a_missing_workgroup_and_recovers(oracle_mod, monkeypatch):
    """Fault injection: one workgroup of the cooperative launch never starts (what a co-tenant on the device
    does to it).  Workgroup 0 calls the launch off after ~100 ms, before any iterate has been touched; the
    SAME call -- `solve` (the reference's four-call path) as well as `solve_node` -- is redone in the
    two-kernel form with one line on stderr, bit-identical to an engine that never was cooperative; the
    engine stays in that form and tries the cooperative one again 256 solves later (and falls back again
    here, the fault being permanent); an engine set up afterwards works."""
    import time
    from miosqp_amd import qp
    pr = problems.random_miqp(60, 120, 30, seed=11)
    A, l, u = problems.extended(pr)
    monkeypatch.setenv("MIOSQP_COOP_NAP", "12")  # calibration would hit the fault first
    monkeypatch.setenv("MIOSQP_COOP_DBG", "64")
    bad = qp.OSQP()
    bad.setup(pr["P"], pr["q"], A, l, u, coop=1, resident=0, **problems.QP_SETTINGS)
    bad2 = qp.OSQP()
    bad2.setup(pr["P"], pr["q"], A, l, u, coop=1, resident=0, **problems.QP_SETTINGS)
    monkeypatch.delenv("MIOSQP_COOP_DBG")
    ref = qp.OSQP()
    ref.setup(pr["P"], pr["q"], A, l, u, coop=0, resident=0, **problems.QP_SETTINGS)
    x0, y0 = np.zeros(60), np.zeros(A.shape[0])
    assert bad.factor_stats()["coop"] is True
    for s_ in (bad, ref):
        s_.warm_start(x=x0, y=y0)
    t0 = time.time()
    r_bad = bad.solve()
    assert time.time() - t0 < 1.5  # ~0.1 s of waiting, not the 2 s of a hung exchange
    r_ref = ref.solve()
    fs = bad.factor_stats()
    assert fs["coop"] is False and fs["coop_fallbacks"] == 1
    assert (r_bad.info.status_val, r_bad.info.iter) == (r_ref.info.status_val, r_ref.info.iter)
    np.testing.assert_array_equal(r_bad.x, r_ref.x)
    np.testing.assert_array_equal(r_bad.y, r_ref.y)
    for s_ in (bad, ref):
        s_.set_integer_rows(pr["i_idx"], pr["A"].shape[0])
    ra, rb = bad.solve_node(l, u, x0, y0), ref.solve_node(l, u, x0, y0)
    assert (ra.status_val, ra.iter) == (rb.status_val, rb.iter)
    np.testing.assert_array_equal(ra.x, rb.x)
    # 256 solves later the engine tries the cooperative form again, is called off again, and still answers
    for k in range(260):
        ra2 = bad.solve_node(l, u, ra.x, ra.y)
    rb2 = ref.solve_node(l, u, rb.x, rb.y)
    assert (ra2.status_val, ra2.iter) == (rb2.status_val, rb2.iter)
    np.testing.assert_array_equal(ra2.x, rb2.x)
    fs = bad.factor_stats()
    assert fs["coop"] is False and fs["coop_fallbacks"] == 2
    # the iterate-level debug entry falls back the same way
    rng = np.random.RandomState(5)
    xw, yw = rng.randn(60), rng.randn(A.shape[0])
    for s_ in (bad2, ref):
        s_.warm_start(x=xw, y=yw)
    xs, zs, ys = bad2.debug_iterate(3)
    xr, zr, yr = ref.debug_iterate(3)
    assert bad2.factor_stats()["coop_fallbacks"] == 1
    np.testing.assert_array_equal(xs, xr)
    np.testing.assert_array_equal(ys, yr)
    g, o = qp.OSQP(), oracle_mod.OSQP()
    g.setup(pr["P"], pr["q"], A, l, u, coop=1, resident=0, **problems.QP_SETTINGS)
    o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
    rg, ro = g.solve(), o.solve()
    assert (rg.info.status_val, rg.info.iter) == (ro.info.status_val, ro.info.iter)
    assert rel(rg.x, ro.x) <= SOL_TOL


def test_persistent_solver_with_resident_columns_of_the_tail_inverse(oracle_mod, monkeypatch):
    """Factor form with the tail as S^-1 (pers = 2): the last columns of a workgroup's rows of S^-1 stay in its LDS for
    the whole launch and only the rest is streamed every iteration (config 5: 392 of 5000 columns).  At sizes a test can
    afford the space is normally taken by the termination test's y and dy; MIOSQP_PERS_TEST_SPARSE=1 frees it.  Odd and
    even n, rows of S^-1 shorter and longer than the resident block can cut, against the oracle and against the same
    engine without the resident block (MIOSQP_PERS_RESIDENT=0): status, iterations, x, y."""
    from miosqp_amd import qp
    monkeypatch.setenv("MIOSQP_PERS_TEST_SPARSE", "1")
    monkeypatch.setenv("MIOSQP_PERS_SYM", "0")  # (the row-streaming form of S^-1: the symmetric tiles keep no resident block)
    rng = np.random.RandomState(99)
    seen = 0
    for trial, (n, m, p) in enumerate([(257, 300, 40), (300, 450, 100), (301, 520, 33), (384, 200, 50), (511, 700, 120),
                                       (640, 900, 200), (777, 400, 10), (130, 300, 20), (900, 300, 60)]):
        pr = problems.random_miqp(n, m, p, density=0.3, seed=7000 + trial)
        A, l, u = problems.extended(pr)
        M = A.shape[0]
        o = oracle_mod.OSQP()
        o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
        x0, y0 = 0.1 * rng.randn(n), 0.1 * rng.randn(M)
        o.warm_start(x=x0, y=y0)
        ro = o.solve()
        res = []
        for keep in ("1", "0"):
            monkeypatch.setenv("MIOSQP_PERS_RESIDENT", keep)
            g = qp.OSQP()
            g.setup(pr["P"], pr["q"], A, l, u, fold=0, resident=0, coop=0, pers=2, **problems.QP_SETTINGS)
            fs = g.factor_stats()
            assert fs["pers"] and fs["tail_inverse"], (n, m, p)
            cols = g.tail_inverse_resident_columns()
            assert (cols == 0) if keep == "0" else (cols == 0 or 128 <= cols <= n // 2 + 1), (n, cols)
            seen += keep == "1" and cols > 0
            g.warm_start(x=x0, y=y0)
            rg = g.solve()
            assert (rg.info.status_val, rg.info.iter) == (ro.info.status_val, ro.info.iter), (n, m, p, keep)
            assert rel(rg.x, ro.x) <= SOL_TOL and rel(rg.y, ro.y) <= SOL_TOL, (n, m, p, keep)
            res.append(rg)
            g.close()
        assert rel(res[0].x, res[1].x) <= 1e-10
    assert seen >= 5  # (the small ones have no room for a block of 128 columns within half a row)


def test_persistent_solver_reads_the_tail_inverse_as_symmetric_tiles(oracle_mod, monkeypatch):
    """Factor form with the tail as S^-1 (pers = 2), n even: S^-1 is symmetric, and the persistent solver reads only its
    tiles on and above the diagonal -- one per workgroup, T (T + 1) / 2 of them, half the bytes per iteration --, the row
    sums of a tile going to the rows' block and its column sums to the columns' block (kernels_pers.inc, pers_sym_rows).
    Tile edges from 6 to 246 columns (one, two and three 128-column pieces per lane and row, last tile shorter, rows per
    wave from one to 31), against the oracle where the CPU can afford it and against the same engine streaming whole rows
    (MIOSQP_PERS_SYM=0): status, iterations, x, y; the tiles with every wave's first rows resident in LDS (the default) and
    with every row streamed (MIOSQP_PERS_RESIDENT=0) agree bit for bit.  Odd n keeps the rows."""
    from miosqp_amd import qp
    rng = np.random.RandomState(123)
    pieces = set()
    for trial, (n, m, p, dens) in enumerate([(130, 300, 20, 0.3), (300, 450, 100, 0.3), (301, 520, 33, 0.3), (384, 200, 50, 0.3),
                                             (640, 900, 200, 0.3), (900, 300, 60, 0.3), (2600, 700, 100, 0.02), (5400, 300, 40, 0.004)]):
        pr = problems.random_miqp(n, m, p, density=dens, seed=7100 + trial)
        A, l, u = problems.extended(pr)
        M = A.shape[0]
        x0, y0 = 0.1 * rng.randn(n), 0.1 * rng.randn(M)
        ro = None
        if n <= 900:
            o = oracle_mod.OSQP()
            o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
            o.warm_start(x=x0, y=y0)
            ro = o.solve()
        res = []
        for sym, keep in (("1", "1"), ("0", "0"), ("1", "0")):
            monkeypatch.setenv("MIOSQP_PERS_SYM", sym)
            monkeypatch.setenv("MIOSQP_PERS_RESIDENT", keep)
            g = qp.OSQP()
            g.setup(pr["P"], pr["q"], A, l, u, fold=0, resident=0, coop=0, pers=2, **problems.QP_SETTINGS)
            fs = g.factor_stats()
            assert fs["pers"] and fs["tail_inverse"], (n, m, p)
            tiles = g.tail_inverse_tiles()
            if sym == "1" and n % 2 == 0:
                G, T = min(256, n // 2), 1  # (one workgroup per CU, at least a pair of variable rows each)
                while (T + 1) * (T + 2) // 2 <= G:
                    T += 1
                C = 2 * ((n + 2 * T - 1) // (2 * T))
                T = (n + C - 1) // C
                assert tiles == T * (T + 1) // 2 and tiles <= 256, (n, tiles)
                pieces.add((C + 14 + 127) // 128)
            else:
                assert tiles == 0, (n, sym, tiles)
            g.warm_start(x=x0, y=y0)
            rg = g.solve()
            if ro is not None:
                assert (rg.info.status_val, rg.info.iter) == (ro.info.status_val, ro.info.iter), (n, m, p, sym)
                assert rel(rg.x, ro.x) <= SOL_TOL and rel(rg.y, ro.y) <= SOL_TOL, (n, m, p, sym)
            res.append((rg, fs["bytes_moved_per_iter"]))
            g.close()
        assert (res[0][0].info.status_val, res[0][0].info.iter) == (res[1][0].info.status_val, res[1][0].info.iter), (n, m, p)
        assert rel(res[0][0].x, res[1][0].x) <= 1e-9 and rel(res[0][0].y, res[1][0].y) <= 1e-9, (n, m, p)
        if n % 2 == 0:  # resident rows or not: the same sums in the same order
            np.testing.assert_array_equal(res[0][0].x, res[2][0].x)
            np.testing.assert_array_equal(res[0][0].y, res[2][0].y)
        if n % 2 == 0 and n >= 2000:  # (the dense tail is most of what moves: about half of it is gone, less what stays in LDS)
            assert res[0][1] < res[2][1] < 0.62 * res[1][1], (n, res[0][1], res[2][1], res[1][1])
    assert pieces == {1, 2, 3}


def test_persistent_solver_reports_a_missing_workgroup_and_recovers(monkeypatch):
    """The fault of the test above on the persistent streaming solver (both kernels: the lean product-form one and the
    general one in factor form): the launch is called off before any iterate is touched, the same call is redone in the
    multi-kernel form -- bit-identical to an engine that never was persistent --, the engine stays there."""
    import time
    from miosqp_amd import qp
    pr = problems.random_miqp(60, 120, 30, seed=11)
    A, l, u = problems.extended(pr)
    x0, y0 = np.zeros(60), np.zeros(A.shape[0])
    for fold in (1, 0):
        monkeypatch.setenv("MIOSQP_COOP_DBG", "64")
        bad = qp.OSQP()
        bad.setup(pr["P"], pr["q"], A, l, u, fold=fold, coop=0, resident=0, pers=1, **problems.QP_SETTINGS)
        monkeypatch.delenv("MIOSQP_COOP_DBG")
        ref = qp.OSQP()
        ref.setup(pr["P"], pr["q"], A, l, u, fold=fold, coop=0, resident=0, pers=0, **problems.QP_SETTINGS)
        assert bad.factor_stats()["pers"] is True and ref.factor_stats()["pers"] is False
        for s_ in (bad, ref):
            s_.warm_start(x=x0, y=y0)
        t0 = time.time()
        r_bad = bad.solve()
        assert time.time() - t0 < 1.5
        r_ref = ref.solve()
        fs = bad.factor_stats()
        assert fs["pers"] is False and fs["coop_fallbacks"] == 1
        assert (r_bad.info.status_val, r_bad.info.iter) == (r_ref.info.status_val, r_ref.info.iter)
        np.testing.assert_array_equal(r_bad.x, r_ref.x)
        np.testing.assert_array_equal(r_bad.y, r_ref.y)
        bad.close()
        ref.close()


def test_persistent_kernels_agree_and_take_sizes_beyond_the_cooperative_grid(oracle_mod, monkeypatch):
    """(a) The lean product-form kernel (one dense task per wave and phase) and the general kernel run the same arithmetic in
    the same order: bit-identical iterates and solutions.  (b) n + M = 2450 > 2048 -- beyond what the cooperative solver
    holds in registers -- takes the persistent form by itself, in either factor form, and matches the oracle: status,
    iteration count, x, y (iterates after 1, 10 and 40 iterations to 1e-9)."""
    from miosqp_amd import qp
    pr = problems.random_miqp(130, 260, 65, seed=3)
    A, l, u = problems.extended(pr)
    lean = qp.OSQP()
    lean.setup(pr["P"], pr["q"], A, l, u, fold=1, coop=0, resident=0, pers=1, **problems.QP_SETTINGS)
    monkeypatch.setenv("MIOSQP_PERS_GENERAL", "1")
    gen = qp.OSQP()
    gen.setup(pr["P"], pr["q"], A, l, u, fold=1, coop=0, resident=0, pers=1, **problems.QP_SETTINGS)
    monkeypatch.delenv("MIOSQP_PERS_GENERAL")
    rng = np.random.RandomState(1)
    xw, yw = rng.randn(130), rng.randn(A.shape[0])
    for k in (1, 7, 60):
        for s_ in (lean, gen):
            s_.warm_start(x=xw, y=yw)
        a, b = lean.debug_iterate(k), gen.debug_iterate(k)
        for va, vb in zip(a, b):
            np.testing.assert_array_equal(va, vb)
    for s_ in (lean, gen):
        s_.warm_start(x=np.zeros(130), y=np.zeros(A.shape[0]))
    ra, rb = lean.solve(), gen.solve()
    assert (ra.info.status_val, ra.info.iter) == (rb.info.status_val, rb.info.iter)
    np.testing.assert_array_equal(ra.x, rb.x)
    np.testing.assert_array_equal(ra.y, rb.y)
    lean.close()
    gen.close()
    pr = problems.random_miqp(700, 1400, 350, density=0.3, seed=5)
    A, l, u = problems.extended(pr)
    n, M = 700, A.shape[0]
    o = oracle_mod.OSQP()
    o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
    xw, yw = rng.randn(n), rng.randn(M)
    for fold in (1, 0):
        g = qp.OSQP()
        g.setup(pr["P"], pr["q"], A, l, u, fold=fold, **problems.QP_SETTINGS)
        fs = g.factor_stats()
        assert fs["pers"] is True and fs["coop"] is False and fs["fold"] == bool(fold)
        assert fs["tail_inverse"] == (not fold)  # the factor form's automatic choice: its dense tail as S^-1
        for k in (1, 10, 40):
            g.warm_start(x=xw, y=yw)
            o.warm_start(x=xw, y=yw)
            xg, zg, yg = g.debug_iterate(k)
            o.iterate(k)
            xo, zo, yo = o.iterates()
            assert rel(xg, xo) <= ITER_TOL and rel(zg, zo) <= ITER_TOL and rel(yg, yo) <= ITER_TOL, (fold, k)
        g.warm_start(x=np.zeros(n), y=np.zeros(M))
        o.warm_start(x=np.zeros(n), y=np.zeros(M))
        rg, ro = g.solve(), o.solve()
        assert (rg.info.status_val, rg.info.iter) == (ro.info.status_val, ro.info.iter), fold
        assert rel(rg.x, ro.x) <= SOL_TOL and rel(rg.y, ro.y) <= SOL_TOL
        g.close()


def test_engine_driven_from_a_worker_thread():
    """the HIP current device is per host thread; every entry point makes the engine's device current,
    so a wave can be solved from a worker thread (the pipelined mode of ShardedSearch does exactly that).  One GPU
    here, so this checks the mechanism (first use of the batched path -- allocation, graph capture -- happens on
    the worker thread), not a device > 0."""
    from concurrent.futures import ThreadPoolExecutor
    from miosqp_amd import qp
    pr = problems.random_miqp(50, 100, 25, seed=2)
    A, l, u = problems.extended(pr)
    g = qp.OSQP()
    g.setup(pr["P"], pr["q"], A, l, u, max_batch=64, **problems.QP_SETTINGS)
    g.set_integer_rows(pr["i_idx"], 100)
    L, U = np.stack([l] * 3), np.stack([u] * 3)
    U[1, 100] = 0.0
    L[2, 101] = 1.0
    X, Y = np.zeros((3, 50)), np.zeros((3, A.shape[0]))
    with ThreadPoolExecutor(max_workers=1) as pool:
        rb = pool.submit(g.solve_batch, L, U, X, Y).result()
        r1 = pool.submit(g.solve_node, L[1], U[1], X[1], Y[1]).result()
    r0 = g.solve_node(L[0], U[0], X[0], Y[0])
    assert (rb.status_val[1], rb.iter[1]) == (r1.status_val, r1.iter)
    assert (rb.status_val[0], rb.iter[0]) == (r0.status_val, r0.iter)
    assert rel(rb.x[1], r1.x) <= SOL_TOL and rel(rb.x[0], r0.x) <= SOL_TOL


def test_single_rank_torchrun_goes_through_rccl(tmp_path):
    """`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` takes the RCCL path
    (process group "nccl", device tensors in barrier / all-reduce / all-gather / broadcast) and reports the same
    work as the plain single-process run: identical node and iteration counts, a rate in the same range."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--gpus", "1", "--steps", "30", "--warmup", "5", "--legs", "none"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    plain = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, env=env, cwd=root,
                           capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0, plain.stderr[-2000:]
    a = json.loads(plain.stdout.strip().splitlines()[-1])
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env2 = dict(env, MIOSQP_FORCE_EXCHANGE="1")  # the incumbent exchange runs although there is one rank
    tr = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py")]
                        + common, env=env2, cwd=root, capture_output=True, text=True, timeout=600)
    assert tr.returncode == 0, tr.stderr[-2000:]
    b = json.loads([ln for ln in tr.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert a["config"]["comm"] == "LocalComm" and b["config"]["comm"] == "TorchComm/nccl"
    # the forced exchange runs the sharded form (replicated ramp-up, deal, all-gathers every step): the same number of
    # nodes per step, another visiting order
    assert a["nodes"] == b["nodes"]
    assert 0.5 * a["iters_per_node"] <= b["iters_per_node"] <= 2.0 * a["iters_per_node"]
    assert 0.5 * a["value"] <= b["value"] <= 1.5 * a["value"]
    assert b["n_gpus"] == 1 and b["roofline"]["kernel"] == a["roofline"]["kernel"]


def _stream_checker(pr, stride, seen, ref=None, oracle_mod=None, oracle_stride=0, oracle_seen=None):
    """observer for StreamSearch: every `stride`-th decided node is replayed through solve_node on a SECOND engine
    from what the pool holds for it (its integer-row bounds, its parent's solution as warm start) and must
    match the digest: status, iterations, bound, integrality count, branching variable, heuristic outcome, and
    the stored solution.  `ref`: that second engine when it exists already (a later MIQP of a sequence must go
    through update(q=) like the engine under test: the cost scaling is fixed at setup).
    With `oracle_mod`, every `oracle_stride`-th of those replays ALSO goes through the CPU oracle in the reference's
    call order (node.py:102-143): status, iterations, x after the clamp, y and the bound of what the stream decided
    are then checked against something that is not this engine."""
    from miosqp_amd import qp
    A, l, u = problems.extended(pr)
    m, p = pr["A"].shape[0], len(pr["i_idx"])
    if ref is None:
        ref = qp.OSQP()
        ref.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
        ref.set_integer_rows(pr["i_idx"], m)
        ref.set_root(l, u, 1e-3, 1e-3)
    orc = None
    if oracle_mod is not None and oracle_stride > 0:
        orc = oracle_mod.OSQP()
        orc.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
    count = [0]
    ocount = [0]

    def obs(search, g):
        if int(g["status_val"]) == -100:
            return
        count[0] += 1
        if count[0] % stride:
            return
        s = int(g["slot"])
        par = int(search.parent[s])
        nd = search.eng.pool_read_node(s, p)
        if par >= 0:
            ws = search.eng.pool_read_node(par, p, want=("x", "y"))
            x0, y0 = ws.x, ws.y
        else:
            x0, y0 = np.zeros(len(pr["q"])), np.zeros(len(l))
        l2, u2 = l.copy(), u.copy()
        l2[m:], u2[m:] = nd.l, nd.u
        r = ref.solve_node(l2, u2, x0, y0)
        assert (int(g["status_val"]), int(g["iter"])) == (r.status_val, r.iter), s
        if orc is not None:
            ocount[0] += 1
            if ocount[0] % oracle_stride == 0:
                orc.update(l=l2, u=u2)
                orc.warm_start(x=x0, y=y0)
                ro = orc.solve()
                assert (int(g["status_val"]), int(g["iter"])) == (ro.info.status_val, ro.info.iter), s
                if ro.info.status_val in (1, -2):
                    xo = ro.x.copy()
                    ii = pr["i_idx"]
                    xo[ii] = np.minimum(np.maximum(xo[ii], l2[m:]), u2[m:])
                    lo = 0.5 * xo.dot(pr["P"].dot(xo)) + pr["q"].dot(xo)
                    assert rel(nd.x, xo) <= SOL_TOL and rel(nd.y, ro.y) <= SOL_TOL
                    assert abs(g["lower"] - lo) <= 1e-9 * max(1.0, abs(lo))
                if oracle_seen is not None:
                    oracle_seen.append(s)
        if r.status_val in (1, -2):
            assert abs(g["lower"] - r.lower) <= 1e-9 * max(1.0, abs(r.lower))
            assert rel(nd.x, r.x) <= SOL_TOL and rel(nd.y, r.y) <= SOL_TOL
            assert int(g["int_inf"]) == r.digest.int_inf
            frac = np.sort(np.abs(r.x[pr["i_idx"]] - np.round(r.x[pr["i_idx"]])))
            if len(frac) > 1 and frac[-1] - frac[-2] > 1e-7:
                assert int(g["nextvar"]) == r.digest.nextvar
            if abs(r.digest.info_viol) > 1e-7:
                assert (g["heur_viol"] <= 0) == r.digest.heur_feasible
            assert abs(g["heur_obj"] - r.digest.heur_obj) <= 1e-9 * max(1.0, abs(r.digest.heur_obj))
            # the children the device wrote: this node's bounds with one entry changed (workspace.py:157-203)
            if int(g["int_inf"]) > 0:
                k = int(g["nextvar"])
                xv = nd.x[pr["i_idx"][k]]
                for side, c in enumerate(search.child[s]):
                    ch = search.eng.pool_read_node(int(c), p, want=("l", "u"))
                    el, eu = nd.l.copy(), nd.u.copy()
                    if side == 0:
                        eu[k] = np.floor(xv)
                    else:
                        el[k] = np.ceil(xv)
                    np.testing.assert_array_equal(ch.l, el)
                    np.testing.assert_array_equal(ch.u, eu)
        seen.append(s)
    obs.ref = ref
    return obs


@pytest.mark.parametrize("n,m,p,seed,cols,fold", [(30, 150, 15, 4, 64, -1), (50, 100, 25, 2, 128, -1), (20, 40, 10, 1, 64, -1),
                                                   (30, 150, 15, 4, 64, 0), (40, 60, 20, 7, 192, 0)])
def test_streaming_search_on_the_leaf_pool(n, m, p, seed, cols, fold):
    """Device-resident leaf pool + streaming batch: the search closes the tree with the
    sequential search's optimum; every node it decides equals solve_node on the same inputs; the children it
    generates on the device are the reference's add_left / add_right; a second MIQP on the same factor
    (update_vectors) reuses the pool.  fold = 0: the factor-form batched kernels (four launches per iteration, the
    form config 5 uses) instead of the matrix-core tiles."""
    from miosqp_amd import bnb, stream
    pr = problems.random_miqp(n, m, p, seed=seed)
    st = dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 6)
    seq = bnb.MIOSQP()
    seq.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st),
              dict(problems.QP_SETTINGS))
    r0 = seq.solve()
    model = bnb.MIOSQP()
    model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st),
                dict(problems.QP_SETTINGS, max_batch=cols, fold=fold))
    assert model.work.solver.factor_stats()["fold"] == (fold != 0)
    seen = []
    srch = stream.StreamSearch(model, columns=cols, observer=_stream_checker(pr, 1, seen))
    r1 = srch.run()
    assert r1.status == r0.status == bnb.MI_SOLVED
    assert abs(r1.upper_glob - r0.upper_glob) <= 1e-3 * max(1.0, abs(r0.upper_glob))
    ii = pr["i_idx"]
    np.testing.assert_array_equal(r1.x[ii], r0.x[ii])
    assert len(seen) == srch.nodes >= 1
    assert len(srch.free) == srch.capacity  # every slot came back
    # a second instance on the same factor and the same pool
    rng = np.random.RandomState(seed)
    q2 = rng.randn(n)
    for mdl in (seq, model):
        mdl.update_vectors(q=q2)
    pr2 = dict(pr, q=q2)
    srch.observer.ref.update(q=q2)
    srch.observer = _stream_checker(pr2, 1, seen, ref=srch.observer.ref)
    srch.begin_instance()
    r0, r1 = seq.solve(), srch.run()
    assert r1.status == r0.status == bnb.MI_SOLVED
    assert abs(r1.upper_glob - r0.upper_glob) <= 1e-3 * max(1.0, abs(r0.upper_glob))
    np.testing.assert_array_equal(r1.x[ii], r0.x[ii])


@pytest.mark.parametrize("n,m,p,seed,cols,rule", [(30, 150, 15, 4, 64, 1), (50, 100, 25, 2, 128, 1), (40, 60, 20, 7, 64, 0),
                                                   (60, 120, 40, 3, 64, 1)])
def test_native_stream_driver_equals_the_python_driver(n, m, p, seed, cols, rule):
    """miosqp_qp_stream_* (the host side of the streaming search compiled into the library) against stream.StreamSearch:
    the same logic, the same optimum, every slot returned -- over two MIQPs on one factor; leaves taken out and put
    back.  (Node counts are close, not equal: how many leaves a round pushes depends on how far the launch in flight
    has got when the host looks, for either driver.)"""
    from miosqp_amd import bnb, stream
    pr = problems.random_miqp(n, m, p, seed=seed)
    st = dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 6, tree_explor_rule=rule)
    mods = []
    for _ in range(2):
        mdl = bnb.MIOSQP()
        mdl.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st),
                  dict(problems.QP_SETTINGS, max_batch=cols))
        mods.append(mdl)
    py = stream.StreamSearch(mods[0], columns=cols, capacity=4096)
    cc = stream.NativeStreamSearch(mods[1], columns=cols, capacity=4096)
    ii = pr["i_idx"]
    rng = np.random.RandomState(seed)
    for inst in range(2):
        r0, r1 = py.run(), cc.run()
        assert r1.status == r0.status == bnb.MI_SOLVED
        assert 0.4 * py.nodes <= cc.nodes <= 2.5 * py.nodes and cc.chunks >= 1
        assert abs(r1.upper_glob - r0.upper_glob) <= 1e-9 * max(1.0, abs(r0.upper_glob))
        np.testing.assert_array_equal(r1.x[ii], r0.x[ii])
        assert len(cc.free) == cc.capacity and len(cc.open) == 0 and cc.in_flight == 0
        q2 = rng.randn(n)
        for mdl in mods:
            mdl.update_vectors(q=q2)
        py.begin_instance()
        cc.begin_instance()
    # a few rounds, every open leaf out and in again, then to the end: still the optimum of the Python driver
    for _ in range(3):
        cc.step()
    recs = [cc.give_leaf() for _ in range(cc.givable())]
    assert cc.givable() == 0
    for rec in recs:
        cc.add_leaf(*rec)
    assert cc.givable() == len(recs)
    r0, r1 = py.run(), cc.run()
    assert r1.status == r0.status
    assert abs(r1.upper_glob - r0.upper_glob) <= 1e-3 * max(1.0, abs(r0.upper_glob))
    np.testing.assert_array_equal(r1.x[ii], r0.x[ii])
    assert len(cc.free) == cc.capacity


def test_sharded_stream_on_the_real_pool_one_rank():
    """dist.ShardedStream with one rank (LocalComm): replicated ramp-up on the host, the leaves written into the
    device pool with explicit vectors (add_leaf), then the stream -- the sequential optimum, every slot returned;
    give_leaf / add_leaf round trip: a leaf taken out of the pool and put back is solved to the same result."""
    from miosqp_amd import bnb, dist
    pr = problems.random_miqp(50, 100, 25, seed=2)
    st = dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 6)
    seq = bnb.MIOSQP()
    seq.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st),
              dict(problems.QP_SETTINGS))
    r0 = seq.solve()
    model = bnb.MIOSQP()
    model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st),
                dict(problems.QP_SETTINGS, max_batch=64))
    sh = dist.ShardedStream(model, columns=64, ramp_leaves=6)
    assert sh.total_alive >= 1 and len(sh.ss.open) == sh.total_alive
    # take every dealt leaf out and put it back: explicit vectors survive the trip through the pool
    recs = [sh.ss.give_leaf() for _ in range(len(sh.ss.open))]
    assert len(sh.ss.free) == sh.ss.capacity
    for rec in recs:
        sh.ss.add_leaf(*rec)
    seen = []
    sh.ss.observer = _stream_checker(pr, 1, seen)
    # leaves put in with explicit vectors have no parent slot: the checker would use a zero warm start for them;
    # tell it where their warm start is (their own slot)
    own = set(int(s) for s in sh.ss.open)
    base = sh.ss.observer

    def obs(search, g):
        if int(g["slot"]) in own:
            return
        base(search, g)
    sh.ss.observer = obs
    sh.run()
    w = model.work
    assert w.status == bnb.MI_SOLVED and sh.total_alive == 0 and len(sh.ss.free) == sh.ss.capacity
    assert abs(w.upper_glob - r0.upper_glob) <= 1e-3 * max(1.0, abs(r0.upper_glob))
    np.testing.assert_array_equal(w.x[pr["i_idx"]], r0.x[pr["i_idx"]])
    assert len(seen) >= 1


def test_two_ranks_stream_on_one_device(tmp_path):
    """The multi-rank streaming leg of bench.py with two processes time-sharing GPU 0 (collectives over gloo,
    MIOSQP_BENCH_ONE_DEVICE=1): exercises dist.ShardedStream against the real pool; not a performance number."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", MIOSQP_BENCH_ONE_DEVICE="1")
    tr = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                         "--gpus", "2", "--steps", "20", "--warmup", "5", "--legs", "batched", "--no-probes",
                         "--stream-warmup", "60", "--stream-chunks", "60", "--batch-width", "64",
                         "--batch-waves", "2"],
                        env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert tr.returncode == 0, tr.stderr[-3000:]
    b = json.loads([ln for ln in tr.stdout.strip().splitlines() if ln.startswith("{")][-1])
    bt = b["batched"]
    assert b["n_gpus"] == 2 and "one pool per rank" in bt["form"]
    assert bt["nodes"] > 0 and bt["node_iters_per_s"] > 0 and 0 < bt["column_occupancy"] <= 1.0


def test_two_ranks_hosted_search_on_one_device(tmp_path):
    """The bench's headline form with more than one rank -- dist.ShardedStream over search.HostedSearch: every rank runs
    the compiled node-at-a-time loop on its share of the tree, incumbent all-gather + broadcast and dry-rank feed after
    every step -- as two processes time-sharing GPU 0 (gloo): both ranks end with the optimum of the sequential search
    on the same engine form (value to 1e-9, integer part exactly), status Solved, and together they visited at least
    the sequential search's nodes."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = str(tmp_path / "rec")
    tr = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", str(port),
                         os.path.join(root, "tests", "dist_worker_gpu.py"), out, "60", "120", "30", "3"],
                        env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert tr.returncode == 0, tr.stderr[-3000:]
    recs = [json.load(open("%s.%d" % (out, r))) for r in (0, 1)]
    r0 = recs[0]
    assert r0["seq_status"] == "Solved"
    ii = problems.random_miqp(60, 120, 30, seed=3)["i_idx"]
    for r in recs:
        assert r["status"] == "Solved"
        assert abs(r["upper"] - r0["seq_upper"]) <= 1e-9 * max(1.0, abs(r0["seq_upper"]))
        np.testing.assert_array_equal(np.array(r["x"])[ii], np.array(r0["seq_x"])[ii])
    assert recs[0]["local_nodes"] > 0 and recs[1]["local_nodes"] > 0  # both ranks worked
    assert r0["nodes_total"] >= r0["seq_nodes"]
    # every leaf that changed hands did so as a device tensor: slot store -> broadcast -> slot store, no host hop
    assert r0["moved_total"] >= 1 and r0["moved_dev_total"] == r0["moved_total"]


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["hosted", "native_stream", "python_stream"])
def test_leaves_leave_and_enter_the_slot_store_as_device_tensors(form):
    """The hand-over dist.ShardedStream uses between ranks, in one process (tests/leaf_dev_worker.py: a process of its own
    because torch brings its own HIP runtime, which must be the first to open the device): every open leaf of a search is
    taken out INTO a device tensor (qp.DevicePtr views: the library copies device to device), the tensors are cloned (what
    a broadcast does), the leaves are put back FROM the clones -- and the search ends where the untouched search ends; the
    numpy path moves the same bytes."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tr = subprocess.run([sys.executable, os.path.join(root, "tests", "leaf_dev_worker.py"), form], cwd=root,
                        capture_output=True, text=True, timeout=600)
    assert tr.returncode == 0 and "leaf round trip ok" in tr.stdout, (tr.stdout[-2000:], tr.stderr[-3000:])


@pytest.mark.gpu
def test_bench_with_four_ranks_on_one_device_prints_the_line_the_driver_parses(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 4 bench.py --gpus 4 ...` -- the command of the driver's SCALE step --
    end to end with four processes time-sharing GPU 0 (MIOSQP_BENCH_ONE_DEVICE=1: collectives over gloo; a one-GPU box has
    no second device for RCCL): rank 0 prints ONE JSON line with the contract's keys, `value` is the whole job's rate,
    `n_gpus` 4, weak scaling, the headline config, nodes and trees of the sharded hosted search and the per-rank pools of
    the batched leg.  Not a performance number: four engines share one chip."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", MIOSQP_BENCH_ONE_DEVICE="1")
    tr = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
                         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                         "--gpus", "4", "--steps", "20", "--warmup", "5", "--legs", "batched", "--no-probes",
                         "--stream-warmup", "40", "--stream-chunks", "40", "--batch-width", "64", "--batch-waves", "2"],
                        env=env, cwd=root, capture_output=True, text=True, timeout=1500)
    assert tr.returncode == 0, tr.stderr[-3000:]
    lines = [ln for ln in tr.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 alone prints
    b = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "nodes_per_s", "nodes", "trees"):
        assert key in b, key
    assert b["n_gpus"] == 4 and b["steps"] == 20 and b["warmup"] == 5 and b["scaling"] == "weak" and b["dtype"] == "f64"
    assert b["higher_is_better"] is True and b["vs_baseline"] is None and b["data"] == "synthetic"
    assert "random_miqp n=500 m=1000 p=250" in b["metric"] and "BASELINE configs[1]" in b["config"]["workload"]
    assert "sharded over 4 GPU(s)" in b["config"]["workload"]
    assert b["value"] > 0 and b["nodes"] >= 20 and b["ms_per_step"] > 0 and b["nodes_per_s"] > 0
    assert abs(b["value"] - b["nodes_per_s"] * b["iters_per_node"]) <= 0.02 * b["value"]
    assert b["config"]["comm"].startswith("TorchComm")
    bt = b["batched"]
    assert "one pool per rank" in bt["form"] and bt["nodes"] > 0 and 0 < bt["column_occupancy"] <= 1.0


def test_bench_with_eight_ranks_on_one_device_reports_every_rank(tmp_path):
    """The driver's SCALE command at its widest -- `torch.distributed.run --nproc-per-node 8 bench.py --gpus 8` -- with eight
    processes time-sharing GPU 0 (collectives over gloo): one line, `n_gpus` 8, and one row per rank in `ranks`: every rank
    took part in the collectives exactly once per row, the rows' nodes add up to the line's, and no rank sat without a leaf
    in more than a fifth of its steps (the leaf-sharded search keeps eight ranks fed on config 2's trees).  Not a performance
    number: eight engines share one chip."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", MIOSQP_BENCH_ONE_DEVICE="1")
    tr = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                         "--gpus", "8", "--steps", "16", "--warmup", "4", "--legs", "none", "--no-probes"],
                        env=env, cwd=root, capture_output=True, text=True, timeout=1500)
    assert tr.returncode == 0, tr.stderr[-3000:]
    lines = [ln for ln in tr.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    b = json.loads(lines[0])
    assert b["n_gpus"] == 8 and b["steps"] == 16 and b["scaling"] == "weak" and "sharded over 8 GPU(s)" in b["config"]["workload"]
    rows = b["ranks"]
    assert sorted(r["rank"] for r in rows) == list(range(8))
    # two ranks per tree by default: four trees of the MIQP stream at a time, each group with a communicator of its own
    assert [r["tree_group"] for r in sorted(rows, key=lambda r: r["rank"])] == [0, 0, 1, 1, 2, 2, 3, 3]
    assert "2 ranks per tree" in b["config"]["workload"] and "4 trees of the MIQP stream at a time" in b["config"]["workload"]
    assert sum(r["nodes"] for r in rows) == int(b["nodes"]) and sum(r["iters"] for r in rows) > 0
    assert all(r["steps"] == 16 for r in rows)
    worst = max(r["idle_frac"] for r in rows)
    print("nodes per rank %s, worst idle fraction %.2f" % ([r["nodes"] for r in rows], worst))
    assert worst <= 0.2, rows


@pytest.mark.parametrize("n,m,p,seed,rule", [(30, 150, 15, 4, 1), (50, 100, 25, 2, 1), (20, 40, 10, 1, 0), (100, 150, 40, 7, 1),
                                              (60, 120, 60, 3, 0)])
def test_hosted_search_equals_the_python_loop(n, m, p, seed, rule):
    """miosqp_qp_search_* (the loop of solver.py:65-172 in the C++ host library, leaves in device slots, children
    written on the device): same nodes, same ADMM iterations, same incumbent as the Python loop driving solve_node;
    a second MIQP on the same factor; leaves taken out and put back; an incumbent handed in from outside."""
    from miosqp_amd import bnb, search
    pr = problems.random_miqp(n, m, p, seed=seed)
    st = dict(problems.BNB_SETTINGS, tree_explor_rule=rule)
    qs = dict(problems.QP_SETTINGS, resident=0)  # (small problems would otherwise run as one launch: kernels_tree.inc)
    py = bnb.MIOSQP()
    py.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
             dict(st, device_search=False, device_tree=False), dict(qs))
    cc = bnb.MIOSQP()
    cc.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st, device_tree=False), dict(qs))
    ii = pr["i_idx"]
    rng = np.random.RandomState(seed)
    for inst in range(2):
        r0, r1 = py.solve(), cc.solve()
        assert getattr(cc.work, "_hosted", None) is not None and getattr(py.work, "_hosted", None) is None
        assert r1.status == r0.status
        assert (cc.work.iter_num, cc.work.osqp_iter) == (py.work.iter_num, py.work.osqp_iter)
        if r0.status == bnb.MI_SOLVED:
            assert abs(r1.upper_glob - r0.upper_glob) <= 1e-9 * max(1.0, abs(r0.upper_glob))
            np.testing.assert_array_equal(r1.x[ii], r0.x[ii])
            assert rel(r1.x, r0.x) <= SOL_TOL
        assert len(cc.work._hosted.free) == cc.work._hosted.capacity  # every slot came back
        q2 = rng.randn(n)
        py.update_vectors(q=q2)
        cc.update_vectors(q=q2)
    # leaves out and in again, stepping node by node; an incumbent from outside prunes
    hs = cc.work._hosted
    hs.begin_instance()
    for _ in range(6):
        if hs.step(1) == 0:
            break
    if hs.givable() >= 2:
        recs = [hs.give_leaf() for _ in range(hs.givable())]
        assert hs.givable() == 0
        for rec in recs:
            hs.add_leaf(*rec)
        assert hs.givable() == len(recs)
    r2 = hs.run()
    r0 = py.solve()
    assert r2.status == r0.status
    if r0.status == bnb.MI_SOLVED:
        assert abs(r2.upper_glob - r0.upper_glob) <= 1e-3 * max(1.0, abs(r0.upper_glob))
        np.testing.assert_array_equal(r2.x[ii], r0.x[ii])
        # the optimum handed in before the search starts: nothing better is found, the tree closes sooner
        cc.update_vectors(q=q2)
        cc.work.upper_glob, cc.work.x = r0.upper_glob - 1e-6, r0.x.copy()
        hs.begin_instance()
        r3 = hs.run()
        assert r3.upper_glob == r0.upper_glob - 1e-6 and cc.work.iter_num <= py.work.iter_num


def _final_results(case, settings_extra, qp_extra, make_runner=None):
    """Drives a golden case as make_bnb_traces.py drove the reference, without an observer (so that MIOSQP.solve() takes
    its device-resident path), and returns per solve the final numbers.  make_runner(model) -> callable returning
    Results replaces model.solve (the native stream driver)."""
    from miosqp_amd import bnb
    prob = case["prob"]
    model = bnb.MIOSQP()
    model.setup(prob["P"], prob["q"], prob["A"], np.copy(prob["l"]), np.copy(prob["u"]), prob["i_idx"], prob["i_l"],
                prob["i_u"], dict(case["settings"], **settings_extra), dict(case["qp_settings"], **qp_extra))
    run = model.solve if make_runner is None else make_runner(model)
    out = []

    def one():
        res = run()
        out.append(dict(x=np.array(res.x, dtype=float), upper_glob=res.upper_glob, status=res.status,
                        osqp_iter=model.work.osqp_iter, iter_num=model.work.iter_num))

    if case["x0"] is not None:
        model.set_x0(np.copy(case["x0"]))
    one()
    for (q, l, u, x0u) in case["updates"]:
        model.update_vectors(q=q, l=l, u=u)
        if x0u is not None:
            model.set_x0(np.copy(x0u))
        one()
    return out, model


@pytest.mark.parametrize("name", case_names())
def test_hosted_search_follows_the_reference_traces(name):
    """miosqp_qp_search_* -- the loop of solver.py:85-123 compiled into the host library, the bench headline's loop --
    directly against what the REFERENCE recorded (tests/golden/bnb_*.npz, 13 cases: both exploration rules, node cap,
    infeasible root, set_x0, update_vectors sequences): status, iter_num and osqp_iter identical (the loop visits the
    same nodes in the same order and every relaxation takes the same iterations), x and upper_glob to tolerance."""
    case = load_case(name)
    got, model = _final_results(case, dict(device_tree=False), dict(resident=0))
    hosted = getattr(model.work, "_hosted", None) is not None
    assert hosted or case["prob"]["i_idx"].size == 0
    assert len(got) == len(case["solves"])
    for g, e in zip(got, case["solves"]):
        assert g["status"] == e["status"]
        assert (g["iter_num"], g["osqp_iter"]) == (e["iter_num"], e["osqp_iter"])
        if e["status"] in ("Solved", "Max-iter feasible"):
            assert rel(g["x"], e["x"]) <= SOL_TOL
            assert abs(g["upper_glob"] - e["upper_glob"]) <= 1e-8 * max(1, abs(e["upper_glob"]))


@pytest.mark.parametrize("name", [c for c in case_names() if "cap" not in c])
def test_native_stream_driver_reaches_the_reference_optimum(name):
    """The compiled streaming driver (miosqp_qp_stream_*: leaves in the device pool, many relaxations in flight) on the
    reference's recorded cases: it explores in another order, so node counts differ by construction -- what must agree
    with the reference is the final status and the optimum (value to 1e-8 relative; the integer part of x exactly).
    (The node-capped case is left out: which incumbent exists when a cap strikes depends on the order.)"""
    from miosqp_amd import stream
    case = load_case(name)
    if case["prob"]["i_idx"].size == 0:
        pytest.skip("no integer variable")
    holder = {}

    def make_runner(model):
        def run():
            ns = holder.get("ns")
            if ns is None:
                ns = holder["ns"] = stream.NativeStreamSearch(model, columns=64, capacity=4096)
            else:
                ns.begin_instance()
            return ns.run()
        return run

    got, model = _final_results(case, dict(max_iter_bb=10 ** 6), dict(max_batch=64), make_runner)
    ii = case["prob"]["i_idx"]
    for g, e in zip(got, case["solves"]):
        assert g["status"] == e["status"], (g["status"], e["status"])
        if e["status"] == "Solved":
            assert abs(g["upper_glob"] - e["upper_glob"]) <= 1e-8 * max(1, abs(e["upper_glob"]))
            np.testing.assert_array_equal(g["x"][ii], e["x"][ii])


@pytest.mark.parametrize("form", [dict(coop=0, resident=0), dict(fold=0, coop=0, resident=0), dict(resident=1),
                                  dict(coop=0, resident=0, pers=1), dict(fold=0, coop=0, resident=0, pers=1)])
def test_hosted_search_on_every_engine_form(form):
    """The hosted loop drives whatever form the engine uses for single nodes (two-kernel product form with host-checked
    chunks, four-kernel factor form, LDS-resident workgroup): same nodes and iterations as the Python loop on that form;
    a store that starts with too few slots grows instead of overwriting leaves or giving up."""
    from miosqp_amd import bnb, search
    pr = problems.random_miqp(40, 60, 20, seed=7)
    st = dict(problems.BNB_SETTINGS, device_tree=False)
    qs = dict(problems.QP_SETTINGS, **form)
    py, cc = bnb.MIOSQP(), bnb.MIOSQP()
    py.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st, device_search=False), dict(qs))
    cc.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st), dict(qs))
    r0, r1 = py.solve(), cc.solve()
    assert getattr(cc.work, "_hosted", None) is not None
    assert (r1.status, cc.work.iter_num, cc.work.osqp_iter) == (r0.status, py.work.iter_num, py.work.osqp_iter)
    assert abs(r1.upper_glob - r0.upper_glob) <= 1e-9 * max(1.0, abs(r0.upper_glob))
    np.testing.assert_array_equal(r1.x[pr["i_idx"]], r0.x[pr["i_idx"]])
    # four slots to start with: the store grows as the tree does (the reference's leaf list is unbounded) -- same search
    tiny = bnb.MIOSQP()
    tiny.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st, device_search=False), dict(qs))
    hs = search.HostedSearch(tiny, capacity=4)
    r2 = hs.run()
    assert (r2.status, tiny.work.iter_num, tiny.work.osqp_iter) == (r0.status, py.work.iter_num, py.work.osqp_iter)
    assert abs(r2.upper_glob - r0.upper_glob) <= 1e-9 * max(1.0, abs(r0.upper_glob))
    if py.work.iter_num > 4:
        assert hs._free + hs._open > 4  # it did grow


def test_hosted_search_survives_a_called_off_cooperative_launch(monkeypatch):
    """The hosted loop with the fault of test_cooperative_solver_reports_a_missing_workgroup_and_recovers: the first
    node's cooperative launch is called off, the SAME node is redone in the two-kernel form, the search goes on there and
    ends like a search on an engine that never was cooperative."""
    from miosqp_amd import bnb
    pr = problems.random_miqp(60, 120, 30, seed=11)
    st = dict(problems.BNB_SETTINGS, device_tree=False)
    monkeypatch.setenv("MIOSQP_COOP_NAP", "12")
    monkeypatch.setenv("MIOSQP_COOP_DBG", "64")
    bad = bnb.MIOSQP()
    bad.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st),
              dict(problems.QP_SETTINGS, coop=1, resident=0))
    monkeypatch.delenv("MIOSQP_COOP_DBG")
    ref = bnb.MIOSQP()
    ref.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st),
              dict(problems.QP_SETTINGS, coop=0, resident=0))
    assert bad.work.solver.factor_stats()["coop"] is True
    r0, r1 = ref.solve(), bad.solve()
    fs = bad.work.solver.factor_stats()
    assert fs["coop"] is False and fs["coop_fallbacks"] >= 1
    assert getattr(bad.work, "_hosted", None) is not None and getattr(ref.work, "_hosted", None) is not None
    assert (r1.status, bad.work.iter_num, bad.work.osqp_iter) == (r0.status, ref.work.iter_num, ref.work.osqp_iter)
    assert r1.upper_glob == r0.upper_glob
    np.testing.assert_array_equal(r1.x, r0.x)


def test_called_off_launch_leaves_slot_children_and_incumbent_alone(monkeypatch):
    """A cooperative launch that is called off has touched no iterate: its epilogue must write the record and nothing
    else.  The case that used to go wrong: a leaf adopted with explicit vectors (its slot IS its warm start) whose warm
    start is integral and cheaper than the incumbent -- the epilogue of the called-off launch took it for an
    integer-feasible solution and overwrote the incumbent's x on the device while the host (rightly) discarded the
    record, so the x handed out at the end no longer belonged to upper_glob."""
    from miosqp_amd import bnb, search
    pr = problems.random_miqp(60, 120, 30, seed=11)
    st = dict(problems.BNB_SETTINGS, device_tree=False)
    ref = bnb.MIOSQP()
    ref.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st),
              dict(problems.QP_SETTINGS, coop=0, resident=0))
    r0 = ref.solve()
    assert r0.status == bnb.MI_SOLVED
    n, ii = 60, pr["i_idx"]
    # an integral point below the optimum: the optimum's integer part, the continuous part minimised WITHOUT the constraints
    cset = np.array([i for i in range(n) if i not in set(ii.tolist())])
    Pd = pr["P"].toarray()
    x0 = r0.x.copy()
    x0[cset] = np.linalg.solve(Pd[np.ix_(cset, cset)] + 1e-9 * np.eye(len(cset)),
                               -(pr["q"][cset] + Pd[np.ix_(cset, ii)].dot(x0[ii])))
    obj0 = 0.5 * x0.dot(pr["P"].dot(x0)) + pr["q"].dot(x0)
    assert obj0 < r0.upper_glob - 1e-3
    upper = 0.5 * (obj0 + r0.upper_glob)  # below every feasible point: nothing the search finds can replace it
    marker = r0.x.copy()
    marker[cset[0]] += 0.125
    monkeypatch.setenv("MIOSQP_COOP_NAP", "12")
    monkeypatch.setenv("MIOSQP_COOP_DBG", "64")  # workgroup 1 never shows up: the first launch is called off
    bad = bnb.MIOSQP()
    bad.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st),
              dict(problems.QP_SETTINGS, coop=1, resident=0))
    monkeypatch.delenv("MIOSQP_COOP_DBG")
    assert bad.work.solver.factor_stats()["coop"] is True
    hs = search.HostedSearch(bad)
    hs.begin_instance(seed_root=False)
    hs.adopt_incumbent(upper, marker)
    root = bad.work._make_root()
    hs.add_leaf(root.l[-30:], root.u[-30:], x0, np.zeros(150), 0, -np.inf)
    hs.run()
    fs = bad.work.solver.factor_stats()
    assert fs["coop"] is False and fs["coop_fallbacks"] >= 1  # the launch was called off and redone
    val, x = bad.work.solver.search_get_incumbent()
    assert val == upper
    np.testing.assert_array_equal(x, marker)


def test_hosted_search_random_sweep():
    """60 random small MIQPs (random shape, density, exploration rule, engine form): the hosted loop and the Python loop
    visit the same number of nodes, spend the same iterations and end with the same incumbent value."""
    from miosqp_amd import bnb
    rng = np.random.RandomState(123)
    done = 0
    for trial in range(60):
        n = int(rng.randint(8, 70)); m = int(rng.randint(5, 120)); p = int(rng.randint(2, max(3, n // 2)))
        rule = int(rng.randint(0, 2)); seed = int(rng.randint(0, 10 ** 6))
        dens = float(rng.choice([0.3, 0.7, 1.0]))
        pr = problems.random_miqp(n, m, p, density=dens, seed=seed)
        st = dict(problems.BNB_SETTINGS, tree_explor_rule=rule, device_tree=False, max_iter_bb=400)
        qs = dict(problems.QP_SETTINGS, resident=int(rng.choice([0, -1])))
        out = []
        for hosted in (False, True):
            mdl = bnb.MIOSQP()
            mdl.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                      dict(st, device_search=hosted), dict(qs))
            r = mdl.solve()
            out.append((r.status, mdl.work.iter_num, mdl.work.osqp_iter, r.upper_glob))
            mdl.work.solver.close()
        assert out[0][:3] == out[1][:3], (n, m, p, rule, seed, dens, out)
        assert out[0][3] == out[1][3] or abs(out[0][3] - out[1][3]) <= 1e-9 * max(1.0, abs(out[0][3])), (seed, out)
        done += 1
    assert done == 60


def test_hosted_search_at_config2_size():
    """Config 2 (n=500, m=1000, p=250) through the hosted search in the engine's cooperative form: the first 40 nodes
    equal the Python loop's (nodes, iterations, incumbent)."""
    from miosqp_amd import bnb
    pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
    st = dict(problems.BNB_SETTINGS, max_iter_bb=41)
    out = []
    for hosted in (False, True):
        mdl = bnb.MIOSQP()
        mdl.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                  dict(st, device_search=hosted), dict(problems.QP_SETTINGS))
        r = mdl.solve()
        assert (getattr(mdl.work, "_hosted", None) is not None) == hosted
        out.append((r.status, mdl.work.iter_num, mdl.work.osqp_iter, r.upper_glob, mdl.work.solver.factor_stats()["coop"]))
    assert out[0][:3] == out[1][:3] and out[1][4]
    if np.isfinite(out[0][3]):
        assert abs(out[0][3] - out[1][3]) <= 1e-9 * max(1.0, abs(out[0][3]))


def test_sharded_hosted_search_one_rank():
    """dist.ShardedStream over search.HostedSearch (node-at-a-time per rank): ramp-up on the host, deal, hosted
    steps with a node budget, closes with the sequential optimum."""
    from miosqp_amd import bnb, dist, search
    pr = problems.random_miqp(50, 100, 25, seed=2)
    st = dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 6)
    seq = bnb.MIOSQP()
    seq.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st, device_search=False),
              dict(problems.QP_SETTINGS, resident=0))
    r0 = seq.solve()
    model = bnb.MIOSQP()
    model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st),
                dict(problems.QP_SETTINGS, resident=0))
    hs = search.HostedSearch(model)
    sh = dist.ShardedStream(model, search=hs, step_kwargs=dict(nodes=3), ramp_leaves=4)
    sh.run()
    w = model.work
    assert w.status == bnb.MI_SOLVED and sh.total_alive == 0 and len(hs.free) == hs.capacity
    assert abs(w.upper_glob - r0.upper_glob) <= 1e-3 * max(1.0, abs(r0.upper_glob))
    np.testing.assert_array_equal(w.x[pr["i_idx"]], r0.x[pr["i_idx"]])


@pytest.mark.parametrize("driver", ["python", "native"])
def test_two_pools_on_one_gpu_share_one_tree(driver):
    """stream.MultiPoolSearch on the real engine: two pools (two engines, two host threads) close one tree with the
    sequential optimum, every slot comes back, and a second MIQP reuses both -- with the rounds of each pool in
    stream.StreamSearch and in the library (miosqp_qp_stream_*)."""
    from miosqp_amd import bnb, stream
    pr = problems.random_miqp(50, 100, 25, seed=2)
    st = dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 6)

    def make():
        m = bnb.MIOSQP()
        m.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st),
                dict(problems.QP_SETTINGS, max_batch=64))
        return m

    seq = bnb.MIOSQP()
    seq.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st), dict(problems.QP_SETTINGS))
    mp = stream.MultiPoolSearch(make, pools=2, columns=64, exchange_every=2, driver=driver)
    ii = pr["i_idx"]
    rng = np.random.RandomState(3)
    for inst in range(2):
        r0, r1 = seq.solve(), mp.run()
        assert r1.status == r0.status == bnb.MI_SOLVED
        assert abs(r1.upper_glob - r0.upper_glob) <= 1e-3 * max(1.0, abs(r0.upper_glob))
        np.testing.assert_array_equal(r1.x[ii], r0.x[ii])
        assert all(len(sh.ss.free) == sh.ss.capacity for sh in mp.sh)
        assert mp.models[0].work.upper_glob == mp.models[1].work.upper_glob
        q2 = rng.randn(50)
        seq.update_vectors(q=q2)
        mp.update_vectors(q=q2)


def test_streaming_batch_at_config3_size(oracle_mod):
    """BASELINE config 3 as a stream: n=500, 256 columns kept full from the device-resident pool; a sample of the
    decided nodes is replayed through solve_node, and a sample of THOSE through the CPU oracle (status, iterations,
    x, y, bound); the columns stay busy (no wave tail)."""
    from miosqp_amd import bnb, stream
    pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
    st = dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 9)
    model = bnb.MIOSQP()
    model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], st,
                dict(problems.QP_SETTINGS, max_batch=256))
    seen, oseen = [], []
    srch = stream.StreamSearch(model, columns=256, observer=_stream_checker(pr, 16, seen, oracle_mod=oracle_mod,
                                                                            oracle_stride=3, oracle_seen=oseen))
    eng = model.work.solver
    alive, steps = 1, 0
    while alive and steps < 1500 and srch.nodes < 600:  # the frontier doubles every ~20 chunks (one node's iterations)
        alive = srch.step()
        steps += 1
    assert srch.nodes >= 256 and len(seen) >= 16 and len(oseen) >= 5
    ms, lock_iters, node_iters = eng.batch_stats()
    assert lock_iters == 25 * srch.chunks or lock_iters == 25 * (srch.chunks - 1)
    # useful column-iterations / (columns x lock-step iterations): the stream keeps the columns busy once the
    # frontier is wide enough (a wave of 256 leaves averages below one half)
    assert node_iters <= 256 * lock_iters
    # the chunks ran as persistent launches (§3e'), or were called off once for the launches when a replayed node's
    # cooperative launch held CUs at the wrong moment -- either way the nodes above were right
    print("persistent sweeps in use at the end: %s, called off %d time(s)" % (eng.factor_stats()["batch_pers"], eng.batch_pers_fallbacks()))
    assert eng.batch_pers_fallbacks() <= 1


@pytest.mark.parametrize("name", [c for c in case_names() if "x0" in c or "n10" in c or "n12" in c or "n20" in c or "mpc" in c])
def test_whole_tree_in_one_launch_follows_the_reference_traces(name):
    """miosqp_qp_solve_tree (csrc/kernels_tree.inc): the small golden cases -- trees the REFERENCE explored, recorded
    node by node -- solved in one launch each end with the reference's node count, iteration count, incumbent
    and status, for both exploration rules, with set_x0 incumbents and update_vectors sequences."""
    from miosqp_amd import bnb, qp
    case = load_case(name)
    prob = case["prob"]
    model = bnb.MIOSQP(backend=qp)
    model.setup(prob["P"], prob["q"], prob["A"], np.copy(prob["l"]), np.copy(prob["u"]), prob["i_idx"], prob["i_l"],
                prob["i_u"], case["settings"], case["qp_settings"])
    if case["x0"] is not None:
        model.set_x0(np.copy(case["x0"]))
    runs = [(None, None)] + [(u_, x0u) for u_ in case["updates"] for x0u in [u_[3]]]
    for k, (upd, x0u) in enumerate(runs):
        if upd is not None:
            model.update_vectors(q=upd[0], l=upd[1], u=upd[2])
            if x0u is not None:
                model.set_x0(np.copy(x0u))
        res = model.solve()
        exp = case["solves"][k]
        used_tree = not getattr(model.work, "_no_tree", False)
        assert used_tree, "these cases are small enough for the LDS-resident form"
        assert res.status == exp["status"]
        assert model.work.iter_num == exp["iter_num"] and model.work.osqp_iter == exp["osqp_iter"]
        if np.isfinite(exp["upper_glob"]):
            assert abs(res.upper_glob - exp["upper_glob"]) <= 1e-8 * max(1.0, abs(exp["upper_glob"]))
            np.testing.assert_allclose(res.x, exp["x"], rtol=0, atol=1e-7)
        else:
            assert not np.isfinite(res.upper_glob)


def test_one_wavefront_tree_equals_the_workgroup_tree():
    """k_tree_w (n + M <= 64: the search in ONE wavefront, explicit KKT inverse in registers, DPP row broadcasts)
    against k_tree (one workgroup, product form in LDS; MIOSQP_TREE_WAVE=0) on a sweep of random small MIQPs, both
    exploration rules: same status, node count, iteration count and incumbent.  Two processes: the switch is read
    once per process."""
    import json
    import subprocess
    script = r'''
import json, sys
import numpy as np
sys.path.insert(0, %r)
from miosqp_amd import bnb, problems, qp
out = []
for seed, (n, m, p) in enumerate([(6, 4, 3), (8, 10, 4), (10, 12, 5), (12, 20, 6), (14, 24, 7), (16, 28, 8), (9, 30, 9),
                                  (18, 20, 9), (12, 8, 12), (20, 22, 10)]):
    pr = problems.random_miqp(n, m, p, seed=100 + seed)
    for rule in (0, 1):
        model = bnb.MIOSQP(backend=qp)
        model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                    dict(problems.BNB_SETTINGS, tree_explor_rule=rule), dict(problems.QP_SETTINGS))
        res = model.solve()
        assert not getattr(model.work, "_no_tree", False)
        out.append(dict(case=[n, m, p, rule], status=res.status, nodes=int(model.work.iter_num), iters=int(model.work.osqp_iter),
                        upper=float(res.upper_glob), x=[float(v) for v in np.atleast_1d(res.x)]))
        model.work.solver.close()
print("RESULT " + json.dumps(out))
''' % ROOT
    runs = []
    for wave in ("1", "0"):
        env = dict(os.environ, MIOSQP_TREE_WAVE=wave)
        p = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1]
        runs.append(json.loads(line[7:]))
    assert len(runs[0]) == len(runs[1]) == 20
    for a, b in zip(*runs):
        assert (a["case"], a["status"], a["nodes"], a["iters"]) == (b["case"], b["status"], b["nodes"], b["iters"])
        if np.isfinite(a["upper"]):
            assert abs(a["upper"] - b["upper"]) <= 1e-8 * max(1.0, abs(b["upper"]))
            np.testing.assert_allclose(a["x"], b["x"], rtol=0, atol=1e-7)
        else:
            assert not np.isfinite(b["upper"])


def test_resident_register_loop_sweep(oracle_mod):
    """The one-workgroup solver with its loop on the explicit inverse in registers (res_admm_w) against the oracle on a
    sweep of shapes that move through its layouts: n+M from 9 to 192 (2 or 4 parts per row, 1-6 chunks of 16 columns per
    lane, widths that are and are not multiples of 16, more variables than constraints and the reverse): same status and
    iteration count, x and y within the solution tolerance; and the product-form sweeps (MIOSQP_RES_W=0 is read per
    setup) give the same on the same inputs."""
    from miosqp_amd import qp
    shapes = [(5, 2, 2), (8, 7, 1), (12, 3, 1), (15, 30, 3), (16, 31, 1), (17, 30, 1), (30, 30, 4), (20, 90, 2), (60, 4, 0),
              (40, 70, 18), (64, 63, 1), (64, 64, 0), (33, 100, 27), (90, 80, 22), (100, 60, 32), (6, 180, 6)]
    for k, (n, m, p) in enumerate(shapes):
        pr = problems.random_miqp(n, m, max(p, 1), seed=200 + k)
        A, l, u = problems.extended(pr)
        N = n + A.shape[0]
        assert N <= 192
        o = oracle_mod.OSQP()
        o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
        rng = np.random.RandomState(k)
        x0, y0 = 0.1 * rng.randn(n), 0.1 * rng.randn(A.shape[0])
        o.warm_start(x=x0, y=y0)
        ro = o.solve()
        for w in ("1", "0"):
            os.environ["MIOSQP_RES_W"] = w
            try:
                g = qp.OSQP()
                g.setup(pr["P"], pr["q"], A, l, u, resident=1, coop=0, **problems.QP_SETTINGS)
            finally:
                os.environ.pop("MIOSQP_RES_W", None)
            if not g.factor_stats()["resident"]:
                assert w == "0", (n, m, p)  # (the product form of this shape does not fit the LDS)
                g.close()
                continue
            g.warm_start(x=x0, y=y0)
            rg = g.solve()
            assert (rg.info.status_val, rg.info.iter) == (ro.info.status_val, ro.info.iter), (n, m, p, w)
            if ro.info.status_val == 1:
                assert rel(rg.x, ro.x) <= SOL_TOL and rel(rg.y, ro.y) <= SOL_TOL, (n, m, p, w)
            g.close()


def test_whole_tree_kernels_random_sweep(oracle_mod):
    """MIOSQP.solve on the HIP engine (whole tree in one launch: k_tree_w for n+M <= 64, k_tree above) against the same
    search on the CPU oracle over random small MIQPs -- odd and even n (compressed rows are padded to even length with
    a zero that repeats a column index: a dense copy must not let it overwrite an entry), few and many constraints, both
    exploration rules, update_vectors re-solves: status, node count, optimum, integer part of x."""
    from miosqp_amd import bnb
    rng = np.random.RandomState(2024)
    for k in range(90):
        n = int(rng.randint(4, 61))
        m = int(rng.randint(1, 91))
        p = int(rng.randint(1, min(n, 12) + 1))
        rule = int(rng.randint(0, 2))
        pr = problems.random_miqp(n, m, p, seed=1000 + k)
        st = dict(problems.BNB_SETTINGS, tree_explor_rule=rule)
        a, b = bnb.MIOSQP(), bnb.MIOSQP(backend=oracle_mod)
        for mdl in (a, b):
            mdl.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st),
                      dict(problems.QP_SETTINGS))
        for step in range(2):
            ra, rb = a.solve(), b.solve()
            assert not getattr(a.work, "_no_tree", False), (n, m, p)
            assert ra.status == rb.status and a.work.iter_num == b.work.iter_num, (k, n, m, p, rule, step)
            if np.isfinite(rb.upper_glob):
                assert abs(ra.upper_glob - rb.upper_glob) <= 1e-6 * max(1.0, abs(rb.upper_glob)), (k, n, m, p, rule, step)
                np.testing.assert_array_equal(np.round(ra.x[pr["i_idx"]]), np.round(rb.x[pr["i_idx"]]))
            q2 = rng.randn(n)
            a.update_vectors(q=q2)
            b.update_vectors(q=q2)
        a.work.solver.close()


def test_native_stream_driver_applies_the_node_cap_per_instance():
    """A compiled streaming driver kept across a sequence of MIQPs: max_iter_bb counts the nodes of the current instance
    (it used to count the driver's lifetime: the first instance after the cap was reached returned at once)."""
    from miosqp_amd import bnb, stream
    pr = problems.random_miqp(30, 60, 12, seed=5)
    probe = bnb.MIOSQP()
    probe.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 6), dict(problems.QP_SETTINGS, max_batch=64))
    s0 = stream.NativeStreamSearch(probe, columns=64, capacity=2048)
    s0.run()
    per_tree = s0.nodes
    assert per_tree >= 3
    probe.work.solver.close()
    cap = 3 * per_tree + 16
    seq, mdl = bnb.MIOSQP(), bnb.MIOSQP()
    for m in (seq, mdl):
        m.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                dict(problems.BNB_SETTINGS, max_iter_bb=cap), dict(problems.QP_SETTINGS, max_batch=64))
    s = stream.NativeStreamSearch(mdl, columns=64, capacity=2048)
    rng = np.random.RandomState(6)
    for inst in range(6):
        r0, r1 = seq.solve(), s.run()
        assert r1.status == r0.status == bnb.MI_SOLVED, inst
        assert abs(r1.upper_glob - r0.upper_glob) <= 1e-3 * max(1.0, abs(r0.upper_glob))
        q2 = rng.randn(30)
        seq.update_vectors(q=q2)
        mdl.update_vectors(q=q2)
        s.begin_instance()
    assert s.nodes > cap


@pytest.mark.parametrize("n,m,p,seed,B", [(50, 100, 10, 0, 24), (12, 30, 6, 8, 16), (40, 60, 20, 7, 5)])
def test_many_trees_in_one_launch_equal_the_sequential_calls(n, m, p, seed, B):
    """miosqp_qp_solve_trees / MIOSQP.solve_many: B MIQPs that share P and A (new q, l, u each, some with an initial
    solution) solved in ONE launch -- a workgroup per instance (a wavefront for n + M <= 64) -- give, instance by
    instance, what update_vectors + set_x0 + solve give one after the other on the same engine: status, node count,
    iteration count, incumbent value and x identical; the model itself is left untouched."""
    from miosqp_amd import bnb
    pr = problems.random_miqp(n, m, p, seed=seed)
    mk = lambda: bnb.MIOSQP()
    seq, bat = mk(), mk()
    for mdl in (seq, bat):
        mdl.setup(pr["P"], pr["q"], pr["A"], pr["l"].copy(), pr["u"].copy(), pr["i_idx"], pr["i_l"], pr["i_u"],
                  dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
    rng = np.random.RandomState(seed + 99)
    inst = []
    for k in range(B):
        d = dict(q=rng.randn(n), l=-2 + rng.rand(m), u=2 + rng.rand(m))
        if k % 3 == 0:
            d = dict(q=rng.randn(n))  # only the cost changes: l, u are the model's
        inst.append(d)
    want = []
    for d in inst:
        seq.update_vectors(q=d["q"].copy(), l=None if "l" not in d else d["l"].copy(), u=None if "u" not in d else d["u"].copy())
        r = seq.solve()
        want.append(dict(x=np.array(r.x, dtype=float), upper=r.upper_glob, status=r.status, nodes=seq.work.iter_num - 1,
                         osqp_iter=seq.work.osqp_iter))
        # (the sequential model keeps the last l, u; the batch builds every instance on the model's ORIGINAL l, u: give the
        #  sequential one its original bounds back before the next cost-only instance)
        seq.update_vectors(l=pr["l"].copy(), u=pr["u"].copy())
    # a quarter of the instances also carry the optimum of the first run as an initial solution: the sequential answer
    # for those is update_vectors + set_x0 + solve
    for k in range(0, B, 4):
        if want[k]["status"] == bnb.MI_SOLVED:
            d = inst[k] = dict(inst[k], x0=want[k]["x"].copy())
            seq.update_vectors(q=d["q"].copy(), l=None if "l" not in d else d["l"].copy(), u=None if "u" not in d else d["u"].copy())
            seq.set_x0(d["x0"].copy())
            r = seq.solve()
            want[k] = dict(x=np.array(r.x, dtype=float), upper=r.upper_glob, status=r.status, nodes=seq.work.iter_num - 1,
                           osqp_iter=seq.work.osqp_iter)
            seq.update_vectors(l=pr["l"].copy(), u=pr["u"].copy())
    q_before, l_before = bat.work.data.q.copy(), bat.work.data.l.copy()
    got = bat.solve_many(inst)
    assert np.array_equal(bat.work.data.q, q_before) and np.array_equal(bat.work.data.l, l_before)
    assert len(got) == B
    for k, (g, w) in enumerate(zip(got, want)):
        assert g["status"] == w["status"], k
        assert (g["nodes"], g["osqp_iter"]) == (w["nodes"], w["osqp_iter"]), k
        if w["status"] == bnb.MI_SOLVED:
            assert g["upper_glob"] == w["upper"], k
            np.testing.assert_array_equal(g["x"], w["x"])


def test_power_converter_steps_as_one_batch():
    """BASELINE config 4 as a batch: the 40 MPC steps of the reference's recorded sequence (each with its own q, u and
    initial solution) in ONE launch of 40 single-wavefront trees: node and iteration counts of every step equal the
    reference's, the incumbents equal the recorded ones."""
    from miosqp_amd import bnb
    pc = problems.load_power_converter()
    model = bnb.MIOSQP()
    model.setup(pc["P"], pc["q"][0].copy(), pc["A"], pc["l"].copy(), pc["u"][0].copy(), pc["i_idx"], pc["i_l"], pc["i_u"],
                pc["settings"], pc["qp_settings"])
    inst = [dict(q=pc["q"][k].copy(), l=pc["l"].copy(), u=pc["u"][k].copy(), x0=pc["x0"][k].copy()) for k in range(len(pc["q"]))]
    got = model.solve_many(inst)
    for k, g in enumerate(got):
        assert g["status"] == pc["status"][k], k
        assert (g["nodes"], g["osqp_iter"]) == (int(pc["nodes"][k]), int(pc["osqp_iter"][k])), k
        assert abs(g["upper_glob"] - pc["upper"][k]) <= 1e-8 * max(1.0, abs(pc["upper"][k]))
        assert rel(g["x"], pc["x"][k]) <= 1e-6
