"""Certification of the CPU oracle itself (the reference pins nothing here: SURVEY.md sec. 8c).

(i) KKT certificates from an independent numpy checker, (ii) scipy trust-constr cross-check,
(iii) exact active-set refinement, (iv) brute-force enumeration of the 2^10 binaries of
BASELINE config 1, (v) infeasibility certificates, (vi) the reference's 49 hard inputs.
"""
import itertools

import numpy as np
import pytest
import scipy.optimize as sopt
import scipy.sparse as spa

from golden_cases import load_case, load_maxiter
from miosqp_amd import problems
from qp_check import active_set_refine, kkt_certificate, osqp_tolerances

SOLVED, MAXIT, PINF, DINF = 1, -2, -3, -4


def _solve(oracle_mod, P, q, A, l, u, x0=None, y0=None, **kw):
    s = oracle_mod.OSQP()
    s.setup(P, q, A, l, u, **kw)
    n, m = A.shape[1], A.shape[0]
    s.warm_start(x=np.zeros(n) if x0 is None else x0, y=np.zeros(m) if y0 is None else y0)
    return s, s.solve()


@pytest.mark.parametrize("n,m,p,seed", [(10, 5, 2, 0), (12, 60, 6, 1), (20, 100, 10, 2),
                                        (50, 100, 10, 3), (60, 30, 20, 4)])
def test_kkt_certificate_tight(oracle_mod, n, m, p, seed):
    pr = problems.random_miqp(n, m, p, seed=seed)
    A, l, u = problems.extended(pr)
    eps = 1e-7
    s, r = _solve(oracle_mod, pr["P"], pr["q"], A, l, u, eps_abs=eps, eps_rel=eps,
                  max_iter=200000, rho=0.03)
    assert r.info.status_val == SOLVED
    c = kkt_certificate(pr["P"], pr["q"], A, l, u, r.x, r.y)
    z = np.clip(A.dot(r.x), l, u)
    ep, ed = osqp_tolerances(pr["P"], pr["q"], A, r.x, r.y, z, eps, eps)
    assert c["pri"] <= ep and c["dua"] <= ed
    assert c["comp"] <= 1e-4 and c["stray"] == 0.0
    assert abs(c["obj"] - r.info.obj_val) <= 1e-9 * max(1, abs(c["obj"]))
    # exact answer on the identified active set
    xs = active_set_refine(pr["P"], pr["q"], A, l, u, r.x, r.y)
    assert np.max(np.abs(xs - r.x)) <= 1e-4 * max(1, np.max(np.abs(xs)))


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_scipy_trust_constr_agrees(oracle_mod, seed):
    pr = problems.random_miqp(12, 20, 4, seed=10 + seed)
    A, l, u = problems.extended(pr)
    s, r = _solve(oracle_mod, pr["P"], pr["q"], A, l, u, eps_abs=1e-6, eps_rel=1e-6,
                  max_iter=100000)
    assert r.info.status_val == SOLVED
    Pd, Ad = pr["P"].toarray(), A.toarray()
    res = sopt.minimize(lambda x: 0.5 * x @ Pd @ x + pr["q"] @ x, np.zeros(12),
                        jac=lambda x: Pd @ x + pr["q"], hess=lambda x: Pd,
                        constraints=[sopt.LinearConstraint(Ad, l, u)], method="trust-constr",
                        options=dict(gtol=1e-10, xtol=1e-12, maxiter=3000))
    assert abs(res.fun - r.info.obj_val) <= 1e-3 * max(1.0, abs(res.fun))
    assert np.max(np.abs(res.x - r.x)) <= 1e-2 * max(1.0, np.max(np.abs(res.x)))


def test_default_eps_meets_its_own_rule(oracle_mod):
    pr = problems.random_miqp(50, 100, 10, seed=0)
    A, l, u = problems.extended(pr)
    s, r = _solve(oracle_mod, pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
    assert r.info.status_val == SOLVED and r.info.iter % 25 == 0
    c = kkt_certificate(pr["P"], pr["q"], A, l, u, r.x, r.y)
    assert abs(c["dua"] - r.info.dua_res) <= 1e-12 + 1e-9 * c["dua"]
    assert c["pri"] <= r.info.pri_res + 1e-12


def test_bruteforce_config1(oracle_mod):
    """BASELINE config 1 (n=50, m=100, 10 binaries): enumerate all 2^10 fixings."""
    case = load_case("cfg1_n50m100p10_s0")
    pr = case["prob"]
    A, l, u = problems.extended(pr)
    m = pr["A"].shape[0]
    s = oracle_mod.OSQP()
    s.setup(pr["P"], pr["q"], A, l, u, eps_abs=1e-5, eps_rel=1e-5, max_iter=20000)
    best, best_x = np.inf, None
    x0, y0 = np.zeros(50), np.zeros(A.shape[0])
    for bits in itertools.product((0.0, 1.0), repeat=10):
        lb, ub = l.copy(), u.copy()
        lb[m:] = bits
        ub[m:] = bits
        s.update(l=lb, u=ub)
        s.warm_start(x=x0, y=y0)
        r = s.solve()
        if r.info.status_val == SOLVED and r.info.obj_val < best:
            best, best_x = r.info.obj_val, r.x.copy()
    e = case["solves"][0]
    assert e["status"] == "Solved"
    # B&B ran at eps 1e-3: same integer assignment, objective within the relaxation tolerance
    np.testing.assert_array_equal(np.round(best_x[pr["i_idx"]]), e["x"][pr["i_idx"]])
    assert abs(best - e["upper_glob"]) <= 1e-2 * abs(best)
    assert np.linalg.norm(best_x - e["x"]) <= 1e-2 * np.linalg.norm(best_x)  # authors' bar


def test_primal_infeasible(oracle_mod):
    P = spa.csc_matrix(np.eye(2))
    A = spa.csc_matrix(np.array([[1.0, 1.0], [1.0, 1.0], [1.0, 0.0]]))
    l = np.array([1.0, -np.inf, -np.inf])
    u = np.array([np.inf, -1.0, np.inf])
    s, r = _solve(oracle_mod, P, np.zeros(2), A, l, u)
    assert r.info.status_val == PINF
    assert np.all(np.isnan(r.x))
    dy = r.y
    # certificate: A'dy ~ 0 and u'dy+ + l'dy- < 0
    assert np.max(np.abs(A.T.dot(dy))) <= 1e-3
    assert dy[0] < 0 and dy[1] > 0


def test_dual_infeasible(oracle_mod):
    P = spa.csc_matrix(np.diag([1.0, 0.0]))
    A = spa.csc_matrix(np.array([[1.0, 0.0], [0.0, 1.0]]))
    l = np.array([-1.0, -np.inf])
    u = np.array([1.0, 5.0])
    s, r = _solve(oracle_mod, P, np.array([0.0, 1.0]), A, l, u)  # unbounded along -e2
    assert r.info.status_val == DINF
    assert r.x[1] < 0 and abs(r.x[0]) <= 1e-6


def test_update_rejects_crossed_bounds(oracle_mod):
    pr = problems.random_miqp(10, 5, 2, seed=0)
    A, l, u = problems.extended(pr)
    s = oracle_mod.OSQP()
    s.setup(pr["P"], pr["q"], A, l, u)
    lb = l.copy()
    lb[0] = u[0] + 1
    with pytest.raises(ValueError):
        s.update(l=lb, u=u)


def test_warm_start_is_pure_function(oracle_mod):
    """With rho fixed a node result depends only on (l, u, x0, y0): visit order is irrelevant."""
    pr = problems.random_miqp(20, 40, 6, seed=5)
    A, l, u = problems.extended(pr)
    s = oracle_mod.OSQP()
    s.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
    rng = np.random.RandomState(0)
    x0, y0 = rng.randn(20), rng.randn(A.shape[0])
    s.warm_start(x=x0, y=y0)
    a = s.solve()
    u2 = u.copy()
    u2[-1] = 0.0
    s.update(l=l, u=u2)
    s.warm_start(x=a.x, y=a.y)
    s.solve()
    s.update(l=l, u=u)
    s.warm_start(x=x0, y=y0)
    b = s.solve()
    assert a.info.iter == b.info.iter
    np.testing.assert_array_equal(a.x, b.x)
    np.testing.assert_array_equal(a.y, b.y)


def test_reference_hard_instances(oracle_mod):
    """The reference's 49 max-iter relaxations (inputs only): every one must end in a defined
    status, and the SOLVED ones must pass the KKT certificate at their own eps."""
    counts = {}
    for inst in load_maxiter():
        st = dict(inst["settings"])
        s, r = _solve(oracle_mod, inst["P"], inst["q"], inst["A"], inst["l"], inst["u"], **st)
        counts[r.info.status_val] = counts.get(r.info.status_val, 0) + 1
        assert r.info.status_val in (SOLVED, MAXIT, PINF, DINF)
        if r.info.status_val == SOLVED:
            c = kkt_certificate(inst["P"], inst["q"], inst["A"], inst["l"], inst["u"], r.x, r.y)
            z = np.clip(inst["A"].dot(r.x), inst["l"], inst["u"])
            ep, ed = osqp_tolerances(inst["P"], inst["q"], inst["A"], r.x, r.y, z,
                                     st["eps_abs"], st["eps_rel"])
            assert c["pri"] <= ep * (1 + 1e-9) and c["dua"] <= ed * (1 + 1e-9)
            assert c["stray"] <= 1e-12
    assert sum(counts.values()) == 49
