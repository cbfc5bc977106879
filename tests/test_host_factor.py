"""Host-side setup logic of the HIP engine (miosqp_amd/csrc/factor.cpp) checked on the CPU.

The C++ is compiled with g++ into a throw-away library together with tests/host_harness.cpp
(a test-only wrapper); nothing here touches a GPU.  Checks: equilibration equals the oracle's,
the block factor reproduces K^-1 against a dense numpy solve, the pre-inverted tail is the
inverse of the triangular factor of the reduced Hessian, padded-row layout invariants.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as spa

from miosqp_amd import problems

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)


@pytest.fixture(scope="module")
def hh(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("hh") / "libhh.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread",
                           os.path.join(HERE, "host_harness.cpp"),
                           os.path.join(ROOT, "miosqp_amd", "csrc", "factor.cpp"), "-o", out])
    L = C.CDLL(out)
    L.hh_build.restype = C.c_void_p
    L.hh_build.argtypes = [C.c_int, C.c_int, ip, ip, dp, ip, ip, dp, dp, C.c_int, C.c_double, C.c_double]
    L.hh_free.argtypes = [C.c_void_p]
    L.hh_scaling.argtypes = [C.c_void_p, dp, dp, dp, dp]
    L.hh_tail.argtypes = [C.c_void_p, dp, dp, dp]
    L.hh_layout_ok.argtypes = [C.c_void_p]
    L.hh_nnz_panel.argtypes = [C.c_void_p]
    L.hh_nnz_panel.restype = C.c_long
    L.hh_apply_kinv.argtypes = [C.c_void_p, dp, dp, dp, dp]
    L.hh_apply_kinv_folded.argtypes = [C.c_void_p, dp, dp, dp, dp]
    L.hh_products.argtypes = [C.c_void_p, dp, dp, dp, dp, dp, dp]
    return L


def _d(a):
    return a.ctypes.data_as(dp)


def _i(a):
    return a.ctypes.data_as(ip)


def _build(hh, pr, passes=10, rho=0.1, sigma=1e-6):
    A, l, u = problems.extended(pr)
    P = spa.csc_matrix(pr["P"]); P.sort_indices(); A.sort_indices()
    keep = [np.ascontiguousarray(P.indptr, np.int32), np.ascontiguousarray(P.indices, np.int32),
            np.ascontiguousarray(P.data, np.float64), np.ascontiguousarray(A.indptr, np.int32),
            np.ascontiguousarray(A.indices, np.int32), np.ascontiguousarray(A.data, np.float64),
            np.ascontiguousarray(pr["q"], np.float64)]
    n, M = A.shape[1], A.shape[0]
    h = hh.hh_build(n, M, _i(keep[0]), _i(keep[1]), _d(keep[2]), _i(keep[3]), _i(keep[4]), _d(keep[5]),
                    _d(keep[6]), passes, rho, sigma)
    assert h
    return h, P, A, n, M


@pytest.mark.parametrize("n,m,p,seed", [(10, 5, 2, 0), (50, 100, 10, 1), (80, 30, 40, 2), (130, 260, 65, 3)])
def test_block_factor_matches_dense_solve(hh, oracle_mod, n, m, p, seed):
    pr = problems.random_miqp(n, m, p, seed=seed)
    rho, sigma = 0.1, 1e-6
    h, P, A, n, M = _build(hh, pr, rho=rho, sigma=sigma)
    try:
        assert hh.hh_layout_ok(h) == 1
        assert hh.hh_nnz_panel(h) == A.nnz
        D, E, qs, c = np.empty(n), np.empty(M), np.empty(n), C.c_double()
        hh.hh_scaling(h, _d(D), _d(E), C.byref(c), _d(qs))
        # same equilibration as the oracle
        s = oracle_mod.OSQP()
        _, l, u = problems.extended(pr)
        s.setup(pr["P"], pr["q"], A, l, u)
        Do, Eo, co = s.scaling()
        np.testing.assert_allclose(D, Do, rtol=1e-14)
        np.testing.assert_allclose(E, Eo, rtol=1e-14)
        assert abs(c.value - co) <= 1e-14 * co
        np.testing.assert_allclose(qs, co * Do * pr["q"], rtol=1e-13, atol=1e-300)
        # dense scaled KKT
        Pb = c.value * (D[:, None] * P.toarray() * D[None, :])
        Ab = E[:, None] * A.toarray() * D[None, :]
        K = np.block([[Pb + sigma * np.eye(n), Ab.T], [Ab, -np.eye(M) / rho]])
        rng = np.random.RandomState(seed)
        rx, rz = rng.randn(n), rng.randn(M)
        xt, nu = np.empty(n), np.empty(M)
        hh.hh_apply_kinv(h, _d(rx), _d(rz), _d(xt), _d(nu))
        ref = np.linalg.solve(K, np.concatenate([rx, rz]))
        scale = np.max(np.abs(ref))
        assert np.max(np.abs(xt - ref[:n])) <= 1e-9 * scale
        assert np.max(np.abs(nu - ref[n:])) <= 1e-9 * scale
        # product-form factor gives the same K^-1
        xt2, nu2 = np.empty(n), np.empty(M)
        hh.hh_apply_kinv_folded(h, _d(rx), _d(rz), _d(xt2), _d(nu2))
        assert np.max(np.abs(xt2 - ref[:n])) <= 1e-9 * scale
        assert np.max(np.abs(nu2 - ref[n:])) <= 1e-9 * scale
        # tail: Linv is the inverse of the unit-lower factor of S = Pb + sigma I + rho Ab'Ab
        Linv, LinvT, d2inv = np.empty((n, n)), np.empty((n, n)), np.empty(n)
        hh.hh_tail(h, _d(Linv), _d(LinvT), _d(d2inv))
        assert np.all(np.triu(Linv) == 0) and np.all(np.tril(LinvT) == 0)
        np.testing.assert_array_equal(LinvT, Linv.T)
        S = Pb + sigma * np.eye(n) + rho * Ab.T @ Ab
        Li = Linv + np.eye(n)
        np.testing.assert_allclose(Li.T @ np.diag(d2inv) @ Li @ S, np.eye(n), atol=1e-8)
        assert np.all(d2inv > 0)
        # products through the padded rows
        x, v = rng.randn(n), rng.randn(M)
        Ax, Atv, Px, Prx = np.empty(M), np.empty(n), np.empty(n), np.empty(n)
        hh.hh_products(h, _d(x), _d(v), _d(Ax), _d(Atv), _d(Px), _d(Prx))
        np.testing.assert_allclose(Ax, Ab @ x, atol=1e-12)
        np.testing.assert_allclose(Atv, Ab.T @ v, atol=1e-12)
        np.testing.assert_allclose(Px, Pb @ x, atol=1e-12)
        np.testing.assert_allclose(Prx, P @ x, atol=1e-11)
    finally:
        hh.hh_free(h)


def test_scaling_off_and_upper_triangular_input(hh):
    pr = problems.random_miqp(20, 30, 5, seed=4)
    pr_u = dict(pr)
    pr_u["P"] = spa.triu(pr["P"]).tocsc()
    for passes in (0, 3):
        h1, P, A, n, M = _build(hh, pr, passes=passes)
        h2, _, _, _, _ = _build(hh, pr_u, passes=passes)
        try:
            out = []
            for h in (h1, h2):
                D, E, qs, c = np.empty(n), np.empty(M), np.empty(n), C.c_double()
                hh.hh_scaling(h, _d(D), _d(E), C.byref(c), _d(qs))
                Linv, LinvT, d2 = np.empty((n, n)), np.empty((n, n)), np.empty(n)
                hh.hh_tail(h, _d(Linv), _d(LinvT), _d(d2))
                out.append((D, E, c.value, Linv, d2))
            if passes == 0:
                assert np.all(out[0][0] == 1) and np.all(out[0][1] == 1) and out[0][2] == 1
            for a, b in zip(out[0], out[1]):
                np.testing.assert_array_equal(a, b)
        finally:
            hh.hh_free(h1); hh.hh_free(h2)


def test_nonconvex_rejected(hh):
    pr = problems.random_miqp(8, 4, 2, seed=5)
    pr["P"] = spa.csc_matrix(-np.eye(8) * 50.0)
    A, l, u = problems.extended(pr)
    P = pr["P"]; P.sort_indices(); A.sort_indices()
    k = [np.ascontiguousarray(P.indptr, np.int32), np.ascontiguousarray(P.indices, np.int32),
         np.ascontiguousarray(P.data, np.float64), np.ascontiguousarray(A.indptr, np.int32),
         np.ascontiguousarray(A.indices, np.int32), np.ascontiguousarray(A.data, np.float64),
         np.ascontiguousarray(pr["q"], np.float64)]
    h = hh.hh_build(8, A.shape[0], _i(k[0]), _i(k[1]), _d(k[2]), _i(k[3]), _i(k[4]), _d(k[5]), _d(k[6]),
                    0, 0.1, 1e-6)
    assert not h


def test_equilibration_of_a_large_matrix_takes_the_threaded_path(hh, oracle_mod):
    """Above 2^20 stored entries of P the column norms and the element-wise products of the Ruiz passes run on
    several host threads (max-merges and independent products: the result cannot depend on the split)."""
    import scipy.sparse as spa
    rng = np.random.RandomState(7)
    n, m = 1500, 40
    B = rng.randn(n, n)
    P = spa.csc_matrix(B @ B.T / n + np.eye(n))
    assert spa.triu(P).nnz > (1 << 20)
    A = spa.random(m, n, density=0.3, random_state=rng, format="csc")
    pr = dict(P=P, q=rng.randn(n), A=A, l=-np.ones(m), u=np.ones(m), i_idx=np.arange(0), i_l=np.zeros(0), i_u=np.zeros(0))
    h, P2, A2, n2, M = _build(hh, pr)
    try:
        D, E, qs, c = np.empty(n), np.empty(M), np.empty(n), C.c_double()
        hh.hh_scaling(h, _d(D), _d(E), C.byref(c), _d(qs))
        s = oracle_mod.OSQP()
        _, l, u = problems.extended(pr)
        s.setup(P, pr["q"], A2, l, u)
        Do, Eo, co = s.scaling()
        np.testing.assert_allclose(D, Do, rtol=1e-14)
        np.testing.assert_allclose(E, Eo, rtol=1e-14)
        assert abs(c.value - co) <= 1e-14 * co
        # the symmetric matrices by row (threaded transpose of the upper triangle: per-chunk histograms and cursors)
        x, v = rng.randn(n), rng.randn(M)
        Ax, Atv, Px, Prx = np.empty(M), np.empty(n), np.empty(n), np.empty(n)
        hh.hh_products(h, _d(x), _d(v), _d(Ax), _d(Atv), _d(Px), _d(Prx))
        Pd = P.toarray()
        np.testing.assert_allclose(Prx, Pd @ x, rtol=0, atol=1e-10 * np.abs(Pd @ x).max())
        Pb = c.value * (D[:, None] * Pd * D[None, :])
        np.testing.assert_allclose(Px, Pb @ x, rtol=0, atol=1e-10 * np.abs(Pb @ x).max())
        assert hh.hh_layout_ok(h)
    finally:
        hh.hh_free(h)
