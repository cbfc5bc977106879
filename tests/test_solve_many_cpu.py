"""MIOSQP.solve_many on a backend without the batched entry point (the CPU oracle): the sequential path gives what
update_vectors + set_x0 + solve give, and leaves the model as it was."""
import numpy as np

from miosqp_amd import bnb, problems


def test_solve_many_falls_back_to_the_sequential_calls(oracle_mod):
    pr = problems.random_miqp(12, 30, 6, seed=8)
    a, b = bnb.MIOSQP(backend=oracle_mod), bnb.MIOSQP(backend=oracle_mod)
    for mdl in (a, b):
        mdl.setup(pr["P"], pr["q"], pr["A"], pr["l"].copy(), pr["u"].copy(), pr["i_idx"], pr["i_l"], pr["i_u"],
                  dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
    rng = np.random.RandomState(5)
    inst = [dict(q=rng.randn(12), l=-2 + rng.rand(30), u=2 + rng.rand(30)) for _ in range(4)]
    want = []
    for d in inst:
        a.update_vectors(q=d["q"].copy(), l=d["l"].copy(), u=d["u"].copy())
        r = a.solve()
        want.append((r.status, r.upper_glob, a.work.iter_num - 1, a.work.osqp_iter, np.array(r.x, dtype=float)))
    q0, l0 = b.work.data.q.copy(), b.work.data.l.copy()
    got = b.solve_many(inst)
    assert np.array_equal(b.work.data.q, q0) and np.array_equal(b.work.data.l, l0)
    for g, w in zip(got, want):
        assert (g["status"], g["nodes"], g["osqp_iter"]) == (w[0], w[2], w[3])
        if w[0] == bnb.MI_SOLVED:
            assert g["upper_glob"] == w[1]
            np.testing.assert_array_equal(g["x"], w[4])
