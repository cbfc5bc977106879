"""The fused-node / digest / wave branches of the host layer on CPU (tests/digest_backend.py)."""
import numpy as np
import pytest

import digest_backend
from golden_cases import case_names, load_case, run_case
from miosqp_amd import bnb, dist, problems


@pytest.mark.parametrize("name", case_names())
def test_traces_through_fused_node_and_digest(name):
    """solve_node + device-style digest instead of the four calls + host numpy: same tree as the
    reference recorded (decisions identical; values to rounding of the objective expression)."""
    case = load_case(name)
    got = run_case(case, digest_backend)
    cols = case["cols"]
    disc = [cols.index(c) for c in ("iter_num", "depth", "status", "num_iter", "n_leaves", "constr_idx",
                                    "nextvar_idx", "intinf")]
    for g, e in zip(got, case["solves"]):
        assert g["status"] == e["status"] and g["osqp_iter"] == e["osqp_iter"]
        np.testing.assert_array_equal(g["trace"][:, disc], e["trace"][:, disc])
        np.testing.assert_allclose(g["trace"], e["trace"], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("width,pipelined", [(1, False), (3, False), (16, False), (4, True), (16, True)])
def test_wave_search_reaches_the_same_optimum(width, pipelined):
    pr = problems.random_miqp(20, 100, 10, seed=3)
    ref = bnb.MIOSQP(backend=digest_backend)
    ref.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
              dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
    r = ref.solve()
    m = bnb.MIOSQP(backend=digest_backend)
    m.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
            dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
    s = dist.ShardedSearch(m)
    s.run(nodes_per_rank=width, batched=True, pipelined=pipelined)
    assert m.work.status == r.status == bnb.MI_SOLVED
    assert abs(m.work.upper_glob - r.upper_glob) <= 1e-3 * max(1.0, abs(r.upper_glob))
    np.testing.assert_array_equal(m.work.x[pr["i_idx"]], r.x[pr["i_idx"]])
    assert s._inflight is None and not m.work.leaves
    if width == 1:  # a wave of one is the sequential search
        assert s.nodes == ref.work.iter_num - 1 and s.iters == ref.work.osqp_iter
