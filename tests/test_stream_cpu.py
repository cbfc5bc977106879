"""Host side of the streaming search (miosqp_amd/stream.py) on CPU: the leaf-pool calls are emulated by
tests/digest_backend.py (oracle relaxations, same slot / ring / harvest semantics as the device), so slot lifetime,
pruning, push order, the one-launch-deep pipeline and termination are exercised without a GPU."""
import numpy as np
import pytest

import digest_backend
from miosqp_amd import bnb, problems, stream


def _models(pr, **st):
    out = []
    for _ in range(2):
        m = bnb.MIOSQP(backend=digest_backend)
        m.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                dict(problems.BNB_SETTINGS, **st), dict(problems.QP_SETTINGS))
        out.append(m)
    return out


@pytest.mark.parametrize("n,m,p,seed,cols,rule", [(20, 100, 10, 3, 4, 1), (30, 150, 15, 4, 64, 1), (12, 60, 6, 2, 8, 0),
                                                  (20, 100, 10, 6, 128, 1)])
def test_stream_search_closes_the_tree_with_the_sequential_optimum(n, m, p, seed, cols, rule):
    pr = problems.random_miqp(n, m, p, seed=seed)
    seq, mdl = _models(pr, tree_explor_rule=rule, max_iter_bb=10 ** 6)
    r0 = seq.solve()
    seen = []
    s = stream.StreamSearch(mdl, columns=cols, capacity=512, observer=lambda srch, g: seen.append(int(g["slot"])))
    r1 = s.run()
    assert r1.status == r0.status == bnb.MI_SOLVED
    assert abs(r1.upper_glob - r0.upper_glob) <= 1e-3 * max(1.0, abs(r0.upper_glob))
    np.testing.assert_array_equal(r1.x[pr["i_idx"]], r0.x[pr["i_idx"]])
    assert len(s.free) == s.capacity and s.in_flight == 0 and not s.open  # every slot came back
    assert s.nodes + s.dropped == len(seen) and s.nodes >= 1
    assert mdl.work.osqp_iter == s.iters and mdl.work.iter_num == s.nodes + 1
    # a second MIQP on the same factor reuses the pool
    rng = np.random.RandomState(seed)
    q2 = rng.randn(n)
    for x in (seq, mdl):
        x.update_vectors(q=q2)
    s.begin_instance()
    r0, r1 = seq.solve(), s.run()
    assert r1.status == r0.status
    assert abs(r1.upper_glob - r0.upper_glob) <= 1e-3 * max(1.0, abs(r0.upper_glob))
    assert len(s.free) == s.capacity


def test_stream_search_with_an_infeasible_root_and_with_a_node_cap():
    import scipy.sparse as spa
    # infeasible: x0 + x1 >= 3 with binaries
    P = spa.csc_matrix(np.eye(4))
    A = spa.csc_matrix(np.array([[1.0, 1.0, 0.0, 0.0]]))
    m = bnb.MIOSQP(backend=digest_backend)
    m.setup(P, np.zeros(4), A, np.array([3.0]), np.array([np.inf]), np.array([0, 1]), np.zeros(2), np.ones(2),
            dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
    s = stream.StreamSearch(m, columns=4, capacity=64)
    r = s.run()
    assert r.status == bnb.MI_PRIMAL_INFEASIBLE and s.nodes == 1 and len(s.free) == s.capacity
    # node cap: the search stops with leaves left and says so
    pr = problems.random_miqp(30, 150, 15, seed=4)
    mdl = _models(pr, max_iter_bb=6)[0]
    s = stream.StreamSearch(mdl, columns=2, capacity=256)
    r = s.run()
    assert r.status in (bnb.MI_MAX_ITER_FEASIBLE, bnb.MI_MAX_ITER_UNSOLVED) and s.nodes >= 5


def test_pool_exhaustion_is_loud():
    pr = problems.random_miqp(30, 150, 15, seed=4)
    mdl = _models(pr, max_iter_bb=10 ** 6)[0]
    s = stream.StreamSearch(mdl, columns=8, capacity=6)
    with pytest.raises(MemoryError):
        s.run()


def test_several_pools_share_one_tree():
    """stream.MultiPoolSearch: two / three pools (threads of this process, poolcomm.PoolComm) on one tree end with the
    sequential optimum; a second instance reuses them."""
    import digest_backend
    from miosqp_amd import bnb, problems, stream
    pr = problems.random_miqp(50, 100, 30, seed=5)

    def make(**st):
        m = bnb.MIOSQP(backend=digest_backend)
        m.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 6, **st), dict(problems.QP_SETTINGS))
        return m

    seq = make(device_search=False)
    for pools in (2, 3):
        mp = stream.MultiPoolSearch(make, pools=pools, columns=4, exchange_every=2, capacity=512)
        r0, r1 = seq.solve(), mp.run()
        assert r1.status == r0.status == bnb.MI_SOLVED
        assert abs(r1.upper_glob - r0.upper_glob) <= 1e-3 * max(1.0, abs(r0.upper_glob))
        np.testing.assert_array_equal(r1.x[pr["i_idx"]], r0.x[pr["i_idx"]])
        assert all(len(sh.ss.free) == sh.ss.capacity for sh in mp.sh)
        q2 = np.random.RandomState(pools).randn(50)
        seq.update_vectors(q=q2)
        mp.update_vectors(q=q2)
        r0, r1 = seq.solve(), mp.run()
        assert r1.status == r0.status
        assert abs(r1.upper_glob - r0.upper_glob) <= 1e-3 * max(1.0, abs(r0.upper_glob))
        seq.update_vectors(q=pr["q"])


def test_the_node_cap_of_run_counts_the_nodes_of_the_current_instance():
    """A driver that lives across a sequence of MIQPs (update_vectors + begin_instance) applies max_iter_bb to every
    instance afresh: with the cap between one tree's size and the sequence's total, every instance still closes."""
    pr = problems.random_miqp(20, 100, 10, seed=3)
    seq, mdl = _models(pr, max_iter_bb=10 ** 6)
    probe = stream.StreamSearch(mdl, columns=8, capacity=512)
    probe.run()
    per_tree = probe.nodes
    assert per_tree >= 3
    seq2, mdl2 = _models(pr, max_iter_bb=2 * per_tree + 8)
    s = stream.StreamSearch(mdl2, columns=8, capacity=512)
    rng = np.random.RandomState(5)
    total = 0
    for inst in range(5):
        r0, r1 = seq2.solve(), s.run()
        assert r1.status == r0.status == bnb.MI_SOLVED, inst
        assert abs(r1.upper_glob - r0.upper_glob) <= 1e-3 * max(1.0, abs(r0.upper_glob))
        total = s.nodes
        q2 = rng.randn(20)
        for x in (seq2, mdl2):
            x.update_vectors(q=q2)
        s.begin_instance()
    assert total > 2 * per_tree + 8  # (the sequence as a whole went past the cap: only a per-instance count lets it)
