"""Host logic around the hosted node-at-a-time search (miosqp_amd/search.py, bnb.MIOSQP._solve_hosted,
dist.ShardedStream over a HostedSearch) on CPU: the engine calls miosqp_qp_search_* are emulated by
tests/digest_backend.py with the oracle doing the relaxations."""
import numpy as np
import pytest

import digest_backend
from miosqp_amd import bnb, dist, problems, search
from test_dist_threads import _threads


def _model(pr, **st):
    m = bnb.MIOSQP(backend=digest_backend)
    m.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
            dict(problems.BNB_SETTINGS, **st), dict(problems.QP_SETTINGS))
    return m


@pytest.mark.parametrize("n,m,p,seed,rule", [(30, 150, 15, 4, 1), (20, 40, 10, 1, 0), (40, 100, 25, 2, 1)])
def test_hosted_search_is_the_python_loop(n, m, p, seed, rule):
    pr = problems.random_miqp(n, m, p, seed=seed)
    py, cc = _model(pr, tree_explor_rule=rule, device_search=False), _model(pr, tree_explor_rule=rule)
    rng = np.random.RandomState(seed)
    for inst in range(2):
        r0, r1 = py.solve(), cc.solve()
        assert getattr(cc.work, "_hosted", None) is not None and getattr(py.work, "_hosted", None) is None
        assert (r1.status, cc.work.iter_num, cc.work.osqp_iter) == (r0.status, py.work.iter_num, py.work.osqp_iter)
        assert abs(r1.upper_glob - r0.upper_glob) <= 1e-12 * max(1.0, abs(r0.upper_glob))
        np.testing.assert_allclose(r1.x, r0.x, rtol=0, atol=1e-12)
        q2 = rng.randn(n)
        py.update_vectors(q=q2)
        cc.update_vectors(q=q2)


def test_node_cap_and_initial_incumbent():
    pr = problems.random_miqp(40, 100, 25, seed=2)
    full = _model(pr)
    r = full.solve()
    capped = _model(pr, max_iter_bb=5)
    rc = capped.solve()
    assert capped.work.iter_num == 5 and rc.status in (bnb.MI_MAX_ITER_FEASIBLE, bnb.MI_MAX_ITER_UNSOLVED)
    # MIOSQP.set_x0 with the optimum: the search starts with that incumbent and visits no more nodes than without
    warm = _model(pr)
    warm.set_x0(r.x)
    rw = warm.solve()
    assert rw.status == bnb.MI_SOLVED and rw.upper_glob <= r.upper_glob + 1e-9 and warm.work.iter_num <= full.work.iter_num


@pytest.mark.parametrize("world,every,deal_to", [(2, 1, None), (3, 2, 0), (4, 1, 1)])
def test_sharded_hosted_search(world, every, deal_to):
    """dist.ShardedStream over HostedSearch: every rank runs the node-at-a-time loop on its share; incumbents are
    exchanged, dry ranks are fed; all end with the sequential optimum."""
    pr = problems.random_miqp(50, 100, 30, seed=5)
    r = _model(pr, device_search=False).solve()

    def body(rank, comm):
        m = _model(pr, max_iter_bb=10 ** 6)
        hs = search.HostedSearch(m, capacity=512)
        s = dist.ShardedStream(m, comm, search=hs, step_kwargs=dict(nodes=2), exchange_every=every, ramp_leaves=2, feed=4,
                               deal_to=deal_to)
        s.run()
        return dict(upper=m.work.upper_glob, x=np.array(m.work.x), status=m.work.status, moved=s.moved, local=hs.nodes,
                    alive=s.total_alive, gnodes=s.global_nodes)

    out = _threads(world, body)
    for o in out:
        assert o["status"] == bnb.MI_SOLVED and o["alive"] == 0
        assert o["upper"] == out[0]["upper"] and o["gnodes"] == out[0]["gnodes"]
        assert abs(o["upper"] - r.upper_glob) <= 1e-3 * max(1.0, abs(r.upper_glob))
        np.testing.assert_array_equal(o["x"][pr["i_idx"]], r.x[pr["i_idx"]])
    if deal_to is not None:
        assert out[deal_to]["moved"] >= 1


def test_dropping_the_model_frees_the_engine_at_once():
    """The hosted search object is kept by the model's workspace and refers back weakly: no reference cycle, so the engine
    (device pool, stream and pinned-buffer bundle on the GPU) goes when the model goes -- not when the cyclic collector
    happens to run (the next setup would pay for a new bundle)."""
    import gc
    import weakref
    pr = problems.random_miqp(20, 40, 10, seed=1)
    was = gc.isenabled()
    gc.disable()
    try:
        m = _model(pr)
        m.solve()
        assert getattr(m.work, "_hosted", None) is not None
        refs = [weakref.ref(m), weakref.ref(m.work), weakref.ref(m.work.solver), weakref.ref(m.work._hosted)]
        m = None
        assert all(r() is None for r in refs)
    finally:
        if was:
            gc.enable()
