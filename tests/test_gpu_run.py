"""The hosted search with the cooperative grid RESIDENT for a whole miosqp_qp_search_run (kernels_coop.inc: k_coop_run; the
host feeds it nodes through a mailbox) against the same search with one cooperative launch per node (MIOSQP_COOP_RUN=0):
the loop of /root/reference/miosqp/solver.py:85-123 around Node.solve (/root/reference/miosqp/node.py:96-143).  A node is a
pure function of (l, u, x0, y0) and both forms run the same instructions on the same operands, so everything is compared
for EQUALITY: nodes, ADMM iterations, incumbent value and vector."""
import numpy as np
import pytest

from miosqp_amd import problems

pytestmark = pytest.mark.gpu


def _search(pr, rule, run, monkeypatch, max_nodes=10 ** 9, capacity=None, step=None, qp=None):
    from miosqp_amd import bnb, search
    monkeypatch.setenv("MIOSQP_COOP_RUN", "1" if run else "0")
    st = dict(problems.BNB_SETTINGS, tree_explor_rule=rule, device_tree=False)
    mdl = bnb.MIOSQP()
    mdl.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], st,
              dict(problems.QP_SETTINGS, coop=1, resident=0, **(qp or {})))
    eng = mdl.work.solver
    assert eng.factor_stats()["coop"] is True
    hs = search.HostedSearch(mdl, capacity=capacity)
    eng.loop_stats(reset=True)
    calls = 0
    while hs.nodes < max_nodes:
        calls += 1
        if hs.step(min(step or 10 ** 9, max_nodes - hs.nodes)) == 0:
            break
    ms, iters = eng.loop_stats()
    out = dict(nodes=hs.nodes, iters=hs.iters, upper=float(mdl.work.upper_glob),
               x=None if mdl.work.x is None else np.array(mdl.work.x, dtype=float), launches=eng.loop_launches(), calls=calls,
               loop_iters=iters, loop_ms=ms, node_stats=eng.node_stats(), fallbacks=eng.factor_stats()["coop_fallbacks"],
               open=hs._open)
    eng.close()
    return out


def _same(a, b):
    assert (a["nodes"], a["iters"], a["open"]) == (b["nodes"], b["iters"], b["open"])
    assert a["upper"] == b["upper"]
    if a["x"] is None:
        assert b["x"] is None
    else:
        np.testing.assert_array_equal(a["x"], b["x"])


@pytest.mark.parametrize("n,m,p,seed,rule", [(120, 200, 60, 3, 1), (150, 300, 40, 5, 0), (200, 150, 100, 9, 1), (96, 100, 30, 1, 1)])
def test_resident_run_equals_a_launch_per_node(n, m, p, seed, rule, monkeypatch):
    pr = problems.random_miqp(n, m, p, seed=seed)
    a = _search(pr, rule, True, monkeypatch, max_nodes=400)
    b = _search(pr, rule, False, monkeypatch, max_nodes=400)
    _same(a, b)
    assert a["fallbacks"] == 0 and b["fallbacks"] == 0
    # the resident grid: one launch per call of search_run; the other form: one per node
    assert b["launches"] == b["nodes"]
    assert a["launches"] == a["calls"] and (a["nodes"] < 3 or a["launches"] < a["nodes"])
    assert a["loop_iters"] == a["iters"] and a["loop_ms"] > 0
    # every node of the run carries its own device time (stamps of the first tester)
    assert a["node_stats"][3] == a["nodes"] and a["node_stats"][0] > 0


def test_resident_run_in_pieces_and_with_a_growing_store(monkeypatch):
    """search_run called for 1, 2, 3 ... nodes at a time (every call its own launch), on a slot store that starts with four
    slots (it grows through copies on the engine's stream: the run is ended and started again around them)."""
    pr = problems.random_miqp(120, 200, 60, seed=3)
    whole = _search(pr, 1, True, monkeypatch, max_nodes=150)
    for step in (1, 7):
        pieces = _search(pr, 1, True, monkeypatch, max_nodes=150, capacity=4, step=step)
        _same(whole, pieces)
        assert pieces["launches"] >= pieces["calls"]


def test_resident_run_at_config2_size(monkeypatch):
    """BASELINE configs[1] (n=500, m=1000, p=250): the first 60 nodes, resident grid against a launch per node -- and the
    headline's kernel is the resident one by default."""
    pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
    a = _search(pr, 1, True, monkeypatch, max_nodes=60)
    b = _search(pr, 1, False, monkeypatch, max_nodes=60)
    _same(a, b)
    assert a["launches"] == 1 and b["launches"] == 60
    monkeypatch.delenv("MIOSQP_COOP_RUN")
    from miosqp_amd import bnb, search
    mdl = bnb.MIOSQP()
    mdl.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(problems.BNB_SETTINGS),
              dict(problems.QP_SETTINGS))
    hs = search.HostedSearch(mdl)
    mdl.work.solver.loop_stats(reset=True)
    hs.step(20)
    assert hs.nodes == 20 and mdl.work.solver.loop_launches() == 1


def test_resident_run_with_rho_chosen_at_setup(monkeypatch):
    """the same with rho="auto" (shorter nodes: the host's turn-around is a larger share of each)"""
    pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
    a = _search(pr, 1, True, monkeypatch, max_nodes=120, qp=dict(rho="auto"))
    b = _search(pr, 1, False, monkeypatch, max_nodes=120, qp=dict(rho="auto"))
    _same(a, b)


def test_resident_run_that_is_called_off_touches_nothing(monkeypatch):
    """Workgroup 1 never shows up (MIOSQP_COOP_DBG=64): the resident grid calls itself off at its registration, the node goes
    through a launch of its own, which is called off the same way, and the search ends in the multi-kernel form with the
    result of an engine that never was cooperative."""
    from miosqp_amd import bnb
    pr = problems.random_miqp(60, 120, 30, seed=11)
    st = dict(problems.BNB_SETTINGS, device_tree=False)
    monkeypatch.setenv("MIOSQP_COOP_RUN", "1")
    monkeypatch.setenv("MIOSQP_COOP_NAP", "12")
    monkeypatch.setenv("MIOSQP_COOP_DBG", "64")
    bad = bnb.MIOSQP()
    bad.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st),
              dict(problems.QP_SETTINGS, coop=1, resident=0))
    monkeypatch.delenv("MIOSQP_COOP_DBG")
    ref = bnb.MIOSQP()
    ref.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st),
              dict(problems.QP_SETTINGS, coop=0, resident=0))
    r0, r1 = ref.solve(), bad.solve()
    fs = bad.work.solver.factor_stats()
    assert fs["coop"] is False and fs["coop_fallbacks"] >= 2  # the run, then the node's own launch
    assert (r1.status, bad.work.iter_num, bad.work.osqp_iter) == (r0.status, ref.work.iter_num, ref.work.osqp_iter)
    assert r1.upper_glob == r0.upper_glob
    np.testing.assert_array_equal(r1.x, r0.x)


def test_resident_run_with_a_time_budget(monkeypatch):
    """budget_s is checked between nodes: the call returns with the grid stopped and the search resumable"""
    from miosqp_amd import bnb, search
    pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
    mdl = bnb.MIOSQP()
    mdl.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(problems.BNB_SETTINGS),
              dict(problems.QP_SETTINGS))
    hs = search.HostedSearch(mdl)
    hs.step(10 ** 9, budget=0.004)
    first = hs.nodes
    assert 1 <= first <= 12
    hs.step(10 ** 9, budget=0.004)
    assert hs.nodes > first


def test_resident_run_calibrates_its_poll_delay_on_the_first_node_and_leaves_no_trace(monkeypatch):
    """MIOSQP_COOP_RUN_CAL=2: the resident grid's own poll delay is MEASURED on the first node of the search (runs of one
    synthetic 300-iteration node each: tolerances nothing can meet, an incumbent value that keeps the epilogue from branching,
    the solution written into a free slot) whatever the caches hold.  The search that follows equals the search with a launch
    per node -- nodes, iterations, incumbent --, the calibration's launches and nodes appear in none of its statistics, and
    the value it chose is within the range the candidates span."""
    pr = problems.random_miqp(170, 260, 30, seed=2)
    monkeypatch.setenv("MIOSQP_COOP_RUN_CAL", "2")
    from miosqp_amd import bnb, search
    monkeypatch.setenv("MIOSQP_COOP_RUN", "1")
    st = dict(problems.BNB_SETTINGS, device_tree=False)
    mdl = bnb.MIOSQP()
    mdl.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], st,
              dict(problems.QP_SETTINGS, coop=1, resident=0))
    eng = mdl.work.solver
    launch_nap = eng.factor_stats()["coop_nap"]
    hs = search.HostedSearch(mdl)
    eng.loop_stats(reset=True)
    calls = 0
    while hs.nodes < 200:
        calls += 1
        if hs.step(200 - hs.nodes) == 0:
            break
    nap, cal_nodes = eng.resident_grid_poll_delay()
    assert cal_nodes >= 8 and launch_nap - 5 <= nap <= launch_nap + 3, (nap, launch_nap, cal_nodes)
    assert eng.loop_launches() == calls and eng.loop_stats()[1] == hs.iters and eng.node_stats()[3] == hs.nodes
    a = dict(nodes=hs.nodes, iters=hs.iters, open=hs._open, upper=float(mdl.work.upper_glob),
             x=None if mdl.work.x is None else np.array(mdl.work.x, dtype=float))
    eng.close()
    monkeypatch.delenv("MIOSQP_COOP_RUN_CAL")
    b = _search(pr, 1, False, monkeypatch, max_nodes=200)
    _same(a, b)


def test_two_resident_searches_of_one_process_take_turns_on_one_chip(monkeypatch):
    """Two "ranks" as host threads of ONE process (miosqp_amd/poolcomm.py), each with an engine of its own on the same GPU and
    the leaf-sharded hosted search (dist.ShardedStream over search.HostedSearch): both engines keep their cooperative grid
    RESIDENT over their search_run calls although they share the chip -- a resident run is one whole-chip launch like any
    other and holds its turn for the call (host_search.inc: run_possible) --, nothing is called off, the launches were
    ordered behind each other, and the tree closes with the optimum of the sequential search.  (The multi-rank GPU tests that
    use separate processes on one device cannot run the headline's kernel: two processes' grids never become co-resident.)"""
    import threading
    from miosqp_amd import bnb, dist, poolcomm, search
    monkeypatch.delenv("MIOSQP_COOP_RUN", raising=False)
    pr = problems.random_miqp(120, 200, 60, seed=3)
    st = dict(problems.BNB_SETTINGS, device_tree=False, max_iter_bb=10 ** 6)
    qs = dict(problems.QP_SETTINGS, coop=1, resident=0)
    seq = bnb.MIOSQP()
    seq.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st), dict(qs))
    r0 = seq.solve()
    seq.work.solver.close()
    world = 2
    tw = poolcomm.PoolWorld(world)
    out, err = [None] * world, []

    def body(rank):
        try:
            m = bnb.MIOSQP()
            m.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st), dict(qs))
            eng = m.work.solver
            hs = search.HostedSearch(m, capacity=1024)
            sh = dist.ShardedStream(m, poolcomm.PoolComm(tw, rank), search=hs, step_kwargs=dict(nodes=6), exchange_every=1,
                                    ramp_leaves=1, feed=4)
            sh.run()
            fs = eng.factor_stats()
            out[rank] = dict(upper=float(m.work.upper_glob), x=np.array(m.work.x), status=m.work.status, local=hs.nodes,
                             resident=fs["search_grid_resident"], coop=fs["coop"], fallbacks=fs["coop_fallbacks"],
                             users=eng.chip_turn_users(), waits=eng.chip_turn_waits(), launches=eng.loop_launches())
            eng.close()
        except BaseException as e:  # noqa: BLE001
            err.append(e)
            tw.fail(e)

    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(600) for t in th]
    assert not err, err
    for o in out:
        assert o["status"] == bnb.MI_SOLVED and o["coop"] and o["resident"] and o["fallbacks"] == 0, o
        assert o["upper"] == out[0]["upper"]
        assert abs(o["upper"] - r0.upper_glob) <= 1e-3 * max(1.0, abs(r0.upper_glob))
        np.testing.assert_array_equal(np.round(o["x"][pr["i_idx"]]), np.round(r0.x[pr["i_idx"]]))
        # a launch per CALL of the search (a few nodes each), not per node
        assert o["local"] == 0 or o["launches"] < o["local"], o
    assert max(o["waits"] for o in out) > 0 and all(o["local"] > 0 for o in out), out


def test_leaf_with_inverted_integer_bounds_is_refused_where_the_host_may_read_them():
    """add_leaf with host vectors (numpy: memory the runtime does not know, or knows as host memory) checks l <= u on the host,
    as update(l, u) does for the reference (/root/reference/miosqp/node.py:102 -> osqp's "Lower bound must be lower than or
    equal to upper bound"); device tensors are never dereferenced on the host (host.inc: host_may_read -- ADVICE r5: the check
    used to run whenever the runtime did not say "device", which a tensor of another runtime instance would not)."""
    from miosqp_amd import bnb, search
    pr = problems.random_miqp(60, 120, 30, seed=11)
    mdl = bnb.MIOSQP()
    mdl.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
              dict(problems.BNB_SETTINGS, device_tree=False), dict(problems.QP_SETTINGS, resident=0))
    hs = search.HostedSearch(mdl)
    hs.begin_instance(seed_root=False)
    root = mdl.work._make_root()
    lo, hi = root.l[-30:].copy(), root.u[-30:].copy()
    lo[3], hi[3] = 1.0, 0.0
    with pytest.raises(ValueError):
        hs.add_leaf(lo, hi, np.zeros(60), np.zeros(150), 0, -np.inf)
    hs.add_leaf(root.l[-30:], root.u[-30:], np.zeros(60), np.zeros(150), 0, -np.inf)
    assert hs.step(1) >= 0 and hs.nodes == 1
    mdl.work.solver.close()
