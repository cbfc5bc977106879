"""The sharded search at world sizes 4 and 8 (threads of one process, CPU oracle as the engine): every rank
ends with the same incumbent, it is the sequential optimum, leaves are dealt once, ranks that run dry are
fed, and the waves terminate."""
import threading

import numpy as np
import pytest

import digest_backend
from miosqp_amd import bnb, dist, problems
from thread_comm import ThreadComm, ThreadWorld


def _run(world, pr, per_rank, batched):
    tw = ThreadWorld(world)
    out = [None] * world
    err = []

    def main(rank):
        try:
            m = bnb.MIOSQP(backend=digest_backend)
            m.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                    dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
            s = dist.ShardedSearch(m, ThreadComm(tw, rank))
            s.run(nodes_per_rank=per_rank, batched=bool(batched), pipelined=batched == "pipelined")
            out[rank] = dict(upper=m.work.upper_glob, x=np.array(m.work.x), nodes=s.nodes, moved=s.moved,
                             status=m.work.status, leaves=len(m.work.leaves))
        except Exception as e:  # a dead rank would leave the others at a barrier
            err.append(e)
            tw.bar.abort()

    th = [threading.Thread(target=main, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(600) for t in th]
    assert not err, err
    return out


@pytest.mark.parametrize("lag", [0, 1])
@pytest.mark.parametrize("world,per_rank,batched", [(4, 1, False), (8, 2, False), (4, 4, True), (4, 4, "pipelined")])
def test_many_ranks_agree_with_the_sequential_search(world, per_rank, batched, lag, monkeypatch):
    """lag 0: blocking exchange after every step; lag 1: the exchange of a step is applied one step later."""
    monkeypatch.setenv("MIOSQP_EXCHANGE_LAG", str(lag))
    pr = problems.random_miqp(30, 150, 15, seed=4)
    ref = bnb.MIOSQP(backend=digest_backend)
    ref.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
              dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
    r = ref.solve()
    out = _run(world, pr, per_rank, batched)
    ii = pr["i_idx"]
    for o in out:
        assert o["status"] == bnb.MI_SOLVED and o["leaves"] == 0
        assert o["upper"] == out[0]["upper"]
        np.testing.assert_array_equal(o["x"], out[0]["x"])
        assert abs(o["upper"] - r.upper_glob) <= 1e-3 * max(1.0, abs(r.upper_glob))
        np.testing.assert_array_equal(o["x"][ii], r.x[ii])
    assert sum(o["nodes"] for o in out) >= 1


@pytest.mark.parametrize("budget", [1e-9, 0.05])
def test_time_budgeted_steps_reach_the_same_optimum(budget):
    """step(budget=...): node relaxations until the budget is spent (at least one) instead of a fixed count."""
    pr = problems.random_miqp(30, 150, 15, seed=4)
    ref = bnb.MIOSQP(backend=digest_backend)
    ref.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
              dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
    r = ref.solve()
    world = 4
    tw = ThreadWorld(world)
    out, err = [None] * world, []

    def main(rank):
        try:
            m = bnb.MIOSQP(backend=digest_backend)
            m.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                    dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
            s = dist.ShardedSearch(m, ThreadComm(tw, rank))
            while s.step(budget=budget) != 0:
                pass
            s.drain()
            out[rank] = (m.work.upper_glob, np.array(m.work.x), len(m.work.leaves))
        except Exception as e:
            err.append(e)
            tw.bar.abort()

    th = [threading.Thread(target=main, args=(k,)) for k in range(world)]
    [t.start() for t in th]
    [t.join(600) for t in th]
    assert not err, err
    for up, x, nl in out:
        assert nl == 0 and up == out[0][0]
        assert abs(up - r.upper_glob) <= 1e-3 * max(1.0, abs(r.upper_glob))
        # the same integer assignment, to the integrality tolerance (which node delivers the incumbent differs)
        np.testing.assert_allclose(x[pr["i_idx"]], r.x[pr["i_idx"]], atol=1e-3, rtol=0)


def _threads(world, body):
    tw = ThreadWorld(world)
    out, err = [None] * world, []

    def main(rank):
        try:
            out[rank] = body(rank, ThreadComm(tw, rank))
        except Exception as e:
            err.append(e)
            tw.bar.abort()

    th = [threading.Thread(target=main, args=(k,)) for k in range(world)]
    [t.start() for t in th]
    [t.join(600) for t in th]
    assert not err, err
    return out


def test_replicated_phase_survives_a_rank_that_disagrees():
    """one rank's relaxations differ slightly during the replicated ramp-up (as after a cooperative
    launch was called off on that GPU).  Rank 2 is made to lose a leaf after its first replicated node: the
    all-gather of (count, checksum, incumbent) notices, everybody adopts rank 0's leaves, the deal partitions ONE
    list, no collective is left unmatched and the optimum is the sequential one."""
    pr = problems.random_miqp(50, 100, 30, seed=5)  # ~100 nodes sequentially
    ref = bnb.MIOSQP(backend=digest_backend)
    ref.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
              dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
    r = ref.solve()

    def body(rank, comm):
        m = bnb.MIOSQP(backend=digest_backend)
        m.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
        s = dist.ShardedSearch(m, comm)
        if rank == 2:
            real, state = s._visit, dict(done=False)

            def bad_visit(rule):
                leaf = real(rule)
                if s.replicated and not state["done"] and len(m.work.leaves) >= 2:
                    m.work.leaves.pop()  # this rank now disagrees with the others
                    state["done"] = True
                return leaf
            s._visit = bad_visit
        s.run(nodes_per_rank=1)
        return dict(upper=m.work.upper_glob, x=np.array(m.work.x), status=m.work.status, resyncs=s.resyncs,
                    leaves=len(m.work.leaves), gnodes=s.global_nodes, nodes=s.nodes)

    out = _threads(4, body)
    assert all(o["resyncs"] == 1 for o in out)
    for o in out:
        assert o["status"] == bnb.MI_SOLVED and o["leaves"] == 0 and o["upper"] == out[0]["upper"]
        assert abs(o["upper"] - r.upper_glob) <= 1e-3 * max(1.0, abs(r.upper_glob))
        np.testing.assert_array_equal(o["x"][pr["i_idx"]], r.x[pr["i_idx"]])
        assert o["gnodes"] == out[0]["gnodes"] == sum(q["nodes"] for q in out)  # every rank knows the global count


def test_global_node_budget_and_uniform_status():
    """max_iter_bb is a budget on the nodes of ALL ranks together, every rank stops after the same step
    and reports the same status (the reference's rule, workspace.py:352-373, applied to the whole tree)."""
    pr = problems.random_miqp(30, 150, 15, seed=4)

    def body(rank, comm):
        st = dict(problems.BNB_SETTINGS, max_iter_bb=12)
        m = bnb.MIOSQP(backend=digest_backend)
        m.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], st,
                dict(problems.QP_SETTINGS))
        s = dist.ShardedSearch(m, comm)
        waves = s.run(nodes_per_rank=1)
        return dict(status=m.work.status, gnodes=s.global_nodes, waves=waves, avg=m.work.osqp_iter_avg)

    out = _threads(4, body)
    assert len(set(o["status"] for o in out)) == 1 and len(set(o["waves"] for o in out)) == 1
    assert out[0]["status"] in (bnb.MI_MAX_ITER_FEASIBLE, bnb.MI_MAX_ITER_UNSOLVED)
    assert 11 <= out[0]["gnodes"] <= 11 + 4  # stops once the budget is reached; overshoot < one step of 4 ranks
    assert out[0]["avg"] > 0 and len(set(o["avg"] for o in out)) == 1


@pytest.mark.parametrize("world,cols,every,deal_to", [(2, 8, 2, None), (4, 4, 3, None), (3, 16, 1, None),
                                                      (2, 2, 1, 1), (4, 2, 2, 0)])
def test_sharded_stream_agrees_with_the_sequential_search(world, cols, every, deal_to):
    """dist.ShardedStream: every rank streams its own leaf pool (CPU emulation of the pool calls), the ranks meet
    every `every` chunks for the incumbent and to feed dry ranks; all end with the same incumbent, the sequential
    optimum, an empty tree and every pool slot returned."""
    pr = problems.random_miqp(50, 100, 30, seed=5)  # ~100 nodes sequentially
    ref = bnb.MIOSQP(backend=digest_backend)
    ref.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
              dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
    r = ref.solve()

    def body(rank, comm):
        m = bnb.MIOSQP(backend=digest_backend)
        m.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 6), dict(problems.QP_SETTINGS))
        s = dist.ShardedStream(m, comm, columns=cols, exchange_every=every, capacity=512, ramp_leaves=2, feed=4,
                               deal_to=deal_to)
        s.run()
        return dict(upper=m.work.upper_glob, x=np.array(m.work.x), status=m.work.status, moved=s.moved,
                    free=len(s.ss.free), cap=s.ss.capacity, gnodes=s.global_nodes, alive=s.total_alive,
                    local=s.ss.nodes)

    out = _threads(world, body)
    for o in out:
        assert o["status"] == bnb.MI_SOLVED and o["alive"] == 0 and o["free"] == o["cap"]
        assert o["upper"] == out[0]["upper"]
        assert abs(o["upper"] - r.upper_glob) <= 1e-3 * max(1.0, abs(r.upper_glob))
        np.testing.assert_array_equal(o["x"][pr["i_idx"]], r.x[pr["i_idx"]])
        assert o["gnodes"] == out[0]["gnodes"]
    assert all(o["local"] >= 1 for o in out)  # every rank solved nodes from its own pool
    if deal_to is not None:  # all leaves dealt to one rank: the others were fed by it
        assert out[deal_to]["moved"] >= world - 1
