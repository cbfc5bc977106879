"""The stream on the leaf pool as ONE persistent launch (miosqp_amd/csrc/kernels_bstream.inc: kbs -- BASELINE configs[2]'s
form: 256 node relaxations side by side, /root/reference/miosqp/node.py:96-143 per column, bound_and_branch of
/root/reference/miosqp/workspace.py:282-334 per decided column) against the same stream as the chunk graph
(MIOSQP_KBS=0: thirteen launches per chunk, the form of rounds 2-4), against the sequential search, and through a
call-off in the middle of a stream.

A node is a pure function of (l, u, x0, y0): the two forms visit the nodes of one tree in different orders (which leaf
is pushed when depends on what the launch in flight has freed), but a node BOTH have decided -- identified by its integer
bounds -- must have come out the same: identical status and iteration count, bound / x / y within the tolerances of
tests/test_gpu_parity.py (SOL_TOL = 1e-8 relative for x and y, 1e-9 relative for the bound)."""
import hashlib

import numpy as np
import pytest

from miosqp_amd import problems

pytestmark = pytest.mark.gpu
SOL_TOL = 1e-8


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) / max(1.0, float(np.max(np.abs(b))))


def _model(pr, cols, rule=1, **qp):
    from miosqp_amd import bnb
    st = dict(problems.BNB_SETTINGS, tree_explor_rule=rule, max_iter_bb=10 ** 9)
    m = bnb.MIOSQP()
    m.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], st,
            dict(problems.QP_SETTINGS, max_batch=cols, **qp))
    return m


def _record_stream(pr, cols, kbs, monkeypatch, max_nodes, read_every=1):
    """the Python stream driver with an observer that keeps, per decided node (keyed by its integer bounds), the
    digest's (status, iterations, bound) and -- every `read_every`-th node -- the slot's solution"""
    from miosqp_amd import stream
    monkeypatch.setenv("MIOSQP_KBS", "1" if kbs else "0")
    model = _model(pr, cols)
    p = len(pr["i_idx"])
    seen = {}
    count = [0]

    def obs(search, g):
        if int(g["status_val"]) == -100:  # pruned before it was solved
            return
        count[0] += 1
        s = int(g["slot"])
        want = ("l", "u", "x", "y") if count[0] % read_every == 0 else ("l", "u")
        nd = search.eng.pool_read_node(s, p, want=want)
        key = hashlib.sha1(np.ascontiguousarray(nd.l).tobytes() + np.ascontiguousarray(nd.u).tobytes()).hexdigest()
        seen[key] = (int(g["status_val"]), int(g["iter"]), float(g["lower"]),
                     None if "x" not in want else np.array(nd.x), None if "y" not in want else np.array(nd.y))

    srch = stream.StreamSearch(model, columns=cols, observer=obs)
    eng = model.work.solver
    alive, steps = 1, 0
    while alive and srch.nodes < max_nodes and steps < 4000:
        alive = srch.step()
        steps += 1
    forms = eng.stream_chunks_by_form()
    out = dict(seen=seen, nodes=srch.nodes, forms=forms, called_off=eng.batch_pers_fallbacks(),
               pers=eng.factor_stats()["batch_pers"], closed=alive == 0, upper=float(model.work.upper_glob))
    eng.close()
    return out


def _same_nodes(a, b, least):
    common = sorted(set(a["seen"]) & set(b["seen"]))
    assert len(common) >= least, (len(common), len(a["seen"]), len(b["seen"]))
    xs = 0
    for k in common:
        (sa, ia, la, xa, ya), (sb, ib, lb, xb, yb) = a["seen"][k], b["seen"][k]
        assert (sa, ia) == (sb, ib), k
        if sa in (1, -2):
            assert abs(la - lb) <= 1e-9 * max(1.0, abs(lb)), k
            if xa is not None and xb is not None:
                assert rel(xa, xb) <= SOL_TOL and rel(ya, yb) <= SOL_TOL, k
                xs += 1
    return len(common), xs


def test_persistent_stream_equals_the_chunk_graph_at_config3_size(monkeypatch):
    """BASELINE configs[2]'s shape (n=500, m=1000, p=250, 256 columns): the first ~700 nodes of the stream through kbs and
    through the chunk graph.  kbs really ran (its launches counted, none called off, no chunk through the graph), the
    other run never launched it; every node both decided agrees."""
    pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
    a = _record_stream(pr, 256, True, monkeypatch, 700, read_every=4)
    b = _record_stream(pr, 256, False, monkeypatch, 700, read_every=4)
    assert a["pers"] and a["forms"][0] >= 2 and a["forms"][1] >= 16 and a["forms"][2] == 0 and a["called_off"] == 0, a["forms"]
    assert b["forms"][0] == 0 and b["forms"][2] >= 16, b["forms"]
    common, xs = _same_nodes(a, b, 300)
    print("config 3: %d / %d nodes decided, %d in common, %d solutions compared" % (a["nodes"], b["nodes"], common, xs))
    assert xs >= 40


SHAPES = [(257, 300, 60, 64), (300, 411, 77, 128), (333, 500, 101, 256), (384, 640, 128, 256), (401, 333, 200, 192),
          (449, 700, 50, 256), (480, 512, 240, 64), (500, 1000, 250, 128), (511, 513, 255, 256), (512, 900, 100, 256),
          (259, 761, 129, 192), (350, 350, 175, 256)]


@pytest.mark.parametrize("n,m,p,cols", SHAPES)
def test_persistent_stream_on_odd_shapes(n, m, p, cols, monkeypatch):
    """a dozen shapes inside kbs's limits (256 <= n <= 512, m <= 1024 general rows -- the integer bound rows are not in the
    products --, n + m <= 1536; odd and even sizes, column counts below the full 256): the first nodes of each stream through
    both forms"""
    assert m <= 1024 and n + m <= 1536
    monkeypatch.setenv("MIOSQP_KBP_MIN_COLS", "1")  # (below 192 columns the engine prefers the launches by itself)
    pr = problems.random_miqp(n, m, p, seed=1000 + n)
    a = _record_stream(pr, cols, True, monkeypatch, 150, read_every=5)
    b = _record_stream(pr, cols, False, monkeypatch, 150, read_every=5)
    assert a["pers"] and a["forms"][0] >= 1 and a["forms"][2] == 0 and a["called_off"] == 0, (a["forms"], a["called_off"])
    assert b["forms"][0] == 0
    _same_nodes(a, b, min(60, min(a["nodes"], b["nodes"]) // 3))


@pytest.mark.parametrize("driver", ["native", "python"])
def test_persistent_stream_closes_trees_with_the_sequential_optimum(driver, monkeypatch):
    """whole trees (MIOSQP_KBP=1 takes the persistent forms below their automatic range, n >= 256) closed by the stream in
    its persistent form and by the hosted node-at-a-time search: status, optimum, integer part; every slot returned; kbs
    ran and was never called off"""
    from miosqp_amd import bnb, stream
    monkeypatch.setenv("MIOSQP_KBP", "1")
    monkeypatch.setenv("MIOSQP_KBP_MIN_COLS", "1")
    monkeypatch.setenv("MIOSQP_KBS", "1")
    ran = 0
    for n, m, p, cols in [(40, 60, 12, 64), (64, 100, 16, 128), (97, 131, 14, 256), (120, 200, 20, 256), (150, 150, 18, 64)]:
        seed = 40 + n
        pr = problems.random_miqp(n, m, p, seed=seed)
        st = dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 6)
        ref = bnb.MIOSQP()
        ref.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st), dict(problems.QP_SETTINGS))
        r0 = ref.solve()
        ref.work.solver.close()
        mdl = _model(pr, cols)
        eng = mdl.work.solver
        s = (stream.NativeStreamSearch if driver == "native" else stream.StreamSearch)(mdl, columns=cols, capacity=4096)
        r1 = s.run()
        forms = eng.stream_chunks_by_form()
        assert r1.status == r0.status and len(s.free) == s.capacity, (n, m, p, r1.status, r0.status)
        if np.isfinite(r0.upper_glob):
            assert abs(r1.upper_glob - r0.upper_glob) <= 1e-3 * max(1.0, abs(r0.upper_glob))
            np.testing.assert_array_equal(np.round(r1.x[pr["i_idx"]]), np.round(r0.x[pr["i_idx"]]))
        if eng.factor_stats()["batch_pers"]:
            assert forms[0] >= 1 and eng.batch_pers_fallbacks() == 0, forms
            ran += 1
        eng.close()
    assert ran >= 3  # (a shape whose factor form rules the persistent sweeps out runs the chunk graph: still checked above)


@pytest.mark.parametrize("driver", ["native", "python"])
def test_stream_survives_a_persistent_launch_called_off_mid_stream(driver, monkeypatch):
    """Workgroup 1 of the THIRD launch of kbs shows up 150 ms late (MIOSQP_KBS_FAULT_AT=3): the launch calls itself off at its
    registration having touched nothing -- the columns' iterates, the slots, the ready ring and the incumbent are as the
    launch before left them --, the engine goes on with the chunk graph, and the tree closes with the sequential optimum
    and every slot returned.  (The twin of test_called_off_launch_leaves_slot_children_and_incumbent_alone for the pool.)"""
    from miosqp_amd import bnb, stream
    pr = problems.random_miqp(300, 400, 40, seed=5)
    st = dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 6)
    ref = bnb.MIOSQP()
    ref.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st), dict(problems.QP_SETTINGS))
    r0 = ref.solve()
    ref.work.solver.close()
    monkeypatch.setenv("MIOSQP_KBS", "1")
    monkeypatch.setenv("MIOSQP_KBS_FAULT_AT", "3")
    monkeypatch.setenv("MIOSQP_KBP_MIN_COLS", "1")
    mdl = _model(pr, 128)
    eng = mdl.work.solver
    s = (stream.NativeStreamSearch if driver == "native" else stream.StreamSearch)(mdl, columns=128, capacity=8192)
    r1 = s.run()
    forms = eng.stream_chunks_by_form()
    assert eng.factor_stats()["batch_pers"]  # (back in the persistent form at the end: the engine tries again after a call-off)
    # (the launch was called off -- once --, and the stream went on: through the chunk graph for a while, or, when nothing was
    #  in flight at the next round, straight back in the persistent form)
    assert forms[0] >= 3 and eng.batch_pers_fallbacks() >= 1, (forms, eng.batch_pers_fallbacks())
    assert r1.status == r0.status == bnb.MI_SOLVED and len(s.free) == s.capacity
    assert abs(r1.upper_glob - r0.upper_glob) <= 1e-3 * max(1.0, abs(r0.upper_glob))
    np.testing.assert_array_equal(np.round(r1.x[pr["i_idx"]]), np.round(r0.x[pr["i_idx"]]))
    eng.close()
