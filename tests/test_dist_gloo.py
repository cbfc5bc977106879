"""Multi-process path (leaf sharding + incumbent exchange) on CPU: gloo, world_size 2.

The GPU engine is replaced by the CPU oracle (test infrastructure); what is under test is
miosqp_amd/dist.py: dealing, per-rank exploration, all-gather/broadcast of the incumbent,
pruning against the global bound, termination when every rank runs dry.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from miosqp_amd import bnb, dist, problems

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n,m,p,seed,per_rank,lag", [(20, 100, 10, 3, 1, 1), (30, 150, 15, 4, 3, 1), (30, 150, 15, 4, 1, 0)])
def test_two_ranks_find_the_same_optimum(tmp_path, oracle_mod, n, m, p, seed, per_rank, lag):
    out = str(tmp_path / "res.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", MIOSQP_EXCHANGE_LAG=str(lag))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), out, str(n), str(m), str(p), str(seed), str(per_rank)]
    subprocess.check_call(cmd, env=env, cwd=ROOT, timeout=600)
    recs = [json.load(open("%s.%d" % (out, r))) for r in range(2)]
    # single-process reference run of the same host logic
    pr = problems.random_miqp(n, m, p, seed=seed)
    model = bnb.MIOSQP(backend=oracle_mod)
    model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
    res = model.solve()
    assert res.status == bnb.MI_SOLVED
    for r in recs:
        assert r["status"] == bnb.MI_SOLVED
        assert abs(r["upper"] - recs[0]["upper"]) == 0.0  # both ranks hold the same incumbent
        np.testing.assert_array_equal(r["x"], recs[0]["x"])
        # same optimum as the sequential search, to the relaxation tolerance
        assert abs(r["upper"] - res.upper_glob) <= 1e-3 * max(1.0, abs(res.upper_glob))
        ii = pr["i_idx"]
        np.testing.assert_array_equal(np.asarray(r["x"])[ii], res.x[ii])
    assert recs[0]["dealt_total"] == recs[0]["leaves_before_deal"]  # every leaf dealt exactly once
    assert recs[0]["nodes_total"] >= 1


@pytest.mark.parametrize("every", [1, 3])
def test_two_ranks_stream_their_own_pools(tmp_path, every):
    """dist.ShardedStream over gloo, world_size 2, all leaves dealt to rank 0 (so rank 1 must be fed): the same
    incumbent on both ranks, the sequential optimum, every pool slot returned."""
    import digest_backend
    n, m, p, seed = 50, 100, 30, 5
    out = str(tmp_path / "res.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", MIOSQP_WORKER_MODE="stream")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), out, str(n), str(m), str(p), str(seed), str(every)]
    subprocess.check_call(cmd, env=env, cwd=ROOT, timeout=600)
    recs = [json.load(open("%s.%d" % (out, r))) for r in range(2)]
    pr = problems.random_miqp(n, m, p, seed=seed)
    model = bnb.MIOSQP(backend=digest_backend)
    model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
    res = model.solve()
    for r in recs:
        assert r["status"] == bnb.MI_SOLVED and r["free"] == 512
        assert r["upper"] == recs[0]["upper"] and r["gnodes"] == recs[0]["gnodes"]
        assert abs(r["upper"] - res.upper_glob) <= 1e-3 * max(1.0, abs(res.upper_glob))
        np.testing.assert_array_equal(np.asarray(r["x"])[pr["i_idx"]], res.x[pr["i_idx"]])
        assert r["local"] >= 1
    assert recs[0]["moved_total"] >= 1


def test_local_comm_is_the_sequential_search(oracle_mod):
    pr = problems.random_miqp(20, 100, 10, seed=3)
    a = bnb.MIOSQP(backend=oracle_mod)
    b = bnb.MIOSQP(backend=oracle_mod)
    for mdl in (a, b):
        mdl.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                  dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
    ra = a.solve()
    s = dist.ShardedSearch(b)
    s.run(nodes_per_rank=1)
    assert b.work.status == ra.status and b.work.upper_glob == ra.upper_glob
    np.testing.assert_array_equal(b.work.x, ra.x)
    assert s.nodes == a.work.iter_num - 1 and s.iters == a.work.osqp_iter


def test_four_ranks_in_two_tree_groups(tmp_path, oracle_mod):
    """bench.py's --ranks-per-tree (r06): four processes over gloo split into two groups of two ranks, each group a
    communicator of its own (dist.TorchComm(group=, ranks=)) that closes ITS OWN MIQP with the leaf-sharded hosted search
    (incumbent all-gather, broadcast of x by the owner's GLOBAL rank, leaf hand-over -- all inside the group) while the other
    group does the same on another MIQP; the whole job's communicator still works beside them.  Each group ends with the
    sequential optimum of its own instance."""
    out = str(tmp_path / "res.json")
    n, m, p, seed = 30, 150, 15, 4
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", MIOSQP_WORKER_MODE="groups")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), out, str(n), str(m), str(p), str(seed), "2"]
    subprocess.check_call(cmd, env=env, cwd=ROOT, timeout=900)
    recs = [json.load(open("%s.%d" % (out, r))) for r in range(4)]
    assert [r["group"] for r in recs] == [0, 0, 1, 1] and [r["group_rank"] for r in recs] == [0, 1, 0, 1]
    assert all(r["group_world"] == 2 and r["status"] == bnb.MI_SOLVED and r["alive"] == 0 for r in recs)
    assert recs[0]["nodes_total"] == sum(r["local"] for r in recs)                 # the job's all-reduce saw all four
    for g in (0, 1):
        a, b = recs[2 * g], recs[2 * g + 1]
        assert a["group_nodes"] == b["group_nodes"] == a["local"] + b["local"]        # the group's saw its two
        assert a["upper"] == b["upper"] and a["x"] == b["x"] and a["gnodes"] == b["gnodes"]
        pr = problems.random_miqp(n, m, p, seed=seed + g)
        model = bnb.MIOSQP(backend=oracle_mod)
        model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                    dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
        res = model.solve()
        assert abs(a["upper"] - res.upper_glob) <= 1e-3 * max(1.0, abs(res.upper_glob))
        np.testing.assert_array_equal(np.asarray(a["x"])[pr["i_idx"]], res.x[pr["i_idx"]])
    assert recs[0]["upper"] != recs[2]["upper"]  # (two different MIQPs)
