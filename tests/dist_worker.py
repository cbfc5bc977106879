"""Worker for tests/test_dist_gloo.py: one rank of a world_size-2 sharded tree search on CPU
(gloo), with the CPU oracle standing in for the GPU engine (test infrastructure)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as td

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from miosqp_amd import bnb, dist, problems  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    out_path, n, m, p, seed, per_rank = sys.argv[1], *map(int, sys.argv[2:7])
    td.init_process_group(backend="gloo")
    comm = dist.TorchComm(torch.device("cpu"))
    pr = problems.random_miqp(n, m, p, seed=seed)
    model = bnb.MIOSQP(backend=oracle)
    st = dict(problems.BNB_SETTINGS)
    model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], st,
                dict(problems.QP_SETTINGS))
    if os.environ.get("MIOSQP_WORKER_MODE") == "groups":
        # bench.py's --ranks-per-tree: the job split into groups of `per_rank` ranks, each group a communicator of its own
        # (dist.TorchComm(group=, ranks=)) closing ITS OWN MIQP with dist.ShardedStream over the hosted search (CPU emulation)
        import digest_backend
        from miosqp_amd import search
        rpt, world, rank = per_rank, td.get_world_size(), td.get_rank()
        mine = None
        for g in range(world // rpt):
            members = list(range(g * rpt, (g + 1) * rpt))
            grp = td.new_group(ranks=members, backend="gloo")
            if rank in members:
                mine = (g, dist.TorchComm(torch.device("cpu"), group=grp, ranks=members))
        g, gcomm = mine
        pr = problems.random_miqp(n, m, p, seed=seed + g)  # (a different MIQP per group)
        model = bnb.MIOSQP(backend=digest_backend)
        model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                    dict(st, max_iter_bb=10 ** 6), dict(problems.QP_SETTINGS))
        hs = search.HostedSearch(model, capacity=512)
        s = dist.ShardedStream(model, gcomm, search=hs, step_kwargs=dict(nodes=2), exchange_every=1, ramp_leaves=1, feed=4)
        s.run()
        w = model.work
        tot = comm.sum([hs.nodes, s.moved])        # the whole job's communicator still works beside the groups'
        gtot = gcomm.sum([hs.nodes])
        rec = dict(rank=rank, group=g, group_rank=gcomm.rank, group_world=gcomm.world, upper=w.upper_glob, x=list(map(float, w.x)),
                   status=w.status, nodes_total=float(tot[0]), group_nodes=float(gtot[0]), gnodes=s.global_nodes, local=hs.nodes,
                   alive=s.total_alive)
        with open("%s.%d" % (out_path, rank), "w") as f:
            json.dump(rec, f)
        td.barrier()
        td.destroy_process_group()
        return
    if os.environ.get("MIOSQP_WORKER_MODE") == "stream":
        # dist.ShardedStream with the CPU emulation of the leaf-pool calls (tests/digest_backend.py)
        import digest_backend
        model = bnb.MIOSQP(backend=digest_backend)
        model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                    dict(st, max_iter_bb=10 ** 6), dict(problems.QP_SETTINGS))
        s = dist.ShardedStream(model, comm, columns=4, exchange_every=per_rank, capacity=512, ramp_leaves=2, feed=4,
                               deal_to=0)
        s.run()
        w = model.work
        tot = comm.sum([s.ss.nodes, s.moved])
        rec = dict(rank=comm.rank, upper=w.upper_glob, x=list(map(float, w.x)), status=w.status,
                   nodes_total=float(tot[0]), moved_total=float(tot[1]), gnodes=s.global_nodes, local=s.ss.nodes,
                   free=len(s.ss.free))
        with open("%s.%d" % (out_path, comm.rank), "w") as f:
            json.dump(rec, f)
        td.barrier()
        td.destroy_process_group()
        return
    s = dist.ShardedSearch(model, comm)
    s.expand_until(2 * comm.world)
    before = len(model.work.leaves)
    s.deal()
    mine = len(model.work.leaves)
    waves = s.run(nodes_per_rank=per_rank)
    tot = comm.sum([s.nodes, s.iters, mine, s.moved])
    w = model.work
    rec = dict(rank=comm.rank, upper=w.upper_glob, x=list(map(float, w.x)), status=w.status, waves=waves,
               nodes_total=float(tot[0]), iters_total=float(tot[1]), dealt_total=float(tot[2]), moved_total=float(tot[3]),
               leaves_before_deal=before)
    with open("%s.%d" % (out_path, comm.rank), "w") as f:
        json.dump(rec, f)
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
