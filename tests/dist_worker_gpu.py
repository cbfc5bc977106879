"""Worker for tests/test_gpu_parity.py::test_two_ranks_hosted_search_on_one_device: one rank of a world_size-2 sharded
HOSTED search (node-at-a-time relaxations on the HIP engine, loop in the C++ host library, dist.ShardedStream over
search.HostedSearch -- the bench's headline form with more than one rank).  Both processes time-share GPU 0 and meet
over gloo (a one-GPU box has no second device for RCCL); the engine runs its multi-kernel form (the single-launch
solvers need the device to themselves)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as td

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from miosqp_amd import bnb, dist, problems, search  # noqa: E402


def main():
    out_path, n, m, p, seed = sys.argv[1], *map(int, sys.argv[2:6])
    td.init_process_group(backend="gloo")
    comm = dist.TorchComm(torch.device("cpu"))
    pr = problems.random_miqp(n, m, p, seed=seed)
    st = dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 6, device_tree=False)
    qs = dict(problems.QP_SETTINGS, device=0, coop=0, pers=0, resident=0)
    model = bnb.MIOSQP()
    model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st), dict(qs))
    hs = search.HostedSearch(model)
    sh = dist.ShardedStream(model, comm, search=hs, exchange_every=1, ramp_leaves=1, step_kwargs=dict(nodes=2))
    sh.run()
    w = model.work
    tot = comm.sum([hs.nodes, sh.moved, sh.moved_dev])
    rec = dict(rank=comm.rank, upper=w.upper_glob, x=list(map(float, w.x)), status=w.status, local_nodes=hs.nodes,
               nodes_total=float(tot[0]), moved_total=float(tot[1]), moved_dev_total=float(tot[2]), gnodes=sh.global_nodes)
    if comm.rank == 0:  # the sequential answer on the same engine form
        ref = bnb.MIOSQP()
        ref.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st), dict(qs))
        r = ref.solve()
        rec.update(seq_upper=r.upper_glob, seq_x=list(map(float, r.x)), seq_nodes=ref.work.iter_num - 1, seq_status=r.status)
    with open("%s.%d" % (out_path, comm.rank), "w") as f:
        json.dump(rec, f)
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
