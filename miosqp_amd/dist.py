"""Leaf sharding across GPUs: one process per GPU, RCCL over xGMI for the incumbent only.

The reference is strictly sequential (one leaf per loop trip, /root/reference/miosqp/solver.py:
85-123), but its open leaves (`Workspace.leaves`, workspace.py:83) are independent relaxations
once rho is fixed, and the factor is read-only.  So the factor is replicated on every GPU, the
open leaves are dealt round-robin to the ranks, and every rank keeps exploring ITS leaves with
the unchanged host logic of miosqp_amd.bnb.  The only exchange is the incumbent: after each wave
an all-gather of one double per rank (the minimum gives `upper_glob`, the first rank holding it
is the owner) and a broadcast of the owner's x (n doubles).  Each rank then prunes its local
leaves against the global bound with the reference's own prune()/bound test semantics
(workspace.py:274-280, 299-300).  Nothing is exchanged per ADMM iteration.

Per-node results are identical to single-GPU mode (a node is a pure function of l,u,x0,y0); the
visiting order differs from the reference's one-at-a-time order, so node counts can differ
(SURVEY.md sec. 8e "parity caveat").
"""
import numpy as np


class LocalComm(object):
    """world_size 1: no collective."""
    rank, world = 0, 1

    def incumbent(self, value, x):
        return value, 0, x

    def sum(self, arr):
        return np.asarray(arr, dtype=np.float64)

    def barrier(self):
        pass


class TorchComm(object):
    """torch.distributed process group: backend "nccl" is RCCL on ROCm; "gloo" in CPU tests."""

    def __init__(self, device):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.device = torch, dist, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def incumbent(self, value, x):
        """(best value over ranks, owner rank, owner's x).  Ties go to the lowest rank."""
        t = self.torch
        mine = t.tensor([value], dtype=t.float64, device=self.device)
        allv = t.empty(self.world, dtype=t.float64, device=self.device)
        self.dist.all_gather_into_tensor(allv, mine)
        vals = allv.cpu().numpy()
        owner = int(np.argmin(vals))
        best = float(vals[owner])
        if not np.isfinite(best):
            return best, owner, x
        buf = t.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(self.device) \
            if self.rank == owner else t.empty(len(x), dtype=t.float64, device=self.device)
        self.dist.broadcast(buf, src=owner)
        return best, owner, buf.cpu().numpy()

    def sum(self, arr):
        t = self.torch
        buf = t.tensor(np.asarray(arr, dtype=np.float64), dtype=t.float64, device=self.device)
        self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM)
        return buf.cpu().numpy()

    def barrier(self):
        self.dist.barrier()


class ShardedSearch(object):
    """Drives one rank's share of the tree of an already set-up MIOSQP model."""

    def __init__(self, model, comm=None):
        self.model = model
        self.work = model.work
        self.comm = comm if comm is not None else LocalComm()
        self.nodes = 0
        self.iters = 0
        self.dealt = False

    # every rank runs this part identically (same data, deterministic relaxations)
    def expand_until(self, n_leaves, max_nodes=10 ** 9):
        """Explore one node at a time, on every rank alike, until `n_leaves` leaves are open."""
        w = self.work
        rule = w.settings['tree_explor_rule']
        done = 0
        while 0 < len(w.leaves) < n_leaves and done < max_nodes:
            self._visit(rule)
            done += 1
        return done

    def deal(self):
        """Round-robin partition of the open leaves; rank r keeps leaves r, r+W, r+2W, ..."""
        w = self.work
        w.leaves = [lf for k, lf in enumerate(w.leaves) if k % self.comm.world == self.comm.rank]
        self.dealt = True

    def _visit(self, rule):
        w = self.work
        leaf = w.choose_leaf(rule)
        leaf.solve()
        w.bound_and_branch(leaf)
        w.iter_num += 1
        self.nodes += 1
        self.iters += leaf.num_iter
        return leaf

    def step(self, nodes_per_rank=1):
        """One wave: up to `nodes_per_rank` local leaves, then the incumbent exchange."""
        w = self.work
        rule = w.settings['tree_explor_rule']
        for _ in range(nodes_per_rank):
            if not w.leaves:
                break
            self._visit(rule)
        self.sync_incumbent()

    def step_batched(self, width):
        """One wave as ONE batched relaxation call: take up to `width` local leaves in the order the
        exploration rule would visit them, solve them together, then bound/branch each in that order."""
        w = self.work
        rule = w.settings['tree_explor_rule']
        wave = []
        while w.leaves and len(wave) < width:
            wave.append(w.choose_leaf(rule))
        if wave:
            w.solve_wave(wave)
            for leaf in wave:
                w.bound_and_branch(leaf)
                w.iter_num += 1
                self.nodes += 1
                self.iters += leaf.num_iter
        self.sync_incumbent()

    def sync_incumbent(self):
        w = self.work
        if self.comm.world == 1:
            return
        best, owner, x = self.comm.incumbent(w.upper_glob, w.x)
        if best < w.upper_glob:
            w.upper_glob = best
            w.x = x
            w.prune()

    def open_leaves(self):
        return int(self.comm.sum([len(self.work.leaves)])[0])

    def run(self, nodes_per_rank=1, max_waves=10 ** 9):
        """Waves until no rank has leaves left (or max_waves)."""
        waves = 0
        while waves < max_waves and self.open_leaves() > 0:
            self.step(nodes_per_rank)
            waves += 1
        w = self.work
        w.get_return_status()
        w.get_return_solution()
        return waves
