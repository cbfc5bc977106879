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
import os

import numpy as np


class LocalComm(object):
    """world_size 1: no collective."""
    rank, world = 0, 1

    def incumbent(self, value, x):
        return value, 0, x

    extra = (0.0, 0.0)  # (nodes, ADMM iterations) summed over the ranks at the last completed exchange

    def exchange(self, value, x, nleaves, have=None, extra=(0.0, 0.0)):
        return self.complete(self.post(value, x, nleaves, extra), have)

    def post(self, value, x, nleaves, extra=(0.0, 0.0)):
        return (value, nleaves, extra)

    def complete(self, h, have=None):
        self.extra = tuple(h[2][:2])
        self.aux = [float(h[2][2])] if len(h[2]) > 2 else None
        return h[0], 0, None, h[1]

    aux = None  # per rank: the optional third value of `extra` at the last completed exchange (ShardedStream: leaves it could give)

    def leaf_counts(self):
        return None

    def gather(self, vec):
        return np.asarray(vec, dtype=np.float64).reshape(1, -1)

    def move(self, arr, size, src):
        raise RuntimeError("single rank")

    def sum(self, arr):
        return np.asarray(arr, dtype=np.float64)

    def barrier(self):
        pass


class TorchComm(object):
    """torch.distributed process group: backend "nccl" is RCCL on ROCm; "gloo" in CPU tests."""

    def __init__(self, device, group=None, ranks=None):
        """group / ranks: a sub-group of the job (torch.distributed.new_group) and the global ranks it consists of, in
        group order -- `rank` and `world` are then the GROUP's (bench.py: a few ranks per tree of the MIQP stream, several
        trees at a time); None: the whole job."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.device = torch, dist, device
        self.group = group
        self.ranks = list(ranks) if ranks is not None else list(range(dist.get_world_size()))
        self.rank, self.world = (self.ranks.index(dist.get_rank()), len(self.ranks)) if group is not None else \
            (dist.get_rank(), dist.get_world_size())
        self.extra = (0.0, 0.0)
        self.n_hint = 0  # length of x for ranks that hold no incumbent yet (set by ShardedSearch)
        # The scalar exchange runs every step: its buffers are made once.  On a GPU the four doubles travel through a
        # pinned staging tensor (non-blocking copies ordered on the current stream) instead of a fresh pageable tensor
        # per call, and the gathered table comes back into pinned memory the same way: one event wait per exchange
        # instead of two implicit synchronisations and three allocations.
        self._gpu = device.type == "cuda"
        self._mine_h = torch.empty(5, dtype=torch.float64, pin_memory=self._gpu)
        self._tab_h = torch.empty(5 * self.world, dtype=torch.float64, pin_memory=self._gpu)
        self.aux = None  # per rank: the optional third value of `extra` (ShardedStream: leaves the rank could give away)
        self._ring = []  # device buffers of exchanges in flight (post() may run ahead of complete() by one)

    def _bufs(self):
        t = self.torch
        if self._ring:
            return self._ring.pop()
        return (t.empty(5, dtype=t.float64, device=self.device), t.empty(5 * self.world, dtype=t.float64, device=self.device))

    def post(self, value, x, nleaves, extra=(0.0, 0.0)):
        """First half of exchange(): the all-gather of (incumbent value, open-leaf count, nodes and ADMM
        iterations of this step) is enqueued (async_op) and nothing is waited for; returns the handle for
        complete().  The incumbent x is snapshotted so that a later broadcast sends the point that belongs
        to `value`."""
        mine, allv = self._bufs()
        h = self._mine_h
        if self._gpu and getattr(self, "_h2d", None) is not None:
            self._h2d.synchronize()  # (the copy out of the staging tensor of the previous post: long done, unless posts run ahead)
        h[0], h[1], h[2], h[3] = float(value), float(nleaves), float(extra[0]), float(extra[1])
        h[4] = float(extra[2]) if len(extra) > 2 else 0.0
        mine.copy_(h, non_blocking=self._gpu)
        if self._gpu:
            self._h2d = self.torch.cuda.Event()
            self._h2d.record()
        work = self.dist.all_gather_into_tensor(allv, mine, group=self.group, async_op=True)
        return (work, allv, mine, None if x is None else np.array(x, dtype=np.float64, copy=True))

    def complete(self, h, have=None):
        """Second half: wait for the all-gather, then -- only when some rank holds a better incumbent
        than `have` -- a broadcast of the owner's x.  Returns (best value, owner rank, owner's x or
        None, total open leaves).  Ties go to the lowest rank, so every rank takes the same decision."""
        t = self.torch
        work, allv, mine, xsnap = h
        work.wait()
        if self._gpu:
            self._tab_h.copy_(allv, non_blocking=True)
            t.cuda.current_stream().synchronize()
            tab = self._tab_h.numpy().reshape(self.world, 5).copy()
        else:
            tab = allv.numpy().reshape(self.world, 5).copy()
        self._ring.append((mine, allv))
        self._counts = [int(round(c)) for c in tab[:, 1]]
        self.extra = (float(tab[:, 2].sum()), float(tab[:, 3].sum()))
        self.aux = [float(v) for v in tab[:, 4]]
        owner = int(np.argmin(tab[:, 0]))
        best = float(tab[owner, 0])
        total = int(round(tab[:, 1].sum()))
        # every rank holds the same previous global incumbent, so this test is rank-uniform
        prev = float(np.max(tab[:, 0])) if have is None else have
        if not np.isfinite(best) or not best < prev:
            return best, owner, None, total
        n = int(self.n_hint if xsnap is None else len(xsnap))
        buf = t.from_numpy(np.ascontiguousarray(xsnap, dtype=np.float64)).to(self.device) \
            if self.rank == owner else t.empty(n, dtype=t.float64, device=self.device)
        self.dist.broadcast(buf, src=self.ranks[owner], group=self.group)
        return best, owner, buf.cpu().numpy(), total

    def exchange(self, value, x, nleaves, have=None, extra=(0.0, 0.0)):
        """Blocking exchange: post() + complete()."""
        return self.complete(self.post(value, x, nleaves, extra), have)

    def gather(self, vec):
        """Blocking all-gather of a short float vector: array [world, len(vec)]."""
        t = self.torch
        mine = t.tensor(np.asarray(vec, dtype=np.float64), dtype=t.float64, device=self.device)
        allv = t.empty(self.world * mine.numel(), dtype=t.float64, device=self.device)
        self.dist.all_gather_into_tensor(allv, mine, group=self.group)
        return allv.cpu().numpy().reshape(self.world, -1)

    def incumbent(self, value, x):
        best, owner, xb, _ = self.exchange(value, x, 0)
        return best, owner, (x if xb is None else xb)

    def leaf_counts(self):
        """Open leaves per rank as of the last exchange()."""
        return list(self._counts)

    def move(self, arr, size, src):
        """One leaf record from rank `src` to everybody (a broadcast on the world communicator: every
        rank takes part, so no extra point-to-point communicator is ever created); `arr` is only
        read on `src`."""
        if self.rank == src:
            buf = self.torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)).to(self.device)
        else:
            buf = self.torch.empty(size, dtype=self.torch.float64, device=self.device)
        self.dist.broadcast(buf, src=self.ranks[src], group=self.group)
        return buf.cpu().numpy()

    def leaf_buffer(self, size):
        """A DEVICE tensor of `size` doubles for leaf records that travel without a host hop (r05), or None when this
        process has no GPU (the CPU tests over gloo).  One tensor per size, kept: the slot store copies into it device to
        device, the broadcast sends it (RCCL over xGMI with backend nccl; gloo stages a CUDA tensor itself), the receiving
        store copies out of it."""
        t = self.torch
        if not t.cuda.is_available() or os.environ.get("MIOSQP_LEAF_VIA_HOST") == "1":
            return None
        bufs = self.__dict__.setdefault("_leaf_bufs", {})
        if size not in bufs:
            try:
                bufs[size] = t.zeros(size, dtype=t.float64, device=t.device("cuda", t.cuda.current_device()))
            except RuntimeError:  # (torch's own runtime cannot open the device in this process: the records go via the host)
                bufs[size] = None
        return bufs[size]

    def move_tensor(self, buf, src):
        """the broadcast of move() on a tensor that stays where it is.  The host returns only when the collective has
        finished with `buf`: RCCL runs on a stream of its own, torch's current stream merely waits for it, and the next writer
        of the buffer is the donor's slot store on the ENGINE's stream (give_leaf(into=...)), which is ordered against
        neither -- without the wait a second leaf of the same exchange could overwrite a record still being sent."""
        self.dist.broadcast(buf, src=self.ranks[src], group=self.group)
        if getattr(buf, "is_cuda", False):
            self.torch.cuda.current_stream(buf.device).synchronize()
        return buf

    def sum(self, arr):
        t = self.torch
        buf = t.tensor(np.asarray(arr, dtype=np.float64), dtype=t.float64, device=self.device)
        self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, group=self.group)
        return buf.cpu().numpy()

    def barrier(self):
        self.dist.barrier(group=self.group)


class ShardedSearch(object):
    """Drives one rank's share of the tree of an already set-up MIOSQP model.

    Every instance starts REPLICATED: all ranks hold the same single root and, the relaxations
    being deterministic, visit the same nodes until at least `world` leaves are open; then the
    leaves are dealt and each rank continues on its own.  Work done while replicated is counted
    once (on rank 0) in `nodes` / `iters`."""

    def __init__(self, model, comm=None):
        self.model = model
        self.work = model.work
        self.comm = comm if comm is not None else LocalComm()
        self.nodes = 0
        self.iters = 0
        self.replicated = True
        self.global_upper = np.inf
        import os
        self.rebalance = os.environ.get("MIOSQP_REBALANCE", "1") != "0"
        # MIOSQP_EXCHANGE_LAG=1: the exchange of step k is completed at step k + 1, so the all-gather travels
        # while the next node is being solved and ranks stop waiting for each other at every node (simulated
        # gain 0.05-0.07 of weak-scaling efficiency plus the hidden exchange latency).  Default 0 (blocking
        # exchange after every step): with lag 1 an RCCL kernel shares the GPU with the cooperative launch of
        # the next node, which could only be exercised over gloo here.
        self.lag = int(os.environ.get("MIOSQP_EXCHANGE_LAG", "0")) if self.comm.world > 1 else 0
        self._pending = None
        import time
        self.clock = time.perf_counter  # what `budget` is measured with (the simulator substitutes its model clock)
        self._inflight = None  # (future, leaves) of the wave being solved on the worker thread
        self._pool = None
        if hasattr(self.comm, "n_hint"):
            self.comm.n_hint = self.work.data.n
        self.moved = 0
        self.resyncs = 0          # times the replicated phase had to adopt rank 0's leaves (see _agree)
        self.global_nodes = 0     # nodes visited by all ranks together (replicated work counted once)
        self.global_iters = 0
        self.global_open = 1      # open leaves over all ranks as of the last exchange
        self._step_nodes = 0      # this rank's nodes / iterations since its last post()
        self._step_iters = 0
        # MIOSQP_FORCE_EXCHANGE=1: run the collectives even with one rank (a 1-GPU torchrun launch then
        # exercises the RCCL path end to end; results are unchanged)
        self.force_exchange = os.environ.get("MIOSQP_FORCE_EXCHANGE") == "1"
        self.feed = int(os.environ.get("MIOSQP_FEED", "1"))  # leaves handed to a dry rank per exchange
        # which leaf a donor hands over: its oldest (shallowest: the largest subtree, keeps the receiver busy
        # longest; simulated +0.02 efficiency at 8 ranks) or its newest (MIOSQP_DONATE=last)
        self.donate_first = os.environ.get("MIOSQP_DONATE", "first") == "first"

    def begin_instance(self):
        """Call after MIOSQP.update_vectors (new root on every rank).  An exchange still in flight belongs
        to the closed tree: it is completed (every rank takes part) and ignored."""
        self.flush_wave()
        self.drain(apply=False)
        self.replicated = True
        self.global_upper = np.inf
        self.global_nodes = self.global_iters = 0
        self.global_open = 1
        self._step_nodes = self._step_iters = 0

    def _count(self, leaf):
        if not self.replicated or self.comm.rank == 0:
            self.nodes += 1
            self.iters += leaf.num_iter
        if self.replicated or self.comm.world == 1:
            self.global_nodes += 1  # the same node on every rank: counted once, no exchange needed
            self.global_iters += leaf.num_iter
        else:
            self._step_nodes += 1   # summed over the ranks by the next exchange
            self._step_iters += leaf.num_iter

    # -- replicated phase: the ranks must hold the SAME leaves before they deal them ---------------
    @staticmethod
    def _fingerprint(leaves):
        """Order-sensitive checksum of the open leaves' bounds (what identifies a leaf), < 2^48 so that it
        survives the float64 all-gather exactly."""
        import hashlib
        h = hashlib.blake2b(digest_size=6)
        for lf in leaves:
            h.update(np.ascontiguousarray(lf.l, dtype=np.float64).tobytes())
            h.update(np.ascontiguousarray(lf.u, dtype=np.float64).tobytes())
            h.update(np.int64(lf.depth).tobytes())
        return float(int.from_bytes(h.digest(), "little"))

    def _leaf_record(self, lf):
        return np.concatenate([lf.l, lf.u, lf.x, lf.y, [float(lf.depth), float(lf.lower), 1.0]])

    def _leaf_from_record(self, msg):
        from miosqp_amd.bnb import Node
        w = self.work
        n, M = w.data.n, w.data.m + w.data.n_int
        return Node(w.data, msg[:M].copy(), msg[M:2 * M].copy(), w.solver, depth=int(msg[-3]), lower=float(msg[-2]),
                    x0=msg[2 * M:2 * M + n].copy(), y0=msg[2 * M + n:3 * M + n].copy(), constant=w.constant)

    def _agree(self):
        """While replicated every rank visits the same nodes and is ASSUMED to get the same results; nothing
        guarantees it (a rank whose cooperative launch was called off continues in the two-kernel form, whose
        summation order differs: one more or one fewer test, another branching variable).  So after every
        replicated node the ranks compare (open leaves, checksum of their bounds, incumbent) in one
        all-gather; on any difference everybody adopts rank 0's leaves and incumbent (broadcast records).
        Afterwards the leaf lists are identical, so `deal()` partitions one list and every rank takes the
        same decision about dealing and about the tree being closed."""
        w, comm = self.work, self.comm
        if comm.world == 1 and not self.force_exchange:
            return
        ug = w.upper_glob if np.isfinite(w.upper_glob) else 1e300
        tab = comm.gather([float(len(w.leaves)), self._fingerprint(w.leaves), ug])
        if np.all(tab == tab[0]):
            return
        self.resyncs += 1
        n, M = w.data.n, w.data.m + w.data.n_int
        size = 3 * M + n + 3
        count = int(tab[0][0])
        mine = w.leaves
        adopted = []
        for k in range(count):
            msg = comm.move(self._leaf_record(mine[k]) if comm.rank == 0 else None, size, 0)
            adopted.append(mine[k] if comm.rank == 0 else self._leaf_from_record(msg))
        w.leaves = adopted
        inc = comm.move(np.concatenate([[w.upper_glob], w.x]) if comm.rank == 0 else None, n + 1, 0)
        if comm.rank != 0:
            w.upper_glob = float(inc[0])
            w.x = inc[1:].copy()

    def deal(self):
        """Round-robin partition of the open leaves; rank r keeps leaves r, r+W, r+2W, ..."""
        w = self.work
        w.leaves = [lf for k, lf in enumerate(w.leaves) if k % self.comm.world == self.comm.rank]
        self.replicated = False
        self.global_upper = w.upper_glob

    def _visit(self, rule):
        w = self.work
        leaf = w.choose_leaf(rule)
        leaf.solve()
        w.bound_and_branch(leaf)
        w.iter_num += 1
        self._count(leaf)
        return leaf

    def _select_wave(self, rule, width):
        """Takes (removes) up to `width` local leaves in the order the exploration rule would visit them."""
        w = self.work
        wave = []
        if w.leaves:
            # the `width` leaves that repeated choose_leaf() calls would pop, in that order: argmax
            # with first-index ties == stable sort by descending key (workspace.py:128-155)
            if rule == 0 or (rule == 1 and np.isinf(w.upper_glob)):
                keys = np.array([lf.depth for lf in w.leaves], dtype=float)
            elif rule == 1:
                keys = np.array([lf.lower for lf in w.leaves], dtype=float)
            else:
                raise ValueError('Tree exploring strategy not recognized')
            order = np.argsort(-keys, kind='stable')[:width]
            wave = [w.leaves[i] for i in order]
            taken = set(int(i) for i in order)
            w.leaves = [lf for i, lf in enumerate(w.leaves) if i not in taken]
        return wave

    def _process_wave(self, wave):
        """Bound and branch the solved leaves of a wave, in the wave's order."""
        w = self.work
        w.defer_lower = True  # one pass over the leaves per wave instead of one per node
        for leaf in wave:
            w.bound_and_branch(leaf)
            w.iter_num += 1
            self._count(leaf)
        w.defer_lower = False
        if w.leaves:
            w.lower_glob = min(lf.lower for lf in w.leaves)

    def _visit_wave(self, rule, width, pipelined=False):
        """Up to `width` local leaves in ONE batched relaxation call, taken in the order the
        exploration rule would visit them, then bound/branch each in that order.
        `pipelined`: the call runs on a worker thread (the engine releases the GIL) while this thread bounds
        and branches the PREVIOUS wave; a wave is then chosen before the previous one's children exist and
        before its incumbents can prune (the search stays exact, it may visit more nodes)."""
        w = self.work
        wave = self._select_wave(rule, width)
        if not pipelined:
            self.flush_wave()
            if wave:
                w.solve_wave(wave)
                self._process_wave(wave)
            return len(wave)
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=1)
        prev, self._inflight = self._inflight, ((self._pool.submit(w.solve_wave, wave), wave) if wave else None)
        if prev is not None:
            prev[0].result()
            self._process_wave(prev[1])
        return len(wave)

    def flush_wave(self):
        """Waits for the wave in flight (pipelined mode) and bounds/branches it."""
        if self._inflight is not None:
            (fut, wave), self._inflight = self._inflight, None
            fut.result()
            self._process_wave(wave)

    def _open(self):
        """Open leaves of this rank, the ones being solved right now included."""
        return len(self.work.leaves) + (len(self._inflight[1]) if self._inflight is not None else 0)

    def expand_until(self, n_leaves, max_nodes=10 ** 9):
        """Node-at-a-time exploration (identical on every rank while replicated) until
        `n_leaves` leaves are open or the tree closes."""
        w = self.work
        rule = w.settings['tree_explor_rule']
        done = 0
        while 0 < len(w.leaves) < n_leaves and done < max_nodes:
            self._visit(rule)
            done += 1
        return done

    def step(self, nodes_per_rank=1, batched=False, pipelined=False, budget=None):
        """One wave.  Returns the number of leaves open over all ranks afterwards.
        `budget` (seconds, node-at-a-time mode): instead of a fixed count, keep visiting local leaves until
        that much time has passed (at least one node) -- ranks then reach the exchange at about the same
        moment whatever their nodes cost, which removes most of the lock-step loss."""
        w = self.work
        rule = w.settings['tree_explor_rule']
        if not (batched and pipelined):
            self.flush_wave()  # the engine is used from this thread below
        if self.replicated:
            if w.leaves and len(w.leaves) < self.comm.world:
                self._visit(rule)  # same node on every rank ...
                self._agree()      # ... checked: identical leaf lists from here on
                self.global_open = len(w.leaves)
                return len(w.leaves)
            if w.leaves:
                self.deal()
        if batched:
            self._visit_wave(rule, nodes_per_rank, pipelined)
        elif budget:
            t0 = self.clock()
            while w.leaves:
                self._visit(rule)
                if self.clock() - t0 >= budget:
                    break
        else:
            for _ in range(nodes_per_rank):
                if not w.leaves:
                    break
                self._visit(rule)
        return self.sync_incumbent()

    def step_batched(self, width):
        return self.step(width, batched=True)

    def sync_incumbent(self):
        """Incumbent exchange + global open-leaf count (one all-gather, plus one broadcast only when
        the incumbent improved somewhere).  With lag 1 this posts the exchange of this step and applies
        the one of the previous step; the returned total is then one step old (and 1 when nothing is
        known yet)."""
        w = self.work
        if self.comm.world == 1 and not self.force_exchange:
            self.global_open = self._open()
            return self.global_open
        h = self.comm.post(w.upper_glob, w.x, self._open(), (self._step_nodes, self._step_iters))
        self._step_nodes = self._step_iters = 0
        if self.lag == 0:
            return self._apply(h)
        prev, self._pending = self._pending, h
        if prev is None:
            return 1
        return self._apply(prev)

    def _apply(self, h):
        w = self.work
        best, owner, x, total = self.comm.complete(h, self.global_upper)
        if self.comm.world > 1:
            self.global_nodes += int(round(self.comm.extra[0]))
            self.global_iters += int(round(self.comm.extra[1]))
        self.global_open = total
        if x is not None:
            self.global_upper = best
            if best < w.upper_glob:
                w.upper_glob = best
                w.x = x
                w.prune()
        self._rebalance()
        return total

    def drain(self, apply=True):
        """Completes the exchange still in flight (same point of the sequence on every rank)."""
        if self._pending is not None:
            h, self._pending = self._pending, None
            if apply:
                self._apply(h)
            else:
                self.comm.complete(h, self.global_upper)

    def _rebalance(self):
        """A rank that ran dry receives one leaf (3M+n+3 doubles; MIOSQP_FEED of them) from the rank holding most.  Every
        rank derives the same transfer plan from the gathered leaf counts (taken before this wave's
        pruning, which is why a donor re-checks that it still has two leaves) and takes part in the
        broadcast that carries the record; only the receiver keeps it."""
        counts = self.comm.leaf_counts()
        if not counts or not self.rebalance:
            return
        w, me = self.work, self.comm.rank
        plan = []
        for r in range(len(counts)):
            if counts[r] == 0:
                for _ in range(self.feed):
                    donor = int(np.argmax(counts))
                    if counts[donor] < 2 or counts[donor] - counts[r] < 2:
                        break
                    plan.append((donor, r))
                    counts[donor] -= 1
                    counts[r] += 1
        n, M = w.data.n, w.data.m + w.data.n_int
        size = 3 * M + n + 3
        for donor, recv in plan:
            msg = None
            if me == donor:
                if len(w.leaves) >= 2:
                    lf = w.leaves.pop(0) if self.donate_first else w.leaves.pop()
                    msg = self._leaf_record(lf)
                    self.moved += 1
                else:  # pruned in the meantime: an empty token keeps the collective matched
                    msg = np.zeros(size)
            msg = self.comm.move(msg, size, donor)
            if me == recv and msg[-1] == 1.0:
                w.leaves.append(self._leaf_from_record(msg))

    def open_leaves(self):
        return int(self.comm.sum([len(self.work.leaves)])[0])

    def run(self, nodes_per_rank=1, max_waves=10 ** 9, batched=False, pipelined=False):
        """Waves until no rank has leaves left (or max_waves)."""
        waves = 0
        total = 1
        cap = self.work.settings['max_iter_bb']
        # the reference visits at most max_iter_bb - 1 nodes (workspace.py:113-126: iter_num starts at 1); here
        # the budget is global -- nodes of all ranks together, known to every rank from the exchange -- so
        # every rank stops after the same step (ranks overshoot by at most one step's nodes)
        while waves < max_waves and total > 0 and self.global_nodes + 1 < cap:
            total = self.step(nodes_per_rank, batched, pipelined)
            waves += 1
        self.flush_wave()
        self.drain()
        w = self.work
        # rank-uniform status: "finished" is a property of the whole tree (no open leaf on any rank), not of
        # this rank's own node count
        w.osqp_iter_avg = self.global_iters / float(max(1, self.global_nodes + 1))
        w.get_return_status(finished=(self.global_open == 0 and self._open() == 0))
        w.get_return_solution()
        return waves


class ShardedStream(object):
    """BASELINE configs[2] across GPUs: every rank keeps its columns busy from its OWN device-resident leaf pool
    (miosqp_amd/stream.py) and the ranks only meet every `exchange_every` chunks for the incumbent -- one all-gather
    of (value, leaves alive, nodes, iterations) plus a broadcast of x when some rank improved it -- and, in the same
    round, to hand leaves to ranks that ran dry (one all-gather of (alive, givable) and one broadcast per moved leaf:
    integer-row bounds + warm start, 2 p + n + M doubles).  Every instance starts replicated like ShardedSearch:
    all ranks expand the same first nodes (checked by `_agree`) until there are a few leaves per rank, then deal.
    Exploration order differs from the sequential search by construction; per-node results do not."""

    def __init__(self, model, comm=None, columns=256, exchange_every=4, capacity=None, ramp_leaves=4, feed=64,
                 deal_to=None, search=None, step_kwargs=None):
        """search: the per-rank search object when it is not a StreamSearch on the leaf pool -- anything with its
        interface, e.g. search.HostedSearch (node-at-a-time relaxations, loop in the C++ library); step_kwargs: what
        its step() is called with (HostedSearch: nodes=, budget=)."""
        from miosqp_amd import stream
        self.model, self.work = model, model.work
        self.comm = comm if comm is not None else LocalComm()
        self.seq = ShardedSearch(model, self.comm)  # replicated ramp-up (its _visit / _agree / counters)
        self.ss = search if search is not None else stream.StreamSearch(model, columns=columns, capacity=capacity)
        self.step_kwargs = dict(step_kwargs or {})
        # MIOSQP_FORCE_EXCHANGE=1: the exchange runs although there is one rank (first contact with a new
        # communicator: `torch.distributed.run --nproc-per-node 1` exercises every collective used below)
        self.force_exchange = os.environ.get("MIOSQP_FORCE_EXCHANGE") == "1"
        self.exchange_every, self.ramp_leaves, self.feed = int(exchange_every), int(ramp_leaves), int(feed)
        self.deal_to = deal_to  # None: leaves dealt round-robin; a rank: all to that one (worst case, for tests)
        self.global_upper = np.inf
        self.global_nodes = self.global_iters = 0
        self.moved = 0
        self.moved_dev = 0  # leaves received as device tensors (no host hop)
        # the per-rank search keeps its leaves on a GPU and takes / hands them out by device pointer
        self._dev_leaves = hasattr(getattr(self.ss, "eng", None), "search_create")
        self.steps = 0
        self._n0 = self._i0 = 0
        self.begin_instance()

    def begin_instance(self):
        """Replicated ramp-up on the host (node at a time, identical on every rank), then the deal."""
        w, comm, seq = self.work, self.comm, self.seq
        if not w.leaves:
            w.leaves = [w._make_root()]
        seq.begin_instance()
        rule = w.settings['tree_explor_rule']
        target = self.ramp_leaves * comm.world
        while 0 < len(w.leaves) < target:
            seq._visit(rule)
            seq._agree()
        self.global_nodes, self.global_iters = seq.global_nodes, seq.global_iters
        if self.deal_to is None:
            mine = [lf for k, lf in enumerate(w.leaves) if k % comm.world == comm.rank]
        else:
            mine = list(w.leaves) if comm.rank == self.deal_to else []
        self.total_alive = len(w.leaves)
        self.global_upper = w.upper_glob
        p = w.data.n_int
        self.ss.begin_instance(seed_root=False)
        for lf in mine:
            self.ss.add_leaf(lf.l[-p:], lf.u[-p:], lf.x, lf.y, lf.depth, lf.lower)
        self._n0, self._i0 = self.ss.nodes, self.ss.iters
        self.steps = 0
        w.leaves = []
        return self.total_alive

    def step(self):
        """One chunk on this rank's stream; every `exchange_every`-th call ends with the exchange.  Returns the number
        of leaves alive over all ranks as of the last exchange (0: the tree is closed everywhere)."""
        alive = self.ss.step(**self.step_kwargs)
        self.steps += 1
        if self.comm.world == 1 and not self.force_exchange:
            self.total_alive = alive
            self.global_nodes += self.ss.nodes - self._n0
            self.global_iters += self.ss.iters - self._i0
            self._n0, self._i0 = self.ss.nodes, self.ss.iters
            return alive
        if self.steps % self.exchange_every == 0:
            self.total_alive = self._exchange(alive)
        return self.total_alive

    def _exchange(self, alive):
        w, comm, ss = self.work, self.comm, self.ss
        # (the third value rides along for the feeding plan below: one collective per exchange instead of two)
        extra = (ss.nodes - self._n0, ss.iters - self._i0, float(ss.givable()))
        self._n0, self._i0 = ss.nodes, ss.iters
        best, owner, x, total = comm.exchange(w.upper_glob, w.x, float(len(ss.open) + ss.in_flight), self.global_upper, extra)
        self.global_nodes += int(round(comm.extra[0]))
        self.global_iters += int(round(comm.extra[1]))
        if x is not None:
            self.global_upper = best
            ss.adopt_incumbent(best, x)
        # leaves for the ranks that ran dry: every rank derives the same plan from the gathered counts
        # (from the exchange's own table -- counts as of just before this exchange's incumbent was adopted: a leaf the plan
        #  counts on may have been pruned since, the donor then sends an empty token; a communicator without the third value
        #  gathers the counts separately)
        counts, aux = comm.leaf_counts(), getattr(comm, "aux", None)
        if counts is not None and aux is not None and len(aux) == comm.world:
            alive_r = [int(v) for v in counts]
            giv = [int(round(v)) for v in aux]
        else:
            tab = comm.gather([float(len(ss.open) + ss.in_flight), float(ss.givable())])
            alive_r = [int(round(v)) for v in tab[:, 0]]
            giv = [int(round(v)) for v in tab[:, 1]]
        n, M, p = w.data.n, w.data.m + w.data.n_int, w.data.n_int
        size = 2 * p + n + M + 3
        # a leaf record = [l_int | u_int | x0 | y0 | depth, lower, valid].  With a GPU under both ends it never visits the
        # host: the donor's slot store copies the vectors into a device tensor, the tensor is broadcast, the receiver's
        # store copies them out (three scalars ride in its tail); otherwise (CPU tests, thread communicators) as numpy
        dev = comm.leaf_buffer(size) if (hasattr(comm, "leaf_buffer") and self._dev_leaves) else None
        views = None
        if dev is not None:
            from miosqp_amd import qp as _qp
            views = _qp.leaf_record_views(dev, p, n, M)
        for r in range(comm.world):
            if alive_r[r] > 0:
                continue
            donor = int(np.argmax(giv))
            count = min(self.feed, giv[donor] // 2)
            for _ in range(count):
                if dev is not None:
                    if comm.rank == donor:
                        tail = [0.0, 0.0, 0.0]
                        if ss.givable() > 0:
                            rec = ss.give_leaf(into=views)
                            tail = [float(rec[4]), float(rec[5]), 1.0]
                            self.moved += 1
                        dev[-3:] = comm.torch.tensor(tail, dtype=comm.torch.float64)  # (ordered before the broadcast: same stream)
                    comm.move_tensor(dev, donor)
                    if comm.rank == r:
                        tail = dev[-3:].cpu().numpy()  # (waits for the broadcast)
                        if tail[2] == 1.0:
                            ss.add_leaf(views[0], views[1], views[2], views[3], int(tail[0]), float(tail[1]))
                            self.moved_dev += 1
                    continue
                msg = None
                if comm.rank == donor:
                    if ss.givable() > 0:
                        l_int, u_int, x0, y0, depth, lower = ss.give_leaf()
                        msg = np.concatenate([l_int, u_int, x0, y0, [float(depth), float(lower), 1.0]])
                        self.moved += 1
                    else:  # pruned in the meantime: an empty token keeps the collective matched
                        msg = np.zeros(size)
                msg = comm.move(msg, size, donor)
                if comm.rank == r and msg[-1] == 1.0:
                    ss.add_leaf(msg[:p], msg[p:2 * p], msg[2 * p:2 * p + n], msg[2 * p + n:2 * p + n + M],
                                int(msg[-3]), float(msg[-2]))
            giv[donor] -= count
            alive_r[r] += count  # (the moved leaves were counted on the donor: `total` is unchanged)
        return total

    def run(self, max_steps=10 ** 9):
        """Until the tree is closed on every rank (or the global node budget is spent)."""
        w = self.work
        cap = w.settings['max_iter_bb']
        steps = 0
        while self.total_alive > 0 and steps < max_steps and self.global_nodes + 1 < cap:
            self.step()
            steps += 1
        # every rank leaves the loop after the same exchange
        w.osqp_iter_avg = self.global_iters / float(max(1, self.global_nodes + 1))
        w.get_return_status(finished=self.total_alive == 0)
        w.get_return_solution()
        return steps
