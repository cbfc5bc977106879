"""Tree search on the device-resident leaf pool (miosqp_qp_pool_*): the host keeps the search logic, the
device keeps the leaves.

What stays on the host is what /root/reference/miosqp/workspace.py does with SCALARS: which open leaf next
(choose_leaf, 128-155), bound test and pruning (274-280, 299-300), incumbent updates (304-327), the return
status (352-373).  What moved to the device is everything that touches vectors: the leaf's bounds and warm
start (Workspace.leaves, 83), the relaxation (Node.solve), the clamp and objective (node.py:131-143), the
integrality test / branching variable / rounding heuristic (workspace.py:205-272) and the creation of the two
children (add_left / add_right, 157-203).  Per decided node 64 bytes come back (a digest); an incumbent's x is
fetched when one is found.

The batch is a STREAM: columns are refilled from a ready ring between chunks of `check_termination`
iterations, so a node never waits for the slowest node of a wave.  Nodes are pushed in the order the
exploration rule would pick them at push time; as in wave mode the visiting order differs from the
reference's one-node-at-a-time order, per-node results do not (tests/test_gpu_parity.py).
"""
import collections
import heapq

import numpy as np

from miosqp_amd import bnb

PRUNED = -100


class OpenLeaves(object):
    """The open leaves that have not been pushed to the device yet, kept so that "the next `k` leaves in the exploration
    rule's order" costs O(k) instead of a sort of all of them at every chunk (thousands at config 2: the sort was the
    largest item of the host's time per chunk).  Same order as a stable sort of the creation-ordered list by the key:
    depth first = the deepest leaves, older first (buckets by depth, each in creation order); best first in the
    reference's sense (workspace.py:142-149: the LARGEST inherited bound) = a heap on (-lower, creation number).
    len(), iteration (creation order) and append() as a list."""

    def __init__(self, depth, lower):
        self._depth, self._lower = depth, lower  # per-slot lists owned by the search
        self.clear()

    def clear(self):
        self._seq = {}       # slot -> creation number, for the leaves present
        self._count = 0
        self._buckets = {}   # depth -> deque of slots (creation order); depth mode
        self._maxd = -1
        self._heap = None    # [(-lower, creation number, slot)]; lower mode (entries of absent slots are skipped)

    def __len__(self):
        return len(self._seq)

    def __iter__(self):
        return iter(sorted(self._seq, key=self._seq.get))

    def __bool__(self):
        return bool(self._seq)

    def append(self, s):
        self._seq[s] = self._count
        if self._heap is not None:
            heapq.heappush(self._heap, (-self._lower[s], self._count, s))
        else:
            d = self._depth[s]
            b = self._buckets.get(d)
            if b is None:
                b = self._buckets[d] = collections.deque()
            b.append(s)
            if d > self._maxd:
                self._maxd = d
        self._count += 1

    def _to_heap(self):
        self._heap = [(-self._lower[s], q, s) for s, q in self._seq.items()]
        heapq.heapify(self._heap)
        self._buckets = {}

    def take(self, k, by_lower):
        """Removes and returns the next (at most) k leaves: by_lower False -> deepest first, True -> largest bound first."""
        out = []
        if k <= 0 or not self._seq:
            return out
        if by_lower:
            if self._heap is None:
                self._to_heap()
            h = self._heap
            while h and len(out) < k:
                _, q, s = heapq.heappop(h)
                if self._seq.get(s) == q:
                    del self._seq[s]
                    out.append(s)
            return out
        if self._heap is not None:  # (a search that went back to depth first: rebuild the buckets)
            self._heap = None
            self._buckets, self._maxd = {}, -1
            for s in sorted(self._seq, key=self._seq.get):
                self._buckets.setdefault(self._depth[s], collections.deque()).append(s)
                self._maxd = max(self._maxd, self._depth[s])
        d = self._maxd
        while d >= 0 and len(out) < k:
            b = self._buckets.get(d)
            if b:
                while b and len(out) < k:
                    s = b.popleft()
                    del self._seq[s]
                    out.append(s)
            if not b:
                self._buckets.pop(d, None)
                if d == self._maxd:
                    self._maxd = d - 1
            d -= 1
        return out

    def prune(self, upper):
        """Removes and returns the leaves whose inherited bound exceeds `upper`."""
        if self._heap is not None:
            out, h = [], self._heap
            while h and -h[0][0] > upper:
                _, q, s = heapq.heappop(h)
                if self._seq.get(s) == q:
                    del self._seq[s]
                    out.append(s)
            return out
        out = [s for s in self._seq if self._lower[s] > upper]
        for s in out:
            del self._seq[s]
            self._buckets[self._depth[s]].remove(s)
        return out

    def pop_shallowest(self):
        """The shallowest leaf, the oldest of them (what is handed to another rank)."""
        s = min(self._seq, key=lambda t: (self._depth[t], self._seq[t]))
        del self._seq[s]
        if self._heap is None:
            self._buckets[self._depth[s]].remove(s)
        return s


class StreamSearch(object):
    def __init__(self, model, columns=256, capacity=None, ring_margin=None, observer=None):
        self.model = model
        self.observer = observer  # observer(search, digest): called before a digest is absorbed (tests)
        self.work = model.work
        w = self.work
        self.eng = w.solver
        if not hasattr(self.eng, "pool_create"):
            raise TypeError("the streaming search needs the HIP engine (miosqp_amd.qp)")
        self.columns = int(columns)
        self.p = w.data.n_int
        # every column dives on its own, and until the first incumbent nothing can be pruned: on config 2 the first
        # incumbent sits ~58 levels down, by which time 256 columns have decided ~20 000 nodes, each leaving one
        # open sibling.  Slots are cheap (config 2: 18 KB each, HBM holds 288 GB): default 65 536 slots, at most 8 GB.
        slot_bytes = 8 * (w.data.n + w.data.m + 3 * self.p + 1)
        self.capacity = int(capacity) if capacity else int(max(4096, min(65536, (8 << 30) // slot_bytes)))
        # ring target = the columns known to be free + this margin.  Entries are pushed in the exploration rule's
        # order AT PUSH TIME: a long ring commits columns to shallow leaves chunks ahead (with columns + columns the
        # search degenerates to breadth first), a short one runs dry -- the host's view of the free columns is one
        # launch old.  Config 2, 256 columns: occupancy 0.82 / 0.87 / 0.95 / 1.00 at margin 16 / 32 / 64 / 128.
        self.margin = int(ring_margin) if ring_margin else max(32, self.columns // 2)
        self.active = 0
        if not getattr(self.eng, "_pool_made", False):
            self.eng.pool_create(self.capacity, self.columns)
            self.eng._pool_made = True
            self.eng._pool_capacity = self.capacity
        self.capacity = self.eng._pool_capacity
        cap = self.capacity
        # per-slot bookkeeping as plain lists (scalar access from the per-node logic: several times faster than numpy)
        self.depth = [0] * cap
        self.lower = [-np.inf] * cap             # bound a node inherited; its own once solved
        self.parent = [-1] * cap
        self.kids_alive = [0] * cap
        self.child = [(-1, -1)] * cap
        self.nodes = 0
        self.iters = 0
        self.dropped = 0
        self.chunks = 0
        self.const_solved = w.constant('OSQP_SOLVED')
        self.const_maxit = w.constant('OSQP_MAX_ITER_REACHED')
        self._decided = [False] * cap
        self.open = OpenLeaves(self.depth, self.lower)
        self.begin_instance()

    # -- instance -------------------------------------------------------------------------------
    def begin_instance(self, seed_root=True):
        """(Re)starts on the model's current root: call after MIOSQP.update_vectors.  seed_root=False starts with
        no leaf at all (the sharded search hands this rank its share through add_leaf)."""
        w = self.work
        self._inst0 = self.nodes  # nodes solved before this instance (run()'s cap counts from here)
        self.eng.pool_reset()
        for k in range(self.capacity):
            self._decided[k] = False
            self.kids_alive[k] = 0
        self.free = list(range(self.capacity - 1, -1, -1))  # pop() hands out slot 0 first
        self.open.clear()     # slots of open leaves not yet pushed (OpenLeaves: creation order, rule-ordered take)
        self.in_flight = 0    # pushed and not yet decided
        if seed_root:
            root = w.leaves[0] if w.leaves else w._make_root()
            self.add_leaf(root.l[-self.p:], root.u[-self.p:], root.x, root.y, 0, root.lower)
        w.leaves = []
        self.eng.pool_set_upper(w.upper_glob)
        self._launched = False
        self.active = 0

    # -- leaves in and out (root, leaves dealt to or taken from this rank) ---------------------------
    def add_leaf(self, l_int, u_int, x0, y0, depth, lower):
        """A leaf given with explicit vectors: its integer-row bounds and its warm start."""
        if not self.free:
            raise MemoryError("leaf pool exhausted (raise `capacity`)")
        s = self.free.pop()
        self.eng.pool_write_node(s, l_int, u_int, x0, y0)
        self.depth[s] = int(depth)
        self.lower[s] = lower
        self.parent[s] = -1
        self.kids_alive[s] = 0
        self._decided[s] = False
        self.open.append(s)
        return s

    def givable(self):
        """Open leaves this rank could hand to another one (not yet pushed to the device)."""
        return len(self.open)

    def give_leaf(self, into=None):
        """Takes the shallowest open leaf out of this rank's tree (the largest subtree: keeps the receiver busy
        longest) and returns it with explicit vectors: (l_int, u_int, x0, y0, depth, lower).  into: four qp.DevicePtr the
        vectors are copied to instead (they stay on the device)."""
        s = self.open.pop_shallowest()
        kw_lu = dict(into=dict(l=into[0], u=into[1])) if into is not None else {}
        kw_xy = dict(into=dict(x=into[2], y=into[3])) if into is not None else {}
        nd = self.eng.pool_read_node(s, self.p, want=("l", "u"), **kw_lu)
        ws = self.parent[s] if self.parent[s] >= 0 else s  # its warm start: the parent's solution
        sol = self.eng.pool_read_node(ws, self.p, want=("x", "y"), **kw_xy)
        rec = (nd.l, nd.u, sol.x, sol.y, int(self.depth[s]), float(self.lower[s]))
        self._done(s)
        return rec

    def adopt_incumbent(self, value, x):
        """An incumbent found by another rank."""
        w = self.work
        if value < w.upper_glob:
            w.upper_glob = value
            w.x = np.array(x, dtype=float)
            self.eng.pool_set_upper(value)
            self._prune_open()

    # -- slots ----------------------------------------------------------------------------------
    def _done(self, s):
        """Node `s` has been decided (or discarded).  Its slot stays allocated while a child may still read its
        solution as a warm start; the last child to be decided frees it.  `s` itself no longer needs its parent."""
        self._decided[s] = True
        if self.kids_alive[s] == 0:
            self.free.append(s)
        par = self.parent[s]
        if par >= 0:
            self.kids_alive[par] -= 1
            if self.kids_alive[par] == 0 and self._decided[par]:
                self.free.append(par)

    # -- host side of bound_and_branch (workspace.py:282-334) on a digest --------------------------
    # digest fields (include/miosqp_amd.h: miosqp_pool_digest), as positions of the row tuples
    G_SLOT, G_STATUS, G_ITER, G_INTINF, G_NEXTVAR, G_RES, G_LOWER, G_HVIOL, G_HOBJ = range(9)

    def _absorb(self, g, rec=None):
        """g: one digest as a tuple (fields above); rec: the same as a numpy record, for the observer."""
        w = self.work
        s = g[0]
        if self.observer is not None:
            self.observer(self, rec)
        self.in_flight -= 1
        c0, c1 = self.child[s]
        st = g[1]
        if st == PRUNED or st not in (self.const_solved, self.const_maxit):
            branch = False
            if st == PRUNED:
                self.dropped += 1
            else:
                self.nodes += 1
                self.iters += g[2]
                w.iter_num += 1
                w.osqp_iter += g[2]
        else:
            self.nodes += 1
            self.iters += g[2]
            w.iter_num += 1
            w.osqp_iter += g[2]
            lower = g[6]
            self.lower[s] = lower
            branch = not lower > w.upper_glob
            if branch and g[3] == 0:
                w.x = self.eng.pool_read_node(s, self.p, want=("x",)).x
                w.upper_glob = lower
                self.eng.pool_set_upper(w.upper_glob)
                self._prune_open()
                branch = False
            elif branch and g[7] <= 0.0 and g[8] < w.upper_glob:
                x = self.eng.pool_read_node(s, self.p, want=("x",)).x
                x_int = w.get_integer_solution(x)
                obj_int = w.data.compute_obj_val(x_int)  # the host's own value enters upper_glob (bnb.py digest branch)
                if obj_int < w.upper_glob:
                    w.upper_glob = obj_int
                    w.x = x_int
                    self.eng.pool_set_upper(w.upper_glob)
                    self._prune_open()
        if not branch:
            if c0 >= 0:
                self.free.append(c0)
            if c1 >= 0:
                self.free.append(c1)
            self.child[s] = (-1, -1)
            self.kids_alive[s] = 0
            self._done(s)
            return
        # both children exist on the device (written by the harvest): they become open leaves
        alive = 0
        d1 = self.depth[s] + 1
        for c in (c0, c1):
            if c < 0:
                continue
            self.depth[c] = d1
            self.lower[c] = lower
            self.parent[c] = s
            self.kids_alive[c] = 0
            self._decided[c] = False
            self.open.append(c)
            alive += 1
        self.kids_alive[s] = alive
        self._done(s)

    def _prune_open(self):
        """Open leaves whose inherited bound exceeds the new incumbent are discarded (workspace.py:274-280;
        the reference's skip-one traversal quirk is not reproduced: every such leaf goes)."""
        for s in self.open.prune(self.work.upper_glob):
            self._done(s)

    def _push(self, ready_left):
        """Keeps the ready ring topped up with the leaves the exploration rule would take next."""
        w = self.work
        room = (self.columns - self.active) + self.margin - int(ready_left)
        if room <= 0 or not self.open:
            return
        rule = w.settings['tree_explor_rule']
        if rule not in (0, 1):
            raise ValueError('Tree exploring strategy not recognized')
        by_lower = rule == 1 and not np.isinf(w.upper_glob)
        slots = self.open.take(min(room, len(self.free) // 2), by_lower)  # two child slots each
        if not slots:
            return
        free, child = self.free, self.child
        c0, c1 = [], []
        for s in slots:
            a, b = free.pop(), free.pop()
            c0.append(a)
            c1.append(b)
            child[s] = (a, b)
        self.eng.pool_push(slots, c0, c1, [self.lower[s] for s in slots])
        self.in_flight += len(slots)

    # -- driver ---------------------------------------------------------------------------------
    def step(self, chunks=1):
        """One round, pipelined one launch deep: a new launch of `chunks` chunks is queued, then the host waits
        for the launch BEFORE it and absorbs the digests of the nodes decided in that one -- the device always
        has a launch to work on while the host bounds, prunes and pushes.  Returns the number of leaves still
        alive (open on the host, waiting in the ring or being solved)."""
        if not self._launched:
            self._push(0)
        if not self.open and self.in_flight == 0:
            return 0  # nothing alive on this rank (it may be handed leaves later): no launch
        self.eng.pool_launch(chunks)
        self._launched = True
        self.chunks += chunks
        # one launch stays in flight while the host works -- except every 64th round, which drains the stream
        # (a full synchronisation point for tools that hook the runtime; costs one bubble in 64 chunks)
        self._rounds = getattr(self, "_rounds", 0) + 1
        dg, self.active, left = self.eng.pool_collect(keep_in_flight=0 if self._rounds % 64 == 0 else 1)
        self._absorb_all(dg)
        self._push(left)
        alive = len(self.open) + self.in_flight
        if alive == 0 or (self.in_flight == 0 and self.open):
            # the tree may be closed -- or the host has leaves it could not push: drain the launch in flight first
            dg, self.active, left = self.eng.pool_collect(keep_in_flight=0)
            self._absorb_all(dg)
            self._push(left)
            alive = len(self.open) + self.in_flight
            if self.in_flight == 0 and self.open:
                raise MemoryError("leaf pool exhausted: %d open leaves, %d free slots of %d (raise `capacity`)"
                                  % (len(self.open), len(self.free), self.capacity))
        return alive

    def _absorb_all(self, dg):
        rows = dg.tolist()  # one conversion per collect: tuples of Python scalars
        if self.observer is None:
            for g in rows:
                self._absorb(g)
        else:
            for k, g in enumerate(rows):
                self._absorb(g, dg[k])

    def run(self, chunks=1, max_nodes=None):
        w = self.work
        cap = w.settings['max_iter_bb'] if max_nodes is None else max_nodes
        alive = 1
        while alive > 0 and self.nodes - self._inst0 + 1 < cap:  # (the cap counts the nodes of THIS instance)
            alive = self.step(chunks)
        w.osqp_iter_avg = w.osqp_iter / float(max(1, w.iter_num))
        w.get_return_status(finished=(alive == 0))
        w.get_return_solution()
        return bnb.Results(w.x, w.upper_glob, w.run_time, w.status, w.osqp_solve_time, w.osqp_iter_avg)


class NativeStreamSearch(object):
    """StreamSearch with its per-round work in the C++ host library (miosqp_qp_stream_*, csrc/host_stream.inc): the
    same exploration order and node counts, no interpreter between two chunks.  Same interface; no observer (use
    StreamSearch for that).  What differs: the value of an incumbent found by the rounding heuristic is the device's
    (StreamSearch recomputes it with numpy; ~1e-12 relative apart)."""

    def __init__(self, model, columns=256, capacity=None, ring_margin=None, rounds=1):
        self.model, self.work = model, model.work
        w = self.work
        self.eng = w.solver
        if not hasattr(self.eng, "stream_create"):
            raise TypeError("the native streaming search needs the HIP engine (miosqp_amd.qp)")
        if w.settings['tree_explor_rule'] not in (0, 1):
            raise ValueError('Tree exploring strategy not recognized')
        self.columns, self.p, self.rounds = int(columns), w.data.n_int, int(rounds)
        slot_bytes = 8 * (w.data.n + w.data.m + 3 * self.p + 1)
        self.capacity = int(capacity) if capacity else int(max(4096, min(65536, (8 << 30) // slot_bytes)))
        if not getattr(self.eng, "_sdriver_made", False):
            if getattr(self.eng, "_pool_made", False):
                self.capacity = self.eng._pool_capacity
            self.eng.stream_create(self.capacity, self.columns, int(ring_margin or 0))
            self.eng._sdriver_made = self.eng._pool_made = True
            self.eng._pool_capacity = self.capacity
        self.capacity = self.eng._pool_capacity
        self.nodes = self.iters = self.chunks = self.dropped = 0
        self._open = self.in_flight = 0
        self._free = self.capacity
        self.begin_instance()

    open = property(lambda self: range(self._open))
    free = property(lambda self: range(self._free))

    def _sync(self, info):
        w = self.work
        w.iter_num += info.nodes - self.nodes
        w.osqp_iter += info.osqp_iter - self.iters
        self.nodes, self.iters, self.chunks, self.dropped = info.nodes, info.osqp_iter, info.chunks, info.dropped
        self._open, self.in_flight, self._free = info.open_leaves, info.in_flight, info.free_slots
        if info.improved:
            w.upper_glob, w.x = self.eng.stream_get_incumbent()
        return info.alive

    def begin_instance(self, seed_root=True):
        w = self.work
        self._inst0 = self.nodes  # nodes solved before this instance (run()'s cap counts from here)
        self.eng.stream_begin()
        self._open, self.in_flight, self._free = 0, 0, self.capacity
        if seed_root:
            root = w.leaves[0] if w.leaves else w._make_root()
            self.add_leaf(root.l[-self.p:], root.u[-self.p:], root.x, root.y, 0, root.lower)
        w.leaves = []
        if np.isfinite(w.upper_glob):
            self.eng.stream_set_incumbent(w.upper_glob, w.x)

    def add_leaf(self, l_int, u_int, x0, y0, depth, lower):
        self.eng.stream_add_leaf(l_int, u_int, x0, y0, depth, lower)
        self._open += 1
        self._free -= 1

    def givable(self):
        return self._open

    def give_leaf(self, into=None):
        rec = self.eng.stream_take_leaf(self.p, into) if into is not None else self.eng.stream_take_leaf(self.p)
        self._sync(self.eng.stream_step(self.work.settings['tree_explor_rule'], 1, 0))  # (no round: the counts)
        return rec

    def adopt_incumbent(self, value, x):
        w = self.work
        if value < w.upper_glob:
            w.upper_glob = value
            w.x = np.array(x, dtype=float)
            self.eng.stream_set_incumbent(value, w.x)

    def step(self, chunks=1, rounds=None, max_nodes=0):
        return self._sync(self.eng.stream_step(self.work.settings['tree_explor_rule'], chunks,
                                               self.rounds if rounds is None else rounds, max_nodes))

    def run(self, chunks=1, max_nodes=None):
        w = self.work
        cap = w.settings['max_iter_bb'] if max_nodes is None else max_nodes
        alive = 1
        while alive > 0 and self.nodes - self._inst0 + 1 < cap:  # (the cap counts the nodes of THIS instance)
            alive = self.step(chunks, rounds=64, max_nodes=self._inst0 + cap)
        w.osqp_iter_avg = w.osqp_iter / float(max(1, w.iter_num))
        w.get_return_status(finished=(alive == 0))
        w.get_return_solution()
        return bnb.Results(w.x, w.upper_glob, w.run_time, w.status, w.osqp_solve_time, w.osqp_iter_avg)


class MultiPoolSearch(object):
    """Several streaming pools on ONE GPU, one tree.  `pools` engines of the same MIQP (each its own stream, batch
    and leaf pool; the factor is set up once per engine) are driven by `pools` host threads and share the tree like
    ranks do (dist.ShardedStream over poolcomm.PoolComm: replicated ramp-up, deal, incumbent + dry-pool feed every
    `exchange_every` chunks).  A chunk of the streaming batch has a serial part -- termination test, harvest, refill:
    twenty small kernels, 190 us of 970 at config 3 -- during which the chip is nearly idle; with two pools the
    sweeps of one fill those gaps of the other: 6.3 -> 8.7 M node-iterations/s at 2 x 256 columns on MI355X.
    The ctypes calls release the interpreter lock while they wait for the device, so the threads overlap where it
    matters.

    make_model: () -> a set-up bnb.MIOSQP (called `pools` times; every model must describe the same problem)."""

    def __init__(self, make_model, pools=2, columns=256, exchange_every=4, capacity=None, driver="python"):
        """driver: "python" -- every pool's rounds in StreamSearch (one interpreter, `pools` threads taking turns on its
        lock between two chunks); "native" -- in NativeStreamSearch (the C++ library runs `exchange_every` rounds per
        call without the lock; the threads only meet the interpreter for the exchange)."""
        from miosqp_amd import dist, poolcomm
        if driver not in ("python", "native"):
            raise ValueError("driver: 'python' or 'native'")
        self.driver = driver
        self.pools = int(pools)
        self.models = [make_model() for _ in range(self.pools)]
        self.world = poolcomm.PoolWorld(self.pools)
        self._poolcomm = poolcomm
        self.columns, self.exchange_every, self.capacity = columns, exchange_every, capacity
        self.sh = [None] * self.pools
        self._dist = dist

    def _each(self, fn):
        import threading
        out, err = [None] * self.pools, []

        def body(k):
            try:
                out[k] = fn(k)
            except BaseException as e:  # noqa: BLE001 -- re-raised in the caller's thread
                err.append(e)
                self.world.fail(e)  # peers waiting in a barrier or an exchange raise at once

        ths = [threading.Thread(target=body, args=(k,)) for k in range(self.pools)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if err:
            raise err[0]
        return out

    def _shard(self, k):
        if self.sh[k] is None:
            comm = self._poolcomm.PoolComm(self.world, k)
            if self.driver == "native":
                ns = NativeStreamSearch(self.models[k], columns=self.columns, capacity=self.capacity,
                                        rounds=self.exchange_every)
                self.sh[k] = self._dist.ShardedStream(self.models[k], comm, exchange_every=1, search=ns)
            else:
                self.sh[k] = self._dist.ShardedStream(self.models[k], comm, columns=self.columns,
                                                      exchange_every=self.exchange_every, capacity=self.capacity)
        return self.sh[k]

    def update_vectors(self, q=None, l=None, u=None):
        """New MIQP on the same factors (every engine)."""
        for m in self.models:
            m.update_vectors(q=q, l=l, u=u)
        self._fresh = True

    def run(self):
        """Closes the current tree.  Returns bnb.Results (identical on every pool; pool 0's)."""
        fresh = getattr(self, "_fresh", False)
        self._fresh = False

        def body(k):
            first = self.sh[k] is None
            sh = self._shard(k)
            if fresh and not first:
                sh.begin_instance()
            sh.run()
            return None

        self._each(body)
        w = self.models[0].work
        return bnb.Results(w.x, w.upper_glob, w.run_time, w.status, w.osqp_solve_time, w.osqp_iter_avg)

    def steps(self, count, next_instance=None):
        """`count` chunks per pool (bench): when the tree closes, next_instance(k, model) must re-root model k -- with the
        same data on every pool -- and the search goes on.  Returns per-pool (nodes, iterations, chunks)."""
        def body(k):
            sh = self._shard(k)
            per_step = self.exchange_every if self.driver == "native" else 1
            for _ in range(max(1, count // per_step)):
                if sh.step() == 0:
                    if next_instance is None:
                        break
                    next_instance(k, self.models[k])
                    sh.begin_instance()
            return (sh.ss.nodes, sh.ss.iters, sh.ss.chunks)

        return self._each(body)
