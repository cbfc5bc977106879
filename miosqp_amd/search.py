"""Node-at-a-time branch and bound with the loop in the C++ host library (miosqp_qp_search_*).

The reference's loop (/root/reference/miosqp/solver.py:65-172) asks the interpreter for every node: choose_leaf,
Node.solve (four vectors over PCIe each way), bound_and_branch with its numpy copies of l and u -- about 125 us next
to a 800 us relaxation on config 2.  Here the open leaves are device slots, the children are written on the device,
a node's outcome is a 96-byte record and the loop itself is compiled code; Python sees the search between calls
(after a node, a batch of nodes or a time budget): for the incumbent exchange of the sharded search, for handing
leaves to other ranks, for an observer.  Same list semantics and decisions as bnb.Workspace (creation order, first
maximum, the prune traversal of workspace.py:278-280).  An incumbent found by the rounding heuristic carries the
device's sum while the compiled loop runs; when the call returns its value is recomputed on the host as
workspace.py:321-327 does and handed back (`search_set_incumbent(value, None)`), so results and later comparisons use
the reference's number (the two agree to ~1e-12 relative).

HostedSearch has the interface of stream.StreamSearch, so dist.ShardedStream shards either."""
import numpy as np

from miosqp_amd import bnb


class HostedSearch(object):
    def __init__(self, model, capacity=None, owned=False):
        """owned: this object is kept BY the model's workspace (bnb.MIOSQP._solve_hosted): it then refers back through
        weak proxies, so that dropping the model frees the engine at once (a reference cycle would keep the device pool and
        the stream / pinned-buffer bundle until the cyclic collector runs -- the next setup then pays for new ones)."""
        if owned:
            import weakref
            self.model, self.work = weakref.proxy(model), weakref.proxy(model.work)
        else:
            self.model, self.work = model, model.work
        w = model.work
        self.eng = w.solver
        if not hasattr(self.eng, "search_create"):
            raise RuntimeError("the engine has no hosted search (miosqp_qp_search_*)")
        if w.settings['branching_rule'] != 0 or w.settings['tree_explor_rule'] not in (0, 1):
            raise ValueError("hosted search: branching_rule 0 and tree_explor_rule 0 / 1 only")
        self.p = w.data.n_int
        if capacity is None:
            slot = 8 * (w.data.n + w.data.m + 3 * self.p)
            capacity = int(max(64, min(16384, (2 << 30) // slot)))
        self.capacity = int(capacity)
        if not getattr(self.eng, "_search_ready", False):
            self.eng.search_create(self.capacity)
            self.eng._search_ready = True
        self.nodes = self.iters = 0
        self.device_time = 0.0
        self.in_flight = 0  # (interface of StreamSearch)
        self.dropped = 0
        self._open = 0
        self._free = self.capacity
        self.begin_instance()

    # list-like views the sharded wrapper looks at
    @property
    def open(self):
        return range(self._open)

    @property
    def free(self):
        return range(self._free)

    def begin_instance(self, seed_root=True):
        """(Re)starts on the model's current root: call after MIOSQP.update_vectors."""
        w = self.work
        self.eng.search_reset()
        # (the slot store grows on demand: ask what it holds now)
        self.capacity = int(self.eng.search_run(w.settings['tree_explor_rule'], 0).free_slots)
        self._open, self._free = 0, self.capacity
        if seed_root:
            root = w.leaves[0] if w.leaves else w._make_root()
            self.add_leaf(root.l[-self.p:], root.u[-self.p:], root.x, root.y, 0, root.lower)
        w.leaves = []
        if np.isfinite(w.upper_glob):
            self.eng.search_set_incumbent(w.upper_glob, w.x)

    def add_leaf(self, l_int, u_int, x0, y0, depth, lower):
        self.eng.search_add_leaf(l_int, u_int, x0, y0, depth, lower)
        self._open += 1
        self._free -= 1

    def givable(self):
        return self._open

    def give_leaf(self, into=None):
        """into: four qp.DevicePtr -- the leaf's vectors stay on the device (dist.ShardedStream)"""
        rec = self.eng.search_take_leaf(self.p, into) if into is not None else self.eng.search_take_leaf(self.p)
        info = self.eng.search_run(self.work.settings['tree_explor_rule'], 0)  # (no node: the counts)
        self._open, self._free = info.open_leaves, info.free_slots
        return rec

    def adopt_incumbent(self, value, x):
        w = self.work
        if value < w.upper_glob:
            w.upper_glob = value
            w.x = np.array(x, dtype=float)
            self.eng.search_set_incumbent(value, w.x)

    def step(self, nodes=1, budget=None):
        """Up to `nodes` nodes (or `budget` seconds of them).  Returns the number of open leaves."""
        w = self.work
        if self._open == 0:
            return 0
        info = self.eng.search_run(w.settings['tree_explor_rule'], nodes, 0.0 if budget is None else budget)
        self.nodes += info.nodes
        self.iters += info.osqp_iter
        self.device_time += info.device_time
        w.iter_num += info.nodes
        w.osqp_iter += info.osqp_iter
        w.osqp_solve_time += info.device_time
        w.lower_glob = info.lower_glob
        self._open, self._free = info.open_leaves, info.free_slots
        if info.improved:
            w.upper_glob, x = self.eng.search_get_incumbent()
            w.x = x
            if info.improved == 2:
                # found by the rounding heuristic: its value is recomputed exactly as workspace.py:321-327 does (the
                # device sums in another order, ~1e-12 relative apart) and handed back, so that what the caller sees
                # and every later comparison use the reference's number
                w.upper_glob = float(w.data.compute_obj_val(x))
                self.eng.search_set_incumbent(w.upper_glob, None)
        if getattr(info, "full", False):
            # the device has no memory left for more open leaves: the counters above are consistent, say what happened
            raise MemoryError("hosted search: the leaf store cannot grow any further (%d open leaves, %d nodes done)"
                              % (self._open, w.iter_num))
        return self._open

    def run(self, max_nodes=None):
        w = self.work
        cap = w.settings['max_iter_bb'] if max_nodes is None else max_nodes
        alive = self._open
        while alive > 0 and w.iter_num < cap:
            alive = self.step(cap - w.iter_num)
        w.osqp_iter_avg = w.osqp_iter / float(max(1, w.iter_num))
        w.get_return_status(finished=(alive == 0))
        w.get_return_solution()
        return bnb.Results(w.x, w.upper_glob, w.run_time, w.status, w.osqp_solve_time, w.osqp_iter_avg)
