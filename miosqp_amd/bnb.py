"""Host-side branch-and-bound control: the caller of the hot path.

This module is the host counterpart of the reference's Python tree search, kept on the CPU
("stays on the host unchanged" in BASELINE.json's north_star).  It reproduces the observable
behaviour of the reference classes so that, given identical relaxation results, the tree is
explored in the same order and the same statistics come out (SURVEY.md sec. 3.4 lists the
quirks that matter; tests/test_bnb_trace.py replays traces recorded from the reference):

  MIOSQP      /root/reference/miosqp/solver.py:32-212
  Workspace   /root/reference/miosqp/workspace.py:18-433
  Node        /root/reference/miosqp/node.py:5-147
  Data        /root/reference/miosqp/data.py:36-126, add_bounds data.py:5-33
  Results     /root/reference/miosqp/results.py:1-12
  MI_*        /root/reference/miosqp/constants.py:1-7

The only thing replaced is what `Node.solve` calls: the relaxation solver is the HIP engine
behind the C ABI (miosqp_amd.qp), reached either through the reference's own four-call
sequence update -> warm_start -> solve (node.py:102-108) or through the fused per-node entry
`solve_node`, which also performs the integer clamp (node.py:131-136) and the objective
evaluation (node.py:143 -> data.py:99-103) on the device.

A different solver module can be passed explicitly as `backend=` (tests pass the CPU oracle);
there is no automatic fallback: without an explicit backend the HIP library must load.
"""
from __future__ import print_function

from time import time

import numpy as np
import scipy.sparse as spa

# status strings, verbatim (constants.py:2 really says 'Unolved')
MI_UNSOLVED = 'Unolved'
MI_SOLVED = 'Solved'
MI_PRIMAL_INFEASIBLE = 'Primal Infeasible'
MI_DUAL_INFEASIBLE = 'Dual Infeasible'
MI_MAX_ITER_FEASIBLE = 'Max-iter feasible'
MI_MAX_ITER_UNSOLVED = 'Max-iter unsolved'


def _default_backend():
    from miosqp_amd import qp  # raises if the HIP library cannot be loaded
    return qp


def add_bounds(i_idx, l_new, u_new, A, l, u):
    """Append l_new <= x[i_idx] <= u_new as identity rows of A (data.py:5-33)."""
    n = A.shape[1]
    rows = spa.identity(n, format='csc')[i_idx, :]
    return spa.vstack([A, rows]).tocsc(), np.append(l, l_new), np.append(u, u_new)


class Data(object):
    """Relaxed-QP data with the integer bounds as trailing constraint rows (data.py:78-97)."""

    def __init__(self, P, q, A, l, u, i_idx, i_l, i_u):
        self.m, self.n = A.shape
        self.n_int = len(i_idx)
        self.A, self.l, self.u = add_bounds(i_idx, i_l, i_u, A, l, u)
        self.P = P.tocsc()
        self.q = q
        self.i_idx = i_idx
        self.i_l = i_l
        self.i_u = i_u

    def compute_obj_val(self, x):
        # data.py:99-103
        return .5 * np.dot(x, self.P.dot(x)) + np.dot(self.q, x)

    def update_vectors(self, q=None, l=None, u=None):
        # data.py:105-126 (l, u are written in place: the root node shares these arrays)
        if q is not None:
            if len(q) != self.n:
                raise ValueError('Wrong q dimension!')
            self.q = q
        if l is not None:
            if len(l) != self.m:
                raise ValueError('Wrong l dimension!')
            self.l[:self.m] = l
        if u is not None:
            if len(u) != self.m:
                raise ValueError('Wrong u dimension!')
            self.u[:self.m] = u


class Node(object):
    """One branch-and-bound node = one relaxation (node.py:41-94)."""

    def __init__(self, data, l, u, solver, depth=0, lower=None, x0=None, y0=None,
                 constant=None):
        self.data = data
        self.l = l
        self.u = u
        self.solver = solver
        self.depth = depth
        self.lower = -np.inf if lower is None else lower
        self.frac_idx = None
        self.intinf = None
        self.num_iter = 0
        self.osqp_solve_time = 0
        self.x = np.zeros(data.n) if x0 is None else x0
        self.y = np.zeros(data.m + data.n_int) if y0 is None else y0
        self._constant = constant if constant is not None else solver.constant
        self.status = self._constant('OSQP_UNSOLVED')
        self.nextvar_idx = None
        self.constr_idx = None
        self.digest = None  # filled by the device epilogue (miosqp_qp_set_root) when available

    def _absorb(self, status, num_iter, run_time, x, y, lower):
        self.status = status
        self.num_iter = num_iter
        self.osqp_solve_time = run_time
        self.x = x
        self.y = y
        if lower is not None:
            self.lower = lower

    def solve(self):
        """Lower bound of this node's relaxation (node.py:96-143)."""
        if hasattr(self.solver, 'solve_node'):
            # fused device path: bounds + warm start + ADMM + clamp + objective in one call
            r = self.solver.solve_node(self.l, self.u, self.x, self.y)
            self._absorb(r.status_val, r.iter, r.run_time, r.x, r.y, r.lower)
            self.digest = getattr(r, 'digest', None)
            return
        self.solver.update(l=self.l, u=self.u)
        self.solver.warm_start(x=self.x, y=self.y)
        res = self.solver.solve()
        self._absorb(res.info.status_val, res.info.iter, res.info.run_time, res.x, res.y, None)
        if self.status in (self._constant('OSQP_SOLVED'),
                           self._constant('OSQP_MAX_ITER_REACHED')):
            k = self.data.n_int
            ii = self.data.i_idx
            self.x[ii] = np.minimum(np.maximum(self.x[ii], self.l[-k:]), self.u[-k:])
            self.lower = self.data.compute_obj_val(self.x)


class Results(object):
    def __init__(self, x, upper_glob, run_time, status, osqp_solve_time, osqp_iter_avg):
        self.x = x
        self.upper_glob = upper_glob
        self.run_time = run_time
        self.status = status
        self.osqp_solve_time = osqp_solve_time
        self.osqp_iter_avg = osqp_iter_avg


class Workspace(object):
    """Tree state + the single shared relaxation solver (workspace.py:58-92)."""

    def __init__(self, data, settings, qp_settings=None, backend=None):
        self.data = data
        self.settings = settings
        self.backend = backend if backend is not None else _default_backend()
        self.constant = self.backend.constant
        self.solver = self.backend.OSQP()
        self.qp_settings = {} if qp_settings is None else qp_settings
        # workspace.py:67-68 expands the *argument*: qp_settings=None is a TypeError there too
        self.solver.setup(data.P, data.q, data.A, data.l, data.u, **qp_settings)
        if hasattr(self.solver, 'set_integer_rows'):
            self.solver.set_integer_rows(data.i_idx, data.m)
        self.push_root()
        self._reset_counters()
        self.first_run = 1
        self.leaves = [self._make_root()]
        self.upper_glob = np.inf
        self.x = np.empty(data.n)
        self.setup_time = 0.
        self.solve_time = 0.
        self.run_time = 0.

    def push_root(self):
        """Hands the root bounds and the two tolerances to the engine so that the x-only part of
        bound_and_branch (integrality test, branching variable, rounding heuristic) is evaluated on
        the device at the end of each node (settings['device_digest'] = False keeps it on the host)."""
        if hasattr(self.solver, 'set_root') and self.settings.get('device_digest', True) \
                and 'eps_abs' in self.qp_settings and self.data.n_int > 0:
            self.solver.set_root(self.data.l, self.data.u, self.settings['eps_int_feas'],
                                 self.qp_settings['eps_abs'])

    # -- bookkeeping ---------------------------------------------------------------------
    def _reset_counters(self):
        self.iter_num = 1
        self.osqp_solve_time = 0.
        self.osqp_iter = 0
        self.osqp_iter_avg = 0
        self.lower_glob = -np.inf
        self.defer_lower = False
        self.status = MI_UNSOLVED

    def _make_root(self):
        return Node(self.data, self.data.l, self.data.u, self.solver, constant=self.constant)

    def _is(self, leaf, *names):
        return any(leaf.status == self.constant(nm) for nm in names)

    def set_x0(self, x0):
        # workspace.py:94-111
        root = self.leaves[0]
        if self.satisfies_lin_constraints(x0, root.l, root.u) and self.is_int_feas(x0, root):
            self.x = x0
            self.upper_glob = self.data.compute_obj_val(x0)
        else:
            print('Invalid initial solution!\n')
            self.upper_glob = np.inf
            self.x = np.empty(self.data.n)

    def can_continue(self):
        # workspace.py:113-126
        return len(self.leaves) > 0 and self.iter_num < self.settings['max_iter_bb']

    # -- tree exploration ----------------------------------------------------------------
    def leaf_index(self, tree_explor_rule):
        """Index of the next leaf (workspace.py:128-149; note argmax of `lower` in phase two)."""
        if tree_explor_rule == 0 or (tree_explor_rule == 1 and np.isinf(self.upper_glob)):
            return int(np.argmax([lf.depth for lf in self.leaves]))
        if tree_explor_rule == 1:
            return int(np.argmax([lf.lower for lf in self.leaves]))
        raise ValueError('Tree exploring strategy not recognized')

    def choose_leaf(self, tree_explor_rule):
        return self.leaves.pop(self.leaf_index(tree_explor_rule))

    def _child(self, leaf, l, u):
        if np.any(l > u):
            # the reference drops into a debugger here (workspace.py:170-171,194-195)
            raise RuntimeError('branching produced l > u')
        self.leaves.append(Node(self.data, l, u, self.solver, depth=leaf.depth + 1,
                                lower=leaf.lower, x0=leaf.x, y0=leaf.y,
                                constant=self.constant))

    def add_left(self, leaf):
        l, u = np.copy(leaf.l), np.copy(leaf.u)
        u[leaf.constr_idx] = np.floor(leaf.x[leaf.nextvar_idx])
        self._child(leaf, l, u)

    def add_right(self, leaf):
        l, u = np.copy(leaf.l), np.copy(leaf.u)
        l[leaf.constr_idx] = np.ceil(leaf.x[leaf.nextvar_idx])
        self._child(leaf, l, u)

    def pick_nextvar(self, leaf):
        # workspace.py:205-230: largest fractional part among the still-fractional integers
        if self.settings['branching_rule'] != 0:
            raise ValueError('No variable selection rule recognized!')
        xf = leaf.x[self.data.i_idx[leaf.frac_idx]]
        nextvar = leaf.frac_idx[int(np.argmax(abs(xf - np.round(xf))))]
        leaf.constr_idx = self.data.m + nextvar
        leaf.nextvar_idx = self.data.i_idx[nextvar]

    def satisfies_lin_constraints(self, x, l, u):
        # workspace.py:232-243 (needs 'eps_abs' in qp_settings)
        z = self.data.A.dot(x)
        tol = self.qp_settings['eps_abs']
        return not (np.any(z < l - tol) or np.any(z > u + tol))

    def is_int_feas(self, x, leaf):
        # workspace.py:245-264
        xi = x[self.data.i_idx]
        bad = abs(xi - np.round(xi)) > self.settings['eps_int_feas']
        leaf.frac_idx = np.where(bad)[0].tolist()
        leaf.intinf = np.sum(bad)
        return not leaf.intinf > 0

    def get_integer_solution(self, x):
        xr = np.copy(x)
        xr[self.data.i_idx] = np.round(x[self.data.i_idx])
        return xr

    def solve_wave(self, leaves):
        """Relaxations of several open leaves at once.  With the HIP engine the wave is ONE
        batched device call sharing the factor (`solve_batch`); every leaf ends up exactly as
        its own Node.solve() would leave it."""
        if len(leaves) > 1 and hasattr(self.solver, 'solve_batch'):
            r = self.solver.solve_batch(np.stack([lf.l for lf in leaves]), np.stack([lf.u for lf in leaves]),
                                        np.stack([lf.x for lf in leaves]), np.stack([lf.y for lf in leaves]))
            for k, lf in enumerate(leaves):
                lower = None if np.isnan(r.lower[k]) else float(r.lower[k])
                lf._absorb(int(r.status_val[k]), int(r.iter[k]), float(r.run_time[k]), r.x[k].copy(),
                           r.y[k].copy(), lower)
                lf.digest = r.digest[k] if getattr(r, 'digest', None) is not None else None
        else:
            for lf in leaves:
                lf.solve()

    def prune(self):
        """Drop leaves whose bound exceeds the incumbent, with the reference's traversal:
        workspace.py:278-280 removes from the list it is iterating, so the element following
        each removed one is never examined."""
        k = 0
        while k < len(self.leaves):
            if self.leaves[k].lower > self.upper_glob:
                del self.leaves[k]
            k += 1

    def branch(self, leaf):
        self.pick_nextvar(leaf)
        self.add_left(leaf)
        self.add_right(leaf)

    def update_lower_glob(self):
        # workspace.py:334: after every branching; a wave defers it to its end (`defer_lower`), the value is
        # only reported, never used for a decision
        if not self.defer_lower:
            self.lower_glob = min(lf.lower for lf in self.leaves)

    def bound_and_branch(self, leaf):
        # workspace.py:282-334
        self.osqp_iter += leaf.num_iter
        self.osqp_solve_time += leaf.osqp_solve_time
        if self._is(leaf, 'OSQP_PRIMAL_INFEASIBLE', 'OSQP_DUAL_INFEASIBLE'):
            return
        if leaf.lower > self.upper_glob:
            return
        dg = leaf.digest
        if dg is not None:
            # Same decisions from the device digest of this node's x.  The device sums in another order than
            # numpy, so its objective values agree with the host's to ~1e-12 relative, not bit for bit; what
            # enters `upper_glob` through the rounding heuristic is therefore recomputed on the host (rare:
            # only when the digest says the incumbent improves), exactly as workspace.py:321-327 computes
            # it.  `leaf.lower` is the device value on this path by design (node.py:143 fused into the
            # epilogue; 1e-9 relative, DESIGN.md "tolerances").
            if self.settings['branching_rule'] != 0:
                raise ValueError('No variable selection rule recognized!')
            leaf.intinf = dg.int_inf
            if dg.int_inf == 0:
                leaf.frac_idx = []
                self.x = leaf.x
                self.upper_glob = leaf.lower
                self.prune()
                return
            if dg.heur_feasible and dg.heur_obj < self.upper_glob:
                x_int = self.get_integer_solution(leaf.x)
                obj_int = self.data.compute_obj_val(x_int)
                if obj_int < self.upper_glob:
                    self.upper_glob = obj_int
                    self.x = x_int
                    self.prune()
            xi = leaf.x[self.data.i_idx]
            leaf.frac_idx = np.where(abs(xi - np.round(xi)) > self.settings['eps_int_feas'])[0].tolist()
            leaf.constr_idx = self.data.m + dg.nextvar
            leaf.nextvar_idx = self.data.i_idx[dg.nextvar]
            self.add_left(leaf)
            self.add_right(leaf)
            self.update_lower_glob()
            return
        if self.is_int_feas(leaf.x, leaf):
            self.x = leaf.x
            self.upper_glob = leaf.lower
            self.prune()
            return
        x_int = self.get_integer_solution(leaf.x)
        if self.satisfies_lin_constraints(x_int, self.data.l, self.data.u):
            obj_int = self.data.compute_obj_val(x_int)
            if obj_int < self.upper_glob:
                self.upper_glob = obj_int
                self.x = x_int
                self.prune()
        self.branch(leaf)
        self.update_lower_glob()

    # -- results -------------------------------------------------------------------------
    def get_return_status(self, finished=None):
        # workspace.py:352-373; the sharded search passes `finished` (no open leaf on any rank)
        if finished is None:
            finished = self.iter_num < self.settings['max_iter_bb']
        if self.upper_glob != np.inf:
            self.status = MI_SOLVED if finished else MI_MAX_ITER_FEASIBLE
        elif self.upper_glob >= 0:
            self.status = MI_PRIMAL_INFEASIBLE if finished else MI_MAX_ITER_UNSOLVED
        else:
            self.status = MI_DUAL_INFEASIBLE

    def get_return_solution(self):
        if self.status in (MI_SOLVED, MI_MAX_ITER_FEASIBLE):
            ii = self.data.i_idx
            self.x[ii] = np.round(self.x[ii])

    # -- progress table (workspace.py:386-433) --------------------------------------------
    def print_headline(self):
        print("     Nodes      |           Current Node        |"
              "             Objective Bounds             |   Cur Node")
        print("Explr\tUnexplr\t|      Obj\tDepth\tIntInf  |    Lower\t   Upper\t"
              "    Gap    |     Iter")

    def print_progress(self, leaf):
        if self.upper_glob == np.inf:
            gap = "    --- "
        else:
            gap = "%8.2f%%" % ((self.upper_glob - self.lower_glob) / abs(self.lower_glob) * 100)
        infeas = self._is(leaf, 'OSQP_PRIMAL_INFEASIBLE', 'OSQP_DUAL_INFEASIBLE')
        obj = np.inf if infeas else leaf.lower
        intinf = "  ---" if leaf.intinf is None else "%5d" % leaf.intinf
        tail = "!" if self._is(leaf, 'OSQP_MAX_ITER_REACHED') else ""
        print("%4d\t%4d\t  %10.2e\t%4d\t%s\t  %10.2e\t%10.2e\t%s\t%5d%s" %
              (self.iter_num, len(self.leaves), obj, leaf.depth, intinf, self.lower_glob,
               self.upper_glob, gap, leaf.num_iter, tail))

    def print_footer(self):
        print("\n")
        print("Status: %s" % self.status)
        if self.status == MI_SOLVED:
            print("Objective bound: %6.3e" % self.upper_glob)
        print("Total number of OSQP iterations: %d" % self.osqp_iter)


class MIOSQP(object):
    """Public facade (solver.py:32-212): setup / solve / update_vectors / set_x0."""

    def __init__(self, backend=None):
        self.data = None
        self.work = None
        self._backend = backend

    def setup(self, P, q, A, l, u, i_idx, i_l, i_u, settings, qp_settings):
        t0 = time()
        if i_l is None:
            i_l = -np.inf * np.ones(len(i_idx))
        if i_u is None:
            i_u = np.inf * np.ones(len(i_idx))
        data = Data(P, q, A, l, u, i_idx, i_l, i_u)
        self.work = Workspace(data, settings, qp_settings, backend=self._backend)
        self.work.setup_time = time() - t0

    def solve(self, observer=None):
        """Run the tree search (solver.py:65-172).  `observer(work, leaf)` is an optional hook
        called after each bound_and_branch; tests use it to record traces."""
        t0 = time()
        work = self.work
        verbose = work.settings['verbose']
        if verbose:
            work.print_headline()
        if observer is None and not verbose:
            self._solve_on_device(work)
        while work.can_continue():
            leaf = work.choose_leaf(work.settings['tree_explor_rule'])
            leaf.solve()
            work.bound_and_branch(leaf)
            if observer is not None:
                observer(work, leaf)
            if verbose and work.iter_num % work.settings['print_interval'] == 0:
                work.print_progress(leaf)
            work.iter_num += 1
        work.osqp_iter_avg = work.osqp_iter / work.iter_num
        work.get_return_status()
        work.get_return_solution()
        if verbose:
            work.print_footer()
        work.solve_time = time() - t0
        if work.first_run:
            work.first_run = 0
            work.run_time = work.setup_time + work.solve_time
        else:
            work.run_time = work.solve_time
        if verbose:
            print("Elapsed time: %.4es" % work.run_time)
        return Results(work.x, work.upper_glob, work.run_time, work.status,
                       work.osqp_solve_time, work.osqp_iter_avg)

    def _solve_on_device(self, work):
        """Small problems (the LDS-resident engine form; BASELINE config 4): the whole loop below runs inside ONE
        device launch with the same decisions (`miosqp_qp_solve_tree`, csrc/kernels_tree.inc); the host only sends
        the root and reads the outcome.  Falls through to the host loop when the engine does not cover the problem
        (too large, no device digest) or the leaf list overflowed.  settings['device_tree'] = False keeps the host loop."""
        st = work.settings
        if not hasattr(work.solver, 'solve_tree') or getattr(work, '_no_tree', False) or not st.get('device_tree', True):
            self._solve_hosted(work)
            return
        if st['branching_rule'] != 0 or st['tree_explor_rule'] not in (0, 1) or len(work.leaves) != 1 \
                or work.iter_num != 1 or work.data.n_int == 0 or 'eps_abs' not in work.qp_settings \
                or not st.get('device_digest', True):
            return
        root = work.leaves[0]
        have = np.isfinite(work.upper_glob)
        r = work.solver.solve_tree(root.l, root.u, root.x, root.y, work.upper_glob, work.x if have else None,
                                   st['tree_explor_rule'], st['max_iter_bb'])
        if r is None:
            work._no_tree = True  # this engine form never will: do not ask again
            self._solve_hosted(work)
            return
        if r.info.overflow:
            return  # more leaves alive than the launch holds: the host loop redoes the search from the root
        work.iter_num = r.info.nodes + 1
        work.osqp_iter = r.info.osqp_iter
        work.osqp_solve_time = r.info.device_time
        work.upper_glob = r.info.upper_glob
        work.lower_glob = r.info.lower_glob
        if r.info.found:
            work.x = r.x
        # the leaves live on the device; what matters afterwards is whether any is left (node cap reached)
        work.leaves = [] if r.info.leaves_left == 0 else [root] * int(r.info.leaves_left)
        if work.leaves:
            work.iter_num = max(work.iter_num, st['max_iter_bb'])

    def _solve_hosted(self, work):
        """Larger problems: the same loop in the C++ host library with the leaves in device slots
        (`miosqp_qp_search_*`, miosqp_amd/search.py) -- no interpreter and no vector traffic between two nodes.
        settings['device_search'] = False keeps the Python loop."""
        st = work.settings
        if not st.get('device_search', True) or not hasattr(work.solver, 'search_create'):
            return
        if st['branching_rule'] != 0 or st['tree_explor_rule'] not in (0, 1) or len(work.leaves) != 1 \
                or work.iter_num != 1 or work.data.n_int == 0 or 'eps_abs' not in work.qp_settings \
                or not st.get('device_digest', True):
            return
        from miosqp_amd import search
        hs = getattr(work, '_hosted', None)
        root = work.leaves[0]
        if hs is None:
            hs = work._hosted = search.HostedSearch(self, owned=True)
        else:
            hs.begin_instance()
        alive = hs._open
        while alive > 0 and work.iter_num < st['max_iter_bb']:
            alive = hs.step(st['max_iter_bb'] - work.iter_num)
        work.leaves = [root] * int(alive)  # the leaves live on the device; what matters is whether any is left

    def solve_many(self, instances):
        """B MIQPs on this model's factorisation, solved TOGETHER: instance k is what
        `update_vectors(q=, l=, u=)` [+ `set_x0(x0)`] + `solve()` would solve (the reference's MPC pattern,
        /root/reference/miosqp/solver.py:174-212, examples/power_converter/power_converter.py:467-476), for a list of
        dicts with keys q, l, u (each optional: the model's current vector otherwise) and x0 (optional).  On the HIP
        engine every tree runs in one launch -- one workgroup (one wavefront for n + M <= 64) per instance,
        `miosqp_qp_solve_trees` --, so that many small independent MIQPs fill the chip instead of one compute unit;
        instances the launch cannot hold (leaf list overflow) and engines without the entry point go through the
        sequential calls.  The model itself is left as it was: the one-launch path touches nothing of it, the sequential
        fallback puts q, l, u, the leaf list and the statistics back when it is done or when an instance raises (leaves of
        an unfinished device-hosted search are placeholders and not resumable either way).
        Returns a list of dicts: x, upper_glob, status, nodes, osqp_iter, run_time."""
        work, data, st = self.work, self.work.data, self.work.settings
        B = len(instances)
        if B == 0:
            return []
        n, m, M = data.n, data.m, data.m + data.n_int
        ok_engine = hasattr(work.solver, 'solve_trees') and st.get('device_tree', True) and st['branching_rule'] == 0 \
            and st['tree_explor_rule'] in (0, 1) and data.n_int > 0 and 'eps_abs' in work.qp_settings \
            and st.get('device_digest', True) and not getattr(work, '_no_trees', False)
        Q = np.empty((B, n)); L = np.empty((B, M)); U = np.empty((B, M))
        up = np.full(B, np.inf); XI = np.zeros((B, n)); any_inc = False
        for k, inst in enumerate(instances):
            Q[k] = data.q if inst.get('q') is None else inst['q']
            L[k] = data.l; U[k] = data.u
            if inst.get('l') is not None:
                L[k, :m] = inst['l']
            if inst.get('u') is not None:
                U[k, :m] = inst['u']
            x0 = inst.get('x0')
            if x0 is not None:
                # Workspace.set_x0 (workspace.py:94-111) on this instance's data
                x0 = np.asarray(x0, dtype=float)
                z = data.A.dot(x0)
                tol = work.qp_settings['eps_abs']
                xi = x0[data.i_idx]
                if not (np.any(z < L[k] - tol) or np.any(z > U[k] + tol)) and \
                        not np.any(abs(xi - np.round(xi)) > st['eps_int_feas']):
                    up[k] = .5 * np.dot(x0, data.P.dot(x0)) + np.dot(Q[k], x0)
                    XI[k] = x0
                    any_inc = True
                else:
                    print('Invalid initial solution!\n')
        out = [None] * B
        redo = list(range(B))
        if ok_engine:
            t0 = time()
            r = work.solver.solve_trees(Q, L, U, np.zeros((B, n)), np.zeros((B, M)), up, XI if any_inc else None,
                                        st['tree_explor_rule'], st['max_iter_bb'])
            if r is None:
                work._no_trees = True
            else:
                X, infos = r
                dt = time() - t0
                redo = []
                for k in range(B):
                    info = infos[k]
                    if info.overflow:
                        redo.append(k)
                        continue
                    upper = info.upper_glob
                    # workspace.py:352-373 decides on the loop counter, not on the leaf list: a tree that closes with its
                    # last permitted node reports the MAX_ITER family, exactly as solve() does (iter_num = nodes + 1)
                    finished = int(info.nodes) + 1 < st['max_iter_bb']
                    if upper != np.inf:
                        status = MI_SOLVED if finished else MI_MAX_ITER_FEASIBLE
                    elif upper >= 0:
                        status = MI_PRIMAL_INFEASIBLE if finished else MI_MAX_ITER_UNSOLVED
                    else:
                        status = MI_DUAL_INFEASIBLE
                    x = X[k].copy() if (info.found or np.isfinite(up[k])) else np.empty(n)
                    if status in (MI_SOLVED, MI_MAX_ITER_FEASIBLE):
                        x[data.i_idx] = np.round(x[data.i_idx])
                    out[k] = dict(x=x, upper_glob=upper, status=status, nodes=int(info.nodes),
                                  osqp_iter=int(info.osqp_iter), run_time=dt / B)
        if redo:
            # sequential path on a copy of the model's vectors, restored afterwards
            q_keep, l_keep, u_keep = data.q, data.l[:m].copy(), data.u[:m].copy()
            # what solve() / update_vectors() overwrite: put back whatever happens (an instance with l > u raises)
            names = ('leaves', 'x', 'upper_glob', 'lower_glob', 'status', 'iter_num', 'osqp_iter', 'osqp_solve_time',
                     'solve_time', 'run_time', 'first_run', 'osqp_iter_avg', 'defer_lower')
            keep = {a: getattr(work, a) for a in names if hasattr(work, a)}
            try:
                for k in redo:
                    inst = instances[k]
                    self.update_vectors(q=Q[k].copy(), l=L[k, :m].copy(), u=U[k, :m].copy())
                    if inst.get('x0') is not None:
                        self.set_x0(np.asarray(inst['x0'], dtype=float).copy())
                    res = self.solve()
                    out[k] = dict(x=np.array(res.x, dtype=float), upper_glob=res.upper_glob, status=res.status,
                                  nodes=work.iter_num - 1, osqp_iter=work.osqp_iter, run_time=res.run_time)
            finally:
                self.update_vectors(q=q_keep, l=l_keep, u=u_keep)
                for a, v in keep.items():
                    setattr(work, a, v)
        return out

    def update_vectors(self, q=None, l=None, u=None):
        # solver.py:174-205: same factorisation, new root, statistics reset
        work = self.work
        work.data.update_vectors(q, l, u)
        if q is not None:
            work.solver.update(q=q)
        work.push_root()
        work.leaves = [work._make_root()]
        work._reset_counters()
        work.solve_time = 0.
        work.run_time = 0.
        work.x = np.empty(work.data.n)
        work.upper_glob = np.inf

    def set_x0(self, x0):
        self.work.set_x0(x0)
