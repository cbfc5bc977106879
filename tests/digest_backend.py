"""Test double with the FULL surface of miosqp_amd.qp (solve_node, solve_batch, set_root, node
digest) built on the CPU oracle, so that the fused / batched / digest branches of the host layer
(miosqp_amd/bnb.py, miosqp_amd/dist.py) are exercised without a GPU.  The digest is computed with
the reference's own numpy expressions (workspace.py:232-272)."""
import types

import numpy as np

from oracle import oracle

constant = oracle.constant


class OSQP(oracle.OSQP):
    def setup(self, P=None, q=None, A=None, l=None, u=None, **kw):
        kw = {k: v for k, v in kw.items() if k not in ("max_batch", "fold", "resident", "device", "setup_on_device", "coop")}
        self._P, self._q, self._A = P.tocsc(), np.array(q, dtype=float), A.tocsc()
        self._root = None
        oracle.OSQP.setup(self, P, q, A, l, u, **kw)

    def update(self, q=None, l=None, u=None):
        if q is not None:
            self._q = np.array(q, dtype=float)
        oracle.OSQP.update(self, q=q, l=l, u=u)

    def set_integer_rows(self, i_idx, m_orig):
        self._ii, self._m = np.asarray(i_idx), int(m_orig)

    def set_root(self, l_root, u_root, eps_int_feas, eps_lin):
        self._root = (np.array(l_root, dtype=float), np.array(u_root, dtype=float), eps_int_feas, eps_lin)

    def solve_node(self, l, u, x0, y0):
        self.update(l=l, u=u)
        self.warm_start(x=x0, y=y0)
        r = self.solve()
        st, lower, digest = r.info.status_val, None, None
        x = r.x
        if st in (1, -2):
            k = len(self._ii)
            x[self._ii] = np.minimum(np.maximum(x[self._ii], np.asarray(l)[-k:]), np.asarray(u)[-k:])
            lower = .5 * np.dot(x, self._P.dot(x)) + np.dot(self._q, x)
            if self._root is not None:
                lr, ur, eps_int, eps_lin = self._root
                xi = x[self._ii]
                frac = abs(xi - np.round(xi))
                xr = np.copy(x)
                xr[self._ii] = np.round(xi)
                z = self._A.dot(xr)
                feas = not (np.any(z < lr - eps_lin) or np.any(z > ur + eps_lin))
                digest = types.SimpleNamespace(int_inf=int(np.sum(frac > eps_int)), nextvar=int(np.argmax(frac)),
                                               heur_feasible=feas,
                                               heur_obj=.5 * np.dot(xr, self._P.dot(xr)) + np.dot(self._q, xr))
        return types.SimpleNamespace(x=x, y=r.y, status_val=st, iter=r.info.iter, run_time=r.info.run_time,
                                     lower=lower, digest=digest)

    def solve_batch(self, l, u, x0, y0):
        rs = [self.solve_node(l[k], u[k], x0[k], y0[k]) for k in range(len(l))]
        return types.SimpleNamespace(
            x=np.stack([r.x for r in rs]), y=np.stack([r.y for r in rs]),
            status_val=np.array([r.status_val for r in rs]), iter=np.array([r.iter for r in rs]),
            lower=np.array([np.nan if r.lower is None else r.lower for r in rs]),
            run_time=np.array([r.run_time for r in rs]), digest=[r.digest for r in rs])

    # ---- CPU emulation of the leaf pool / streaming batch (miosqp_qp_pool_*): same call surface and semantics as
    #      miosqp_amd.qp -- slots, ready ring with drop-at-refill, at most 64 refills and harvests per chunk, nodes
    #      that take a node-dependent number of chunks, children written at harvest, digests handed out on collect,
    #      one launch possibly still in flight -- with the oracle doing the relaxations.  Test infrastructure only.
    POOL_PRUNED = -100

    def pool_create(self, capacity, columns):
        self._pc = dict(cap=int(capacity), cols=int(columns))
        self.pool_reset()

    def pool_reset(self):
        pc = self._pc
        pc.update(lo={}, hi={}, ws={}, sol={}, ring=[], upper=np.inf, cols_state=[None] * pc["cols"], launched=[],
                  chunk=0)

    def pool_write_node(self, slot, l_int, u_int, x0, y0):
        pc = self._pc
        pc["lo"][slot], pc["hi"][slot] = np.array(l_int, dtype=float), np.array(u_int, dtype=float)
        pc["sol"][slot] = (np.array(x0, dtype=float), np.array(y0, dtype=float))
        pc["ws"][slot] = slot

    def pool_read_node(self, slot, n_int, want=("l", "u", "x", "y")):
        pc = self._pc
        vals = dict(l=pc["lo"].get(slot), u=pc["hi"].get(slot), x=pc["sol"].get(slot, (None, None))[0],
                    y=pc["sol"].get(slot, (None, None))[1])
        return types.SimpleNamespace(**{k: np.array(vals[k]) for k in want})

    def pool_push(self, slot, child0, child1, lower):
        for s, c0, c1, lo in zip(slot, child0, child1, lower):
            self._pc["ring"].append((int(s), int(c0), int(c1), float(lo)))

    def pool_set_upper(self, upper):
        self._pc["upper"] = float(upper)

    def _pool_chunk(self):
        """One chunk: refill (<= 64, dropping entries whose bound exceeds the incumbent), one chunk of progress for
        every column, harvest (<= 64) of the columns whose node is decided."""
        pc = self._pc
        out = []
        refills = 0
        for b in range(pc["cols"]):
            while pc["cols_state"][b] is None and pc["ring"] and refills < 64:
                s, c0, c1, lo = pc["ring"].pop(0)
                if lo > pc["upper"]:
                    out.append(dict(slot=s, status_val=self.POOL_PRUNED, iter=0, int_inf=-1, nextvar=-1, lower=lo,
                                    heur_viol=np.nan, heur_obj=np.nan))
                    continue
                ws = pc["ws"][s]
                l = np.concatenate([self._root[0][:self._m], pc["lo"][s]])
                u = np.concatenate([self._root[1][:self._m], pc["hi"][s]])
                r = self.solve_node(l, u, pc["sol"][ws][0].copy(), pc["sol"][ws][1].copy())
                chunks = max(1, -(-int(r.iter) // 25))  # the node occupies its column for that many chunks
                pc["cols_state"][b] = [s, c0, c1, r, chunks]
                refills += 1
        harvested = 0
        for b in range(pc["cols"]):
            stt = pc["cols_state"][b]
            if stt is None:
                continue
            stt[4] -= 1
            if stt[4] > 0 or harvested >= 64:
                stt[4] = max(stt[4], 0)
                continue
            s, c0, c1, r, _ = stt
            harvested += 1
            pc["cols_state"][b] = None
            ok = r.status_val in (1, -2)
            pc["sol"][s] = (np.array(r.x), np.array(r.y))
            dg = r.digest
            if ok and dg is not None and dg.int_inf > 0:
                xv = r.x[self._ii[dg.nextvar]]
                for c, side in ((c0, 0), (c1, 1)):
                    if c < 0:
                        continue
                    lo_c, hi_c = pc["lo"][s].copy(), pc["hi"][s].copy()
                    if side == 0:
                        hi_c[dg.nextvar] = np.floor(xv)
                    else:
                        lo_c[dg.nextvar] = np.ceil(xv)
                    pc["lo"][c], pc["hi"][c], pc["ws"][c] = lo_c, hi_c, s
            out.append(dict(slot=s, status_val=r.status_val, iter=r.iter, int_inf=dg.int_inf if ok else -1,
                            nextvar=dg.nextvar if ok else -1, lower=r.lower if ok else np.nan,
                            heur_viol=(-1.0 if dg.heur_feasible else 1.0) if ok else np.nan,
                            heur_obj=dg.heur_obj if ok else np.nan))
        return out

    def pool_launch(self, chunks=1):
        # executed eagerly; the digests stay "in flight" until a collect asks for them
        self._pc["launched"].append([d for _ in range(chunks) for d in self._pool_chunk()])

    def pool_collect(self, keep_in_flight=0):
        pc = self._pc
        got = []
        while len(pc["launched"]) > keep_in_flight:
            got.extend(pc["launched"].pop(0))
        dt = np.dtype([("slot", "i4"), ("status_val", "i4"), ("iter", "i4"), ("int_inf", "i4"), ("nextvar", "i4"),
                       ("reserved", "i4"), ("lower", "f8"), ("heur_viol", "f8"), ("heur_obj", "f8"), ("pri_res", "f8"),
                       ("dua_res", "f8")])
        arr = np.zeros(len(got), dtype=dt)
        for k, g in enumerate(got):
            for name in ("slot", "status_val", "iter", "int_inf", "nextvar", "lower", "heur_viol", "heur_obj"):
                arr[k][name] = g[name]
        active = sum(1 for c in pc["cols_state"] if c is not None)
        return arr, active, len(pc["ring"])

    # ---- CPU emulation of the hosted node-at-a-time search (miosqp_qp_search_*, csrc/host_search.inc): the same
    #      list semantics (creation order, first maximum, prune traversal) on solve_node above.  Test infrastructure.
    def search_create(self, capacity):
        self._sc = dict(cap=int(capacity))
        self.search_reset()

    def search_reset(self):
        self._sc.update(open=[], upper=np.inf, inc=None, used=0)

    def search_add_leaf(self, l_int, u_int, x0, y0, depth, lower):
        if np.any(np.asarray(l_int) > np.asarray(u_int)):
            raise ValueError("Lower bound must be lower than or equal to upper bound")
        self._sc["open"].append(dict(l=np.array(l_int, dtype=float), u=np.array(u_int, dtype=float),
                                     x=np.array(x0, dtype=float), y=np.array(y0, dtype=float), depth=int(depth),
                                     lower=float(lower)))

    def search_take_leaf(self, n_int):
        op = self._sc["open"]
        if not op:
            return None
        k = int(np.argmin([lf["depth"] for lf in op]))
        lf = op.pop(k)
        return lf["l"], lf["u"], lf["x"], lf["y"], lf["depth"], lf["lower"]

    def _search_prune(self):
        sc, k = self._sc, 0
        while k < len(sc["open"]):
            if sc["open"][k]["lower"] > sc["upper"]:
                del sc["open"][k]
            k += 1

    def search_set_incumbent(self, upper, x):
        sc = self._sc
        if x is None:  # only the value of the incumbent held (a heuristic value recomputed by the caller)
            assert sc["inc"] is not None
            sc["upper"] = float(upper)
            return
        if upper < sc["upper"]:
            sc["upper"], sc["inc"] = float(upper), np.array(x, dtype=float)
            self._search_prune()

    def search_get_incumbent(self):
        sc = self._sc
        return (sc["upper"], None if sc["inc"] is None else sc["inc"].copy())

    def search_run(self, tree_explor_rule, max_nodes, budget_s=0.0):
        import time
        sc, t0 = self._sc, time.perf_counter()
        lr, ur = self._root[0], self._root[1]
        m, done, iters, improved = self._m, 0, 0, 0
        while sc["open"] and done < max_nodes:
            if budget_s > 0 and done > 0 and time.perf_counter() - t0 >= budget_s:
                break
            op = sc["open"]
            if tree_explor_rule == 0 or not np.isfinite(sc["upper"]):
                k = int(np.argmax([lf["depth"] for lf in op]))
            else:
                k = int(np.argmax([lf["lower"] for lf in op]))
            lf = op.pop(k)
            l, u = lr.copy(), ur.copy()
            l[m:], u[m:] = lf["l"], lf["u"]
            r = self.solve_node(l, u, lf["x"], lf["y"])
            done += 1
            iters += int(r.iter)
            if r.status_val not in (1, -2):
                continue
            lower = float(r.lower)
            if lower > sc["upper"]:
                continue
            dg = r.digest
            if dg.int_inf == 0:
                sc["upper"], sc["inc"], improved = lower, r.x.copy(), 1
                self._search_prune()
                continue
            if dg.heur_feasible and dg.heur_obj < sc["upper"]:
                xr = r.x.copy()
                xr[self._ii] = np.round(xr[self._ii])
                sc["upper"], sc["inc"], improved = float(dg.heur_obj), xr, 2
                self._search_prune()
            xv = r.x[self._ii[dg.nextvar]]
            for side in (0, 1):
                cl, cu = lf["l"].copy(), lf["u"].copy()
                if side == 0:
                    cu[dg.nextvar] = np.floor(xv)
                else:
                    cl[dg.nextvar] = np.ceil(xv)
                op.append(dict(l=cl, u=cu, x=r.x, y=r.y, depth=lf["depth"] + 1, lower=lower))
        lg = min([lf["lower"] for lf in sc["open"]]) if sc["open"] else sc["upper"]
        return types.SimpleNamespace(nodes=done, osqp_iter=iters, open_leaves=len(sc["open"]),
                                     free_slots=sc["cap"] - len(sc["open"]), improved=improved, upper_glob=sc["upper"],
                                     lower_glob=lg, device_time=0.0, run_time=time.perf_counter() - t0)
