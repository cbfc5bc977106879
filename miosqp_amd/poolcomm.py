"""The communicator of several streaming pools on ONE GPU (stream.MultiPoolSearch): "ranks" that are host
threads of one process.  Same interface as dist.TorchComm, so dist.ShardedSearch / dist.ShardedStream run on it
unchanged: while one pool is in the serial part of its chunk -- termination test, harvest, refill: small kernels --
the sweeps of another fill the chip.  The CPU suite and tools/sim_scaling.py also drive the sharded searches through it
at world sizes up to 8 (no process group needed).

An exchange round's deposits are dropped once every rank has read them (long-running searches post millions of
rounds), and a rank that fails wakes everybody who is waiting for it at once (`PoolWorld.fail`)."""
import threading

import numpy as np


class PoolWorld(object):
    def __init__(self, world):
        self.world = world
        self.bar = threading.Barrier(world, timeout=300)
        self.slots = [None] * world
        self.rounds = {}   # exchange number -> per-rank deposits of post()
        self.readers = {}  # exchange number -> ranks that have not read it yet
        self.cv = threading.Condition()
        self.failed = None

    def fail(self, exc):
        """Called by the owner of the threads when one of them raised: peers blocked in a barrier or in complete()
        raise at once instead of sitting out the timeout."""
        with self.cv:
            if self.failed is None:
                self.failed = exc
            self.cv.notify_all()
        try:
            self.bar.abort()
        except Exception:  # noqa: BLE001
            pass


class PoolComm(object):
    def __init__(self, tw, rank):
        self.tw, self.rank, self.world = tw, rank, tw.world
        self._counts = None
        self._posted = 0

    def _all(self, obj):
        tw = self.tw
        tw.slots[self.rank] = obj
        tw.bar.wait()
        out = list(tw.slots)
        tw.bar.wait()
        return out

    extra = (0.0, 0.0)
    aux = None  # per rank: the optional third value of `extra` at the last completed exchange

    def gather(self, vec):
        return np.array(self._all(np.asarray(vec, dtype=np.float64)))

    def post(self, value, x, nleaves, extra=(0.0, 0.0)):
        """Non-blocking half: deposit this rank's entry of exchange number k."""
        tw = self.tw
        k = self._posted
        self._posted += 1
        with tw.cv:
            if k not in tw.rounds:
                tw.rounds[k] = [None] * self.world
                tw.readers[k] = self.world
            tw.rounds[k][self.rank] = (value, nleaves, None if x is None else np.array(x), tuple(extra))
            tw.cv.notify_all()
        return k

    def complete(self, k, have=None):
        tw = self.tw
        with tw.cv:
            ok = tw.cv.wait_for(lambda: tw.failed is not None or all(e is not None for e in tw.rounds.get(k, [None])),
                                timeout=300)
            if tw.failed is not None:
                raise RuntimeError("a peer of this in-process world failed: %r" % (tw.failed,))
            assert ok, "exchange %d never completed" % k
            tab = list(tw.rounds[k])
            tw.readers[k] -= 1
            if tw.readers[k] == 0:  # the last reader drops the round
                del tw.rounds[k]
                del tw.readers[k]
        return self._decide(tab, have)

    def exchange(self, value, x, nleaves, have=None, extra=(0.0, 0.0)):
        return self.complete(self.post(value, x, nleaves, extra), have)

    def _decide(self, tab, have):
        self._counts = [int(t[1]) for t in tab]
        self.extra = (float(sum(t[3][0] for t in tab)), float(sum(t[3][1] for t in tab)))
        self.aux = [float(t[3][2]) if len(t[3]) > 2 else 0.0 for t in tab]
        vals = np.array([t[0] for t in tab])
        owner = int(np.argmin(vals))
        best = float(vals[owner])
        total = sum(self._counts)
        prev = float(np.max(vals)) if have is None else have
        if not np.isfinite(best) or not best < prev:
            return best, owner, None, total
        return best, owner, np.array(tab[owner][2]), total

    def incumbent(self, value, x):
        best, owner, xb, _ = self.exchange(value, x, 0)
        return best, owner, (x if xb is None else xb)

    def leaf_counts(self):
        return list(self._counts)

    def move(self, arr, size, src):
        tab = self._all(None if arr is None else np.array(arr))
        return np.array(tab[src])

    def sum(self, arr):
        return np.sum(self._all(np.asarray(arr, dtype=np.float64)), axis=0)

    def barrier(self):
        self.tw.bar.wait()
