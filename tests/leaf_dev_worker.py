"""Worker of tests/test_gpu_parity.py::test_leaves_leave_and_enter_the_slot_store_as_device_tensors (a process of its own:
torch's bundled HIP runtime has to open the device before the library's does)."""
import os
import sys

import numpy as np
import torch

torch.cuda.init()
torch.zeros(1, device="cuda")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from miosqp_amd import problems  # noqa: E402

form = sys.argv[1]
from miosqp_amd import bnb, qp, search, stream
# (the stream forms push every open leaf into a free column at once: only a tree wider than the 64 columns + the ready
#  ring's margin leaves leaves to give)
n, m, p, seed = (40, 60, 20, 7) if form == "hosted" else (70, 140, 50, 3)
pr = problems.random_miqp(n, m, p, seed=seed)
st = dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 6, device_tree=False)
M = m + p

def make():
    mdl = bnb.MIOSQP()
    mdl.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st),
              dict(problems.QP_SETTINGS, max_batch=64))
    if form == "hosted":
        return mdl, search.HostedSearch(mdl)
    return mdl, (stream.NativeStreamSearch if form == "native_stream" else stream.StreamSearch)(mdl, columns=64, capacity=8192)

ref_model, ref = make()
r0 = ref.run()
model, hs = make()
for _ in range(40 if form == "hosted" else 3000):
    if (hs.step(1) if form == "hosted" else hs.step()) == 0 or hs.givable() >= 3:
        break
k = hs.givable()
assert k >= 1
recs, tails = [], []
for _ in range(k):
    t = torch.zeros(2 * p + n + M, dtype=torch.float64, device="cuda")
    out = hs.give_leaf(into=qp.leaf_record_views(t, p, n, M))
    recs.append(t.clone())
    tails.append((out[4], out[5]))
assert hs.givable() == 0
for t in recs:  # (bounds of integer rows: 0 / 1 fixings or the root's, never garbage)
    h = t.cpu().numpy()
    assert np.all(h[:p] <= h[p:2 * p]) and np.all(np.isfinite(h))
for t, (depth, lower) in zip(recs, tails):
    v = qp.leaf_record_views(t, p, n, M)
    hs.add_leaf(v[0], v[1], v[2], v[3], depth, lower)
assert hs.givable() == k
# the numpy path moves the same bytes
h0 = recs[0].cpu().numpy()
l_int, u_int, x0, y0, depth, lower = hs.give_leaf()
# (the shallowest leaf comes out first both times)
np.testing.assert_array_equal(np.concatenate([l_int, u_int, x0, y0]), h0)
hs.add_leaf(l_int, u_int, x0, y0, depth, lower)
r1 = hs.run()
assert r1.status == r0.status == bnb.MI_SOLVED
assert abs(r1.upper_glob - r0.upper_glob) <= 1e-9 * max(1.0, abs(r0.upper_glob))
np.testing.assert_array_equal(r1.x[pr["i_idx"]], r0.x[pr["i_idx"]])
print("leaf round trip ok", form, k)
