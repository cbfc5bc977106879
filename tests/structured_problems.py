"""Structured (non-random-dense) MIQP instances for the parity tests of the register-resident solvers.

Every GPU parity instance of n + M in 193 .. 2048 used to be `problems.random_miqp` (positive definite P = Pt Pt',
finite two-sided rows, density >= 0.7).  These generators produce what a relaxation engine meets in practice and the
reference's own examples contain (/root/reference/examples/power_converter/quadratic_program.py:11-136: l = -inf rows,
:98; an almost singular P): no quadratic term, a rank-deficient one, equality rows, one-sided rows, sparse A with empty
rows, badly scaled rows, and the power converter's block structure at longer horizons.  Same dict layout as
`problems.random_miqp`.
"""
import os

import numpy as np
import scipy.sparse as spa

HERE = os.path.dirname(os.path.abspath(__file__))


def _base(n, m, p, density, rng, pd_rank=None):
    i_idx = rng.choice(np.arange(n), p, replace=False)
    r = n if pd_rank is None else pd_rank
    if r > 0:
        Pt = spa.random(n, r, density=density, random_state=rng)
        P = spa.csc_matrix(Pt.dot(Pt.T))
    else:
        P = spa.csc_matrix((n, n))
    q = rng.randn(n)
    A = spa.csc_matrix(spa.random(m, n, density=density, random_state=rng))
    u = 2 + rng.rand(m)
    l = -2 + rng.rand(m)
    return dict(P=P, q=q, A=A, l=l, u=u, i_idx=i_idx, i_l=np.zeros(p), i_u=np.ones(p))


def milp_relaxation(n, m, p, density=0.5, seed=0):
    """P = 0: the LP relaxation of a MILP (the KKT matrix's (1,1) block is sigma I alone)."""
    return _base(n, m, p, density, np.random.RandomState(seed), pd_rank=0)


def low_rank_quadratic(n, m, p, density=0.5, seed=0):
    """P = Pt Pt' with Pt n x n/4: three quarters of P's eigenvalues are zero."""
    return _base(n, m, p, density, np.random.RandomState(seed), pd_rank=max(1, n // 4))


def equality_rows(n, m, p, density=0.5, seed=0, frac=0.1):
    """l = u on `frac` of the general rows (at a point the other rows admit)."""
    rng = np.random.RandomState(seed)
    pr = _base(n, m, p, density, rng)
    x_in = 0.1 * rng.rand(n)
    ax = pr["A"].dot(x_in)
    rows = rng.choice(m, max(1, int(frac * m)), replace=False)
    pr["l"][rows] = ax[rows]
    pr["u"][rows] = ax[rows]
    return pr


def one_sided_rows(n, m, p, density=0.5, seed=0, frac=0.3):
    """-inf / +inf on one side of `frac` of the general rows, both sides on a few (free rows)."""
    rng = np.random.RandomState(seed)
    pr = _base(n, m, p, density, rng)
    rows = rng.choice(m, max(2, int(frac * m)), replace=False)
    h = len(rows) // 2
    pr["l"][rows[:h]] = -np.inf
    pr["u"][rows[h:]] = np.inf
    free = rows[:max(1, len(rows) // 10)]
    pr["u"][free] = np.inf
    return pr


def sparse_rows(n, m, p, density=0.01, seed=0):
    """A at 1 % / 5 % density: a handful of entries per row, some rows and columns empty."""
    return _base(n, m, p, density, np.random.RandomState(seed))


def badly_scaled_rows(n, m, p, density=0.5, seed=0):
    """row norms of A spread over 1e-4 .. 1e4 (bounds scaled with their rows)."""
    rng = np.random.RandomState(seed)
    pr = _base(n, m, p, density, rng)
    s = 10.0 ** rng.uniform(-4, 4, m)
    pr["A"] = spa.csc_matrix(spa.diags(s).dot(pr["A"]))
    pr["l"] = pr["l"] * s
    pr["u"] = pr["u"] * s
    return pr


def few_rows_milp(n, m, p, density=0.5, seed=0):
    """P = 0 and fewer rows than variables: S = sigma I + rho A'A keeps n - (m + p) eigenvalues at sigma = 1e-6 -- the
    KKT matrix is as ill-conditioned as the frozen spec allows (what the set-up guard of the explicit inverse is for)."""
    assert m + p < n
    return _base(n, m, p, density, np.random.RandomState(seed), pd_rank=0)


def few_rows_tiny_quadratic(n, m, p, density=0.5, seed=0):
    """as few_rows_milp with P = 1e-6 diag(0.5 .. 1.5): bounded, and K just as ill-conditioned."""
    rng = np.random.RandomState(seed)
    pr = _base(n, m, p, density, rng, pd_rank=0)
    pr["P"] = spa.csc_matrix(spa.diags(1e-6 * (0.5 + rng.rand(n))))
    return pr


def power_converter_horizon(K, step=0):
    """The power converter's structure (tests/golden/power_converter_N3.npz: n = 18, all variables integer in
    [-1, 1], 27 rows with l = -inf, P with eigenvalues 1e-22 .. 4e-3) at K times the horizon: K copies of the N = 3
    blocks on the diagonal, consecutive copies chained by the same kind of row the blocks use internally
    (u_a - u_b <= t, l = -inf) and by a small rank-one term in P; q and u are the recorded vectors of K consecutive MPC
    steps starting at `step`.  n = 18 K, 27 K + 3 (K - 1) general rows."""
    z = np.load(os.path.join(HERE, "golden", "power_converter_N3.npz"), allow_pickle=False)
    P3 = spa.csc_matrix((z["P_data"], z["P_indices"], z["P_indptr"]), shape=tuple(z["P_shape"]))
    A3 = spa.csc_matrix((z["A_data"], z["A_indices"], z["A_indptr"]), shape=tuple(z["A_shape"]))
    n3, m3 = A3.shape[1], A3.shape[0]
    steps = [(step + k) % len(z["q"]) for k in range(K)]
    P = spa.block_diag([P3] * K, format="lil")
    for b in range(K - 1):  # chain the blocks: one rank-one coupling per seam, of the size of P's own entries
        v = np.zeros(n3 * K)
        v[n3 * b + n3 - 3:n3 * b + n3] = 1.0
        v[n3 * (b + 1):n3 * (b + 1) + 3] = -1.0
        P = P + spa.lil_matrix(1e-4 * np.outer(v, v))
    rows = []
    for b in range(K - 1):
        for c in range(3):
            r = np.zeros(n3 * K)
            r[n3 * b + n3 - 6 + c] = 1.0
            r[n3 * (b + 1) + c] = -1.0
            rows.append(r)
    A = spa.vstack([spa.block_diag([A3] * K)] + ([spa.csr_matrix(np.array(rows))] if rows else [])).tocsc()
    l = np.concatenate([np.tile(z["l"], K), np.full(len(rows), -np.inf)])
    u = np.concatenate([np.concatenate([z["u"][s] for s in steps]), np.full(len(rows), 1.0)])
    q = np.concatenate([z["q"][s] for s in steps])
    i_idx = np.arange(n3 * K)
    return dict(P=spa.csc_matrix(P), q=q, A=A, l=l, u=u, i_idx=i_idx, i_l=np.tile(z["i_l"], K), i_u=np.tile(z["i_u"], K))


# name -> (generator, kwargs); all with 193 <= n + M <= 2048 (the cooperative range)
CASES = {
    "milp": (milp_relaxation, dict(n=300, m=500, p=120, density=0.4, seed=11)),
    "low_rank_P": (low_rank_quadratic, dict(n=400, m=600, p=150, density=0.5, seed=12)),
    "equality_rows": (equality_rows, dict(n=300, m=500, p=100, density=0.5, seed=13)),
    "one_sided_rows": (one_sided_rows, dict(n=350, m=700, p=150, density=0.5, seed=14)),
    "A_1pct": (sparse_rows, dict(n=500, m=800, p=200, density=0.01, seed=15)),
    "A_5pct": (sparse_rows, dict(n=450, m=900, p=150, density=0.05, seed=16)),
    "badly_scaled": (badly_scaled_rows, dict(n=300, m=500, p=100, density=0.5, seed=17)),
    "few_rows_milp": (few_rows_milp, dict(n=400, m=80, p=100, density=0.5, seed=18)),
    "few_rows_tinyP": (few_rows_tiny_quadratic, dict(n=400, m=80, p=100, density=0.5, seed=19)),
    "power_converter_K10": (power_converter_horizon, dict(K=10)),
    "power_converter_K20": (power_converter_horizon, dict(K=20, step=7)),
}


def make(name):
    fn, kw = CASES[name]
    return fn(**kw)
