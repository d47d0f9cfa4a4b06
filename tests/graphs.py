"""Seeded synthetic CSR graphs for the parity tests (numpy only)."""
import numpy as np


def random_csr(n, avg_deg, seed, n_cols=None, hubs=(), empty_frac=0.0, sort_cols=True):
    """Random row-stochastic-ish CSR.  hubs: list of (row, degree) forced long rows."""
    rng = np.random.default_rng(seed)
    n_cols = n if n_cols is None else n_cols
    deg = rng.poisson(avg_deg, n).astype(np.int64)
    deg[rng.random(n) < empty_frac] = 0
    for r, dg in hubs:
        deg[r] = dg
    rowptr = np.zeros(n + 1, dtype=np.uint64)
    rowptr[1:] = np.cumsum(deg).astype(np.uint64)
    nnz = int(rowptr[-1])
    col = rng.integers(0, n_cols, nnz, dtype=np.int64).astype(np.uint32)
    if sort_cols:  # the reference stores each row sorted by column (sparse_matrix_builder.rs:292)
        rows = np.repeat(np.arange(n), deg)
        order = np.lexsort((col, rows))
        col = col[order]
    val = rng.random(nnz, dtype=np.float32) + np.float32(0.05)
    rs = np.add.reduceat(val, np.minimum(rowptr[:-1].astype(np.int64), max(nnz - 1, 0))) if nnz else np.zeros(n)
    rs = np.where(deg > 0, rs, 1.0).astype(np.float32)
    val_left = (val / np.repeat(rs, deg)).astype(np.float32)
    val_sym = (val / np.sqrt(np.repeat(rs, deg) * (1.0 + rng.random(nnz, dtype=np.float32)))).astype(np.float32)
    return rowptr, col, val_left, val_sym
