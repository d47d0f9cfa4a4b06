"""GPU fuzz: random shapes / degrees / flags through cleora_propagate_dev against the oracle.
Every row is bit-exact (reference-order sums, hub rows through hub_inorder_kernel, exact-order norm); CLEORA_F_HUB_SEGMENTS (drawn in a
quarter of the cases with hub rows) and FASTNORM carry tolerances."""
import numpy as np
import pytest

import oracle
from cleora_amd import _hip

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 2500))
    n_cols = n if rng.random() < 0.7 else int(rng.integers(1, 3000))
    d = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 31, 32, 33, 60, 64, 100, 128, 129, 200, 256, 260, 300, 512, 520]))
    mode = rng.integers(0, 3)
    if mode == 0:
        deg = rng.poisson(rng.uniform(0.3, 30), n)
    elif mode == 1:
        deg = np.minimum((rng.pareto(1.2, n) * 3).astype(np.int64), 6000)
    else:
        deg = rng.integers(0, 130, n)
    deg = deg.astype(np.int64)
    rowptr = np.zeros(n + 1, np.uint64)
    rowptr[1:] = np.cumsum(deg).astype(np.uint64)
    nnz = int(rowptr[-1])
    col = rng.integers(0, n_cols, nnz).astype(np.uint32)
    val = (rng.standard_normal(nnz) * rng.uniform(0.01, 3)).astype(np.float32)
    x = (rng.standard_normal((n_cols, d)) * rng.uniform(0.1, 10)).astype(np.float32)
    thr = int(rng.choice([0, 0, 8, 64, 300]))
    seg = int(rng.choice([0, 4, 32, 256])) if thr else 0
    return n, n_cols, d, rowptr, col, val, x, thr, seg, deg, rng


@pytest.mark.parametrize("seed", range(60))
def test_random_case(seed):
    n, n_cols, d, rowptr, col, val, x, thr, seg, deg, rng = _case(seed)
    L = _hip.lib()
    g = _hip.Graph.from_host(rowptr, col, val, None, n_cols=n_cols, hub_threshold=thr, hub_segment=seg)
    eff_thr = g.info().hub_threshold
    if seed % 3 == 0:
        g.set_hub_inorder_min(0)               # every long row on the in-order hub launch instead of first in the main launch
    if seed % 5 == 0:
        g.set_hub_lanes(2)
    square = n == n_cols
    flags = 0
    if rng.random() < 0.7:
        flags |= _hip.F_L2NORM
    fast = bool(flags & _hip.F_L2NORM) and rng.random() < 0.3
    if fast:
        flags |= _hip.F_FASTNORM
    rw = 0.0
    xs = x[:n] if square else (rng.standard_normal((n, d)).astype(np.float32))
    if rng.random() < 0.4:
        flags |= _hip.F_RESIDUAL
        rw = float(rng.uniform(0.05, 0.95))
    want_sq = rng.random() < 0.4
    if want_sq:
        flags |= _hip.F_SQDIFF
    segmented = bool((deg > eff_thr).any()) and seed % 4 == 3      # the opt-in segment sum of the hub rows
    if segmented:
        flags |= _hip.F_HUB_SEGMENTS
    dx, dxs = _hip.DevArray.from_host(x), _hip.DevArray.from_host(np.ascontiguousarray(xs))
    dy = _hip.DevArray((n, d), np.float32)
    L.cleora_memset(dy.ptr, 0xFF, dy.nbytes, None)
    dsq = _hip.DevArray((n,), np.float64)
    _hip.check(L.cleora_propagate_dev(g.handle, 0, dx.ptr, d, d, dy.ptr, d, flags, rw, dxs.ptr,
                                      dsq.ptr if want_sq else None, None, None))
    _hip.check(L.cleora_stream_sync(None))
    got = dy.to_host()
    y = oracle.spmm(rowptr, col, val, x)
    if flags & _hip.F_RESIDUAL:
        y = (np.float32(1.0) - np.float32(rw)) * y + np.float32(rw) * xs
    want = oracle.l2_normalize(y) if flags & _hip.F_L2NORM else y
    split = (deg > eff_thr) if segmented else np.zeros(n, bool)
    scale = np.abs(want).max() + 1e-30
    if not fast:
        np.testing.assert_array_equal(got[~split], want[~split])
    else:
        np.testing.assert_allclose(got[~split], want[~split], rtol=0, atol=4e-7 * max(scale, 1.0))
    if split.any():
        bound = oracle.spmm(rowptr, col, np.abs(val), np.abs(x))[split].max() + 1e-30
        tol = (3e-6 * bound) if not (flags & _hip.F_L2NORM) else 3e-6
        assert np.abs(got[split] - want[split]).max() <= tol
    if want_sq:
        delta = (got.astype(np.float64) - xs.astype(np.float64))
        np.testing.assert_allclose(dsq.to_host(), (delta * delta).sum(axis=1), rtol=1e-6, atol=1e-12)
