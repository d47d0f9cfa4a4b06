"""GPU parity: SpMM + fused epilogue through the C ABI vs the CPU oracle.

Bit-exact (assert_array_equal) for every row: the kernels consume each row's edges in stored order with
separate f32 multiply/add exactly as src/embedding.rs:80-82 — hub rows included (hub_inorder_kernel).  Only with
CLEORA_F_HUB_SEGMENTS are hub rows summed in segments; those are compared with a tolerance of 2e-6 * sum|terms|.
"""
import ctypes

import numpy as np
import pytest

import oracle
from cleora_amd import _hip
from tests.graphs import random_csr

pytestmark = pytest.mark.gpu

L = None


def setup_module(module):
    global L
    L = _hip.lib()
    assert _hip.device_count() >= 1, "no GPU visible"


def run_dev(g, kind, x, flags=0, rw=0.0, x_self=None, want_sqdiff=False):
    n_rows = g.info().n_rows
    d = x.shape[1]
    dx = _hip.DevArray.from_host(x)
    dy = _hip.DevArray((n_rows, d), np.float32)
    L.cleora_memset(dy.ptr, 0xFF, dy.nbytes, None)  # poison: every row must be written
    dxs = _hip.DevArray.from_host(x_self) if x_self is not None else None
    dsq = _hip.DevArray((n_rows,), np.float64) if want_sqdiff else None
    _hip.check(L.cleora_propagate_dev(g.handle, kind, dx.ptr, d, d, dy.ptr, d, flags, rw,
                                      dxs.ptr if dxs else None, dsq.ptr if dsq else None, None, None))
    _hip.check(L.cleora_stream_sync(None))
    y = dy.to_host()
    return (y, dsq.to_host()) if want_sqdiff else y


@pytest.mark.parametrize("d", [256, 128, 32, 8, 64, 512, 1024, 2048, 100, 4, 20, 260])
def test_spmm_bit_exact_vs_oracle(d):
    n = 3000 if d <= 512 else 600
    rowptr, col, vl, vs = random_csr(n, 12, seed=d, empty_frac=0.05)
    x = np.random.default_rng(d + 1).standard_normal((n, d)).astype(np.float32)
    g = _hip.Graph.from_host(rowptr, col, vl, vs)
    for kind, val in ((_hip.LEFT, vl), (_hip.SYMMETRIC, vs)):
        want = oracle.spmm(rowptr, col, val, x)
        got = run_dev(g, kind, x)
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("d", [3, 7, 33, 65, 129, 250, 1000])
def test_spmm_scalar_path_bit_exact(d):
    n = 1200
    rowptr, col, vl, _ = random_csr(n, 9, seed=100 + d, empty_frac=0.1)
    x = np.random.default_rng(d).standard_normal((n, d)).astype(np.float32)
    g = _hip.Graph.from_host(rowptr, col, vl)
    np.testing.assert_array_equal(run_dev(g, _hip.LEFT, x), oracle.spmm(rowptr, col, vl, x))


@pytest.mark.parametrize("d", [256, 64, 1024, 48])
def test_fused_l2_bit_exact(d):
    n = 2500
    rowptr, col, vl, _ = random_csr(n, 15, seed=7 * d, empty_frac=0.02)
    x = np.random.default_rng(d).standard_normal((n, d)).astype(np.float32)
    g = _hip.Graph.from_host(rowptr, col, vl)
    want = oracle.l2_normalize(oracle.spmm(rowptr, col, vl, x))
    got = run_dev(g, _hip.LEFT, x, flags=_hip.F_L2NORM)
    np.testing.assert_array_equal(got, want)
    fast = run_dev(g, _hip.LEFT, x, flags=_hip.F_L2NORM | _hip.F_FASTNORM)
    np.testing.assert_allclose(fast, want, rtol=0, atol=3e-7)


def test_residual_and_sqdiff():
    n, d = 2000, 256
    rowptr, col, vl, _ = random_csr(n, 10, seed=5)
    x = np.random.default_rng(6).standard_normal((n, d)).astype(np.float32)
    g = _hip.Graph.from_host(rowptr, col, vl)
    rw = np.float32(0.3)
    y = oracle.spmm(rowptr, col, vl, x)
    blended = (np.float32(1.0) - rw) * y + rw * x      # f32: alpha*dst + rw*src
    want = oracle.l2_normalize(blended)
    got, sq = run_dev(g, _hip.LEFT, x, flags=_hip.F_L2NORM | _hip.F_RESIDUAL | _hip.F_SQDIFF,
                      rw=float(rw), want_sqdiff=True)
    np.testing.assert_array_equal(got, want)
    delta = (want - x).astype(np.float64)
    np.testing.assert_allclose(sq, (delta * delta).sum(axis=1), rtol=1e-12)
    # rw outside (0,1) disables the blend (src/embedding.rs:116)
    for bad in (0.0, 1.0, 1.5):
        got = run_dev(g, _hip.LEFT, x, flags=_hip.F_L2NORM | _hip.F_RESIDUAL, rw=bad)
        np.testing.assert_array_equal(got, oracle.l2_normalize(y))


@pytest.mark.parametrize("d", [256, 64, 1024, 100, 30, 7, 1280, 4096])
def test_hub_rows_in_reference_order(d):
    """Rows longer than hub_threshold are added edge by edge like every other row (src/embedding.rs:76-83): bit-equal to the
    oracle, for the 16-byte-aligned quad form (d % 4 == 0), the lane form (any d), partial last slabs, panels (d > 2048),
    row lengths around the chunk (192 / 64 edges) and step (4 edges) boundaries."""
    n = 4000
    hubs = [(17, 5000), (1234, 1025), (3999, 20000), (2000, 1024),            # 1024 = threshold: not a hub row
            (5, 1026), (6, 1027), (7, 1028), (8, 1152), (9, 1153), (10, 1151), (11, 1343), (12, 1344), (13, 1345), (14, 1088)]
    rowptr, col, vl, _ = random_csr(n, 8, seed=11, hubs=hubs)
    x = np.random.default_rng(12).standard_normal((n, d)).astype(np.float32)
    assert _hip.Graph.from_host(rowptr, col, vl).info().hub_threshold == 256           # the default
    g = _hip.Graph.from_host(rowptr, col, vl, hub_threshold=1024)
    info = g.info()
    assert info.n_hub_rows == len(hubs) - 1 and info.hub_threshold == 1024
    want = oracle.spmm(rowptr, col, vl, x)
    got = run_dev(g, _hip.LEFT, x)
    np.testing.assert_array_equal(got, want)
    # which long rows take the in-order hub launch and which are the first work items of the main launch is a scheduling choice
    # (default: by graph size, nnz / 8192 edges): any split gives the same bits
    assert info.hub_inorder_min == 1024 and info.n_inorder_rows == len(hubs) - 1        # a small graph: every long row on the hub launch
    for min_edges in (0, 1100, 8192):
        g.set_hub_inorder_min(min_edges)
        assert g.info().n_inorder_rows == sum(1 for _, deg in hubs if deg > max(min_edges, 1024))
        for lanes in (2, 4, 0):                  # both shapes of the hub launch (8 / 4 edges per load): the same bits
            g.set_hub_lanes(lanes)
            np.testing.assert_array_equal(run_dev(g, _hip.LEFT, x), want)
    g.set_hub_inorder_min(0)
    # with the epilogue (exact-order L2 norm over the whole row, residual blend, squared difference)
    got = run_dev(g, _hip.LEFT, x, flags=_hip.F_L2NORM)
    np.testing.assert_array_equal(got, oracle.l2_normalize(want))
    rw = np.float32(0.25)
    got = run_dev(g, _hip.LEFT, x, flags=_hip.F_L2NORM | _hip.F_RESIDUAL, rw=float(rw))
    np.testing.assert_array_equal(got, oracle.l2_normalize((np.float32(1.0) - rw) * want + rw * x))
    # every row a hub row, chunk tails of every length
    g2 = _hip.Graph.from_host(rowptr, col, vl, hub_threshold=4, hub_segment=16)
    np.testing.assert_array_equal(run_dev(g2, _hip.LEFT, x), want)
    g2.set_hub_lanes(2)
    np.testing.assert_array_equal(run_dev(g2, _hip.LEFT, x), want)
    np.testing.assert_array_equal(run_dev(g2, _hip.LEFT, x, flags=_hip.F_L2NORM), oracle.l2_normalize(want))
    g2.set_hub_inorder_min(0)                    # every row of more than 4 edges on the hub launch
    for lanes in (4, 2):
        g2.set_hub_lanes(lanes)
        np.testing.assert_array_equal(run_dev(g2, _hip.LEFT, x), want)


@pytest.mark.parametrize("d", [256, 64, 1024, 100, 1280])
def test_longest_rows_on_the_chain_kernel_keep_the_bits(d):
    """hub_chain_kernel (csrc/spmm.hip: seven producer waves gather and multiply, one consumer wave adds the products in stored
    order) for the longest rows of the hub launch: the same bits as the oracle whichever rows take it — every row of the hub launch
    (chain_min = 1: rows shorter than one 112-edge chunk, chunk tails of every length, rows of many trips), a mix with
    hub_inorder_kernel (both shapes), none — with the fused epilogue, and on a width with a partial last slab / unaligned rows
    (d = 100 keeps the 16-byte form, d % 4 == 0; the scalar path never takes the chain kernel)."""
    n = 3000
    hubs = [(17, 5000), (1234, 1025), (2999, 20011), (5, 1026), (6, 1137), (7, 1138), (8, 1120), (9, 1121), (10, 1119), (11, 1343), (12, 1344),
            (13, 896 * 3), (14, 896 * 3 + 1), (15, 896 * 3 - 1), (16, 1232)]
    rowptr, col, vl, _ = random_csr(n, 8, seed=21, hubs=hubs)
    x = np.random.default_rng(22).standard_normal((n, d)).astype(np.float32)
    g = _hip.Graph.from_host(rowptr, col, vl, hub_threshold=1024)
    assert g.info().n_inorder_rows == len(hubs)
    want = oracle.spmm(rowptr, col, vl, x)
    for chain_min in (1, 1300, 5000, 2 ** 64 - 1, 0):
        g.set_hub_chain_min(chain_min)
        for lanes in (4, 2):
            g.set_hub_lanes(lanes)
            np.testing.assert_array_equal(run_dev(g, _hip.LEFT, x), want, err_msg=f"chain_min {chain_min}, lanes {lanes}")
    g.set_hub_chain_min(1)
    np.testing.assert_array_equal(run_dev(g, _hip.LEFT, x, flags=_hip.F_L2NORM), oracle.l2_normalize(want))
    # every row of more than 4 edges on the hub launch and on the chain kernel: 8-edge rows, one-chunk rows
    g2 = _hip.Graph.from_host(rowptr, col, vl, hub_threshold=4, hub_segment=16)
    g2.set_hub_inorder_min(0)
    g2.set_hub_chain_min(1)
    np.testing.assert_array_equal(run_dev(g2, _hip.LEFT, x), want)
    g.close()
    g2.close()


def test_hub_rows_signed_zero_and_specials():
    """The last step of a hub row adds only its real edges (the padding lanes hold the NEXT row's edges), and NaN / inf /
    denormal / signed-zero terms travel through the DPP chain like through the reference's scalar loop."""
    n, d = 64, 64
    lens = [9, 10, 11, 12, 193, 194, 195, 196, 385]
    rowptr = np.zeros(n + 1, np.uint64)
    rowptr[1:len(lens) + 1] = np.cumsum(lens)
    rowptr[len(lens) + 1:] = rowptr[len(lens)]
    nnz = int(rowptr[-1])
    rng = np.random.default_rng(5)
    col = rng.integers(0, n, nnz).astype(np.uint32)
    vl = rng.standard_normal(nnz).astype(np.float32)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x[:, 0] = -0.0                      # every product is +-0: the sum keeps -0 only if nothing else is added
    vl_pos = np.abs(vl)
    x[:, 1] = np.float32(1e-41)         # denormal terms
    x[3, 2] = np.inf
    x[4, 3] = np.nan
    g = _hip.Graph.from_host(rowptr, col, vl_pos, hub_threshold=4, hub_segment=16)
    g.set_hub_inorder_min(0)                     # all nine rows on the hub launch (the default would keep the four short ones in the main launch)
    want = oracle.spmm(rowptr, col, vl_pos, x)
    g.set_hub_lanes(2)
    got2 = run_dev(g, _hip.LEFT, x)
    g.set_hub_lanes(4)
    got = run_dev(g, _hip.LEFT, x)
    np.testing.assert_array_equal(got.view(np.uint32), got2.view(np.uint32))
    g.set_hub_chain_min(1)                       # ... and through the chain kernel's producer / consumer waves
    got3 = run_dev(g, _hip.LEFT, x)
    g.set_hub_chain_min(0)
    nan = np.isnan(got)
    np.testing.assert_array_equal(got.view(np.uint32)[~nan], got3.view(np.uint32)[~nan])
    np.testing.assert_array_equal(nan, np.isnan(got3))
    np.testing.assert_array_equal(got.view(np.uint32)[:, :3], want.view(np.uint32)[:, :3])     # bit patterns: signs of zero, denormals, inf
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_array_equal(got[~np.isnan(want)], want[~np.isnan(want)])


@pytest.mark.parametrize("d", [256, 64, 1024])
def test_hub_rows_segmented(d):
    """CLEORA_F_HUB_SEGMENTS: hub rows as 256-edge segment sums added in a fixed order — same terms, another order."""
    n = 4000
    hubs = [(17, 5000), (1234, 1025), (3999, 20000), (2000, 1024)]  # 1024 = threshold: not split
    rowptr, col, vl, _ = random_csr(n, 8, seed=11, hubs=hubs)
    x = np.random.default_rng(12).standard_normal((n, d)).astype(np.float32)
    g = _hip.Graph.from_host(rowptr, col, vl, hub_threshold=1024)
    want = oracle.spmm(rowptr, col, vl, x)
    got = run_dev(g, _hip.LEFT, x, flags=_hip.F_HUB_SEGMENTS)
    hub_rows = np.array([17, 1234, 3999])
    mask = np.ones(n, bool)
    mask[hub_rows] = False
    np.testing.assert_array_equal(got[mask], want[mask])          # unsplit rows: bit exact
    # split rows: different summation order, same terms
    absx = np.abs(x)
    bound = oracle.spmm(rowptr, col, np.abs(vl), absx)[hub_rows]
    assert np.all(np.abs(got[hub_rows] - want[hub_rows]) <= 2e-6 * bound + 1e-30)
    # with the epilogue
    got = run_dev(g, _hip.LEFT, x, flags=_hip.F_L2NORM | _hip.F_HUB_SEGMENTS)
    np.testing.assert_allclose(got, oracle.l2_normalize(want), rtol=0, atol=2e-6)
    # custom thresholds: everything split at 16 edges/segment still agrees; the two forms can alternate on one handle
    g2 = _hip.Graph.from_host(rowptr, col, vl, hub_threshold=4, hub_segment=16)
    np.testing.assert_allclose(run_dev(g2, _hip.LEFT, x, flags=_hip.F_HUB_SEGMENTS), want, rtol=0, atol=1e-4 * np.abs(want).max())
    np.testing.assert_array_equal(run_dev(g2, _hip.LEFT, x), want)
    np.testing.assert_array_equal(run_dev(g, _hip.LEFT, x), want)


def test_row_shard_rectangular():
    # a row block of a bigger graph: n_rows < n_cols, x_self offset
    n, d = 3000, 128
    rowptr, col, vl, _ = random_csr(n, 10, seed=21)
    x = np.random.default_rng(22).standard_normal((n, d)).astype(np.float32)
    r0, r1 = 1000, 1800
    rp = (rowptr[r0:r1 + 1] - rowptr[r0]).astype(np.uint64)
    e0, e1 = int(rowptr[r0]), int(rowptr[r1])
    g = _hip.Graph.from_host(rp, col[e0:e1], vl[e0:e1], n_cols=n)
    want = oracle.l2_normalize(oracle.spmm(rowptr, col, vl, x))[r0:r1]
    got = run_dev(g, _hip.LEFT, x, flags=_hip.F_L2NORM)
    np.testing.assert_array_equal(got, want)


def test_host_entry_points_and_errors():
    n, d = 500, 32
    rowptr, col, vl, vs = random_csr(n, 6, seed=31)
    x = np.random.default_rng(32).standard_normal((n, d)).astype(np.float32)
    g = _hip.Graph.from_host(rowptr, col, vl, vs)
    y = np.empty((n, d), np.float32)
    _hip.check(L.cleora_propagate(g.handle, _hip.SYMMETRIC, _hip.ptr(x), d, _hip.ptr(y)))
    np.testing.assert_array_equal(y, oracle.spmm(rowptr, col, vs, x))
    _hip.check(L.cleora_l2_normalize(_hip.ptr(x), n, d, _hip.ptr(y)))
    np.testing.assert_array_equal(y, oracle.l2_normalize(x))
    with pytest.raises(ValueError):
        _hip.check(L.cleora_propagate(g.handle, 7, _hip.ptr(x), d, _hip.ptr(y)))
    g1 = _hip.Graph.from_host(rowptr, col, vl)  # no symmetric values
    with pytest.raises(ValueError):
        _hip.check(L.cleora_propagate(g1.handle, _hip.SYMMETRIC, _hip.ptr(x), d, _hip.ptr(y)))
    bad = col.copy()
    bad[3] = n + 5
    with pytest.raises(ValueError):
        _hip.Graph.from_host(rowptr, bad, vl)


def test_init_bit_exact():
    rng = np.random.default_rng(41)
    hashes = rng.integers(0, 1 << 63, 3000, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 3000, dtype=np.uint64)
    for d, seed in ((256, 0), (128, 42), (33, -7), (1, 1 << 40)):
        out = np.empty((hashes.shape[0], d), np.float32)
        _hip.check(L.cleora_init(_hip.ptr(hashes), hashes.shape[0], d, seed, _hip.ptr(out)))
        np.testing.assert_array_equal(out, oracle.init(hashes, d, seed))


def test_embed_loop_bit_exact():
    n, d = 2000, 64
    rowptr, col, vl, vs = random_csr(n, 10, seed=51)
    hashes = np.random.default_rng(52).integers(0, 1 << 62, n, dtype=np.uint64)
    g = _hip.Graph.from_host(rowptr, col, vl, vs)
    x0 = oracle.init(hashes, d, 3)
    for kind, val, rw in ((_hip.LEFT, vl, 0.0), (_hip.SYMMETRIC, vs, 0.25)):
        want, _ = oracle.embed(rowptr, col, val, x0, 7, residual_weight=rw)
        out = np.empty((n, d), np.float32)
        it = ctypes.c_uint64(0)
        _hip.check(L.cleora_embed(g.handle, _hip.ptr(hashes), None, kind, d, 7, 3, rw, 0.0, 0,
                                  _hip.ptr(out), ctypes.byref(it)))
        assert it.value == 7
        np.testing.assert_array_equal(out, want)
    # convergence: same stopping iteration as the oracle for a threshold away from the boundary
    want, it_want = oracle.embed(rowptr, col, vl, x0, 50, convergence_threshold=1e-3)
    out = np.empty((n, d), np.float32)
    it = ctypes.c_uint64(0)
    _hip.check(L.cleora_embed(g.handle, _hip.ptr(hashes), None, _hip.LEFT, d, 50, 3, 0.0, 1e-3, 0,
                              _hip.ptr(out), ctypes.byref(it)))
    assert it.value == it_want and it_want < 50
    np.testing.assert_array_equal(out, want)


@pytest.mark.parametrize("d", [256, 512, 1024, 2048, 128, 64, 32, 8, 100, 260, 1000])
def test_hot_column_policy_changes_no_bit(d):
    """The gather cache policy (the most referenced rows keep the default cache policy, all others are
    loaded `nt` through a buffer descriptor) must not change any result bit; forced on with a small
    byte budget."""
    n = 3000 if d <= 512 else 700
    rng = np.random.default_rng(900 + d)
    deg = rng.poisson(10, n).astype(np.int64)
    deg[5] = 2500                                                    # one split row
    rowptr = np.zeros(n + 1, np.uint64)
    rowptr[1:] = np.cumsum(deg).astype(np.uint64)
    nnz = int(rowptr[-1])
    col = np.minimum((rng.pareto(1.0, nnz) * 20).astype(np.int64), n - 1).astype(np.uint32)   # skewed popularity
    val = rng.random(nnz, dtype=np.float32)
    x = rng.standard_normal((n, d)).astype(np.float32)
    g = _hip.Graph.from_host(rowptr, col, val)
    base = run_dev(g, _hip.LEFT, x, flags=_hip.F_L2NORM)
    assert g.info().hot_rows == 0                                    # automatic mode: X is far below 1 GiB
    indeg = np.bincount(col, minlength=n)
    for rows in (50, 1000):
        g.set_hot_cache(d * 4 * rows)
        np.testing.assert_array_equal(run_dev(g, _hip.LEFT, x, flags=_hip.F_L2NORM), base)
        marked = g.info().hot_rows
        if d % 4 == 0:                                               # the policy lives in the float4 path
            assert marked > 0                                        # whole in-degree classes from the top
            thr = np.sort(indeg)[::-1][marked - 1]
            assert marked == int((indeg >= thr).sum())
        else:
            assert marked == 0
    g.set_hot_cache(0)
    np.testing.assert_array_equal(run_dev(g, _hip.LEFT, x, flags=_hip.F_L2NORM), base)
    assert g.info().hot_rows == 0
    mask = np.ones(n, bool)
    mask[5] = False
    np.testing.assert_array_equal(base[mask], oracle.l2_normalize(oracle.spmm(rowptr, col, val, x))[mask])


@pytest.mark.parametrize("d", [64, 128, 256, 512, 1024, 192, 320])
@pytest.mark.parametrize("n", [1, 15, 16, 17, 1001])
def test_standalone_l2_normalize_bit_exact(n, d):
    """cleora_l2_normalize / cleora_rowops_dev(L2NORM): the 16-lanes-per-row exact-order kernel (d = 64, 128, 256,
    512, 1024) and the general path (192, 320) against the oracle's sequential sum, bit for bit; zero rows stay zero;
    also in place."""
    rng = np.random.default_rng(n * 1000 + d)
    x = (rng.standard_normal((n, d)) * rng.uniform(1e-3, 1e3, (n, 1))).astype(np.float32)
    if n > 2:
        x[2] = 0.0
    want = oracle.l2_normalize(x)
    y = np.full_like(x, np.nan)
    _hip.check(L.cleora_l2_normalize(_hip.ptr(x), n, d, _hip.ptr(y)))
    np.testing.assert_array_equal(y, want)
    dx = _hip.DevArray.from_host(x)
    _hip.check(L.cleora_rowops_dev(dx.ptr, d, n, d, dx.ptr, d, _hip.F_L2NORM, 0.0, None, None, None, None))
    _hip.check(L.cleora_stream_sync(None))
    np.testing.assert_array_equal(dx.to_host(), want)
