"""GPU: the HIP backend of cleora_amd/sharded.py on one device (world 1, several row blocks per
iteration): block decomposition, padding and the whitened loop against the oracle.  The
multi-rank collective logic is covered on CPU by tests/test_sharded_cpu.py (gloo, world 2)."""
import numpy as np
import pytest
import torch

import oracle
from cleora_amd import _hip, sharded
from tests import sharded_model as model
from oracle import whiten as ow
from tests.graphs import random_csr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("steps", [1, 3, 4])
def test_blocks_reproduce_whole_graph(steps):
    dev = torch.device("cuda:0")
    n, d = 5003, 256
    rowptr, col, vl, vs = random_csr(n, 9, seed=4, empty_frac=0.03, hubs=[(11, 2500), (4000, 1500)])
    t = lambda a, dt: torch.from_numpy(a.view(dt) if a.dtype.kind == "u" else a).to(dev)
    sg = model.ShardedGraph(n, t(rowptr, np.int64), t(col, np.int32), t(vl, None), t(vs, None), 0, 1,
                              steps, sharded.HipBackend(dev))
    assert sg.n_pad % 4 == 0 and sg.n_pad >= n and sg.local_nnz == int(rowptr[-1])
    x0 = np.zeros((sg.n_pad, d), np.float32)
    x0[:n] = np.random.default_rng(5).standard_normal((n, d)).astype(np.float32)
    hub = np.zeros(n, bool)
    hub[[11, 4000]] = True
    for kind, val, rw in ((0, vl, 0.0), (1, vs, 0.35)):
        x, ran = model.embed_sharded(sg, kind, torch.from_numpy(x0).to(dev), 1, rw)
        want, _ = oracle.embed(rowptr, col, val, x0[:n], 1, residual_weight=rw)
        got = x[:n].cpu().numpy()
        np.testing.assert_array_equal(got, want)                      # every row bit-exact, the two hub rows included
        assert float(x[n:].abs().max()) == 0.0 if sg.n_pad > n else True
    x, ran = model.embed_sharded(sg, 0, torch.from_numpy(x0).to(dev), 6, 0.0, 0.0)
    want, _ = oracle.embed(rowptr, col, vl, x0[:n], 6)
    np.testing.assert_array_equal(x[:n].cpu().numpy(), want)


def test_whitened_loop_on_blocks():
    dev = torch.device("cuda:0")
    n, d = 3001, 32
    rowptr, col, vl, vs = random_csr(n, 8, seed=6)
    t = lambda a, dt: torch.from_numpy(a.view(dt) if a.dtype.kind == "u" else a).to(dev)
    sg = model.ShardedGraph(n, t(rowptr, np.int64), t(col, np.int32), t(vl, None), None, 0, 1, 3,
                              sharded.HipBackend(dev))
    x0 = np.zeros((sg.n_pad, d), np.float32)
    x0[:n] = np.random.default_rng(7).standard_normal((n, d)).astype(np.float32)
    x, _ = model.embed_sharded(sg, 0, torch.from_numpy(x0).to(dev), 3, whiten=True)
    want, _ = ow.embed_slow(lambda v: oracle.spmm(rowptr, col, vl, v), x0[:n], 3, whiten=True)
    got = x[:n].cpu().numpy()
    s = np.sign((got * want).sum(axis=0))
    assert np.abs(got * s - want).max() <= 2e-3 * np.abs(want).max()


@pytest.mark.parametrize("d,parts", [(256, 2), (256, 8), (64, 4), (24, 2)])
def test_column_slices_rowsq_then_scale(d, parts):
    """The two-pass form a column-partitioned rank runs: SpMM + ROWSQ per slice, sum the partial
    sums of squares (what the all-reduce does), SCALE.  SpMM elements are bit-exact; the norm is a
    sum of per-slice partials (tolerance)."""
    dev = torch.device("cuda:0")
    n = 3000
    rowptr, col, vl, vs = random_csr(n, 9, seed=21, empty_frac=0.03, hubs=[(5, 1500)])
    t = lambda a, dt: torch.from_numpy(a.view(dt) if a.dtype.kind == "u" else a).to(dev)
    be = sharded.HipBackend(dev)
    x0 = np.random.default_rng(22).standard_normal((n, d)).astype(np.float32)
    dl = d // parts
    rw = 0.25
    graphs = [sharded.ColumnShardedGraph(n, t(rowptr, np.int64), t(col, np.int32), t(vl, None), t(vs, None),
                                         d, r, parts, be) for r in range(parts)]
    xs = [torch.from_numpy(np.ascontiguousarray(x0[:, g.c0:g.c0 + dl])).to(dev) for g in graphs]
    ys = [torch.empty_like(v) for v in xs]
    sqs = [torch.zeros(n, dtype=torch.float32, device=dev) for _ in graphs]
    flags = _hip.F_RESIDUAL | _hip.F_ROWSQ
    for g, xl, yl, sq in zip(graphs, xs, ys, sqs):
        be.propagate(g.block, 0, xl, yl, flags, rw, xl, None, sq)
    raw = torch.cat(ys, dim=1).cpu().numpy()
    y = oracle.spmm(rowptr, col, vl, x0)
    blended = (np.float32(1.0) - np.float32(rw)) * y + np.float32(rw) * x0
    hub = np.zeros(n, bool)
    hub[5] = True
    np.testing.assert_array_equal(raw, blended)                         # unnormalised rows: bit-exact (hub row 5 too)
    total = torch.stack(sqs).sum(0)                                      # the all-reduce
    np.testing.assert_allclose(total.cpu().numpy(), (blended.astype(np.float64) ** 2).sum(1), rtol=2e-6)
    sqd = torch.zeros(n, dtype=torch.float64, device=dev)
    acc = 0.0
    for g, xl, yl in zip(graphs, xs, ys):
        be.rowops(yl, yl, _hip.F_SCALE | _hip.F_SQDIFF, 0.0, xl, sqd, total)
        acc += float(sqd.sum())
    got = torch.cat(ys, dim=1).cpu().numpy()
    want = oracle.l2_normalize(blended)
    np.testing.assert_allclose(got, want, rtol=0, atol=3e-7)
    delta = (want - x0).astype(np.float64)
    assert abs(acc - (delta * delta).sum()) <= 1e-5 * (delta * delta).sum()
    # world == 1 goes through the fused single-pass kernel and stays bit-exact
    g1 = sharded.ColumnShardedGraph(n, t(rowptr, np.int64), t(col, np.int32), t(vl, None), None, d, 0, 1, be)
    x1, _ = sharded.embed_column_sharded(g1, 0, torch.from_numpy(x0).to(dev), 2)
    want2, _ = oracle.embed(rowptr, col, vl, x0, 2)
    np.testing.assert_allclose(x1.cpu().numpy(), want2, rtol=0, atol=2e-6)


def test_column_layout_whitened_loop_single_rank():
    """ColumnShardedGraph.whiten on the HIP backend (world 1: the row-layout switch is the identity;
    the all-to-all path itself is covered under gloo in tests/test_sharded_cpu.py)."""
    dev = torch.device("cuda:0")
    n, d = 2501, 32
    rowptr, col, vl, vs = random_csr(n, 8, seed=31)
    t = lambda a, dt: torch.from_numpy(a.view(dt) if a.dtype.kind == "u" else a).to(dev)
    cg = sharded.ColumnShardedGraph(n, t(rowptr, np.int64), t(col, np.int32), t(vl, None), None, d, 0, 1,
                                    sharded.HipBackend(dev))
    x0 = np.random.default_rng(32).standard_normal((n, d)).astype(np.float32)
    xp = np.zeros((cg.n_pad, d), np.float32)
    xp[:n] = x0
    x, _ = sharded.embed_column_sharded(cg, 0, torch.from_numpy(xp).to(dev), 3, whiten=True)
    want, _ = ow.embed_slow(lambda v: oracle.spmm(rowptr, col, vl, v), x0, 3, whiten=True)
    got = x[:n].cpu().numpy()
    s = np.sign((got * want).sum(axis=0))
    assert np.abs(got * s - want).max() <= 2e-3 * np.abs(want).max()


def test_column_layout_row_blocks_single_rank():
    """steps > 1 (the row blocks whose all-reduces overlap the next block's SpMM) on the HIP backend."""
    dev = torch.device("cuda:0")
    n, d = 4001, 64
    rowptr, col, vl, vs = random_csr(n, 9, seed=41, empty_frac=0.02, hubs=[(100, 1800)])
    t = lambda a, dt: torch.from_numpy(a.view(dt) if a.dtype.kind == "u" else a).to(dev)
    cg = sharded.ColumnShardedGraph(n, t(rowptr, np.int64), t(col, np.int32), t(vl, None), t(vs, None), d, 0, 1,
                                    sharded.HipBackend(dev), steps=3)
    assert len(cg.blocks) == 3 and cg.row_blocks[-1][1] == n
    x0 = np.random.default_rng(42).standard_normal((n, d)).astype(np.float32)
    for kind, val, rw in ((0, vl, 0.0), (1, vs, 0.3)):
        x, _ = sharded.embed_column_sharded(cg, kind, torch.from_numpy(x0).to(dev), 3, rw)
        want, _ = oracle.embed(rowptr, col, val, x0, 3, residual_weight=rw)
        np.testing.assert_allclose(x.cpu().numpy(), want, rtol=0, atol=2e-6)
