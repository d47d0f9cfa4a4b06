"""GPU: degenerate inputs the reference handles (empty / ragged / all-empty rows / zero norms) and
size-independent properties at BASELINE sizes, where the CPU oracle would not finish in seconds."""
import ctypes

import numpy as np
import pytest

import oracle
from cleora_amd import _hip
from cleora_amd.pycleora import SparseMatrix

pytestmark = pytest.mark.gpu


def dev_propagate(g, x, flags=0, kind=0):
    L = _hip.lib()
    n_rows, d = g.info().n_rows, x.shape[1]
    dx, dy = _hip.DevArray.from_host(x), _hip.DevArray((n_rows, d), np.float32)
    L.cleora_memset(dy.ptr, 0xFF, dy.nbytes, None)
    _hip.check(L.cleora_propagate_dev(g.handle, kind, dx.ptr, d, d, dy.ptr, d, flags, 0.0, None, None, None, None))
    _hip.check(L.cleora_stream_sync(None))
    return dy.to_host()


def test_all_rows_empty_and_zero_norm_rows():
    n, d = 300, 64
    rowptr = np.zeros(n + 1, np.uint64)
    g = _hip.Graph.from_host(rowptr, np.zeros(0, np.uint32), np.zeros(0, np.float32))
    x = np.random.default_rng(0).standard_normal((n, d)).astype(np.float32)
    # no edges: zeros (the reference zero-fills), and normalising a zero row keeps zeros (norm clamp 1e-10)
    np.testing.assert_array_equal(dev_propagate(g, x), np.zeros((n, d), np.float32))
    np.testing.assert_array_equal(dev_propagate(g, x, _hip.F_L2NORM), np.zeros((n, d), np.float32))


def test_single_entity_and_d1():
    rowptr = np.array([0, 1], np.uint64)
    g = _hip.Graph.from_host(rowptr, np.array([0], np.uint32), np.array([1.0], np.float32))
    for d in (1, 2, 256):
        x = np.full((1, d), -3.0, np.float32)
        np.testing.assert_array_equal(dev_propagate(g, x), x)
        np.testing.assert_array_equal(dev_propagate(g, x, _hip.F_L2NORM), oracle.l2_normalize(x))


def test_ragged_rows_every_length_up_to_200():
    # row r has exactly r edges: covers every remainder of the 8-wide unroll and the 64-wide chunking
    n, d = 201, 256
    deg = np.arange(n, dtype=np.int64)
    rowptr = np.zeros(n + 1, np.uint64)
    rowptr[1:] = np.cumsum(deg).astype(np.uint64)
    rng = np.random.default_rng(1)
    col = rng.integers(0, n, int(rowptr[-1])).astype(np.uint32)
    val = rng.standard_normal(int(rowptr[-1])).astype(np.float32)
    x = rng.standard_normal((n, d)).astype(np.float32)
    g = _hip.Graph.from_host(rowptr, col, val)
    np.testing.assert_array_equal(dev_propagate(g, x), oracle.spmm(rowptr, col, val, x))
    for dd in (64, 8, 20):   # sub-wave groups with ragged neighbours in one wavefront
        xx = np.ascontiguousarray(x[:, :dd])
        np.testing.assert_array_equal(dev_propagate(g, xx, _hip.F_L2NORM),
                                      oracle.l2_normalize(oracle.spmm(rowptr, col, val, xx)))


def test_special_values_propagate_like_ieee():
    rowptr = np.array([0, 2, 3, 3], np.uint64)
    col = np.array([1, 2, 0], np.uint32)
    val = np.array([0.5, -2.0, 1e30], np.float32)
    x = np.array([[1e30, 1.0, -0.0, 5.0], [np.inf, 1e-45, 0.0, -1.0], [1.0, np.nan, 0.0, 2.0]], np.float32)
    g = _hip.Graph.from_host(rowptr, col, val)
    got = dev_propagate(g, x)
    want = oracle.spmm(rowptr, col, val, x)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))   # inf, nan, denormals, signed zeros


def test_empty_graph_object():
    g = SparseMatrix()
    assert g.left_markov_propagate(np.zeros((0, 8), np.float32)).shape == (0, 8)
    assert g.initialize_deterministically(4).shape == (0, 4)
    assert g.embed_fast(4, 3).shape == (0, 4)


def test_wide_rows_panel_path():
    n, d = 300, 2304   # > 2048: column panels + wide row epilogue
    rng = np.random.default_rng(3)
    deg = rng.poisson(6, n).astype(np.int64)
    rowptr = np.zeros(n + 1, np.uint64)
    rowptr[1:] = np.cumsum(deg).astype(np.uint64)
    col = rng.integers(0, n, int(rowptr[-1])).astype(np.uint32)
    val = rng.random(int(rowptr[-1]), dtype=np.float32)
    x = rng.standard_normal((n, d)).astype(np.float32)
    g = _hip.Graph.from_host(rowptr, col, val)
    want = oracle.spmm(rowptr, col, val, x)
    np.testing.assert_array_equal(dev_propagate(g, x), want)
    np.testing.assert_array_equal(dev_propagate(g, x, _hip.F_L2NORM), oracle.l2_normalize(want))


@pytest.mark.parametrize("config", ["C2_bipartite_1M_20M", "C3_powerlaw_10M_200M"])
def test_properties_at_baseline_sizes(config):
    """BASELINE configs 2 and 3 at full size: row-stochasticity, linearity, determinism, unit norms,
    independence from the hub threshold, and whole-graph bit-equality with the oracle (hub rows included)."""
    import torch
    from cleora_amd import synth
    dev = torch.device("cuda:0")
    if config.startswith("C2"):
        g = synth.bipartite_graph(500_000, 500_000, 10_000_000, 1, dev)
    else:
        g = synth.power_law_graph(10_000_000, 95_000_000, 2, dev)
    n, nnz, d = g["n"], g["nnz"], 256
    L = _hip.lib()
    mk = lambda thr, seg: _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(),
                                                 g["val_left"].data_ptr(), g["val_sym"].data_ptr(), 0, thr, seg, keepalive=g)
    graph, graph2 = mk(0, 0), mk(300, 64)
    s = torch.cuda.current_stream().cuda_stream

    def prop(gr, x, flags=0, kind=0):
        y = torch.empty_like(x)
        _hip.check(L.cleora_propagate_dev(gr.handle, kind, x.data_ptr(), d, d, y.data_ptr(), d, flags, 0.0, None, None, None, s))
        return y

    # 1. left Markov matrix is row-stochastic: A @ c = c (each row sums to 1 within f32 rounding)
    #    — as far as the reference's own sum goes: one f32 accumulator over the row's edges in stored order (src/embedding.rs:76-83)
    #    drifts by up to ~deg * 2^-24 of the sum, which for C3's longest row (431 465 edges) is ~1e-3; rows of ordinary length
    #    stay within 2e-5.  (The segmented hub sum of rounds 1-4 hid this; the in-order kernel reproduces it bit for bit — check 7.)
    c = torch.full((n, d), 0.375, dtype=torch.float32, device=dev)
    err = (prop(graph, c) - 0.375).abs().amax(dim=1)
    deg_all = torch.diff(g["rowptr"])
    assert float(err[deg_all <= 1024].max()) < 2e-5
    assert bool((err <= 0.375 * 2.0 ** -24 * (deg_all.double() + 2) + 2e-5).all())
    del err
    # 2. determinism: two launches are bit-identical
    x = torch.randn((n, d), dtype=torch.float32, device=dev)
    y1, y2 = prop(graph, x), prop(graph, x)
    assert torch.equal(y1, y2)
    # 2b. the automatic gather cache policy (hot.hip) arms on the third launch when X >= 1 GiB and d >= 256:
    #     y1 ran without it, y2 with it (C3), and switching it off again changes no bit either
    armed = graph.info().hot_rows
    assert (armed > 0) == config.startswith("C3")
    graph.set_hot_cache(0)
    assert torch.equal(prop(graph, x), y1) and graph.info().hot_rows == 0
    graph.set_hot_cache(256 << 20)
    assert torch.equal(prop(graph, x), y1) and 0 < graph.info().hot_rows <= (256 << 20) // (d * 4)
    graph.set_hot_cache(-1)
    # 3. linearity: A(2x) = 2 A x exactly (power-of-two scaling commutes with every rounding)
    assert torch.equal(prop(graph, x * 2.0), y1 * 2.0)
    # 4. the hub threshold decides which KERNEL adds a row, never the result: rows of 257..300 edges are scheduled first on graph2 and in row order on graph, the longest go through
    #    hub_inorder_kernel on graph2 and through the main kernel on graph — same bits; the segmented form
    #    (CLEORA_F_HUB_SEGMENTS) changes only the split rows, within summation-order tolerance
    y3 = prop(graph2, x)
    assert torch.equal(y3, y1)
    deg = torch.diff(g["rowptr"])
    y4 = prop(graph2, x, _hip.F_HUB_SEGMENTS)
    unsplit = deg <= 300
    assert torch.equal(y4[unsplit], y1[unsplit])
    assert float((y4 - y1).abs().max()) < 1e-4
    del y3, y4
    # 5. fused L2: unit rows
    yn = prop(graph, x, _hip.F_L2NORM)
    assert float((yn.double().pow(2).sum(1).sqrt() - 1).abs().max()) < 1e-6
    # 6. symmetric propagation: sampled rows recomputed on the host in the reference's order, bit-exact
    rows = torch.randint(0, n, (2000,), device=dev).cpu().numpy()
    rp = g["rowptr"].cpu().numpy()
    colh, vsh = g["col"].cpu().numpy().view(np.uint32), g["val_sym"].cpu().numpy()
    xh = x.cpu().numpy()
    ys_dev = prop(graph, x, 0, 1)
    ys = ys_dev.cpu().numpy()
    for r in rows[:400]:
        b, e = int(rp[r]), int(rp[r + 1])
        acc = np.zeros(d, np.float32)   # reference order: acc += v * x, separate f32 mul/add
        for k in range(b, e):
            acc += vsh[k] * xh[colh[k]]
        np.testing.assert_array_equal(ys[r], acc)
    # 7. WHOLE-GRAPH parity with the oracle (src/embedding.rs:52-104), left and symmetric, same graph and X:
    #    one fused SpMM + L2 iteration — EVERY row bit-identical, the hub rows (C3: 1 750, longest 431 465 edges) included
    rp64 = rp.astype(np.uint64)
    edges = np.empty(nnz, dtype=oracle.EDGE_DTYPE)
    edges["col"], edges["left"], edges["sym"] = colh, g["val_left"].cpu().numpy(), vsh
    hub = np.diff(rp.astype(np.int64)) > graph.info().hub_threshold
    threads = oracle.max_threads()
    want = np.empty((n, d), np.float32)
    for kind in (0, 1):
        oracle.spmm_aos_l2_inplace(rp64, edges, kind == 1, xh, want, threads)
        got = prop(graph, x, _hip.F_L2NORM, kind).cpu().numpy()
        same = (got.view(np.uint32) == want.view(np.uint32)).all(axis=1)
        assert same.all(), f"{(~same).sum()} of {n} rows differ from the oracle ({(~same[hub]).sum()} of them hub rows)"
        del got
    del want, edges


def test_whitening_at_c2_size_against_numpy_oracle():
    """whiten_embeddings (pycleora/__init__.py:130-164) at BASELINE config 2's shape, 1M x 256: the device chain
    (cleora_whiten_dev) against the numpy fp64 restatement, columns sign-aligned; tolerance 2e-4 relative like the
    small-size tests (f32 projection with a different summation order)."""
    import torch
    from oracle import whiten as ow
    dev = torch.device("cuda:0")
    n, d = 1_000_000, 256
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    x = torch.randn((n, d), generator=gen, device=dev, dtype=torch.float32)
    x = x * torch.linspace(0.5, 3.0, d, device=dev) + torch.randn(d, generator=gen, device=dev)
    x = (x / x.norm(dim=1, keepdim=True)).contiguous()             # unit rows, anisotropic, non-centred
    L = _hip.lib()
    ws = torch.empty(L.cleora_whiten_workspace(n, d), dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    eig = torch.empty(d, dtype=torch.float64, device=dev)
    _hip.check(L.cleora_whiten_dev(x.data_ptr(), d, n, d, d, y.data_ptr(), d, ws.data_ptr(), eig.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream))
    got = y.cpu().numpy()
    xh = x.cpu().numpy()
    mean, cov = ow.whiten_stats(xh)
    _, w = ow.whiten_transform(cov)
    assert np.abs(eig.cpu().numpy() - w).max() <= 1e-10 * w[0]
    want = ow.whiten_embeddings(xh)
    sgn = np.sign((got[:200_000] * want[:200_000]).sum(axis=0))
    sgn[sgn == 0] = 1
    assert np.abs(got * sgn - want).max() <= 2e-4 * np.abs(want).max()


def test_adopted_device_csr_is_bounds_checked():
    import torch
    dev = torch.device("cuda:0")
    rowptr = torch.tensor([0, 2, 3], dtype=torch.int64, device=dev)
    val = torch.ones(3, dtype=torch.float32, device=dev)
    good = torch.tensor([0, 1, 1], dtype=torch.int32, device=dev)
    _hip.Graph.from_device(2, 2, 3, rowptr.data_ptr(), good.data_ptr(), val.data_ptr(), None, 0, keepalive=(rowptr, good, val))
    bad = torch.tensor([0, 7, 1], dtype=torch.int32, device=dev)
    with pytest.raises(ValueError, match="column index out of range"):
        _hip.Graph.from_device(2, 2, 3, rowptr.data_ptr(), bad.data_ptr(), val.data_ptr(), None, 0)
    bad_rp = torch.tensor([0, 3, 2], dtype=torch.int64, device=dev)
    with pytest.raises(ValueError, match="rowptr"):
        _hip.Graph.from_device(2, 2, 3, bad_rp.data_ptr(), good.data_ptr(), val.data_ptr(), None, 0)


def test_more_than_2_pow_32_edges():
    """Maximum-size indexing: nnz > 2^32 (the papers100M regime, BASELINE config 4) needs 64-bit edge
    offsets end to end.  4.4M rows x 1000 edges = 4.4e9 stored entries, d = 4 so X stays small."""
    import torch
    dev = torch.device("cuda:0")
    if torch.cuda.get_device_properties(0).total_memory < 100 * 2 ** 30:
        pytest.skip("needs ~40 GB of HBM")
    n, deg, d = 4_400_000, 1000, 4
    nnz = n * deg
    assert nnz > 2 ** 32
    gen = torch.Generator(device=dev)
    gen.manual_seed(9)
    col = torch.randint(0, n, (nnz,), generator=gen, device=dev, dtype=torch.int32)
    val = torch.full((nnz,), 1.0 / deg, dtype=torch.float32, device=dev)
    rowptr = torch.arange(0, nnz + 1, deg, dtype=torch.int64, device=dev)
    g = _hip.Graph.from_device(n, n, nnz, rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), None, 0,
                               keepalive=(rowptr, col, val))
    x = torch.randn((n, d), dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    L = _hip.lib()
    s = torch.cuda.current_stream().cuda_stream
    _hip.check(L.cleora_propagate_dev(g.handle, 0, x.data_ptr(), d, d, y.data_ptr(), d, 0, 0.0, None, None, None, s))
    torch.cuda.synchronize()
    # rows from the first, middle and LAST part of the edge array (offsets beyond 2^32)
    for r in (0, 1, n // 2, n - 2, n - 1):
        b = r * deg
        want = (x[col[b:b + deg].long()].double() * (1.0 / deg)).sum(0)
        assert float((y[r].double() - want).abs().max()) < 1e-5, r
    # row-stochastic: a constant propagates to itself on every row, including the tail
    c = torch.full((n, d), 0.5, dtype=torch.float32, device=dev)
    _hip.check(L.cleora_propagate_dev(g.handle, 0, c.data_ptr(), d, d, y.data_ptr(), d, 0, 0.0, None, None, None, s))
    torch.cuda.synchronize()
    assert float((y - 0.5).abs().max()) < 1e-4


def test_more_than_2_pow_24_blocks():
    """Launches of more than 2^24 workgroups (|V| > 67M at one row per wavefront — the papers100M
    regime): the work-item extent of a 1-D grid is a 32-bit field and silently truncates, so the
    big kernels linearise a 2-D grid.  72M rows, d = 3 (scalar path: one row per wavefront)."""
    import torch
    dev = torch.device("cuda:0")
    n, deg, d = 72_000_000, 2, 3
    nnz = n * deg
    gen = torch.Generator(device=dev)
    gen.manual_seed(10)
    col = torch.randint(0, n, (nnz,), generator=gen, device=dev, dtype=torch.int32)
    val = torch.full((nnz,), 0.5, dtype=torch.float32, device=dev)
    rowptr = torch.arange(0, nnz + 1, deg, dtype=torch.int64, device=dev)
    g = _hip.Graph.from_device(n, n, nnz, rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), None, 0,
                               keepalive=(rowptr, col, val))
    L = _hip.lib()
    s = torch.cuda.current_stream().cuda_stream
    hashes = torch.arange(n, device=dev, dtype=torch.int64) * 2654435761
    x = torch.empty((n, d), dtype=torch.float32, device=dev)
    y = torch.full((n, d), float("nan"), dtype=torch.float32, device=dev)
    _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 5, x.data_ptr(), d, s))
    tail = oracle.init(hashes[-3:].cpu().numpy().view(np.uint64), d, 5)
    np.testing.assert_array_equal(x[-3:].cpu().numpy(), tail)                       # init reaches the last rows
    _hip.check(L.cleora_propagate_dev(g.handle, 0, x.data_ptr(), d, d, y.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None, s))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y).all())                                             # every row was written
    c2 = col.view(n, deg).long()
    want = (x[c2[:, 0]] * 0.5 + x[c2[:, 1]] * 0.5)
    want = want / want.norm(dim=1, keepdim=True).clamp(min=1e-10)
    assert float((y - want).abs().max()) < 1e-6
    q = torch.tensor([0.6, 0.0, 0.8], dtype=torch.float32, device=dev)
    sc = torch.full((n,), float("nan"), dtype=torch.float32, device=dev)
    _hip.check(L.cleora_cosine_scores_dev(y.data_ptr(), d, n, d, q.data_ptr(), sc.data_ptr(), s))
    torch.cuda.synchronize()
    assert float((sc - (y * q).sum(1)).abs().max()) < 1e-5          # (not y @ q: the BLAS gemv is itself wrong beyond 2^24 rows)


def test_generator_values_on_the_gpu_equal_the_cpu_generator():
    """bench.py builds its graph with cleora_amd/synth.py ON THE GPU; tests/test_wire_and_generator.py pins the same
    code on the CPU against the string builder.  Same pairs on both devices: the Markov values
    (f32 divide, correctly rounded f32 sqrt) must be bit-identical, so the pin carries over to the bench graph."""
    import torch
    from cleora_amd import synth
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 8))       # torch's CPU sort / unique with 256 threads on 50k elements: 44 s of this suite went here
    try:
        _generator_values(torch, synth)
    finally:
        torch.set_num_threads(threads)


def _generator_values(torch, synth):
    cpu = synth.power_law_graph(50_000, 100_000, 3, torch.device("cpu"))
    rp = cpu["rowptr"].numpy()
    rows = np.repeat(np.arange(cpu["n"]), np.diff(rp))
    cols = cpu["col"].numpy().astype(np.int64)
    keep = rows < cols
    a, b = torch.from_numpy(rows[keep]), torch.from_numpy(cols[keep])
    dev = torch.device("cuda:0")
    for reflexive in (True, False):
        c = synth._csr_from_undirected(a, b, cpu["n"], reflexive)
        g = synth._csr_from_undirected(a.to(dev), b.to(dev), cpu["n"], reflexive)
        assert c["n"] == g["n"] and c["nnz"] == g["nnz"]
        for key in ("rowptr", "col", "val_left", "val_sym"):
            assert torch.equal(c[key], g[key].cpu()), key
