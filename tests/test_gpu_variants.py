"""GPU: the device-resident embed variants and predict_links (cleora_amd/variants.py) against the outputs of
the reference's own functions (tests/golden/variants_ref.npz, produced by tests/golden/make_golden.py from
/root/reference/pycleora/__init__.py:206-410, 636-681, 784-852 on karate club).

Tolerances (floating point, stated):
  whiten=False runs: the reference accumulates `adj @ X` in f64 and rounds to f32 once per iteration, the
      device accumulates in f32 in edge order: |delta| <= 2e-5 on unit-norm rows after <= 5 iterations.
  whiten=True runs (d = 8 < rank 33, well conditioned): columns compared after sign alignment (eigh's sign
      is arbitrary), |delta| <= 5e-3 * max|ref|, and the pairwise-cosine matrices agree to 1e-4.
  predict_links: same (source, target) pairs in the same order, scores to 2e-6.
"""
import os

import numpy as np
import pytest

from cleora_amd import _hip, variants
from cleora_amd.pycleora import SparseMatrix

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref(golden_dir):
    k = np.load(os.path.join(golden_dir, "karate_ref.npz"))
    v = np.load(os.path.join(golden_dir, "variants_ref.npz"))
    edges = [str(s) for s in k["edges"]]
    g = SparseMatrix.from_iterator(iter(edges), str(k["columns"]))
    return k, v, edges, str(k["columns"]), g


def cosine_matrix(e):
    e = e.astype(np.float64)
    e = e / np.maximum(np.linalg.norm(e, axis=1, keepdims=True), 1e-300)
    return e @ e.T


def assert_close(got, want, whitened, blocks=1, lead=None):
    """lead: compare only the `lead` leading whitened components (the trailing ones of a nearly singular
    covariance are amplified rounding noise in the reference itself)."""
    assert got.shape == want.shape and got.dtype == np.float32
    if not whitened:
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)
        return
    w = want.shape[1] // blocks
    for b in range(blocks):                                 # concatenated outputs: every block is whitened separately
        a, r = got[:, b * w:(b + 1) * w][:, :lead], want[:, b * w:(b + 1) * w][:, :lead]
        s = np.sign((a * r).sum(axis=0))
        assert np.abs(a * s - r).max() <= 5e-3 * np.abs(r).max()
        assert np.abs(cosine_matrix(a) - cosine_matrix(r)).max() < 1e-4


@pytest.mark.parametrize("tag,wh", [("w", True), ("n", False)])
def test_embed_multiscale(ref, tag, wh):
    k, v, edges, columns, g = ref
    got = variants.embed_multiscale(g, 8, scales=[2, 5, 3], whiten=wh)
    assert_close(got, v[f"multiscale_{tag}"], wh, blocks=3)
    with pytest.raises(ValueError, match="scales must be"):
        variants.embed_multiscale(g, 8, scales=[])


@pytest.mark.parametrize("tag,wh", [("w", True), ("n", False)])
def test_embed_weighted(ref, tag, wh):
    k, v, edges, columns, g = ref
    graph, got = variants.embed_weighted(list(zip(edges, v["weights"].tolist())), columns, 8, 5,
                                         propagation="symmetric", whiten=wh)
    assert graph.entity_ids == g.entity_ids
    assert_close(got, v[f"weighted_{tag}"], wh)


@pytest.mark.parametrize("tag,wh", [("w", True), ("n", False)])
def test_embed_directed(ref, tag, wh):
    k, v, edges, columns, g = ref
    graph, got = variants.embed_directed(edges, columns, 8, 5, whiten=wh)
    assert graph.num_entities == g.num_entities
    # 15 of the 34 karate nodes have no outgoing edge in the directed graph (zero rows): the covariance is nearly
    # singular and the last two whitened components are noise; the first six are compared
    assert_close(got, v[f"directed_{tag}"], wh, lead=6)


@pytest.mark.parametrize("tag,wh", [("w", True), ("n", False)])
def test_embed_with_attention(ref, tag, wh):
    k, v, edges, columns, g = ref
    seen = []
    got = variants.embed_with_attention(g, 8, 4, attention_temperature=0.7, whiten=wh,
                                        callback=lambda i, e: seen.append(i))
    assert seen == [0, 1, 2, 3]
    assert_close(got, v[f"attention_{tag}"], wh)
    got = variants.embed_with_attention(g, 8, 3, propagation="symmetric", attention_temperature=2.0, whiten=wh)
    assert_close(got, v[f"attention_sym_{tag}"], wh)
    with pytest.raises(ValueError, match="attention_temperature must be positive"):
        variants.embed_with_attention(g, 8, 3, attention_temperature=0.0)
    with pytest.raises(ValueError, match="num_iterations must be positive"):
        variants.embed_with_attention(g, 8, 0)


def test_edge_attention_weights_vs_numpy(ref):
    """cleora_edge_attention_dev alone against a numpy restatement of pycleora/__init__.py:241-268 on a random
    iterate: rows sum to 1, weights to 1e-6 relative."""
    k, v, edges, columns, g = ref
    rowptr, col, adj = k["rowptr"].astype(np.int64), k["col"].astype(np.int64), k["val_left"].astype(np.float64)
    n, d, temp = g.num_entities, 24, 0.5
    x = np.random.default_rng(3).standard_normal((n, d)).astype(np.float32)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    xn = x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-10)
    score = np.sum(xn[rows] * xn[col], axis=1) / temp
    mx = np.full(n, -np.inf)
    np.maximum.at(mx, rows, score)
    ex = np.exp(score - mx[rows])
    a = ex / np.maximum(np.bincount(rows, weights=ex, minlength=n), 1e-10)[rows] * adj
    want = a / np.maximum(np.bincount(rows, weights=a, minlength=n), 1e-10)[rows]
    L = _hip.lib()
    dx = _hip.DevArray.from_host(x)
    dv = _hip.DevArray((col.shape[0],), np.float32)
    _hip.check(L.cleora_edge_attention_dev(g._graph().handle, _hip.LEFT, dx.ptr, d, d, temp, dv.ptr, None))
    _hip.check(L.cleora_stream_sync(None))
    got = dv.to_host()
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(np.bincount(rows, weights=got, minlength=n), 1.0, atol=1e-6)
    # and the SpMM with caller-supplied values
    dy = _hip.DevArray((n, d), np.float32)
    _hip.check(L.cleora_propagate_vals_dev(g._graph().handle, dv.ptr, dx.ptr, d, d, dy.ptr, d, 0, 0.0, None, None,
                                           None, None))
    _hip.check(L.cleora_stream_sync(None))
    import oracle
    np.testing.assert_array_equal(dy.to_host(), oracle.spmm(k["rowptr"], k["col"], got, x))


@pytest.mark.parametrize("tag,wh", [("w", True), ("n", False)])
def test_embed_edge_features(ref, tag, wh):
    k, v, edges, columns, g = ref
    feats = dict(zip([str(s) for s in v["feat_keys"]], v["feats"]))
    got = variants.embed_edge_features(g, feats, 8, 3, whiten=wh)
    want = v[f"edgefeat_{tag}"]
    assert got.shape == want.shape == (g.num_entities, 11)
    assert_close(got[:, :8], want[:, :8], wh)
    assert_close(got[:, 8:], want[:, 8:], wh)
    only = variants.embed_edge_features(g, feats, 8, 3, combine="edge_only", whiten=wh)
    np.testing.assert_array_equal(only, got[:, 8:])
    with pytest.raises(ValueError, match="Unknown combine mode"):
        variants.embed_edge_features(g, feats, 8, 1, combine="sum")


@pytest.mark.parametrize("tag,kw", [("all", dict(top_k=12)),
                                    ("some", dict(top_k=7, source_entities=["0", "33", "5"])),
                                    ("keep", dict(top_k=9, exclude_existing=False, source_entities=["1", "2"]))])
def test_predict_links(ref, tag, kw):
    k, v, edges, columns, g = ref
    got = variants.predict_links(g, k["embed_whiten_d16"], **kw)
    want_s, want_t, want_sc = v[f"pred_{tag}_source"], v[f"pred_{tag}_target"], v[f"pred_{tag}_score"]
    assert len(got) == len(want_sc)
    np.testing.assert_allclose([p["score"] for p in got], want_sc, rtol=0, atol=2e-6)
    # candidates whose scores tie within the tolerance may swap places; everything else is in order
    for i, p in enumerate(got):
        if (p["source"], p["target"]) != (str(want_s[i]), str(want_t[i])):
            close = np.abs(want_sc - p["score"]) <= 2e-6
            assert (p["source"], p["target"]) in {(str(a), str(b)) for a, b in zip(want_s[close], want_t[close])}
    with pytest.raises(ValueError, match="not found"):
        variants.predict_links(g, k["embed_whiten_d16"], source_entities=["no-such-entity"])


@pytest.mark.parametrize("n,d,k,nq", [(50_000, 64, 17, 11), (5000, 256, 5, 3), (3000, 20, 40, 9), (100, 8, 100, 2),
                                      (4000, 256, 10, 70), (2000, 96, 5, 64), (700, 33, 3, 12)])
def test_topk_cosine_on_device_against_numpy(n, d, k, nq):
    """cleora_topk_cosine_dev: scores, the -2 masks (self, both directions of stored edges) and the selection order of
    numpy's `argsort()[::-1][:k]` (ties: larger index first), for a batch of query rows, several selection levels.
    Up to 8 queries take the vector-unit form, more the matrix-core form (X . Q through the projection kernel: its
    rows-in-LDS form at d = 64 / 256 / 96, the generic one at d = 20 / 33; 70 queries = two passes)."""
    import ctypes
    from tests.graphs import random_csr
    rng = np.random.default_rng(n + d)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x[7] = x[3]                                   # exact ties
    x[11] = 0.0                                   # zero row: norm clamp
    rowptr, col, vl, vs = random_csr(n, 6, seed=n, empty_frac=0.05)
    g = _hip.Graph.from_host(rowptr, col, vl)
    queries = rng.choice(n, nq, replace=False).astype(np.uint32)
    queries[0] = 3
    L = _hip.lib()
    dx, dq = _hip.DevArray.from_host(x), _hip.DevArray.from_host(queries)
    oi, os_ = _hip.DevArray((nq, k), np.uint32), _hip.DevArray((nq, k), np.float32)
    ws = _hip.DevArray((L.cleora_topk_workspace(n, k),), np.uint8)
    normed = x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-10)
    rows = np.repeat(np.arange(n), np.diff(rowptr.astype(np.int64)))
    for excl_edges in (1, 0):
        _hip.check(L.cleora_topk_cosine_dev(g.handle if excl_edges else None, dx.ptr, d, n, d, dq.ptr, nq, k, 1, excl_edges,
                                            oi.ptr, os_.ptr, ws.ptr, None))
        _hip.check(L.cleora_stream_sync(None))
        idx, sc = oi.to_host(), os_.to_host()
        for qi, q in enumerate(queries):
            sims = normed @ normed[q]
            sims[q] = -2.0
            if excl_edges:
                sims[col[rows == q]] = -2.0
                sims[rows[col == q]] = -2.0
            want = np.argsort(sims)[::-1][:k]
            np.testing.assert_allclose(sc[qi], sims[want], rtol=0, atol=3e-6)
            assert (np.diff(sc[qi]) <= 0).all()                                     # descending
            assert len(set(idx[qi].tolist())) == k                                    # no row twice
            np.testing.assert_allclose(sims[idx[qi]], sc[qi], rtol=0, atol=3e-6)     # every index carries its own score
            masked = sims[idx[qi]] <= -2.0
            assert (sc[qi][masked] == -2.0).all()
    g.close()


def _topk_against_numpy(x, queries, k, g=None, rowptr=None, col=None, atol=3e-6):
    """cleora_topk_cosine_dev (self excluded; stored edges too when a graph is given) against numpy's argsort()[::-1][:k]."""
    n, d = x.shape
    nq = len(queries)
    L = _hip.lib()
    dx, dq = _hip.DevArray.from_host(x), _hip.DevArray.from_host(np.asarray(queries, np.uint32))
    oi, os_ = _hip.DevArray((nq, k), np.uint32), _hip.DevArray((nq, k), np.float32)
    ws = _hip.DevArray((L.cleora_topk_workspace_for(n, k, nq),), np.uint8)
    _hip.check(L.cleora_topk_cosine_dev(g.handle if g is not None else None, dx.ptr, d, n, d, dq.ptr, nq, k, 1, 1 if g is not None else 0,
                                        oi.ptr, os_.ptr, ws.ptr, None))
    _hip.check(L.cleora_stream_sync(None))
    route = L.cleora_topk_last_route()
    idx, sc = oi.to_host(), os_.to_host()
    normed = x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-10)
    rows = np.repeat(np.arange(n), np.diff(rowptr.astype(np.int64))) if g is not None else None
    for qi, q in enumerate(queries):
        sims = normed @ normed[q]
        sims[q] = -2.0
        if g is not None:
            sims[col[rows == q]] = -2.0
            sims[rows[col == q]] = -2.0
        want = np.argsort(sims)[::-1][:k]
        np.testing.assert_allclose(sc[qi], sims[want], rtol=0, atol=atol)
        assert (np.diff(sc[qi]) <= 0).all()
        assert len(set(idx[qi].tolist())) == k
        np.testing.assert_allclose(sims[idx[qi]], sc[qi], rtol=0, atol=atol)
    return route, idx, sc


@pytest.mark.parametrize("n,d,k,nq", [(300_000, 32, 10, 3), (300_000, 32, 100, 70), (400_000, 64, 1, 1), (270_000, 16, 1024, 9)])
def test_topk_selection_from_a_short_list(n, d, k, nq):
    """From 256 Ki rows on (and 512 results per batch: below that cleora_topk_set_route(2) asks for it) the selection runs on a short list: threshold = the r-th largest of a stratified sample of
    the scores, one compaction pass, k rounds over what passed (csrc/similarity.hip).  Same contract as the full selection
    (numpy's `argsort()[::-1][:k]`, pycleora/__init__.py:663, 771; scores to 3e-6), both score layouts ([query][row] up to
    8 queries, [row][query] from the matrix cores beyond), with the -2 masks, and the route is the short list."""
    from tests.graphs import random_csr
    rng = np.random.default_rng(n + k)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x[7] = x[3]
    x[11] = 0.0
    rowptr, col, vl, vs = random_csr(n, 4, seed=k, empty_frac=0.05)
    g = _hip.Graph.from_host(rowptr, col, vl)
    queries = rng.choice(n, nq, replace=False).astype(np.uint32)
    queries[0] = 3
    L = _hip.lib()
    if k * nq < 512:
        _hip.check(L.cleora_topk_set_route(2))                  # the short list for a batch the plan would leave to the rounds
    try:
        route, _, _ = _topk_against_numpy(x, queries, k, g, rowptr, col)
    finally:
        _hip.check(L.cleora_topk_set_route(0))
    assert route == 1
    g.close()


def test_topk_short_list_on_adversarial_score_layouts():
    """Where a sampled threshold could go wrong: (a) scores that grow with the row index (a strided sample of a sorted
    sequence); (b) all rows equal — every score ties, the list overflows its buffer and the batch falls back to the full
    selection (ties: the larger row index first, like numpy's reversed argsort); (c) the k best all inside one stratum.
    The result is numpy's in every case; only (b) may leave the short list."""
    n, d, k = 300_000, 8, 25
    rng = np.random.default_rng(5)
    L = _hip.lib()
    _hip.check(L.cleora_topk_set_route(2))
    try:
        _adversarial_layouts(n, d, k, rng, L)
    finally:
        _hip.check(L.cleora_topk_set_route(0))


def _adversarial_layouts(n, d, k, rng, L):
    t = np.linspace(0.0, 1.5, n, dtype=np.float64)
    x = np.zeros((n, d), np.float32)
    x[:, 0], x[:, 1] = np.cos(t), np.sin(t)                      # cosine with row n-1 grows monotonically with the index
    route, idx, _ = _topk_against_numpy(x, [n - 1, 0, n // 2], k, atol=1e-6)
    assert route == 1
    x = np.tile(rng.standard_normal(d).astype(np.float32), (n, 1))
    route, idx, sc = _topk_against_numpy(x, [17, n - 1], k, atol=1e-6)
    assert route in (0, 2)
    assert idx[0].tolist() == list(range(n - 1, n - 1 - k, -1))                  # all tie at 1: the largest indices first
    x = (rng.standard_normal((n, d)) * 0.1).astype(np.float32)
    x[:, 0] -= 1.0
    x[1000:1000 + k] = np.array([1.0] + [0.0] * (d - 1), np.float32) + (rng.standard_normal((k, d)) * 1e-3).astype(np.float32)
    route, idx, _ = _topk_against_numpy(x, [1000], k - 1, atol=1e-6)
    assert route == 1 and set(idx[0].tolist()) == set(range(1001, 1000 + k))
    # the forced forms at a size both can take give identical arrays
    xs = rng.standard_normal((20_000, 24)).astype(np.float32)
    _hip.check(L.cleora_topk_set_route(2))
    r1, i1, s1 = _topk_against_numpy(xs, [5, 6, 7, 8, 9, 10, 11, 12, 13, 14], 8)
    _hip.check(L.cleora_topk_set_route(1))
    r0, i0, s0 = _topk_against_numpy(xs, [5, 6, 7, 8, 9, 10, 11, 12, 13, 14], 8)
    assert (r1, r0) == (1, 0) and np.array_equal(i1, i0) and np.array_equal(s1, s0)


@pytest.mark.parametrize("n,d", [(6000, 256), (3000, 512), (2000, 260), (1500, 30)])
def test_edge_attention_kernels_on_random_graphs(n, d):
    """Both forms of the attention kernel (16-byte gathers with the scores in registers; the scalar one for widths that
    are not a multiple of 4) against the numpy restatement of pycleora/__init__.py:241-268: rows of every length class —
    empty, a few edges, > 64 (several chunks), > 512 (scores through `out`) — weights to 2e-5 relative, rows sum to 1."""
    from tests.graphs import random_csr
    rowptr, col, vl, vs = random_csr(n, 10, seed=n + d, empty_frac=0.05, hubs=[(3, 700), (40, 130)])
    rowptr64, col64, adj = rowptr.astype(np.int64), col.astype(np.int64), vl.astype(np.float64)
    x = np.random.default_rng(d).standard_normal((n, d)).astype(np.float32)
    temp = 0.8
    rows = np.repeat(np.arange(n), np.diff(rowptr64))
    xn = x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-10)
    score = np.sum(xn[rows].astype(np.float64) * xn[col64], axis=1) / temp
    mx = np.full(n, -np.inf)
    np.maximum.at(mx, rows, score)
    ex = np.exp(score - mx[rows])
    a = ex / np.maximum(np.bincount(rows, weights=ex, minlength=n), 1e-10)[rows] * adj
    want = a / np.maximum(np.bincount(rows, weights=a, minlength=n), 1e-10)[rows]
    g = _hip.Graph.from_host(rowptr, col, vl)
    L = _hip.lib()
    dx = _hip.DevArray.from_host(x)
    dv = _hip.DevArray((col.shape[0],), np.float32)
    _hip.check(L.cleora_edge_attention_dev(g.handle, _hip.LEFT, dx.ptr, d, d, temp, dv.ptr, None))
    _hip.check(L.cleora_stream_sync(None))
    got = dv.to_host()
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-9)
    sums = np.bincount(rows, weights=got, minlength=n)
    nonempty = np.diff(rowptr64) > 0
    np.testing.assert_allclose(sums[nonempty], 1.0, atol=2e-6)
    g.close()


def test_top_k_beyond_the_device_selection_limit():
    """find_most_similar / predict_links accept any top_k like the reference (pycleora/__init__.py:636-681, 753-781); beyond
    the 1024 the device selection handles the scores still come from the device and the selection runs on
    the host.  Against a numpy restatement of the reference's own lines."""
    from cleora_amd import embed as dev_embed
    rng = np.random.default_rng(5)
    n = 2500
    edges = [f"e{a} e{b}" for a, b in zip(rng.integers(0, n, 12_000), rng.integers(0, n, 12_000)) if a != b]
    edges += [f"e{i} e{(i + 1) % n}" for i in range(n)]            # every entity exists
    g = SparseMatrix.from_iterator(iter(edges), "complex::reflexive::x")
    n = g.num_entities
    emb = rng.standard_normal((n, 24)).astype(np.float32)
    normed = emb / np.maximum(np.linalg.norm(emb, axis=1, keepdims=True), 1e-10)
    ids = g.entity_ids
    # find_most_similar, top_k = 1500
    q = 17
    got = dev_embed.find_most_similar(g, emb, ids[q], top_k=1500)
    sims = normed @ (emb[q] / max(np.linalg.norm(emb[q]), 1e-10))
    sims[q] = -1.0
    top = np.argsort(sims)[::-1][:1500]
    assert len(got) == 1500
    assert [r["index"] for r in got[:50]] == [int(i) for i in top[:50]]
    assert set(r["index"] for r in got) ^ set(int(i) for i in top) <= set(int(i) for i in top[-5:]) | set(r["index"] for r in got[-5:])
    np.testing.assert_allclose([r["similarity"] for r in got], sims[[r["index"] for r in got]], rtol=0, atol=2e-6)
    # predict_links, top_k = 1200 over two sources, existing edges excluded
    rows, cols = g.to_sparse_csr()[:2]
    existing = set(zip(rows.tolist(), cols.tolist()))
    src = [3, 40]
    got = variants.predict_links(g, emb, top_k=1200, exclude_existing=True, source_entities=[ids[s] for s in src])
    cand = []
    for s in src:
        sm = normed @ normed[s]
        sm[s] = -2.0
        for o in range(n):
            if (s, o) in existing or (o, s) in existing:
                sm[o] = -2.0
        for tgt in np.argsort(sm)[::-1][:1200]:
            if sm[tgt] > -2.0:
                cand.append((ids[s], ids[int(tgt)], float(sm[tgt])))
    cand.sort(key=lambda c: c[2], reverse=True)
    cand = cand[:1200]
    assert len(got) == len(cand) == 1200
    np.testing.assert_allclose([r["score"] for r in got], [c[2] for c in cand], rtol=0, atol=2e-6)
    assert sum((r["source"], r["target"]) == (c[0], c[1]) for r, c in zip(got, cand)) >= 1190     # near-ties may swap


@pytest.mark.parametrize("n,d,flags,rw", [(6000, 256, _hip.F_L2NORM, 0.0), (3000, 512, _hip.F_L2NORM | _hip.F_RESIDUAL | _hip.F_BLEND_ANY, 0.3),
                                          (2000, 260, 0, 0.0), (2500, 1024, _hip.F_L1NORM, 0.0), (1800, 64, _hip.F_L2NORM, 0.0)])
def test_fused_attention_spmm_against_the_reference_formula(n, d, flags, rw):
    """cleora_propagate_attention_dev — weights and weighted sum in one pass, softmax accumulated online — against a numpy fp64
    restatement of pycleora/__init__.py:241-270 (weights in f64, `weighted_adj @ embeddings`), and against the two-kernel route
    (cleora_edge_attention_dev + cleora_propagate_vals_dev) it replaces: rows of every length class (empty, a few edges, several
    64-edge chunks, a 700-edge row).  Stated: 2e-5 absolute on the epilogue's output (unit rows / blended rows of norm <= 1)."""
    from tests.graphs import random_csr
    rowptr, col, vl, vs = random_csr(n, 10, seed=n + d, empty_frac=0.05, hubs=[(3, 700), (40, 130)])
    rowptr64, col64, adj = rowptr.astype(np.int64), col.astype(np.int64), vl.astype(np.float64)
    x = np.random.default_rng(d).standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    temp = 0.7
    rows = np.repeat(np.arange(n), np.diff(rowptr64))
    xn = x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-10)
    score = np.sum(xn[rows].astype(np.float64) * xn[col64], axis=1) / temp
    mx = np.full(n, -np.inf)
    np.maximum.at(mx, rows, score)
    ex = np.exp(score - mx[rows])
    a = ex / np.maximum(np.bincount(rows, weights=ex, minlength=n), 1e-10)[rows] * adj
    w = a / np.maximum(np.bincount(rows, weights=a, minlength=n), 1e-10)[rows]
    y = np.zeros((n, d))
    np.add.at(y, rows, w[:, None] * x[col64].astype(np.float64))
    y = y.astype(np.float32)
    if flags & _hip.F_RESIDUAL:
        y = (np.float32(1 - rw) * y + np.float32(rw) * x).astype(np.float32)
    if flags & _hip.F_L2NORM:
        y = y / np.maximum(np.linalg.norm(y, axis=1, keepdims=True), 1e-10)
    if flags & _hip.F_L1NORM:
        y = y / np.maximum(np.abs(y).sum(axis=1, keepdims=True), 1e-10)
    g = _hip.Graph.from_host(rowptr, col, vl)
    L = _hip.lib()
    dx = _hip.DevArray.from_host(x)
    dy, dy2 = _hip.DevArray((n, d), np.float32), _hip.DevArray((n, d), np.float32)
    dv = _hip.DevArray((col.shape[0],), np.float32)
    _hip.check(L.cleora_propagate_attention_dev(g.handle, _hip.LEFT, dx.ptr, d, d, temp, dy.ptr, d, flags, rw, dx.ptr, None, None))
    _hip.check(L.cleora_edge_attention_dev(g.handle, _hip.LEFT, dx.ptr, d, d, temp, dv.ptr, None))
    _hip.check(L.cleora_propagate_vals_dev(g.handle, dv.ptr, dx.ptr, d, d, dy2.ptr, d, flags, rw, dx.ptr, None, None, None))
    _hip.check(L.cleora_stream_sync(None))
    got, two = dy.to_host(), dy2.to_host()
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, y, rtol=0, atol=2e-5)
    np.testing.assert_allclose(got, two, rtol=0, atol=2e-5)
    g.close()
    # a shape the fused kernel does not take is refused, not mangled
    with pytest.raises(ValueError):
        _hip.check(L.cleora_propagate_attention_dev(_hip.Graph.from_host(rowptr, col, vl).handle, _hip.LEFT, dx.ptr, d, 30, temp, dy.ptr, d, 0,
                                                    0.0, None, None, None))
