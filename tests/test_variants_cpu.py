"""CPU: the HOST side of cleora_amd/variants.py (who is connected to whom, with which weight; argument
checks; output assembly) against the reference's own outputs (tests/golden/variants_ref.npz), with the device
loop replaced by the CPU oracle (oracle SpMM + the numpy restatement of _postprocess_iteration).  The device
loop itself is covered by tests/test_gpu_variants.py; this file runs without a GPU.

Tolerances: whiten=False 2e-5 absolute on unit rows (f64 scipy SpMM in the reference vs the oracle's f32
order); whiten=True: sign-aligned columns 5e-3 * max|ref| and pairwise cosines 1e-4 (as in the GPU tests).
"""
import os

import numpy as np
import pytest

import oracle
from cleora_amd import _hip, variants
from cleora_amd.pycleora import SparseMatrix
from oracle import whiten as ow


def fake_loop_csr(rowptr, col, val, x0, iters, normalization, whiten, snapshots=None, temperature=None, adj=None):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.uint64)
    col = np.ascontiguousarray(col, dtype=np.uint32)
    val = np.ascontiguousarray(val, dtype=np.float32)
    emb = np.asarray(x0, dtype=np.float32)
    n = emb.shape[0]
    rows = np.repeat(np.arange(n), np.diff(rowptr.astype(np.int64)))
    taken = []
    for i in range(iters):
        v = val
        if temperature is not None and i > 0:            # numpy restatement of pycleora/__init__.py:241-268
            xn = emb / np.maximum(np.linalg.norm(emb, axis=1, keepdims=True), 1e-10)
            score = np.sum(xn[rows] * xn[col.astype(np.int64)], axis=1) / temperature
            mx = np.full(n, -np.inf)
            np.maximum.at(mx, rows, score)
            ex = np.exp(score - mx[rows])
            a = ex / np.maximum(np.bincount(rows, weights=ex, minlength=n), 1e-10)[rows] * val.astype(np.float64)
            v = (a / np.maximum(np.bincount(rows, weights=a, minlength=n), 1e-10)[rows]).astype(np.float32)
        emb = oracle.spmm(rowptr, col, v, emb)
        if normalization == "l2":
            emb = ow.normalize_l2(emb)
        if whiten:
            emb = ow.whiten_embeddings(emb)
        if snapshots is not None and (i + 1) in snapshots:
            taken.append(emb.copy())
    return taken if snapshots is not None else emb


@pytest.fixture()
def cpu_device(monkeypatch):
    """Replace the three device entry points variants.py uses by oracle-backed ones."""
    def loop(g, n, x0, kind, iters, normalization, callback, rw, conv, whiten, snapshots=None,
             attention_temperature=None):
        a = g._arr
        val = a["val_left"] if kind == _hip.LEFT else a["val_sym"]
        assert callback is None and rw == 0.0 and conv == 0.0
        return fake_loop_csr(a["rowptr"], a["col"], val, x0, iters, normalization, whiten, snapshots, attention_temperature)

    def embed_csr(rowptr, col, val, x0, num_iterations, normalization="l2", callback=None, residual_weight=0.0,
                  convergence_threshold=0.0, whiten=True, device=0):
        return fake_loop_csr(rowptr, col, val, x0, num_iterations, normalization, whiten)

    def embed(graph, feature_dim=256, num_iterations=40, propagation="left", normalization="l2", num_workers=None,
              whiten=True, **kw):
        assert not kw
        a = graph._arr
        x0 = oracle.init(a["hashes"], feature_dim, 0)
        val = a["val_left"] if propagation == "left" else a["val_sym"]
        if not whiten:                                       # the reference takes its fast (Rust-order) path here
            return oracle.embed(a["rowptr"], a["col"], val, x0, num_iterations, 0.0)[0]
        return fake_loop_csr(a["rowptr"], a["col"], val, x0, num_iterations, normalization, whiten)

    monkeypatch.setattr(variants, "_device_loop", loop)
    monkeypatch.setattr(variants, "embed_csr", embed_csr)
    monkeypatch.setattr(variants, "embed", embed)
    monkeypatch.setattr(SparseMatrix, "_graph", lambda self: self)
    # initialize_deterministically is a device call in the drop-in; the oracle's init is bit-identical
    monkeypatch.setattr(SparseMatrix, "initialize_deterministically",
                        lambda self, d, seed=0: oracle.init(self._arr["hashes"], d, seed))


@pytest.fixture(scope="module")
def ref(golden_dir):
    k = np.load(os.path.join(golden_dir, "karate_ref.npz"))
    v = np.load(os.path.join(golden_dir, "variants_ref.npz"))
    edges = [str(s) for s in k["edges"]]
    return k, v, edges, str(k["columns"]), SparseMatrix.from_iterator(iter(edges), str(k["columns"]))


def cosine_matrix(e):
    e = e.astype(np.float64)
    e = e / np.maximum(np.linalg.norm(e, axis=1, keepdims=True), 1e-300)
    return e @ e.T


def assert_close(got, want, whitened, blocks=1, lead=None):
    assert got.shape == want.shape
    if not whitened:
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)
        return
    w = want.shape[1] // blocks
    for b in range(blocks):
        a, r = got[:, b * w:(b + 1) * w][:, :lead], want[:, b * w:(b + 1) * w][:, :lead]
        s = np.sign((a * r).sum(axis=0))
        assert np.abs(a * s - r).max() <= 5e-3 * np.abs(r).max()
        assert np.abs(cosine_matrix(a) - cosine_matrix(r)).max() < 1e-4


@pytest.mark.parametrize("tag,wh", [("w", True), ("n", False)])
def test_host_side_of_the_variants(ref, cpu_device, tag, wh):
    k, v, edges, columns, g = ref
    assert_close(variants.embed_multiscale(g, 8, scales=[2, 5, 3], whiten=wh), v[f"multiscale_{tag}"], wh, blocks=3)
    graph, got = variants.embed_weighted(list(zip(edges, v["weights"].tolist())), columns, 8, 5,
                                         propagation="symmetric", whiten=wh)
    assert graph.entity_ids == g.entity_ids
    assert_close(got, v[f"weighted_{tag}"], wh)
    assert_close(variants.embed_directed(edges, columns, 8, 5, whiten=wh)[1], v[f"directed_{tag}"], wh, lead=6)
    assert_close(variants.embed_with_attention(g, 8, 4, attention_temperature=0.7, whiten=wh), v[f"attention_{tag}"], wh)
    feats = dict(zip([str(s) for s in v["feat_keys"]], v["feats"]))
    got = variants.embed_edge_features(g, feats, 8, 3, whiten=wh)
    assert_close(got[:, :8], v[f"edgefeat_{tag}"][:, :8], wh)
    assert_close(got[:, 8:], v[f"edgefeat_{tag}"][:, 8:], wh)
    mean = variants.embed_edge_features(g, feats, 8, 3, combine="mean", whiten=wh)
    np.testing.assert_allclose(mean, (got[:, :3] + got[:, 8:]) / 2.0, rtol=0, atol=1e-7)


def test_argument_checks_need_no_device(ref):
    k, v, edges, columns, g = ref
    with pytest.raises(ValueError, match="scales must be"):
        variants.embed_multiscale(g, 8, scales=[3, -1])
    with pytest.raises(ValueError, match="Unknown propagation type"):
        variants.embed_multiscale(g, 8, propagation="right")
    with pytest.raises(ValueError, match="attention_temperature must be positive"):
        variants.embed_with_attention(g, 8, 3, attention_temperature=-1.0)
    with pytest.raises(ValueError, match="num_iterations must be positive"):
        variants.embed_with_attention(g, 8, 0)
    with pytest.raises(ValueError, match="normalization"):
        variants.embed_directed(edges, columns, 8, 2, normalization="spectral")
