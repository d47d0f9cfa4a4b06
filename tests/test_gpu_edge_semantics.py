"""GPU: the corners of embed() where the reference's Python loop and its Rust loop differ,
against outputs of the reference's own embed() / whiten_embeddings on karate club
(tests/golden/edge_semantics_ref.npz, written by tests/golden/make_golden.py in the build container):

  residual_weight >= 1   the Python loop blends for any rw > 0 (pycleora/__init__.py:111-115); the Rust loop only
                         for 0 < rw < 1 (src/embedding.rs:116) — CLEORA_F_BLEND_ANY selects the former
  normalization          'l1' (:947-950) and 'none' (:957-958) run on the device (CLEORA_F_L1NORM)
  early stop             f64 RMSE like _compute_rmse (:974-976, CLEORA_F_SQDIFF64), same stopping iteration
  n_components           whiten_embeddings slices `[:n_components]` (:151-153): 0 -> no columns, negative -> from the end

Tolerances: the reference normalises with numpy (pairwise f32 sum, true division), the device in index order —
last-ulp differences per iteration, 5e-6 relative to the largest element after <= 8 iterations; whitened outputs are
compared up to column sign and through their pairwise-cosine matrix (eigenvectors of close eigenvalues rotate)."""
import ctypes
import os
import types

import numpy as np
import pytest

from cleora_amd import _hip
from cleora_amd import embed as dev_embed
from cleora_amd.pycleora import SparseMatrix

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fx(golden_dir):
    k = np.load(os.path.join(golden_dir, "karate_ref.npz"))
    g = SparseMatrix.from_iterator(iter(str(s) for s in k["edges"]), str(k["columns"]))
    return np.load(os.path.join(golden_dir, "edge_semantics_ref.npz")), g


def close(got, want, rel=5e-6):
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= rel * np.abs(want).max()


def cosine_matrix(e):
    e = e.astype(np.float64)
    e = e / np.maximum(np.linalg.norm(e, axis=1, keepdims=True), 1e-300)
    return e @ e.T


def test_l1_and_none_normalisation_on_the_device(fx):
    k, g = fx
    close(dev_embed.embed(g, 16, 6, normalization="l1", whiten=False), k["l1_nowhiten"])
    close(dev_embed.embed(g, 16, 4, normalization="none", whiten=False), k["none_nowhiten"])
    got = dev_embed.embed(g, 16, 6, normalization="l1")
    assert np.abs(cosine_matrix(got) - cosine_matrix(k["l1_whiten"])).max() < 1e-4
    # rows of an L1-normalised iterate sum to one in absolute value
    x = dev_embed.embed(g, 24, 3, normalization="l1", whiten=False)
    np.testing.assert_allclose(np.abs(x).sum(axis=1), 1.0, rtol=0, atol=2e-6)


def test_residual_weight_at_and_above_one_blends_like_the_python_loop(fx):
    k, g = fx
    noop = lambda i, e: None
    close(dev_embed.embed(g, 16, 6, residual_weight=1.0, whiten=False, callback=noop), k["rw10_nowhiten"])
    close(dev_embed.embed(g, 16, 6, residual_weight=1.5, whiten=False, callback=noop), k["rw15_nowhiten"])
    close(dev_embed.embed(g, 16, 5, propagation="symmetric", normalization="l1", residual_weight=1.5, whiten=False),
          k["rw15_sym_l1"])
    got = dev_embed.embed(g, 16, 6, residual_weight=1.5)
    assert np.abs(cosine_matrix(got) - cosine_matrix(k["rw15_whiten"])).max() < 1e-4
    # the same loop as ONE C-ABI call (cleora_embed + CLEORA_F_WHITEN)
    n = g.num_entities
    out = np.empty((n, 16), np.float32)
    it = ctypes.c_uint64(0)
    with g._lock:
        _hip.check(_hip.lib().cleora_embed(g._graph().handle, _hip.ptr(g._arr["hashes"]), None, _hip.LEFT, 16, 6, 0,
                                           1.5, 0.0, _hip.F_WHITEN, _hip.ptr(out), ctypes.byref(it)))
    assert it.value == 6
    assert np.abs(cosine_matrix(out) - cosine_matrix(k["rw15_whiten"])).max() < 1e-4
    # the all-native fast path keeps the RUST gate: rw >= 1 does not blend (src/embedding.rs:116)
    np.testing.assert_array_equal(g.embed_fast(16, 4, residual_weight=1.5), g.embed_fast(16, 4, residual_weight=0.0))


def test_early_stop_iteration_matches_the_python_loop(fx):
    k, g = fx
    seen = []
    got = dev_embed.embed(g, 16, 40, convergence_threshold=0.02, whiten=False, callback=lambda i, e: seen.append(i))
    assert len(seen) == int(k["conv_nowhiten_iters"][0])
    close(got, k["conv_nowhiten"])


def test_whiten_n_components_slicing(fx):
    k, _ = fx
    x = k["nc_x"]
    assert dev_embed.whiten_embeddings(x, n_components=0).shape == (300, 0)          # was a heap overflow
    for key, nc in (("nc_m3", -3), ("nc_40", 40)):
        got, want = dev_embed.whiten_embeddings(x, n_components=nc), k[key]
        assert got.shape == want.shape
        s = np.sign((got * want).sum(axis=0))
        assert np.abs(got * s - want).max() <= 2e-4 * np.abs(want).max()


def test_accelerate_forwards_only_spectral(fx):
    """cleora_amd.accelerate() rebinds pycleora.embed: every normalisation but 'spectral' stays on the device."""
    import cleora_amd
    k, g = fx
    calls = []
    pkg = types.ModuleType("pycleora_stand_in")
    pkg.embed = lambda graph, *a, **kw: calls.append(kw.get("normalization")) or "original"
    pkg.whiten_embeddings = None
    cleora_amd.accelerate(pkg)
    assert pkg.embed(g, 16, 2, normalization="spectral") == "original" and calls == ["spectral"]
    close(pkg.embed(g, 16, 6, normalization="l1", whiten=False), k["l1_nowhiten"])
    assert calls == ["spectral"]
