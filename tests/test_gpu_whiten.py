"""GPU parity of the whitening kernels (pycleora/__init__.py:130-164) through the C ABI.

Tolerances (stated, floating point):
  column sums   f64, fixed-order tree vs numpy pairwise:   rel 1e-13
  centred Gram  f64 MFMA vs numpy f64 dgemm:               ||G - G_ref||_F / ||G_ref||_F <= 1e-13
  projection    f32 MFMA vs numpy sgemm:                   |delta| <= 2e-6 * sum_k |a_k b_k| (+1e-30)
  whiten_embeddings end to end vs the reference's own output (tests/golden/whiten_ref.npz):
                columns compared after sign alignment (eigh's sign is arbitrary),
                |delta| <= 2e-4 * max|ref|   (f32 projection through 1/sqrt(lambda) scaling)
"""
import os

import numpy as np
import pytest

from cleora_amd import _hip
from cleora_amd import embed as dev_embed
from oracle import whiten as ow

pytestmark = pytest.mark.gpu


def case_input(seed, n, d):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((n, d)) * np.linspace(0.5, 3.0, d) + rng.standard_normal(d)).astype(np.float32)


def sign_align(a, ref):
    s = np.sign((a * ref).sum(axis=0))
    s[s == 0] = 1
    return a * s


@pytest.mark.parametrize("n,d", [(500, 16), (257, 64), (3000, 32), (60001, 8), (1000, 100), (777, 130),
                                  (5000, 256), (900, 300), (4096, 7), (300, 513)])
def test_colsum_and_gram(n, d):
    x = case_input(n + d, n, d)
    L = _hip.lib()
    dx = _hip.DevArray.from_host(x)
    w = dev_embed.DeviceWhitener(n, d)
    mean, cov = w.stats(dx.ptr, d)
    mean_ref, cov_ref = ow.whiten_stats(x)
    np.testing.assert_allclose(mean, mean_ref, rtol=1e-13, atol=1e-300)
    assert np.array_equal(cov, cov.T)
    assert np.linalg.norm(cov - cov_ref) <= 1e-13 * np.linalg.norm(cov_ref)


@pytest.mark.parametrize("n,d,k", [(500, 16, 16), (1000, 100, 100), (3000, 256, 256), (700, 130, 17),
                                    (513, 64, 200), (2000, 33, 5)])
def test_projection(n, d, k):
    rng = np.random.default_rng(n + d + k)
    x = rng.standard_normal((n, d)).astype(np.float32)
    mean = rng.standard_normal(d).astype(np.float32)
    t = rng.standard_normal((d, k)).astype(np.float32)
    L = _hip.lib()
    dx, dm, dt = _hip.DevArray.from_host(x), _hip.DevArray.from_host(mean), _hip.DevArray.from_host(t)
    out = _hip.DevArray((n, k), np.float32)
    L.cleora_memset(out.ptr, 0xFF, out.nbytes, None)
    _hip.check(L.cleora_project_dev(dx.ptr, d, n, d, dm.ptr, dt.ptr, k, out.ptr, k, None))
    _hip.check(L.cleora_stream_sync(None))
    got = out.to_host()
    blk = x - mean
    want = blk.astype(np.float64) @ t.astype(np.float64)
    bound = np.abs(blk).astype(np.float64) @ np.abs(t).astype(np.float64)
    assert np.all(np.abs(got - want) <= 2e-6 * bound + 1e-30)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_whiten_vs_reference_golden(tag, golden_dir):
    ref = np.load(os.path.join(golden_dir, "whiten_ref.npz"))
    seed, n, d, step = (int(v) for v in ref[f"{tag}_seed"])
    x = case_input(seed, n, d)
    got = dev_embed.whiten_embeddings(x)[::step]
    want = ref[f"{tag}_whiten"]
    got = sign_align(got, want)
    assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max()


def test_whiten_truncated_and_degenerate(golden_dir):
    ref = np.load(os.path.join(golden_dir, "whiten_ref.npz"))
    x = np.random.default_rng(15).standard_normal((400, 24)).astype(np.float32)
    got = dev_embed.whiten_embeddings(x, n_components=8)
    want = ref["trunc_whiten_k8"]
    assert got.shape == (400, 8)
    assert np.abs(sign_align(got, want) - want).max() <= 2e-4 * np.abs(want).max()
    one = np.ones((1, 5), np.float32)
    np.testing.assert_array_equal(dev_embed.whiten_embeddings(one), one)   # n <= 1: copy (:132-133)


def test_whiten_property_identity_covariance_at_scale():
    """Size-independent property at a size the CPU oracle would not finish quickly:
    the whitened matrix has zero mean and identity covariance."""
    n, d = 400_000, 256
    rng = np.random.default_rng(5)
    mix = rng.standard_normal((d, d)).astype(np.float32) / np.sqrt(d)
    x = (rng.standard_normal((n, d)).astype(np.float32) @ mix + 0.5).astype(np.float32)
    out = dev_embed.whiten_embeddings(x)
    mean = out.mean(axis=0, dtype=np.float64)
    cov = (out.astype(np.float64) - mean).T @ (out.astype(np.float64) - mean) / (n - 1)
    assert np.abs(mean).max() < 2e-4
    assert np.abs(cov - np.eye(d)).max() < 2e-3
