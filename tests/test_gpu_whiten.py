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


@pytest.mark.parametrize("d,k", [(16, 16), (64, 10), (256, 256), (130, 130), (513, 40)])
def test_whiten_transform_dev_vs_lapack(d, k):
    """cleora_whiten_transform_dev (rocSOLVER dsyevd + scaling on the device) vs np.linalg.eigh, the routine
    the reference calls: eigenvalues rel 1e-11 of the largest; transform columns after sign alignment
    |delta| <= 1e-4 * max|column| for well-separated eigenvalues (columns of close eigenvalues rotate)."""
    rng = np.random.default_rng(d * 7 + k)
    n = 5000
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    lam = np.geomspace(1.0, 1e-3, d) * (1 + 0.3 * rng.random(d))
    lam[::7] *= 3.0                                        # spread, keep gaps
    cov = (q * lam) @ q.T
    cov = (cov + cov.T) / 2
    gram = np.ascontiguousarray(cov * (n - 1))
    L = _hip.lib()
    dg = _hip.DevArray.from_host(gram)
    dt = _hip.DevArray((d, k), np.float32)
    de = _hip.DevArray((d,), np.float64)
    ws = _hip.DevArray((L.cleora_eigh_workspace(d),), np.uint8)
    _hip.check(L.cleora_whiten_transform_dev(dg.ptr, n, d, k, dt.ptr, de.ptr, ws.ptr, None))
    _hip.check(L.cleora_stream_sync(None))
    np.testing.assert_array_equal(dg.to_host(), gram)      # input is not modified
    w, v = np.linalg.eigh(cov)
    w, v = w[::-1], v[:, ::-1]
    got_w, got_t = de.to_host(), dt.to_host()
    assert np.abs(got_w - w).max() <= 1e-11 * w[0]
    want_t = (v * (1.0 / np.sqrt(np.maximum(w, 1e-10))))[:, :k].astype(np.float32)
    gaps = np.minimum(np.abs(np.diff(w, prepend=np.inf)), np.abs(np.diff(w, append=-np.inf)))[:k] / w[:k]
    ok = gaps > 1e-3
    got_a = sign_align(got_t, want_t)
    assert ok.sum() >= k // 2
    assert np.all(np.abs(got_a - want_t)[:, ok].max(axis=0) <= 1e-4 * np.abs(want_t)[:, ok].max(axis=0))
    # the transform whitens cov whatever the rotation inside close eigenvalue groups: T^T cov T = I
    t64 = got_t.astype(np.float64)
    assert np.abs(t64.T @ cov @ t64 - np.eye(k)).max() < 5e-6


@pytest.mark.parametrize("n,d,k", [(3000, 32, None), (2000, 100, 17), (5000, 256, None), (2, 8, None), (700, 130, 130)])
def test_whiten_dev_matches_host_statistics_route(n, d, k):
    """The device-only chain (cleora_whiten_dev) vs the route with numpy's LAPACK eigh between the kernels."""
    x = case_input(3 * n + d, n, d)
    dx = _hip.DevArray.from_host(x)
    kk = d if k is None else k
    L = _hip.lib()
    w = dev_embed.DeviceWhitener(n, d)
    out = _hip.DevArray((n, kk), np.float32)
    assert w.whiten(dx.ptr, d, out.ptr, kk, k) == kk
    _hip.check(L.cleora_stream_sync(None))
    outs = [(out.to_host(), w.last_eigenvalues)]
    # the host-statistics route, assembled here from the public pieces: device mean / covariance, numpy's LAPACK
    # eigh (the routine the reference itself calls, pycleora/__init__.py:145), device projection
    mean, cov = w.stats(dx.ptr, d)
    ev, evec = np.linalg.eigh(cov)
    idx = np.argsort(ev)[::-1][:kk]
    transform = np.ascontiguousarray((evec[:, idx] / np.sqrt(np.maximum(ev[idx], 1e-10))).astype(np.float32))
    dt, dm = _hip.DevArray.from_host(transform), _hip.DevArray.from_host(mean.astype(np.float32))
    out2 = _hip.DevArray((n, kk), np.float32)
    _hip.check(L.cleora_project_dev(dx.ptr, d, n, d, dm.ptr, dt.ptr, kk, out2.ptr, kk, None))
    _hip.check(L.cleora_stream_sync(None))
    outs.append((out2.to_host(), ev[np.argsort(ev)[::-1]]))
    (a, wa), (b, wb) = outs
    assert np.abs(wa[:kk] - wb[:kk]).max() <= 1e-10 * max(wb[0], 1e-300)
    if n > d:        # full-rank covariance: columns are defined up to sign
        a = sign_align(a, b)
        assert np.abs(a - b).max() <= 2e-4 * np.abs(b).max()


def test_cleora_whiten_errors_and_single_row():
    L = _hip.lib()
    x = np.arange(6, dtype=np.float32).reshape(1, 6)
    y = np.zeros_like(x)
    _hip.check(L.cleora_whiten(_hip.ptr(x), 1, 6, 3, _hip.ptr(y)))       # n <= 1: the row comes back unchanged
    np.testing.assert_array_equal(y, x)
    _hip.check(L.cleora_whiten(None, 0, 6, 0, None))                     # empty input is a no-op
    with pytest.raises(ValueError):
        _hip.check(L.cleora_whiten(_hip.ptr(x), 1, 0, 0, _hip.ptr(y)))
    dx = _hip.DevArray.from_host(np.ones((4, 8), np.float32))
    ws = _hip.DevArray((L.cleora_whiten_workspace(4, 8),), np.uint8)
    with pytest.raises(ValueError, match="alias"):
        _hip.check(L.cleora_whiten_dev(dx.ptr, 8, 4, 8, 0, dx.ptr, 8, ws.ptr, None, None))


@pytest.mark.parametrize("n,d,iters,rw,kind", [
    # the d x d step of an intermediate iteration: on one host core up to d = 256 (tiny, odd, one below the limit, the limit) ...
    (20_000, 64, 6, 0.0, 0), (6000, 256, 4, 0.3, 1), (3000, 32, 5, 1.5, 0), (500, 8, 4, 0.0, 0), (4000, 33, 4, 0.0, 0),
    (3000, 255, 3, 0.0, 1), (2500, 1, 3, 0.0, 0), (3000, 130, 3, 0.0, 0),
    # ... rocSOLVER's potrf / trtri beyond
    (3000, 320, 3, 0.0, 0), (5000, 512, 3, 0.2, 0)])
def test_overlapped_whitened_loop_equals_the_sequential_order(n, d, iters, rw, kind):
    """cleora_embed + CLEORA_F_WHITEN without a convergence test runs SpMM(t+1) beside Gram / eigh(t), taking the SpMM
    before the projection (A ((Y - mu) T) = (A Y - (A 1) mu^T) T).  With a (never met) convergence threshold the same
    call keeps the reference's sequential order: both must agree to f32 rounding — columns up to sign, 2e-3 relative
    after several whitenings, and the pairwise cosines of a row sample to 1e-4.  Also against the numpy oracle loop."""
    import ctypes
    from tests.graphs import random_csr
    import oracle
    rowptr, col, vl, vs = random_csr(n, 9, seed=n + d, empty_frac=0.02, hubs=[(13, 1400)])
    g = _hip.Graph.from_host(rowptr, col, vl, vs)
    x0 = np.random.default_rng(d).standard_normal((n, d)).astype(np.float32)
    L = _hip.lib()
    outs = []
    for thr in (0.0, 1e-30):
        out = np.empty((n, d), np.float32)
        ran = ctypes.c_uint64(0)
        _hip.check(L.cleora_embed(g.handle, None, _hip.ptr(x0), kind, d, iters, 0, rw, thr, _hip.F_WHITEN, _hip.ptr(out),
                                  ctypes.byref(ran)))
        assert ran.value == iters
        outs.append(out)
    a, b = outs
    assert np.isfinite(a).all()
    a = sign_align(a, b)
    assert np.abs(a - b).max() <= 2e-3 * np.abs(b).max()
    rows = np.random.default_rng(1).choice(n, 400, replace=False)
    cos = lambda e: (lambda u: u @ u.T)(e[rows].astype(np.float64) / np.linalg.norm(e[rows].astype(np.float64), axis=1, keepdims=True))
    assert np.abs(cos(a) - cos(b)).max() < 1e-4
    # the same loop through the device-pointer entry point
    dx = _hip.DevArray.from_host(x0)
    _hip.check(L.cleora_embed_dev(g.handle, dx.ptr, kind, d, iters, rw, 0.0, _hip.F_WHITEN, None))
    np.testing.assert_array_equal(dx.to_host(), outs[0])
    if rw < 1.0 and rw == 0.0:
        val = vl if kind == 0 else vs
        want, _ = ow.embed_slow(lambda v: oracle.spmm(rowptr, col, val, v), x0, iters, whiten=True)
        assert np.abs(cos(outs[0]) - cos(want)).max() < 1e-3
    g.close()


@pytest.mark.parametrize("d", [256, 320])                 # the host route (d <= 256) and rocSOLVER's potrf + trtri
def test_cholesky_guard_implies_the_reference_clamp(d):
    """Intermediate iterations may take the Cholesky whitening only when the reference's clamp max(lambda, 1e-10)
    (pycleora/__init__.py:155) is provably inactive.  The smallest pivot of the factor only bounds lambda_min from ABOVE
    (a covariance with lambda_min = 5e-11 and every squared pivot >= 1e-8 exists: below); the guard therefore also
    requires trace(cov^-1) = ||L^-T||_F^2 <= 1e10, which implies lambda_min >= 1e-10.  Checked on synthetic
    covariances Q diag(lambda) Q^T through cleora_whiten_transform_any_dev:
      lambda_min = 1e-9  -> Cholesky form (form = 1), T^T C T = I
      lambda_min = 5e-11 -> PCA form (form = 0) with the clamp: T^T C T = diag(1, ..., 1, 0.5)."""
    import ctypes
    L = _hip.lib()
    n = 10_000
    rng = np.random.default_rng(3)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    ws = _hip.DevArray((L.cleora_eigh_workspace(d),), np.uint8)
    dt = _hip.DevArray((d, d), np.float32)
    for lam_min, want_form in ((1e-9, 1), (5e-11, 0)):
        lam = np.sort(rng.uniform(0.5, 2.0, d) / d)[::-1].copy()
        lam[-1] = lam_min
        cov = (q * lam) @ q.T
        cov = (cov + cov.T) / 2
        piv = np.diag(np.linalg.cholesky(cov)) ** 2
        if want_form == 0:
            assert piv.min() >= 1e-8, "the case must pass the pivot test alone (it is the case the old guard let through)"
        dg = _hip.DevArray.from_host(np.ascontiguousarray(cov * (n - 1)))
        form = ctypes.c_int(-1)
        _hip.check(L.cleora_whiten_transform_any_dev(dg.ptr, n, d, dt.ptr, ws.ptr, None, ctypes.byref(form)))
        _hip.check(L.cleora_stream_sync(None))
        assert form.value == want_form, (lam_min, form.value, piv.min())
        t = dt.to_host().astype(np.float64)
        m = t.T @ cov @ t
        if want_form == 1:
            assert np.abs(m - np.eye(d)).max() <= 2e-3          # f32 transform of a 1e7-conditioned matrix
        else:
            dm = np.diag(m)
            assert np.abs(dm[:-1] - 1).max() <= 2e-3 and abs(dm[-1] - 0.5) <= 2e-3      # lambda_min / 1e-10
            assert np.abs(m - np.diag(dm)).max() <= 2e-3


@pytest.mark.parametrize("n,d", [(70_001, 256), (40_003, 512), (30_001, 1024), (5_003, 256), (300_007, 256), (1_100_003, 256), (1_050_001, 512)])
def test_intermediate_gram_from_the_bf16_matrix_cores(n, d):
    """cleora_whiten_stats_dev(intermediate = 1) at d = 256 S: the centred Gram from the bf16 matrix cores with split f32 operands
    (gram16_kernel; the matrix cores sum 32 rows, f32 sums over <= 2048 rows, f64 across; S diagonal super-tile blocks and
    S (S - 1) off-diagonal ones) against the f64 form (intermediate = 0) and numpy fp64.  Two forms:
      fewer than 2^20 rows: six v_mfma_f32_32x32x16_bf16 per f32 product (three-way split) — stated <= 5e-7 of the Gram's Frobenius norm;
      from 2^20 rows:       THREE — (1,1) (1,2) (2,1) of a two-way split — plus the systematic r1^2 term of the diagonal accumulated on
                            the vector unit; what they drop is random-signed per row and falls as sqrt(d) 2^-18 / sqrt(n) — stated <= 1e-7.
    Both: every diagonal entry to the same tolerance of the largest, and NO systematic bias of the variances: |mean relative diagonal
    error| <= 2e-8 (dropping the r1^2 term would show as -6e-7; a long f32 accumulation chain on the bf16 MFMA as -1.3e-6: docs/history.md §3.5).
    Mean: 1e-12 for the f64 form; the split forms centre in f32 (y = x - c32, one rounding of 3e-8 |y|): <= 1e-8."""
    L = _hip.lib()                                             # n: not a multiple of the 32-row stage or of the slice count
    rng = np.random.default_rng(8)
    x = (rng.standard_normal((n, d)) * np.linspace(0.3, 2.0, d) + rng.standard_normal(d) * 0.2).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    dx = _hip.DevArray.from_host(x)
    ws = _hip.DevArray((L.cleora_whiten_workspace(n, d),), np.uint8)
    dm, dg = _hip.DevArray((d,), np.float64), _hip.DevArray((d, d), np.float64)
    x64 = x.astype(np.float64)
    mean = x64.mean(axis=0)
    gram = (x64 - mean).T @ (x64 - mean)
    measured = {}
    for intermediate, tol in ((0, 1e-12), (1, 1e-7 if n >= (1 << 20) else 5e-7)):
        _hip.check(L.cleora_whiten_stats_dev(dx.ptr, d, n, d, ws.ptr, intermediate, dm.ptr, dg.ptr, None))
        _hip.check(L.cleora_stream_sync(None))
        gm, gg = dm.to_host(), dg.to_host()
        rel = float(np.linalg.norm(gg - gram) / np.linalg.norm(gram))
        bias = float(((np.diag(gg) - np.diag(gram)) / np.diag(gram)).mean())
        measured[intermediate] = (rel, bias)
        assert np.abs(gm - mean).max() <= (1e-8 if intermediate else 1e-12)
        assert rel <= tol, (intermediate, rel)
        assert np.abs(np.diag(gg) - np.diag(gram)).max() <= tol * np.diag(gram).max()
        assert abs(bias) <= (2e-8 if intermediate else 1e-12), (intermediate, bias)
        np.testing.assert_array_equal(gg, gg.T)
    try:
        import json
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r04_gram_error.jsonl")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(json.dumps({"n": n, "d": d, "f64_rel_frobenius": measured[0][0], "split_rel_frobenius": measured[1][0],
                                "split_mean_rel_diag_bias": measured[1][1]}) + "\n")
    except OSError:
        pass


def test_overlapped_loop_recomputes_the_statistics_when_the_guard_refuses_them():
    """The intermediate iterations take their statistics from the bf16 matrix cores (~1e-8 of the f64 Gram) and the Cholesky
    transform — but only while sum_i (trace(cov) / d) / lambda_i <= 1e4 (csrc/eigh.hip kMaxRelativeSpread, ADVICE round 3: in a direction of variance
    lambda such a Gram is off by ~2e-8 (trace / d) / lambda).  A start whose last 16 columns are scaled by 1e-3 gives the FIRST
    whitening a covariance with 16 eigenvalues near 1e-6 of the rest: the clamp (1e-10) is inactive, the relative spread is 1.6e7 —
    the loop must take that iteration's statistics again in f64 and the PCA form, and still agree with the reference's order."""
    import ctypes
    from tests.graphs import random_csr
    n, d, iters = 6000, 256, 4
    rowptr, col, vl, vs = random_csr(n, 9, seed=77, empty_frac=0.0, hubs=[(13, 1400)])
    g = _hip.Graph.from_host(rowptr, col, vl, vs)
    x0 = np.random.default_rng(5).standard_normal((n, d)).astype(np.float32)
    x0[:, -16:] *= 1e-3
    L = _hip.lib()
    outs = []
    for thr in (0.0, 1e-30):
        out = np.empty((n, d), np.float32)
        _hip.check(L.cleora_embed(g.handle, None, _hip.ptr(x0), 0, d, iters, 0, 0.0, thr, _hip.F_WHITEN, _hip.ptr(out), None))
        outs.append(out)
    a, b = outs
    assert np.isfinite(a).all()
    rows = np.random.default_rng(1).choice(n, 400, replace=False)
    cos = lambda e: (lambda u: u @ u.T)(e[rows].astype(np.float64) / np.linalg.norm(e[rows].astype(np.float64), axis=1, keepdims=True))
    assert np.abs(cos(a) - cos(b)).max() < 1e-4
    assert np.abs(np.cov(a.astype(np.float64).T) - np.eye(d)).max() < 5e-3
    g.close()


@pytest.mark.parametrize("n,d,k", [(50_000, 256, 256), (40_000, 1024, 1024), (36_000, 64, 64), (40_000, 256, 100), (20_000, 256, 100),
                                   (9_000, 1024, 1024), (5_000, 96, 96),
                                   (3_000, 4096, 128),       # a row of 16 KiB: the block asks for > 64 KiB of LDS
                                   (2_000, 40, 24)])         # d % 32 != 0: the tiled f32-MFMA kernel
def test_projection_error_against_an_f64_product(n, d, k):
    """The projection computes every f32 product from six bf16 MFMAs (three-way split operands, whiten.hip) for d % 32 == 0, on the
    f32 matrix cores otherwise.  Against an f64 product of the same f32 inputs: maximum error <= 2e-6 of max|out| and RMS error
    <= 3e-7 of it (x sqrt(d / 1024) beyond d = 1024) — the class of the numpy sgemm of pycleora/__init__.py:163 (~1e-7 of max|out| at d = 256; round 3 measured the
    split form at or below the f32 matrix cores' own error on these shapes, profiles/r03a_kernel_probes.jsonl).
    >= 32 768 rows take 128-row tiles, fewer 64-row tiles."""
    L = _hip.lib()
    rng = np.random.default_rng(d + k)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    mean = x.mean(axis=0).astype(np.float32)
    t = (rng.standard_normal((d, k)) * np.sqrt(d)).astype(np.float32)
    dx, dm, dt = (_hip.DevArray.from_host(a) for a in (x, mean, t))
    do = _hip.DevArray((n, k), np.float32)
    _hip.check(L.cleora_project_dev(dx.ptr, d, n, d, dm.ptr, dt.ptr, k, do.ptr, k, None))
    _hip.check(L.cleora_stream_sync(None))
    got = do.to_host().astype(np.float64)
    ref = (x - mean).astype(np.float64) @ t.astype(np.float64)
    emax = float(np.abs(got - ref).max() / np.abs(ref).max())
    erms = float(np.sqrt(((got - ref) ** 2).mean()) / np.abs(ref).max())
    grow = max(1.0, (d / 1024) ** 0.5)                    # f32 rounding of a d-term sum grows like sqrt(d): measured 2.07e-6 / 2.07e-7 at d = 4096
    assert emax <= 2e-6 * grow and erms <= 3e-7 * grow, (emax, erms)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 127, 5003, 16_384, 70_001])
@pytest.mark.parametrize("norm,scaled", [(1, True), (0, False), (2, True), (1, False)])
def test_bounded_projection_on_the_f16_matrix_cores(n, norm, scaled):
    """cleora_project_bounded_dev at d = k = 256 (csrc/project_f16.hip: the intermediate projections of the whitened loop — operands
    bounded row by row, every f32 product from three f16 MFMAs of two-way split operands scaled by powers of two, the transform
    resident in registers) against an f64 product of the same f32 inputs: row bounds over five decades (what symmetric Markov values or
    trimmed hyperedges give), transform columns over five decades (1 / sqrt(eigenvalue)), all norms, with and without row scales,
    ragged last tiles.  Stated: every row within 2e-6 of its own norm (measured 5e-7; the six-product bf16 form reads 7e-7 on the
    same data, scripts/r06/project_probe.py) — the error class of the f32 GEMM of pycleora/__init__.py:163."""
    import ctypes
    L = _hip.lib()
    d = 256
    rng = np.random.default_rng(n + 7 * norm + scaled)
    bound = np.where(rng.random(n) < 0.8, 1.0, 10 ** (rng.random(n) * 4 - 1)).astype(np.float32)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    x *= (bound * (0.2 + 0.8 * rng.random(n)))[:, None].astype(np.float32)
    x = np.minimum(np.maximum(x, -bound[:, None]), bound[:, None]).astype(np.float32)
    rowscale = (bound * (2 * rng.random(n) - 1)).astype(np.float32)
    mean = np.clip(rng.standard_normal(d) * 0.05, -1, 1).astype(np.float32)
    t = (rng.standard_normal((d, d)) * 10 ** (rng.random(d) * 5 - 2)[None, :]).astype(np.float32)
    dx, dm, dt, ds, db = (_hip.DevArray.from_host(a) for a in (x, mean, t, rowscale, bound))
    do = _hip.DevArray.from_host(np.full((n + 1, d), 7.0, np.float32))          # one row more than the call may touch
    nd, form = ctypes.c_int(0), ctypes.c_int(-1)
    _hip.check(L.cleora_project_bounded_dev(dx.ptr, d, n, d, dm.ptr, dt.ptr, d, do.ptr, d, ds.ptr if scaled else None, db.ptr, norm,
                                            ctypes.byref(nd), ctypes.byref(form), None))
    _hip.check(L.cleora_stream_sync(None))
    assert form.value == 1 and nd.value == (1 if norm else 0)
    got = do.to_host()
    assert (got[n] == 7.0).all(), "the ragged last tile wrote past row n"
    got = got[:n].astype(np.float64)
    ref = (x.astype(np.float64) - (rowscale.astype(np.float64)[:, None] if scaled else 1.0) * mean.astype(np.float64)[None, :]) @ t.astype(np.float64)
    if norm == 1:
        ref /= np.maximum(np.linalg.norm(ref, axis=1, keepdims=True), 1e-10)
    elif norm == 2:
        ref /= np.maximum(np.abs(ref).sum(axis=1, keepdims=True), 1e-10)
    assert np.isfinite(got).all()
    err = np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-300)
    assert err.max() <= 2e-6, float(err.max())


@pytest.mark.parametrize("n", [1, 63, 130, 5003, 40_001])
@pytest.mark.parametrize("d,k", [(128, 128), (512, 512), (1024, 1024), (96, 160), (32, 32), (320, 64)])
@pytest.mark.parametrize("norm,scaled", [(1, True), (0, False), (2, True)])
def test_bounded_projection_at_other_widths_takes_the_three_product_mode_of_the_split_form(n, d, k, norm, scaled):
    """cleora_project_bounded_dev with row bounds at shapes other than d = k = 256 (config 5: d = 1024): the split form's bounded-operand
    mode (csrc/whiten.hip: two-way f16 splits scaled by powers of two, three MFMAs per product, *form == 2) against an f64 product of the
    same f32 inputs; row bounds and transform columns over several decades, ragged tiles, several column passes (k > 256: no norm in the
    epilogue, *norm_done == 0).  Stated: every row within 2e-6 of its own norm."""
    import ctypes
    L = _hip.lib()
    rng = np.random.default_rng(n + 7 * norm + scaled + 13 * d + k)
    bound = np.where(rng.random(n) < 0.8, 1.0, 10 ** (rng.random(n) * 4 - 1)).astype(np.float32)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    x *= (bound * (0.2 + 0.8 * rng.random(n)))[:, None].astype(np.float32)
    x = np.minimum(np.maximum(x, -bound[:, None]), bound[:, None]).astype(np.float32)
    rowscale = (bound * (2 * rng.random(n) - 1)).astype(np.float32)
    mean = np.clip(rng.standard_normal(d) * 0.05, -1, 1).astype(np.float32)
    t = (rng.standard_normal((d, k)) * 10 ** (rng.random(k) * 5 - 2)[None, :]).astype(np.float32)
    dx, dm, dt, ds, db = (_hip.DevArray.from_host(a) for a in (x, mean, t, rowscale, bound))
    do = _hip.DevArray.from_host(np.full((n + 1, k), 7.0, np.float32))
    nd, form = ctypes.c_int(0), ctypes.c_int(-1)
    _hip.check(L.cleora_project_bounded_dev(dx.ptr, d, n, d, dm.ptr, dt.ptr, k, do.ptr, k, ds.ptr if scaled else None, db.ptr, norm,
                                            ctypes.byref(nd), ctypes.byref(form), None))
    _hip.check(L.cleora_stream_sync(None))
    assert form.value == 2 and nd.value == (1 if (norm and k <= 256) else 0)
    got = do.to_host()
    assert (got[n] == 7.0).all(), "wrote past row n"
    got = got[:n].astype(np.float64)
    ref = (x.astype(np.float64) - (rowscale.astype(np.float64)[:, None] if scaled else 1.0) * mean.astype(np.float64)[None, :]) @ t.astype(np.float64)
    if nd.value and norm == 1:
        ref /= np.maximum(np.linalg.norm(ref, axis=1, keepdims=True), 1e-10)
    elif nd.value and norm == 2:
        ref /= np.maximum(np.abs(ref).sum(axis=1, keepdims=True), 1e-10)
    assert np.isfinite(got).all()
    err = np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-300)
    assert err.max() <= 2e-6, float(err.max())


def test_bounded_projection_falls_back_for_other_shapes_and_checks_its_arguments():
    """Any shape but d = k = 256 takes cleora_project_general_dev's kernel (*form == 0) with the same result contract."""
    import ctypes
    L = _hip.lib()
    n, d = 3000, 128
    rng = np.random.default_rng(5)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    mean = x.mean(axis=0).astype(np.float32)
    t = rng.standard_normal((d, d)).astype(np.float32)
    dx, dm, dt = (_hip.DevArray.from_host(a) for a in (x, mean, t))
    do = _hip.DevArray((n, d), np.float32)
    nd, form = ctypes.c_int(0), ctypes.c_int(-1)
    _hip.check(L.cleora_project_bounded_dev(dx.ptr, d, n, d, dm.ptr, dt.ptr, d, do.ptr, d, None, None, 1, ctypes.byref(nd), ctypes.byref(form), None))
    _hip.check(L.cleora_stream_sync(None))
    assert form.value == 0 and nd.value == 1
    ref = (x - mean).astype(np.float64) @ t.astype(np.float64)
    ref /= np.linalg.norm(ref, axis=1, keepdims=True)
    assert np.abs(do.to_host() - ref).max() <= 2e-6
    # d = 256 without row scales or bounds (both NULL: ones; unit rows keep the bound): still the f16 kernel
    n2, d2 = 1000, 256
    x2 = rng.standard_normal((n2, d2)).astype(np.float32)
    x2 /= np.linalg.norm(x2, axis=1, keepdims=True)
    mean2 = x2.mean(axis=0).astype(np.float32)
    t2 = rng.standard_normal((d2, d2)).astype(np.float32)
    dx2, dm2, dt2 = (_hip.DevArray.from_host(a) for a in (x2, mean2, t2))
    do2 = _hip.DevArray((n2, d2), np.float32)
    _hip.check(L.cleora_project_bounded_dev(dx2.ptr, d2, n2, d2, dm2.ptr, dt2.ptr, d2, do2.ptr, d2, None, None, 1, ctypes.byref(nd), ctypes.byref(form), None))
    _hip.check(L.cleora_stream_sync(None))
    assert form.value == 1 and nd.value == 1
    ref2 = (x2 - mean2).astype(np.float64) @ t2.astype(np.float64)
    ref2 /= np.linalg.norm(ref2, axis=1, keepdims=True)
    assert np.abs(do2.to_host() - ref2).max() <= 2e-6
    assert L.cleora_project_bounded_dev(dx.ptr, d, n, d, dm.ptr, dt.ptr, d, do.ptr, d, None, None, 3, None, None, None) == _hip.E_INVALID
    assert L.cleora_project_bounded_dev(dx.ptr, d, n, d, dm.ptr, dt.ptr, d, dx.ptr, d, None, None, 1, None, None, None) == _hip.E_INVALID
