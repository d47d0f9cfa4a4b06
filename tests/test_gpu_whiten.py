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


@pytest.mark.parametrize("n,d,iters,rw,kind,route", [
    (20_000, 64, 6, 0.0, 0, "library"), (6000, 256, 4, 0.3, 1, "library"), (3000, 32, 5, 1.5, 0, "library"),
    (3000, 320, 3, 0.0, 0, "library"),                       # d > 256: rocSOLVER's potrf / trtri whatever the switch says
    # the in-house Cholesky kernel (d <= 256): tiny, odd, one below the limit, the limit
    (20_000, 64, 6, 0.0, 0, "kernel"), (6000, 256, 4, 0.3, 1, "kernel"), (500, 8, 4, 0.0, 0, "kernel"),
    (4000, 33, 4, 0.0, 0, "kernel"), (3000, 255, 3, 0.0, 1, "kernel"), (2500, 1, 3, 0.0, 0, "kernel"),
    # the factorisation on the host (d <= 256; the default): the same shapes
    (20_000, 64, 6, 0.0, 0, "host"), (6000, 256, 4, 0.3, 1, "host"), (500, 8, 4, 0.0, 0, "host"), (2500, 1, 3, 0.0, 0, "host"),
    (3000, 255, 3, 0.0, 1, "host"), (3000, 320, 3, 0.0, 0, "host"),      # (d > 256: the library whatever the switch says)
    (3000, 130, 3, 0.0, 0, None)])                           # None: the library's own choice (the host for d <= 256)
def test_overlapped_whitened_loop_equals_the_sequential_order(n, d, iters, rw, kind, route, monkeypatch):
    """cleora_embed + CLEORA_F_WHITEN without a convergence test runs SpMM(t+1) beside Gram / eigh(t), taking the SpMM
    before the projection (A ((Y - mu) T) = (A Y - (A 1) mu^T) T).  With a (never met) convergence threshold the same
    call keeps the reference's sequential order: both must agree to f32 rounding — columns up to sign, 2e-3 relative
    after several whitenings, and the pairwise cosines of a row sample to 1e-4.  Also against the numpy oracle loop."""
    import ctypes
    from tests.graphs import random_csr
    import oracle
    if route:
        monkeypatch.setenv("CLEORA_CHOLESKY", route)         # read per call by the library
    else:
        monkeypatch.delenv("CLEORA_CHOLESKY", raising=False)
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


@pytest.mark.parametrize("route", ["library", "kernel", "host"])
def test_cholesky_guard_implies_the_reference_clamp(route, monkeypatch):
    """Intermediate iterations may take the Cholesky whitening only when the reference's clamp max(lambda, 1e-10)
    (pycleora/__init__.py:155) is provably inactive.  The smallest pivot of the factor only bounds lambda_min from ABOVE
    (a covariance with lambda_min = 5e-11 and every squared pivot >= 1e-8 exists: below); the guard therefore also
    requires trace(cov^-1) = ||L^-T||_F^2 <= 1e10, which implies lambda_min >= 1e-10.  Checked on synthetic
    covariances Q diag(lambda) Q^T through cleora_whiten_transform_any_dev:
      lambda_min = 1e-9  -> Cholesky form (form = 1), T^T C T = I
      lambda_min = 5e-11 -> PCA form (form = 0) with the clamp: T^T C T = diag(1, ..., 1, 0.5)."""
    import ctypes
    monkeypatch.setenv("CLEORA_CHOLESKY", route)
    L = _hip.lib()
    d, n = 256, 10_000
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


@pytest.mark.parametrize("form", ["split", "f32"])
@pytest.mark.parametrize("n,d", [(70_001, 256), (40_003, 512), (30_001, 1024), (5_003, 256)])
def test_intermediate_gram_on_the_f32_matrix_cores(n, d, form, monkeypatch):
    """cleora_whiten_stats_dev(intermediate = 1) at d = 256 S: centred Gram from the bf16 matrix cores with three-way split
    operands (six v_mfma_f32_32x32x16_bf16 per f32 product, the default: gram16_kernel) or on v_mfma_f32_32x32x2_f32
    (CLEORA_GRAM=f32: gram32_kernel) — f32 sums over <= 2048
    rows, f64 across; S diagonal super-tile blocks and S (S - 1) off-diagonal ones — against the f64 form (intermediate = 0) and
    numpy fp64.  Stated: the f64 form 1e-13 relative
    Frobenius as before; the f32 form <= 5e-7 of the Gram's Frobenius norm and of its diagonal entry by entry.  Mean: 1e-12 for
    the f64 form; the f32 form centres in f32 (y = x - c32, one rounding of 3e-8 |y|), so its mean carries that: <= 1e-8."""
    L = _hip.lib()                                             # n: not a multiple of the 16-row chunk or of the slice count
    if form == "f32":
        monkeypatch.setenv("CLEORA_GRAM", "f32")               # read per call
    else:
        monkeypatch.delenv("CLEORA_GRAM", raising=False)
    rng = np.random.default_rng(8)
    x = (rng.standard_normal((n, d)) * np.linspace(0.3, 2.0, d) + rng.standard_normal(d) * 0.2).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    dx = _hip.DevArray.from_host(x)
    ws = _hip.DevArray((L.cleora_whiten_workspace(n, d),), np.uint8)
    dm, dg = _hip.DevArray((d,), np.float64), _hip.DevArray((d, d), np.float64)
    x64 = x.astype(np.float64)
    mean = x64.mean(axis=0)
    gram = (x64 - mean).T @ (x64 - mean)
    for intermediate, tol in ((0, 1e-12), (1, 5e-7)):
        _hip.check(L.cleora_whiten_stats_dev(dx.ptr, d, n, d, ws.ptr, intermediate, dm.ptr, dg.ptr, None))
        _hip.check(L.cleora_stream_sync(None))
        gm, gg = dm.to_host(), dg.to_host()
        assert np.abs(gm - mean).max() <= (1e-8 if intermediate else 1e-12)
        assert np.linalg.norm(gg - gram) <= tol * np.linalg.norm(gram), (intermediate, np.linalg.norm(gg - gram) / np.linalg.norm(gram))
        assert np.abs(np.diag(gg) - np.diag(gram)).max() <= tol * np.diag(gram).max()
        np.testing.assert_array_equal(gg, gg.T)


PROJECTION_ERROR_SCRIPT = r'''
import json, sys
import numpy as np
from cleora_amd import _hip
L = _hip.lib()
out = {}
# >= 32 768 rows take the 128-row / one-wave-per-SIMD form (project_fat.hip), fewer the 64-row form (whiten.hip)
for n, d, k in ((50_000, 256, 256), (40_000, 1024, 1024), (36_000, 64, 64), (40_000, 256, 100), (20_000, 256, 100), (9_000, 1024, 1024),
                (5_000, 96, 96)):
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
    out[f"{n}x{d}x{k}"] = [float(np.abs(got - ref).max() / np.abs(ref).max()), float(np.sqrt(((got - ref) ** 2).mean()) / np.abs(ref).max())]
print("RESULT " + json.dumps(out))
'''


def test_split_projection_is_as_accurate_as_the_f32_matrix_cores():
    """The projection's default form computes every f32 product from six bf16 MFMAs (three-way split operands, whiten.hip);
    CLEORA_PROJECT=f32 keeps the v_mfma_f32_32x32x2_f32 forms.  Both against an f64 product of the same f32 inputs, several
    shapes: the split form's maximum and RMS error must not exceed 1.5x the f32 matrix cores' (the numpy sgemm of
    pycleora/__init__.py:163 sits in the same class: ~1e-7 of max|out| at d = 256)."""
    import json
    import subprocess
    import sys
    res = {}
    for form in ("split", "f32"):
        env = dict(os.environ)
        env.pop("CLEORA_PROJECT", None)
        if form == "f32":
            env["CLEORA_PROJECT"] = "f32"
        p = subprocess.run([sys.executable, "-c", PROJECTION_ERROR_SCRIPT], env=env, capture_output=True, text=True,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res[form] = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    for shape, (emax, erms) in res["split"].items():
        fmax, frms = res["f32"][shape]
        assert emax <= 1.5 * fmax + 1e-9 and erms <= 1.5 * frms + 1e-10, (shape, emax, fmax, erms, frms)
        assert emax <= 2e-6, (shape, emax)
