"""The d x d step of the intermediate whitened iterations on the HOST (csrc/dxd_host.cpp, cleora_cholesky_whiten_host): plain host
math inside the HIP library, so it is checked here without a GPU.  Reference operation: the transform of whiten_embeddings
(pycleora/__init__.py:145-156); inside the loop any W with W^T C W = I serves (docs/history.md §3.7), and this one is W = L^-T."""
import ctypes

import numpy as np
import pytest

from cleora_amd import _hip


def _whiten(gram, n):
    d = gram.shape[0]
    t = np.zeros((d, d), np.float32)
    tr = ctypes.c_double(-1.0)
    rc = _hip.lib().cleora_cholesky_whiten_host(gram.ctypes.data, n, d, t.ctypes.data, ctypes.byref(tr))
    return rc, t, tr.value


@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 6, 7, 8, 9, 31, 64, 100, 255, 256])
def test_host_cholesky_whitens_like_lapack(d):
    """T = L^-T of cov = gram / (n - 1): T^T cov T = I to f32 rounding of T (1e-6), T equal to numpy's inv(cholesky).T
    to 2e-7 of its largest entry, upper triangular, and trace(cov^-1) as reported = numpy's.  Every block remainder of the
    four-row passes is covered (d mod 4 = 0..3)."""
    rng = np.random.default_rng(d)
    n = 3000 + d
    x = rng.standard_normal((n, d)) * np.linspace(0.3, 3.0, d) + rng.standard_normal((n, 1)) * 0.2
    x -= x.mean(0)
    gram = np.ascontiguousarray(x.T @ x)
    rc, t, trace = _whiten(gram, n)
    assert rc == 0
    cov = gram / (n - 1)
    w = t.astype(np.float64)
    assert np.abs(w.T @ cov @ w - np.eye(d)).max() < 1e-6
    ref = np.linalg.inv(np.linalg.cholesky(cov)).T
    assert np.abs(w - ref).max() <= 2e-7 * np.abs(ref).max()
    assert np.abs(np.tril(t, -1)).max() == 0.0
    assert trace == pytest.approx(np.trace(np.linalg.inv(cov)), rel=1e-9)


def test_host_cholesky_verdicts():
    """The guard of docs/history.md §3.7 on the host: lambda_min = 1e-9 is accepted, 5e-11 is not (trace(cov^-1) > 0.999e10 although
    every squared pivot is >= 1e-8), an indefinite and a NaN matrix are not, bad arguments are refused."""
    d, n = 64, 1000
    q, _ = np.linalg.qr(np.random.default_rng(1).standard_normal((d, d)))
    for lam, want in ((1e-9, 0), (5e-11, 1)):
        ev = np.ones(d)
        ev[-1] = lam
        cov = (q * ev) @ q.T
        cov = (cov + cov.T) / 2
        rc, t, trace = _whiten(np.ascontiguousarray(cov * (n - 1)), n)
        assert rc == want
        if want == 0:
            assert trace == pytest.approx(d - 1 + 1 / lam, rel=1e-4)
            w = t.astype(np.float64)
            assert np.abs(w.T @ cov @ w - np.eye(d)).max() < 1e-3      # f32 rounding of entries ~ 3e4 at this condition number
    ev = np.ones(d)
    ev[3] = -0.5
    assert _whiten(np.ascontiguousarray((q * ev) @ q.T * (n - 1)), n)[0] == 1
    bad = np.eye(d)
    bad[5, 5] = np.nan
    assert _whiten(bad, n)[0] == 1
    assert _whiten(np.eye(d), 1)[0] < 0                       # n < 2
