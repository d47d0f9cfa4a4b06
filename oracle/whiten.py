"""ORACLE (test infrastructure): numpy restatement of the Python-side hot-path
pieces of pycleora 3.2.1 — used on the GPU box, where /root/reference does not
exist.  In the build container it is validated against the reference's own
functions imported from /root/reference (tests/golden/make_golden.py writes
the fixtures; tests/test_oracle_golden.py replays them).

  normalize_l2        pycleora/__init__.py:942-946   (_normalize(…, "l2"))
  whiten_embeddings   pycleora/__init__.py:130-164   (PCA whitening, fp64 stats)
  postprocess         pycleora/__init__.py:963-971   (normalise, then whiten)
  compute_rmse        pycleora/__init__.py:974-976
  embed_slow          pycleora/__init__.py:97-125    (the whiten=True loop)
"""
import numpy as np

CHUNK = 50000  # pycleora/__init__.py:134


def normalize_l2(emb):
    norms = np.linalg.norm(emb, ord=2, axis=-1, keepdims=True)
    norms = np.maximum(norms, 1e-10)
    return emb / norms


def whiten_stats(emb):
    """mean (fp64, :136) and covariance (fp64, 50k-row chunks, :138-143)."""
    n, d = emb.shape
    mean = emb.mean(axis=0, dtype=np.float64)
    cov = np.zeros((d, d), dtype=np.float64)
    for i in range(0, n, CHUNK):
        block = emb[i:min(i + CHUNK, n)].astype(np.float64) - mean
        cov += block.T @ block
    cov *= 1.0 / (n - 1)
    return mean, cov


def whiten_transform(cov, n_components=None):
    """eigh, descending sort, optional truncation, clamp 1e-10 (:145-156)."""
    w, v = np.linalg.eigh(cov)
    idx = np.argsort(w)[::-1]
    w, v = w[idx], v[:, idx]
    if n_components is not None:
        w, v = w[:n_components], v[:, :n_components]
    scale = 1.0 / np.sqrt(np.maximum(w, 1e-10))
    return (v * scale).astype(np.float32), w


def whiten_embeddings(emb, n_components=None):
    n, d = emb.shape
    if n <= 1:
        return emb.copy()
    mean, cov = whiten_stats(emb)
    transform, _ = whiten_transform(cov, n_components)
    mean_f32 = mean.astype(np.float32)
    out = np.empty((n, transform.shape[1]), dtype=np.float32)
    for i in range(0, n, CHUNK):
        end = min(i + CHUNK, n)
        np.dot(emb[i:end] - mean_f32, transform, out=out[i:end])
    return out


def postprocess(emb, whiten):
    emb = normalize_l2(emb)
    return whiten_embeddings(emb) if whiten else emb


def compute_rmse(cur, prev):
    diff = cur.astype(np.float64, copy=False) - prev.astype(np.float64, copy=False)
    return float(np.sqrt(np.mean(diff * diff)))


def embed_slow(propagate, x0, iterations, residual_weight=0.0, convergence_threshold=0.0,
               whiten=True):
    """The non-fast-path loop of embed() with normalization='l2'.  `propagate`
    is a callable X -> A·X (the oracle SpMM in tests)."""
    emb = np.asarray(x0, dtype=np.float32)
    ran = 0
    for i in range(iterations):
        prev = emb
        emb = propagate(emb)
        if residual_weight > 0:
            emb = (1 - residual_weight) * emb + residual_weight * prev
        emb = postprocess(emb, whiten)
        ran = i + 1
        if convergence_threshold > 0 and i > 0:
            if compute_rmse(emb, prev) < convergence_threshold:
                break
    return emb, ran
