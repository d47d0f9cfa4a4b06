"""Device-resident version of the reference's `embed()` loop and `whiten_embeddings`
(pycleora/__init__.py:51-164, 942-976): the iterate stays in HBM for all iterations; only the
d-vector of column sums and the d x d Gram matrix cross PCIe per whitening (for the host
LAPACK `eigh`, exactly the routine the reference itself calls, :145), plus the final result.

Per iteration (reference order, :109-125):
    propagate (SpMM)  -> residual blend -> L2 normalise      one fused kernel   (cleora_propagate_dev)
    whiten: column sums (f64) -> centred Gram (f64 MFMA) -> eigh (host) -> project (f32 MFMA)
    callback(i, X) if given (forces a device->host copy) ; RMSE early stop (f64 on device)

The reference's own `pycleora.embed()` also runs unmodified over cleora_amd.pycleora.SparseMatrix
(cleora_amd.install()); it then crosses PCIe twice per iteration and whitens in numpy.  This
module is the fast path for the same arithmetic.
"""
import ctypes

import numpy as np

from . import _hip
from .pycleora import SparseMatrix

DEFAULT_FEATURE_DIM = 256       # pycleora/__init__.py:12
DEFAULT_NUM_ITERATIONS = 40     # pycleora/__init__.py:13


def eigh_descending(cov, backend="auto"):
    """Eigen-decomposition of the covariance on host arrays, eigenvalues descending
    (pycleora/__init__.py:145-149).  Used by the host-statistics routes (DeviceWhitener(eigh="host" |
    "device"), the partitioned whitening in sharded.py); the default single-GPU route never leaves the
    device (cleora_whiten_dev).  backend "host": numpy/LAPACK, the routine the reference itself calls.  "device":
    torch.linalg.eigh on the GPU (rocSOLVER).  "auto": device when torch sees a GPU, else host.
    Measured on the MI355X box: d = 256: host 5.1 ms / device 6.4 ms; d = 1024: host 322 ms /
    device 23 ms.  The device route is the default because the host route is fragile inside a
    GPU loop: when the kernels between two eigh calls are short (C2 scale) the BLAS threads collide
    with still-spinning OpenMP workers of the previous CPU op and the same 256 x 256 eigh takes
    80 ms instead of 5 ms.  Eigenvector signs may differ between the two; whitening is defined only
    up to that."""
    use_device = backend == "device"
    if backend == "auto":
        try:
            import torch
            use_device = torch.cuda.is_available()
        except ImportError:
            use_device = False
    if use_device:
        import torch
        w, v = torch.linalg.eigh(torch.from_numpy(cov).cuda())
        w, v = w.cpu().numpy(), v.cpu().numpy()
    else:
        w, v = np.linalg.eigh(cov)
    idx = np.argsort(w)[::-1]
    return w[idx], v[:, idx]


class DeviceWhitener:
    """whiten_embeddings on device buffers.  Workspaces are sized once per (n, d).

    eigh = "library" (what "auto" means): the whole chain — column sums, mean, centred Gram, rocSOLVER
    dsyevd, transform, projection — is enqueued on one stream by cleora_whiten_dev with no host round
    trip.  "host" / "device" keep the statistics on the host between the kernels and call
    np.linalg.eigh / torch.linalg.eigh (see eigh_descending); they exist for A/B comparison with the
    LAPACK routine the reference itself calls."""

    def __init__(self, n, d, eigh="auto"):
        L = _hip.lib()
        self.n, self.d, self.L = n, d, L
        self.eigh = "library" if eigh == "auto" else eigh
        self.transform = None
        self._eigenvalues = None
        self._split = None          # buffers of the host-statistics route, allocated on first use
        if self.eigh == "library":
            self.ws = _hip.DevArray((L.cleora_whiten_workspace(n, d),), np.uint8)
            self.eig_dev = _hip.DevArray((d,), np.float64)

    def _split_buffers(self):
        if self._split is None:
            L, n, d = self.L, self.n, self.d
            self._split = dict(
                colsum_ws=_hip.DevArray((L.cleora_colsum_workspace(n, d),), np.float64),
                colsum=_hip.DevArray((d,), np.float64), mean64=_hip.DevArray((d,), np.float64),
                mean32=_hip.DevArray((d,), np.float32),
                gram_ws=_hip.DevArray((L.cleora_gram_workspace(n, d),), np.float64),
                gram=_hip.DevArray((d, d), np.float64))
        return self._split

    @property
    def last_eigenvalues(self):
        """Eigenvalues of the last covariance, descending (downloaded on demand)."""
        if self.eigh == "library" and self._eigenvalues is None:
            _hip.check(self.L.cleora_stream_sync(None))
            self._eigenvalues = self.eig_dev.to_host()
        return self._eigenvalues

    def stats(self, x_ptr, ldx, stream=None):
        """(mean f64[d], cov f64[d,d]) as pycleora/__init__.py:136-143."""
        L, n, d, b = self.L, self.n, self.d, self._split_buffers()
        _hip.check(L.cleora_colsum_dev(x_ptr, ldx, n, d, b["colsum_ws"].ptr, b["colsum"].ptr, stream))
        _hip.check(L.cleora_mean_dev(b["colsum"].ptr, n, d, b["mean64"].ptr, b["mean32"].ptr, stream))
        _hip.check(L.cleora_centered_gram_dev(x_ptr, ldx, n, d, b["mean64"].ptr, b["gram_ws"].ptr,
                                              b["gram"].ptr, stream))
        _hip.check(L.cleora_stream_sync(stream))
        cov = b["gram"].to_host()
        cov *= 1.0 / (n - 1)
        return b["mean64"].to_host(), cov

    def whiten(self, x_ptr, ldx, out_ptr, ldo, n_components=None, stream=None):
        """out = whiten_embeddings(x).  Returns k (columns written)."""
        L, n, d = self.L, self.n, self.d
        if self.eigh == "library":
            k = d if n_components is None else min(int(n_components), d)
            _hip.check(L.cleora_whiten_dev(x_ptr, ldx, n, d, k, out_ptr, ldo, self.ws.ptr, self.eig_dev.ptr,
                                           stream))
            self._eigenvalues = None
            return k
        mean, cov = self.stats(x_ptr, ldx, stream)
        w, v = eigh_descending(cov, self.eigh)           # :145-149
        if n_components is not None:                     # :151-153
            w, v = w[:n_components], v[:, :n_components]
        scale = 1.0 / np.sqrt(np.maximum(w, 1e-10))      # :155
        transform = np.ascontiguousarray((v * scale).astype(np.float32))
        k = transform.shape[1]
        if self.transform is None or self.transform.shape != transform.shape:
            self.transform = _hip.DevArray(transform.shape, np.float32)
        _hip.check(L.cleora_memcpy_h2d(self.transform.ptr, _hip.ptr(transform), transform.nbytes, stream))
        _hip.check(L.cleora_project_dev(x_ptr, ldx, n, d, self._split["mean32"].ptr, self.transform.ptr, k,
                                        out_ptr, ldo, stream))
        self._eigenvalues = w
        return k


def whiten_embeddings(embeddings, n_components=None):
    """Drop-in for pycleora.whiten_embeddings on host arrays: cleora_whiten (one upload, one download)."""
    x = np.ascontiguousarray(embeddings, dtype=np.float32)
    n, d = x.shape
    if n <= 1:
        return x.copy()                                   # :132-133
    k = d if n_components is None else min(int(n_components), d)
    out = np.empty((n, k), np.float32)
    _hip.check(_hip.lib().cleora_whiten(_hip.ptr(x), n, d, k, _hip.ptr(out)))
    return out


def embed(graph, feature_dim=DEFAULT_FEATURE_DIM, num_iterations=DEFAULT_NUM_ITERATIONS,
          propagation="left", normalization="l2", seed=0, initial_embeddings=None, num_workers=None,
          callback=None, residual_weight=0.0, convergence_threshold=0.0, whiten=True):
    """Same signature and semantics as pycleora.embed (pycleora/__init__.py:51-127)."""
    if isinstance(num_iterations, str):
        if num_iterations == "auto":
            num_iterations = DEFAULT_NUM_ITERATIONS
        else:
            raise ValueError(f"num_iterations must be an int or 'auto', got '{num_iterations}'")
    if propagation not in ("left", "symmetric"):
        raise ValueError(f"Unknown propagation type: '{propagation}'. Use 'left' or 'symmetric'.")
    if normalization not in ("l2", "none"):
        raise ValueError(f"cleora_amd.embed runs normalization 'l2' or 'none' on the device; got "
                         f"'{normalization}' (use the reference's pycleora.embed for 'l1'/'spectral')")
    if not isinstance(graph, SparseMatrix):
        raise TypeError("graph must be a cleora_amd.pycleora.SparseMatrix")
    kind = _hip.LEFT if propagation == "left" else _hip.SYMMETRIC
    L = _hip.lib()
    n = graph.num_entities

    fast = initial_embeddings is None and callback is None and normalization == "l2" and not whiten
    if fast:                                              # :70-96 — the all-native loop
        if convergence_threshold > 0:
            return graph.embed_fast_convergence(feature_dim, num_iterations, propagation=propagation,
                                                seed=seed, residual_weight=residual_weight,
                                                convergence_threshold=convergence_threshold)[0]
        return graph.embed_fast(feature_dim, num_iterations, propagation=propagation, seed=seed,
                                residual_weight=residual_weight)

    if initial_embeddings is not None:                    # :100-105
        x0 = np.ascontiguousarray(np.asarray(initial_embeddings).astype(np.float32))
        if x0.shape[0] != n:
            raise ValueError(f"initial_embeddings has {x0.shape[0]} rows but graph has {n} entities")
    else:
        x0 = graph.initialize_deterministically(feature_dim, seed)
    d = x0.shape[1]
    if n == 0 or d == 0 or num_iterations <= 0:
        return x0

    with graph._lock:
        return _device_loop(graph._graph(), n, x0, kind, int(num_iterations), normalization, callback,
                            float(residual_weight), float(convergence_threshold), whiten)


def embed_csr(rowptr, col, val, initial_embeddings, num_iterations=DEFAULT_NUM_ITERATIONS,
              normalization="l2", callback=None, residual_weight=0.0, convergence_threshold=0.0,
              whiten=True, device=0):
    """The same device-resident loop over a USER-SUPPLIED CSR adjacency: the
    reference's embed_weighted / embed_directed / embed_edge_features / attention variants
    (pycleora/__init__.py:206-410, 784-852) each build a scipy CSR and then run exactly
    `adj @ X` + `_postprocess_iteration` per iteration — this is that loop on the MI355X for any
    adjacency they construct.  rowptr[n+1], col[nnz], val[nnz]; initial_embeddings n x d."""
    x0 = np.ascontiguousarray(np.asarray(initial_embeddings).astype(np.float32))
    n = int(np.asarray(rowptr).shape[0]) - 1
    if x0.ndim != 2 or x0.shape[0] != n:
        raise ValueError(f"initial_embeddings has shape {x0.shape} but the adjacency has {n} rows")
    if normalization not in ("l2", "none"):
        raise ValueError("normalization must be 'l2' or 'none' on the device path")
    if n == 0 or x0.shape[1] == 0 or num_iterations <= 0:
        return x0
    g = _hip.Graph.from_host(rowptr, col, val, None, n_cols=n, device=device)
    try:
        return _device_loop(g, n, x0, _hip.LEFT, int(num_iterations), normalization, callback,
                            float(residual_weight), float(convergence_threshold), whiten)
    finally:
        g.close()


def _device_loop(g, n, x0, kind, num_iterations, normalization, callback, residual_weight,
                 convergence_threshold, whiten, snapshots=None, attention_temperature=None):
    """The device-resident iteration shared by embed(), embed_csr() and cleora_amd.variants.
    snapshots: iteration counts after which a host copy is also kept (embed_multiscale) — the return
    value is then the list of those copies.  attention_temperature: from iteration 1 on, the edge
    values are recomputed from the current iterate (embed_with_attention)."""
    L = _hip.lib()
    d = x0.shape[1]
    cur = _hip.DevArray.from_host(x0)
    nxt = _hip.DevArray((n, d), np.float32)
    wht = _hip.DevArray((n, d), np.float32) if whiten else None
    whitener = DeviceWhitener(n, d) if (whiten and n > 1) else None
    check = convergence_threshold > 0
    sq = _hip.DevArray((n,), np.float64) if check else None
    ws = _hip.DevArray((L.cleora_reduce_workspace(n),), np.float64) if check else None
    tot = _hip.DevArray((1,), np.float64) if check else None
    attn = _hip.DevArray((max(int(g.info().nnz), 1),), np.float32) if attention_temperature is not None else None
    flags = (_hip.F_L2NORM if normalization == "l2" else 0)
    # the reference's slow path blends for any rw > 0 (:114); the kernel gates on 0 < rw < 1
    # like the Rust loop (src/embedding.rs:116).  rw >= 1 is rejected rather than guessed.
    if residual_weight >= 1.0:
        raise ValueError("residual_weight must be < 1 on the device path")
    if residual_weight > 0:
        flags |= _hip.F_RESIDUAL
    taken = []
    for i in range(int(num_iterations)):
        if attn is not None and i > 0:                # pycleora/__init__.py:241-269
            _hip.check(L.cleora_edge_attention_dev(g.handle, kind, cur.ptr, d, d, float(attention_temperature),
                                                   attn.ptr, None))
            _hip.check(L.cleora_propagate_vals_dev(g.handle, attn.ptr, cur.ptr, d, d, nxt.ptr, d, flags,
                                                   float(residual_weight), cur.ptr, None, None, None))
        else:
            _hip.check(L.cleora_propagate_dev(g.handle, kind, cur.ptr, d, d, nxt.ptr, d, flags,
                                              float(residual_weight), cur.ptr, None, None, None))
        result = nxt
        if whitener is not None:
            whitener.whiten(nxt.ptr, d, wht.ptr, d)
            result = wht
        if callback is not None:
            _hip.check(L.cleora_stream_sync(None))
            callback(i, result.to_host())
        if snapshots is not None and (i + 1) in snapshots:
            _hip.check(L.cleora_stream_sync(None))
            taken.append(result.to_host())
        stop = False
        if check and i > 0:                           # :122-125, f64 RMSE vs the previous iterate
            _hip.check(L.cleora_rowops_dev(result.ptr, d, n, d, result.ptr, d, _hip.F_SQDIFF, 0.0,
                                           cur.ptr, sq.ptr, None, None))
            _hip.check(L.cleora_reduce_sum_f64_dev(sq.ptr, n, ws.ptr, tot.ptr, None))
            _hip.check(L.cleora_stream_sync(None))
            rmse = float(np.sqrt(tot.to_host()[0] / (float(n) * d)))
            stop = rmse < convergence_threshold
        # rotate buffers: `result` becomes the current iterate
        if result is wht:
            cur, wht = wht, cur
        else:
            cur, nxt = nxt, cur
        if stop:
            break
    _hip.check(L.cleora_stream_sync(None))
    if snapshots is not None:
        return taken
    return cur.to_host()


def find_most_similar(graph, embeddings, query_entity, top_k=10, exclude_self=True):
    """Same contract as pycleora.find_most_similar (pycleora/__init__.py:753-781): cosine
    similarity of every entity to the query, top_k as a list of dicts.  The normalise + GEMV runs
    as one pass over X on the device; the top-k selection of the n scores stays on the host."""
    query_idx = graph.get_entity_index(query_entity)
    x = np.ascontiguousarray(embeddings, dtype=np.float32)
    n, d = x.shape
    q = x[query_idx]
    q = (q / max(float(np.linalg.norm(q)), 1e-10)).astype(np.float32)
    L = _hip.lib()
    dx, dq = _hip.DevArray.from_host(x), _hip.DevArray.from_host(q)
    ds = _hip.DevArray((n,), np.float32)
    _hip.check(L.cleora_cosine_scores_dev(dx.ptr, d, n, d, dq.ptr, ds.ptr, None))
    _hip.check(L.cleora_stream_sync(None))
    sims = ds.to_host()
    if exclude_self:
        sims[query_idx] = -1.0
    k = min(int(top_k), n)
    part = np.argpartition(-sims, k - 1)[:k] if k < n else np.arange(n)
    order = part[np.argsort(-sims[part], kind="stable")]
    ids = graph.entity_ids
    return [{"entity_id": ids[int(i)], "index": int(i), "similarity": float(sims[int(i)])} for i in order]
