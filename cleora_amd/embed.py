"""Device-resident version of the reference's `embed()` loop and `whiten_embeddings`
(pycleora/__init__.py:51-164, 942-976): the iterate stays in HBM for all iterations; nothing but the
final result (and the iterate handed to a callback) crosses PCIe.

Per iteration (reference order, :109-125):
    propagate (SpMM)  -> residual blend -> L2 / L1 normalise   one fused kernel   (cleora_propagate_dev)
    whiten: column statistics (f64) -> centred Gram (f64 MFMA) -> eigh (rocSOLVER) -> project (f32 MFMA)
    callback(i, X) if given (forces a device->host copy) ; RMSE early stop (f64 on device)

The reference's own `pycleora.embed()` also runs unmodified over cleora_amd.pycleora.SparseMatrix
(cleora_amd.install()); it then crosses PCIe twice per iteration and whitens in numpy.  This
module is the fast path for the same arithmetic.
"""
import ctypes

import numpy as np

from . import _hip
from . import pycleora as _pyc
from .pycleora import SparseMatrix

DEFAULT_FEATURE_DIM = 256       # pycleora/__init__.py:12
DEFAULT_NUM_ITERATIONS = 40     # pycleora/__init__.py:13


def _n_components(d, n_components):
    """Columns kept by `eigenvectors[:, :n_components]` (pycleora/__init__.py:151-153): None keeps all d,
    a count beyond d keeps d, 0 keeps none, a negative count drops that many from the end."""
    return d if n_components is None else len(range(d)[:int(n_components)])


class DeviceWhitener:
    """whiten_embeddings on device buffers: the whole chain — column statistics, centred Gram, rocSOLVER dsyevd,
    transform, projection — is enqueued on one stream by cleora_whiten_dev with no host round trip.
    Workspaces are sized once per (n, d)."""

    def __init__(self, n, d):
        L = _hip.lib()
        self.n, self.d, self.L = n, d, L
        self._eigenvalues = None
        self._split = None          # buffers of stats(), allocated on first use
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
        if self._eigenvalues is None:
            _hip.check(self.L.cleora_stream_sync(None))
            self._eigenvalues = self.eig_dev.to_host()
        return self._eigenvalues

    def stats(self, x_ptr, ldx, stream=None):
        """(mean f64[d], cov f64[d,d]) as pycleora/__init__.py:136-143, computed by the device kernels."""
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
        """out = whiten_embeddings(x).  Returns k (columns written; 0 writes nothing)."""
        k = _n_components(self.d, n_components)
        if k == 0:
            return 0
        _hip.check(self.L.cleora_whiten_dev(x_ptr, ldx, self.n, self.d, k, out_ptr, ldo, self.ws.ptr,
                                            self.eig_dev.ptr, stream))
        self._eigenvalues = None
        return k


def whiten_embeddings(embeddings, n_components=None):
    """Drop-in for pycleora.whiten_embeddings on host arrays: cleora_whiten (one upload, one download)."""
    x = np.ascontiguousarray(embeddings, dtype=np.float32)
    n, d = x.shape
    if n <= 1:
        return x.copy()                                   # :132-133
    k = _n_components(d, n_components)
    out = np.empty((n, k), np.float32)
    if k:                                                 # the C ABI reads n_components = 0 as "all d"
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
    if normalization not in ("l2", "l1", "none"):
        if normalization == "spectral":
            raise ValueError("cleora_amd.embed runs normalization 'l2', 'l1' or 'none' on the device; 'spectral' "
                             "(a full SVD per iteration, pycleora/__init__.py:951-956) stays with the reference's pycleora.embed")
        raise ValueError(f"Unknown normalization method: {normalization}. Use 'l2', 'l1', 'spectral', or 'none'.")
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

    x0 = None
    if initial_embeddings is not None:                    # :100-105
        x0 = np.ascontiguousarray(np.asarray(initial_embeddings).astype(np.float32))
        if x0.shape[0] != n:
            raise ValueError(f"initial_embeddings has {x0.shape[0]} rows but graph has {n} entities")
    one_call = whiten and callback is None and normalization in ("l2", "l1")
    if x0 is None and not (one_call and n > 0 and int(feature_dim) > 0 and num_iterations > 0):
        x0 = graph.initialize_deterministically(feature_dim, seed)
    d = x0.shape[1] if x0 is not None else int(feature_dim)
    if n == 0 or d == 0 or num_iterations <= 0:
        return x0

    if one_call:
        # the default loop as ONE C-ABI call (cleora_embed + CLEORA_F_WHITEN): nobody looks at the intermediate
        # iterates, so the library may run the SpMM of iteration t+1 beside the Gram / eigensolver of iteration t.
        # Without initial_embeddings E_0 is made ON the device from the entity hashes (init_value, src/lib.rs:478-488: bit-equal to
        # initialize_deterministically) — it does not travel to the host and back (10 GB each way at |V| = 10M, d = 256).
        out = np.empty((n, d), np.float32)
        ran = ctypes.c_uint64(0)
        flags = _hip.F_WHITEN | (_hip.F_L1NORM if normalization == "l1" else 0) | _pyc.loop_flags()
        hashes = graph._arr["hashes"] if x0 is None else None
        with graph._lock:
            m = graph._multi()
            if m is not None:                        # several devices configured (cleora_amd.install(devices=...)): the row partition
                return m.embed(hashes, x0, kind, d, int(num_iterations), int(seed), float(residual_weight), float(max(convergence_threshold, 0.0)), flags)[0]
            _hip.check(L.cleora_embed(graph._graph().handle, _hip.ptr(hashes), _hip.ptr(x0), kind, d, int(num_iterations), int(seed),
                                      float(residual_weight), float(max(convergence_threshold, 0.0)), flags,
                                      _hip.ptr(out), ctypes.byref(ran)))
        return out

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
    if normalization not in ("l2", "l1", "none"):
        raise ValueError("normalization must be 'l2', 'l1' or 'none' on the device path")
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
    whitener = DeviceWhitener(n, d) if (whiten and n > 1) else None
    # iterate buffers placed for the SpMM (cleora_alloc_iterates): the SpMM always WRITES `nxt` and reads `cur` — which
    # is, in turn, each of the other buffers — so `nxt` is the buffer the partners are tuned against
    (nxt, cur, *rest), _ = _hip.DevArray.iterates(g, n, d, 3 if whitener is not None else 2, iterations=int(num_iterations))
    wht = rest[0] if rest else None
    _hip.check(L.cleora_memcpy_h2d(cur.ptr, _hip.ptr(x0), cur.nbytes, None))
    check = convergence_threshold > 0
    sq = _hip.DevArray((n,), np.float64) if check else None
    ws = _hip.DevArray((L.cleora_reduce_workspace(n),), np.float64) if check else None
    tot = _hip.DevArray((1,), np.float64) if check else None
    fused_attention = attention_temperature is not None and d % 4 == 0 and d <= 2048      # cleora_propagate_attention_dev's shapes
    attn = (_hip.DevArray((max(int(g.info().nnz), 1),), np.float32)
            if attention_temperature is not None and not fused_attention else None)
    flags = {"l2": _hip.F_L2NORM, "l1": _hip.F_L1NORM, "none": 0}[normalization]   # _normalize, :942-959
    # this is the Python loop of embed(): it blends for ANY rw > 0 (:111-115), unlike the Rust loop's
    # 0 < rw < 1 (src/embedding.rs:116) that the kernel applies without CLEORA_F_BLEND_ANY
    if residual_weight > 0:
        flags |= _hip.F_RESIDUAL | _hip.F_BLEND_ANY
    taken = []
    for i in range(int(num_iterations)):
        if attention_temperature is not None and i > 0 and fused_attention:      # pycleora/__init__.py:241-269, one pass
            _hip.check(L.cleora_propagate_attention_dev(g.handle, kind, cur.ptr, d, d, float(attention_temperature), nxt.ptr, d,
                                                        flags, float(residual_weight), cur.ptr, None, None))
        elif attn is not None and i > 0:              # shapes the fused kernel does not take: weights, then the SpMM
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
            _hip.check(L.cleora_rowops_dev(result.ptr, d, n, d, result.ptr, d, _hip.F_SQDIFF | _hip.F_SQDIFF64, 0.0,
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
    """Same contract as pycleora.find_most_similar (pycleora/__init__.py:753-781): cosine similarity of every entity
    to the query, top_k as a list of dicts.  Normalise + GEMV + selection in one device call
    (cleora_topk_cosine_dev); only the top_k (index, score) pairs come back."""
    query_idx = graph.get_entity_index(query_entity)
    x = np.ascontiguousarray(embeddings, dtype=np.float32)
    n, d = x.shape
    k = min(int(top_k), n)
    if k <= 0:
        return []
    ids = graph.entity_ids
    if k > 1024:        # beyond the device selection's limit: device scores, host selection (variants._topk_by_host_selection)
        from .variants import _topk_by_host_selection
        idx, sims = _topk_by_host_selection(graph, x, [query_idx], k, exclude_self, False, self_score=-1.0)
        return [{"entity_id": ids[int(i)], "index": int(i), "similarity": float(v)} for i, v in zip(idx[0], sims[0])]
    L = _hip.lib()
    dx = _hip.DevArray.from_host(x)
    dq = _hip.DevArray.from_host(np.asarray([query_idx], dtype=np.uint32))
    kk = k
    oi, os_ = _hip.DevArray((1, kk), np.uint32), _hip.DevArray((1, kk), np.float32)
    ws = _hip.DevArray((L.cleora_topk_workspace_for(n, kk, 1),), np.uint8)
    _hip.check(L.cleora_topk_cosine_dev(None, dx.ptr, d, n, d, dq.ptr, 1, kk, 1 if exclude_self else 0, 0, oi.ptr, os_.ptr,
                                        ws.ptr, None))
    _hip.check(L.cleora_stream_sync(None))
    idx, sims = oi.to_host()[0], os_.to_host()[0]
    # the reference sets the query's own similarity to -1 (:768-769) and still lists it if it ranks; the device masks
    # with -2: report -1 for it like the reference
    return [{"entity_id": ids[int(i)], "index": int(i), "similarity": float(-1.0 if (exclude_self and int(i) == query_idx) else v)}
            for i, v in zip(idx, sims)][:k]
