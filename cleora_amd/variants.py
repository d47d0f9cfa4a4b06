"""Device-resident forms of the reference's embed variants and link prediction.

Each of these reference functions builds a scipy CSR on the host and then iterates
`adj @ X` + `_postprocess_iteration` in numpy (pycleora/__init__.py:206-410, 784-852); here the host part
(who is connected to whom, with which weight) is done once with vectorised numpy and the iteration runs
in HBM through the same kernels as embed().  Signatures, defaults, return values and error messages are
the reference's.  Arithmetic differences, stated once: the reference multiplies with f64 scipy matrices
and rounds to f32 after every iteration; the device accumulates in f32 in edge order (the Rust kernel's
order) — a relative difference of ~1e-6 per iteration, which the parity tests carry as their tolerance.

    embed_multiscale      (:279-309)    embed_weighted        (:312-359)
    embed_directed        (:362-410)    embed_with_attention  (:206-276)
    embed_edge_features   (:784-852)    predict_links         (:636-681)
"""
from itertools import combinations

import numpy as np

from . import _hip
from .embed import (DEFAULT_FEATURE_DIM, DEFAULT_NUM_ITERATIONS, _device_loop, embed, embed_csr)
from .pycleora import SparseMatrix


def _validate_propagation(propagation):
    if propagation not in ("left", "symmetric"):                         # :24-26
        raise ValueError(f"Unknown propagation type: '{propagation}'. Use 'left' or 'symmetric'.")


def _check_device_normalization(normalization):
    if normalization not in ("l2", "l1", "none"):
        raise ValueError(f"the device path runs normalization 'l2', 'l1' or 'none'; got '{normalization}'")


def _row_normalised(rowptr, vals64):
    """diags(1 / max(row_sums, 1e-10)) @ adj on CSR values, in f64 (:351-353, :400-402)."""
    n = rowptr.shape[0] - 1
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    sums = np.bincount(rows, weights=vals64, minlength=n)
    return vals64 / np.maximum(sums, 1e-10)[rows]


def embed_multiscale(graph, feature_dim=DEFAULT_FEATURE_DIM, scales=None, propagation="left",
                     normalization="l2", seed=0, num_workers=None, whiten=True):
    """Snapshots of ONE propagation run after `scales` iterations, concatenated column-wise (:279-309)."""
    _validate_propagation(propagation)
    _check_device_normalization(normalization)
    if scales is None:
        scales = [10, 20, 30, 40]
    if not scales or not all(isinstance(s, int) and s > 0 for s in scales):
        raise ValueError("scales must be a non-empty list of positive integers")
    x0 = graph.initialize_deterministically(feature_dim, seed)
    ordered = sorted(scales)
    kind = _hip.LEFT if propagation == "left" else _hip.SYMMETRIC
    if graph.num_entities == 0:
        return np.concatenate([x0.copy() for _ in ordered], axis=1)
    with graph._lock:
        taken = _device_loop(graph._graph(), graph.num_entities, x0, kind, ordered[-1], normalization, None,
                             0.0, 0.0, whiten, snapshots=set(ordered))
    by_count = dict(zip(sorted(set(ordered)), taken))
    return np.concatenate([by_count[s] for s in ordered], axis=1)


def embed_weighted(edges_with_weights, columns, feature_dim=DEFAULT_FEATURE_DIM,
                   num_iterations=DEFAULT_NUM_ITERATIONS, propagation="left", normalization="l2", seed=0,
                   hyperedge_trim_n=16, num_workers=None, whiten=True):
    """(:312-359).  Every entity takes the largest weight (at least 1) of the lines it occurs in, the rows of
    the Markov matrix are scaled by it and renormalised."""
    _check_device_normalization(normalization)
    edge_strs = [e for e, w in edges_with_weights]
    graph = SparseMatrix.from_iterator(iter(edge_strs), columns, hyperedge_trim_n, num_workers)
    x0 = graph.initialize_deterministically(feature_dim, seed)
    rows, cols, vals, n, _ = graph.to_sparse_csr(propagation)
    # every entity takes the largest weight of the lines it occurs in, never less than 1 (:335-343):
    # one (entity, weight) pair per token occurrence, reduced with a scatter-max
    index_map = {eid: i for i, eid in enumerate(graph.entity_ids)}
    occ_idx, occ_w = [], []
    for line, w in edges_with_weights:
        hits = [index_map[t] for t in line.split() if t in index_map]
        occ_idx.extend(hits)
        occ_w.extend([w] * len(hits))
    weight_diag = np.ones(n, dtype=np.float64)
    if occ_idx:
        np.maximum.at(weight_diag, np.asarray(occ_idx, dtype=np.int64), np.asarray(occ_w, dtype=np.float64))
    rowptr = graph._arr["rowptr"].astype(np.int64)
    weighted = _row_normalised(rowptr, weight_diag[rows.astype(np.int64)] * vals.astype(np.float64))
    emb = embed_csr(rowptr.astype(np.uint64), cols, weighted.astype(np.float32), x0, num_iterations, normalization,
                    whiten=whiten)
    return graph, emb


def embed_directed(edges, columns, feature_dim=DEFAULT_FEATURE_DIM, num_iterations=DEFAULT_NUM_ITERATIONS,
                   normalization="l2", seed=0, hyperedge_trim_n=16, num_workers=None, whiten=True):
    """(:362-410).  Keeps the stored edge (r, c) only if some line lists r's token before c's, then
    renormalises the rows of the left Markov matrix."""
    _check_device_normalization(normalization)
    edges = list(edges)
    graph = SparseMatrix.from_iterator(iter(edges), columns, hyperedge_trim_n, num_workers)
    n = graph.num_entities
    index_map = {eid: i for i, eid in enumerate(graph.entity_ids)}
    # (earlier token, later token) of every line as one integer key per ordered entity pair (:376-382);
    # tokens that are not entities cannot match a stored edge and are dropped here
    keys = set()
    for line in edges:
        seq = [index_map[t] for t in line.split() if t in index_map]
        keys.update(a * n + b for a, b in combinations(seq, 2))
    rows, cols, vals, _, _ = graph.to_sparse_csr("left")
    edge_keys = rows.astype(np.int64) * n + cols.astype(np.int64)
    keep = np.isin(edge_keys, np.fromiter(keys, dtype=np.int64, count=len(keys)))
    kept_rows = rows[keep].astype(np.int64)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(kept_rows, minlength=n), out=rowptr[1:])
    vals64 = _row_normalised(rowptr, vals[keep].astype(np.float64))
    x0 = graph.initialize_deterministically(feature_dim, seed)
    emb = embed_csr(rowptr.astype(np.uint64), cols[keep], vals64.astype(np.float32), x0, num_iterations,
                    normalization, whiten=whiten)
    return graph, emb


def embed_with_attention(graph, feature_dim=DEFAULT_FEATURE_DIM, num_iterations=DEFAULT_NUM_ITERATIONS,
                         propagation="left", normalization="l2", attention_temperature=1.0, seed=0,
                         num_workers=None, callback=None, whiten=True):
    """(:206-276).  Iteration 0 is a plain propagation; from iteration 1 on every edge is re-weighted by a
    softmax over its row of cosine(x_r, x_c) / temperature (cleora_edge_attention_dev) before the SpMM."""
    _validate_propagation(propagation)
    _check_device_normalization(normalization)
    if attention_temperature <= 0:
        raise ValueError(f"attention_temperature must be positive, got {attention_temperature}")
    if num_iterations <= 0:
        raise ValueError(f"num_iterations must be positive, got {num_iterations}")
    x0 = graph.initialize_deterministically(feature_dim, seed)
    if graph.num_entities == 0:
        return x0
    kind = _hip.LEFT if propagation == "left" else _hip.SYMMETRIC
    with graph._lock:
        return _device_loop(graph._graph(), graph.num_entities, x0, kind, int(num_iterations), normalization,
                            callback, 0.0, 0.0, whiten, attention_temperature=float(attention_temperature))


def embed_edge_features(graph, edge_features, feature_dim=DEFAULT_FEATURE_DIM,
                        num_iterations=DEFAULT_NUM_ITERATIONS, propagation="left", normalization="l2",
                        combine="concat", num_workers=None, whiten=True):
    """(:784-852).  Structural embedding + the propagation of per-node means of the edge feature vectors."""
    _validate_propagation(propagation)
    struct_emb = embed(graph, feature_dim=feature_dim, num_iterations=num_iterations, propagation=propagation,
                       normalization=normalization, num_workers=num_workers, whiten=whiten)
    if not edge_features:
        return struct_emb
    edge_feat_dim = len(next(iter(edge_features.values())))
    n = graph.num_entities
    index_map = {eid: i for i, eid in enumerate(graph.entity_ids)}
    # per-node mean of the feature vectors of the two-entity keys it takes part in (:812-828): the usable keys
    # become an endpoint list and a feature matrix, scattered with np.add.at (a key "a a" counts twice for a)
    ends, rows = [], []
    for key, feat in edge_features.items():
        tokens = key.split()
        if len(tokens) == 2 and tokens[0] in index_map and tokens[1] in index_map:
            ends.append((index_map[tokens[0]], index_map[tokens[1]]))
            rows.append(np.asarray(feat, dtype=np.float64))
    node_feats = np.zeros((n, edge_feat_dim), dtype=np.float64)
    node_counts = np.zeros(n, dtype=np.float64)
    if ends:
        ends = np.asarray(ends, dtype=np.int64)
        feats = np.stack(rows)
        for side in (0, 1):
            np.add.at(node_feats, ends[:, side], feats)
            np.add.at(node_counts, ends[:, side], 1.0)
    node_feats /= np.maximum(node_counts, 1.0)[:, None]
    kind = _hip.LEFT if propagation == "left" else _hip.SYMMETRIC
    # H is rounded to f32 by _postprocess_iteration after every iteration in the reference too (:839); only its
    # very first SpMM sees the f64 means, which the f32 start rounds (relative 6e-8)
    with graph._lock:
        edge_emb = _device_loop(graph._graph(), n, np.ascontiguousarray(node_feats.astype(np.float32)), kind,
                                int(num_iterations), "l2", None, 0.0, 0.0, whiten) if num_iterations > 0 \
            else node_feats.astype(np.float32)
    if combine == "concat":
        return np.concatenate([struct_emb, edge_emb], axis=1)
    if combine == "mean":
        min_dim = min(struct_emb.shape[1], edge_emb.shape[1])
        return (struct_emb[:, :min_dim] + edge_emb[:, :min_dim]) / 2.0
    if combine == "edge_only":
        return edge_emb
    raise ValueError(f"Unknown combine mode: '{combine}'. Use 'concat', 'mean', or 'edge_only'.")


def predict_links(graph, embeddings, top_k=10, exclude_existing=True, source_entities=None):
    """(:636-681).  For every source entity: cosine similarity to all entities, the source itself and — optionally —
    its stored neighbours in either direction masked with -2, the top_k of the rest; finally the top_k of all
    candidates by score.  Scores, masks and the per-source selection run on the device for all sources in ONE call
    (cleora_topk_cosine_dev: X is read once per 8 sources, one synchronisation at the end); only the
    n_sources x top_k candidates come back."""
    x = np.ascontiguousarray(embeddings, dtype=np.float32)
    n, d = x.shape
    if source_entities is not None:
        source_indices = [graph.get_entity_index(eid) for eid in source_entities]
    else:
        source_indices = list(range(graph.num_entities))
    top_k = int(top_k)
    if not source_indices or top_k <= 0 or n == 0:
        return []
    idx, score = _topk_neighbours(graph, x, source_indices, min(top_k, n), True, bool(exclude_existing))
    # candidates in the reference's append order: source by source, each source's list in descending score (ties: larger
    # index first) without the masked ones (:663-664); then a STABLE sort by score, descending (:679), and the first top_k
    src = np.repeat(np.asarray(source_indices, dtype=np.int64), idx.shape[1])
    keep = score.ravel() > -2.0
    src, tgt, sc = src[keep], idx.ravel()[keep].astype(np.int64), score.ravel()[keep]
    order = np.argsort(-sc.astype(np.float64), kind="stable")[:top_k]
    ids = graph.entity_ids
    return [{"source": ids[int(src[i])], "target": ids[int(tgt[i])], "score": float(sc[i])} for i in order]


TOPK_DEVICE_MAX = 1024      # cleora_topk_cosine_dev selects at most this many per query (csrc/similarity.hip)


def _topk_by_host_selection(graph, x, query_rows, k, exclude_self, exclude_existing, self_score=-2.0):
    """The same result for k > TOPK_DEVICE_MAX (the reference accepts any top_k): the scores still come from the device
    (cleora_cosine_scores_dev, one pass over X per query), the masks and the selection run on the host like the
    reference's `np.argsort(sims)[::-1][:top_k]` (pycleora/__init__.py:663, 771)."""
    L = _hip.lib()
    n, d = x.shape
    dx = _hip.DevArray.from_host(x)
    dq, ds = _hip.DevArray((d,), np.float32), _hip.DevArray((n,), np.float32)
    out_i, out_s = np.empty((len(query_rows), k), np.uint32), np.empty((len(query_rows), k), np.float32)
    nbr_out = nbr_in = None
    if exclude_existing:
        rows, cols = (np.asarray(a, dtype=np.int64) for a in graph.to_sparse_csr()[:2])
        by_row, by_col = np.argsort(rows, kind="stable"), np.argsort(cols, kind="stable")
        nbr_out = (np.searchsorted(rows[by_row], np.arange(n + 1)), cols[by_row])     # (r, *) stored
        nbr_in = (np.searchsorted(cols[by_col], np.arange(n + 1)), rows[by_col])      # (*, r) stored
    for j, q in enumerate(query_rows):
        v = x[q].astype(np.float32)
        v = v / max(float(np.linalg.norm(v)), 1e-10)
        _hip.check(L.cleora_memcpy_h2d(dq.ptr, _hip.ptr(np.ascontiguousarray(v)), dq.nbytes, None))
        _hip.check(L.cleora_cosine_scores_dev(dx.ptr, d, n, d, dq.ptr, ds.ptr, None))
        _hip.check(L.cleora_stream_sync(None))
        sims = ds.to_host()
        if exclude_self:
            sims[q] = self_score
        if exclude_existing:
            for ptr, other in (nbr_out, nbr_in):
                sims[other[ptr[q]:ptr[q + 1]]] = -2.0
        top = np.argsort(sims)[::-1][:k]
        out_i[j], out_s[j] = top, sims[top]
    return out_i, out_s


def _topk_neighbours(graph, x, query_rows, k, exclude_self, exclude_existing):
    """(index uint32[nq, k], score f32[nq, k]) of the k most cosine-similar rows of x for every query row."""
    if k > TOPK_DEVICE_MAX:
        return _topk_by_host_selection(graph, x, query_rows, k, exclude_self, exclude_existing)
    L = _hip.lib()
    n, d = x.shape
    nq = len(query_rows)
    dx = _hip.DevArray.from_host(x)
    dq = _hip.DevArray.from_host(np.asarray(query_rows, dtype=np.uint32))
    oi, os_ = _hip.DevArray((nq, k), np.uint32), _hip.DevArray((nq, k), np.float32)
    ws = _hip.DevArray((L.cleora_topk_workspace_for(n, k, nq),), np.uint8)
    with graph._lock:
        g = graph._graph().handle if exclude_existing else None
        _hip.check(L.cleora_topk_cosine_dev(g, dx.ptr, d, n, d, dq.ptr, nq, k, 1 if exclude_self else 0,
                                            1 if exclude_existing else 0, oi.ptr, os_.ptr, ws.ptr, None))
        _hip.check(L.cleora_stream_sync(None))
    return oi.to_host(), os_.to_host()
