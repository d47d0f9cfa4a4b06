"""GPU: end-to-end parity of the two loops of pycleora.embed() at BASELINE config 2's size (|V| = 1M, |E| = 20M,
d = 256) against the CPU oracle's loops on the same graph and the same E_0 — the checks VERDICT round 2 found missing.

  * the DEFAULT loop (whiten=True: propagate, L2-normalise, whiten_embeddings every iteration,
    pycleora/__init__.py:109-117): `cleora_embed_dev(..., CLEORA_F_WHITEN)` — the product's reorganised loop (SpMM before
    the projection, Cholesky whitening and f32-matrix-core Gram in the intermediate iterations, split-bf16 projection) —
    against oracle.whiten.embed_slow over oracle.spmm.  The result of PCA whitening is defined up to the sign of every
    column (and up to a rotation inside a cluster of nearly equal eigenvalues), so the comparison uses what is invariant:
    the pairwise cosines of a 2 000-row sample, the row norms (Mahalanobis distances), and the covariance of the result.
  * the PLAIN loop (embed_fast, src/embedding.rs:106-136) for the full 40 iterations: the drift of the GPU iterate from
    the oracle's.  Rows longer than 1024 edges are summed in segment order on the GPU (more accurate than, but not equal
    to, the reference's one-accumulator order), and after one iteration every row depends on them, so end-to-end equality
    is not bit-exact on a graph with hub rows: this test MEASURES the drift and holds it to the stated fp32 tolerance.

Tolerances (stated; the measured values of the last GPU run are written to gpurun_out/r04_parity_at_scale.json and quoted in
DESIGN.md §4):
  whitened loop, 4 iterations:  max |cos_gpu - cos_oracle| <= 1e-4, relative row-norm difference <= 1e-4,
                                max |cov(E_gpu) - I| <= 1e-3 over all rows      (measured round 3: 1.3e-6, 6.1e-7)
  plain loop, 40 iterations:    max |E_gpu - E_oracle| <= 5e-5 on unit-norm rows (measured round 3: 2.4e-5 max, 1.1e-7 rms,
                                49 hub rows, longest 14 170 edges) — north_star's "stated fp32 tolerance for embedding values"
"""
import ctypes
import json
import os

import numpy as np
import pytest

import oracle
from cleora_amd import _hip
from oracle import whiten as ow

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(key, values):
    """Measured values for DESIGN.md / profiles/ (gpurun_out/ is merged back from the GPU box)."""
    path = os.path.join(ROOT, "gpurun_out", "r04_parity_at_scale.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[key] = values
        json.dump(data, open(path, "w"), indent=1)
    except OSError:
        pass


@pytest.fixture(scope="module")
def c2():
    import torch
    from cleora_amd import synth
    dev = torch.device("cuda:0")
    g = synth.bipartite_graph(500_000, 500_000, 10_000_000, 1, dev)
    n, nnz = g["n"], g["nnz"]
    graph = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(),
                                   g["val_sym"].data_ptr(), 0, keepalive=g)
    host = dict(rowptr=g["rowptr"].cpu().numpy().astype(np.uint64), col=g["col"].cpu().numpy().view(np.uint32),
                val=g["val_left"].cpu().numpy())
    hashes = synth.entity_hashes(n, 0, dev).cpu().numpy().view(np.uint64)
    yield n, nnz, graph, host, hashes
    graph.close()


def test_whitened_loop_at_c2_size_against_the_oracle_loop(c2):
    n, nnz, graph, host, hashes = c2
    d, iters = 256, 4
    L = _hip.lib()
    x0 = oracle.init(hashes, d, 0)
    dx = _hip.DevArray.from_host(x0)
    ran = ctypes.c_uint64(0)
    _hip.check(L.cleora_embed_dev(graph.handle, dx.ptr, _hip.LEFT, d, iters, 0.0, 0.0, _hip.F_WHITEN, ctypes.byref(ran)))
    assert ran.value == iters
    got = dx.to_host()
    assert np.isfinite(got).all()
    threads = oracle.max_threads()
    want, _ = ow.embed_slow(lambda v: oracle.spmm(host["rowptr"], host["col"], host["val"], v, threads), x0, iters, whiten=True)

    rows = np.random.default_rng(11).choice(n, 2000, replace=False)

    def cosines(e):
        s = e[rows].astype(np.float64)
        s /= np.linalg.norm(s, axis=1, keepdims=True)
        return s @ s.T

    cos_err = float(np.abs(cosines(got) - cosines(want)).max())
    ng, nw = np.linalg.norm(got.astype(np.float64), axis=1), np.linalg.norm(want.astype(np.float64), axis=1)
    norm_err = float((np.abs(ng - nw) / nw).max())
    cov = np.cov(got.astype(np.float64).T)                  # ALL rows: a prefix of a bipartite graph is one side of it
    cov_err = float(np.abs(cov - np.eye(d)).max())
    cov_ref_err = float(np.abs(np.cov(want.astype(np.float64).T) - np.eye(d)).max())
    # column-wise agreement where the spectrum separates the columns (informative, not asserted: trailing columns of nearly
    # equal eigenvalues may come out rotated against each other)
    sgn = np.sign((got[:100_000] * want[:100_000]).sum(axis=0))
    col_err = np.abs(got[:100_000] * sgn - want[:100_000]).max(axis=0) / np.abs(want).max()
    _record("whitened_loop_c2", {"n": n, "nnz": nnz, "d": d, "iterations": iters, "max_abs_cosine_diff_2000_rows": cos_err,
                                 "max_rel_row_norm_diff": norm_err, "max_abs_cov_minus_identity_gpu": cov_err,
                                 "max_abs_cov_minus_identity_oracle": cov_ref_err,
                                 "sign_aligned_columns_within_1e-2": int((col_err < 1e-2).sum()),
                                 "median_sign_aligned_column_error": float(np.median(col_err))})
    assert cos_err <= 1e-4, cos_err
    assert norm_err <= 1e-4, norm_err
    assert cov_err <= 1e-3, cov_err


def test_default_loop_with_a_residual_blend_at_c2_size(c2):
    """The blended form of the default loop (residual_weight > 0: pycleora/__init__.py:111-115 blends for ANY rw > 0) at C2 size,
    20 iterations: the reorganised loop carries the blend through the projection (alpha (Z - s mu^T) + rw (Y - mu)) T — a different
    kernel instantiation and operand ring than rw = 0 — against the reference's order on the same GPU.  Stated: cosines <= 1e-4."""
    n, nnz, graph, host, hashes = c2
    d, iters, rw = 256, 20, 0.25
    L = _hip.lib()
    x0 = oracle.init(hashes, d, 0)
    rows = np.random.default_rng(12).choice(n, 2000, replace=False)
    outs = []
    for thr in (0.0, 1e-30):
        dx = _hip.DevArray.from_host(x0)
        ran = ctypes.c_uint64(0)
        _hip.check(L.cleora_embed_dev(graph.handle, dx.ptr, _hip.LEFT, d, iters, rw, thr, _hip.F_WHITEN, ctypes.byref(ran)))
        assert ran.value == iters
        outs.append(dx.to_host())
    (cg, ng), (cr, nr) = _invariants(outs[0], rows), _invariants(outs[1], rows)
    res = {"iterations": iters, "residual_weight": rw, "max_abs_cosine_diff_2000_rows": float(np.abs(cg - cr).max()),
           "max_rel_row_norm_diff": float((np.abs(ng - nr) / nr).max())}
    _record("default_loop_blend_c2", res)
    assert np.isfinite(outs[0]).all()
    assert res["max_abs_cosine_diff_2000_rows"] <= 1e-4 and res["max_rel_row_norm_diff"] <= 1e-4, res


def _invariants(e, rows):
    s = e[rows].astype(np.float64)
    s /= np.linalg.norm(s, axis=1, keepdims=True)
    return s @ s.T, np.linalg.norm(e.astype(np.float64), axis=1)


def test_default_loop_for_forty_iterations_against_the_reference_order(c2):
    """the shipped default loop is a RE-ORDERING of the reference's operations (SpMM commuted past
    the projection, Cholesky whitening and split-bf16 Gram in the intermediate iterations, normalisation in the projection's
    epilogue); its equivalence holds in exact arithmetic, and round 3 showed it for 3-4 iterations only.  Here: the reference
    default of 40 iterations (pycleora/__init__.py:12-13), at BASELINE config 2's size, at 10, 20 and 40 iterations, against

      (a) the REFERENCE'S ORDER on the same GPU — SpMM, L2 norm, f64 Gram, eigensolver, projection, every iteration
          (pycleora/__init__.py:109-117) — which `cleora_embed_dev` runs when a convergence threshold is set (a threshold that
          is never met: 1e-30) and which test_whitened_loop_at_c2_size_against_the_oracle_loop's sibling below pins to the
          oracle loop;
      (b) oracle.whiten.embed_slow (numpy fp64 statistics, LAPACK eigh: the reference's own arithmetic) for 10 iterations,
          inside a time budget (the CPU side is 10 x (20 GB of gathers + an fp64 Gram of 1M x 256)).

    PCA whitening is defined up to column signs and rotations inside clusters of equal eigenvalues, so the comparison is on
    invariants: pairwise cosines of 2 000 rows, row norms, covariance of the result.
    Stated tolerances: cosines <= 1e-4, relative row norms <= 1e-4, |cov - I| <= 1e-3, at every checkpoint."""
    import time
    n, nnz, graph, host, hashes = c2
    d = 256
    L = _hip.lib()
    x0 = oracle.init(hashes, d, 0)
    rows = np.random.default_rng(11).choice(n, 2000, replace=False)

    def gpu(iters, threshold):
        dx = _hip.DevArray.from_host(x0)
        ran = ctypes.c_uint64(0)
        _hip.check(L.cleora_embed_dev(graph.handle, dx.ptr, _hip.LEFT, d, iters, 0.0, threshold, _hip.F_WHITEN, ctypes.byref(ran)))
        assert ran.value == iters
        return dx.to_host()

    results = {}
    for iters in (10, 20, 40):
        got, ref = gpu(iters, 0.0), gpu(iters, 1e-30)
        assert np.isfinite(got).all() and np.isfinite(ref).all()
        cg, ng = _invariants(got, rows)
        cr, nr = _invariants(ref, rows)
        cov_err = float(np.abs(np.cov(got.astype(np.float64).T) - np.eye(d)).max())
        results[iters] = {"max_abs_cosine_diff_2000_rows": float(np.abs(cg - cr).max()), "max_rel_row_norm_diff": float((np.abs(ng - nr) / nr).max()),
                          "max_abs_cov_minus_identity": cov_err}
    # (b) the oracle's loop, 10 iterations, time-boxed
    threads, t0, x, oracle_iters = oracle.max_threads(), time.perf_counter(), x0, 0
    per_iter = None
    for it in range(10):
        t1 = time.perf_counter()
        x, _ = ow.embed_slow(lambda v: oracle.spmm(host["rowptr"], host["col"], host["val"], v, threads), x, 1, whiten=True)
        oracle_iters += 1
        per_iter = time.perf_counter() - t1
        if time.perf_counter() - t0 + per_iter > 150.0 and oracle_iters < 10:
            break
    got = gpu(oracle_iters, 0.0)
    cg, ng = _invariants(got, rows)
    co, no = _invariants(x, rows)
    results["oracle"] = {"iterations": oracle_iters, "oracle_seconds": round(time.perf_counter() - t0, 1), "oracle_threads": threads,
                         "max_abs_cosine_diff_2000_rows": float(np.abs(cg - co).max()), "max_rel_row_norm_diff": float((np.abs(ng - no) / no).max())}
    _record("default_loop_40_iterations_c2", {"n": n, "nnz": nnz, "d": d, "vs_reference_order_on_gpu": {str(k): v for k, v in results.items() if k != "oracle"},
                                              "vs_oracle_loop": results["oracle"]})
    for iters in (10, 20, 40):
        r = results[iters]
        assert r["max_abs_cosine_diff_2000_rows"] <= 1e-4, (iters, r)
        assert r["max_rel_row_norm_diff"] <= 1e-4, (iters, r)
        assert r["max_abs_cov_minus_identity"] <= 1e-3, (iters, r)
    assert results["oracle"]["iterations"] >= 4
    assert results["oracle"]["max_abs_cosine_diff_2000_rows"] <= 1e-4, results["oracle"]
    assert results["oracle"]["max_rel_row_norm_diff"] <= 1e-4, results["oracle"]


def test_forty_iteration_drift_of_the_plain_loop_at_c2_size(c2):
    """40 iterations of embed_fast on the GPU against 40 iterations of the oracle (the same arithmetic as oracle.embed:
    oracle.spmm + oracle.l2_normalize per iteration), compared at iterations 1, 10, 20 and 40.  The oracle side is a CPU job
    of 20 GB of random gathers per iteration: on a box that grants this process only a few cores it is cut off after
    ORACLE_BUDGET_S seconds and the comparison is made at the last checkpoint reached (at least 10 iterations)."""
    import time
    n, nnz, graph, host, hashes = c2
    d, checkpoints, budget_s = 256, (1, 10, 20, 40), 75.0
    L = _hip.lib()
    x0 = oracle.init(hashes, d, 0)
    threads = oracle.max_threads()
    deg = np.diff(host["rowptr"].astype(np.int64))
    hub = deg > graph.info().hub_threshold

    def gpu(iters):
        dx = _hip.DevArray.from_host(x0)
        ran = ctypes.c_uint64(0)
        _hip.check(L.cleora_embed_dev(graph.handle, dx.ptr, _hip.LEFT, d, iters, 0.0, 0.0, 0, ctypes.byref(ran)))
        assert ran.value == iters
        return dx.to_host()

    x, t0, results = x0, time.perf_counter(), {}
    for it in range(1, checkpoints[-1] + 1):
        x = oracle.l2_normalize(oracle.spmm(host["rowptr"], host["col"], host["val"], x, threads), threads)   # src/embedding.rs:106-136, rw = 0
        if it in checkpoints:
            got = gpu(it)
            diff = np.abs(got.astype(np.float64) - x.astype(np.float64))
            results[it] = {"max_abs_diff": float(diff.max()), "rms_diff": float(np.sqrt((diff ** 2).mean()))}
            if it == 1:
                same = (got.view(np.uint32) == x.view(np.uint32)).all(axis=1)
                results[it]["rows_bit_equal"] = int(same.sum())
                assert same[~hub].all()                    # one iteration: every row without a split is bit-equal
            assert np.isfinite(got).all()
            if time.perf_counter() - t0 > budget_s and it >= 10:
                break
    last = max(results)
    _record("plain_loop_drift_c2", {"n": n, "nnz": nnz, "d": d, "hub_rows": int(hub.sum()), "non_hub_rows": int((~hub).sum()),
                                    "longest_row": int(deg.max()), "oracle_threads": threads, "compared_at_iterations": results,
                                    "oracle_seconds": round(time.perf_counter() - t0, 1)})
    for it, r in results.items():
        assert r["max_abs_diff"] <= 5e-5, (it, r)
    assert last >= 10


def test_config5_hypergraph_d1024_whitened_loop_against_the_oracle_loop():
    """BASELINE config 5's flavour at a size the oracle finishes in seconds: `complex::reflexive::product` hyperedges (the bench's
    generator: arity 2..14, Zipf-like products) through the C++ host builder, d = 1024 — the shapes this path takes there and
    nowhere else: spmm_rows_kernel<64,4,4>, the split projection in four column passes, the f64 Gram in 8 x 8 block tiles,
    rocSOLVER's Cholesky / eigensolver at d = 1024 — three iterations of the default loop against oracle.whiten.embed_slow.
    Stated: pairwise cosines of 1 500 rows to 1e-4, row norms to 1e-4 relative, covariance of the result within 2e-3 of I."""
    import bench
    from cleora_amd import _host
    n_lines, products, d, iters = 60_000, 20_000, 1024, 3      # ~20k rows: the numpy fp64 whitening at d = 1024 stays in seconds
    data, offsets, _ = bench.hypergraph_lines(n_lines, products, 5)
    h = _host.vp()
    assert _host.lib().cleora_host_build_from_lines(data, offsets.ctypes.data_as(_host.vp), n_lines, b"complex::reflexive::product", 16,
                                                    ctypes.byref(h)) == 0
    hg = _host.HostGraph(h)
    a = hg.arrays()
    n = a["rowptr"].shape[0] - 1
    assert n > 4 * d                                     # full-rank covariance
    g = _hip.Graph.from_host(a["rowptr"], a["col"], a["val_left"], a["val_sym"])
    L = _hip.lib()
    x0 = oracle.init(a["hashes"], d, 0)
    dx = _hip.DevArray.from_host(x0)
    _hip.check(L.cleora_embed_dev(g.handle, dx.ptr, _hip.LEFT, d, iters, 0.0, 0.0, _hip.F_WHITEN, None))
    got = dx.to_host()
    threads = oracle.max_threads()
    want, _ = ow.embed_slow(lambda v: oracle.spmm(a["rowptr"], a["col"], a["val_left"], v, threads), x0, iters, whiten=True)
    rows = np.random.default_rng(2).choice(n, 1500, replace=False)

    def cosines(e):
        s = e[rows].astype(np.float64)
        s /= np.linalg.norm(s, axis=1, keepdims=True)
        return s @ s.T

    cos_err = float(np.abs(cosines(got) - cosines(want)).max())
    ng, nw = np.linalg.norm(got.astype(np.float64), axis=1), np.linalg.norm(want.astype(np.float64), axis=1)
    norm_err = float((np.abs(ng - nw) / nw).max())
    cov_err = float(np.abs(np.cov(got.astype(np.float64).T) - np.eye(d)).max())
    _record("whitened_loop_c5_flavour", {"n": int(n), "nnz": int(a["col"].shape[0]), "d": d, "iterations": iters,
                                         "longest_row": int(np.diff(a["rowptr"].astype(np.int64)).max()),
                                         "max_abs_cosine_diff_1500_rows": cos_err, "max_rel_row_norm_diff": norm_err,
                                         "max_abs_cov_minus_identity_gpu": cov_err})
    g.close()
    assert np.isfinite(got).all()
    assert cos_err <= 1e-4, cos_err
    assert norm_err <= 1e-4, norm_err
    assert cov_err <= 2e-3, cov_err
