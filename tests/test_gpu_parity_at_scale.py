"""GPU: end-to-end parity of the two loops of pycleora.embed() at BASELINE config 2's size (|V| = 1M, |E| = 20M,
d = 256) against the CPU oracle's loops on the same graph and the same E_0 — the checks VERDICT round 2 found missing.

  * the DEFAULT loop (whiten=True: propagate, L2-normalise, whiten_embeddings every iteration,
    pycleora/__init__.py:109-117): `cleora_embed_dev(..., CLEORA_F_WHITEN)` — the product's reorganised loop (SpMM before
    the projection, Cholesky whitening and f32-matrix-core Gram in the intermediate iterations, split-bf16 projection) —
    against oracle.whiten.embed_slow over oracle.spmm.  The result of PCA whitening is defined up to the sign of every
    column (and up to a rotation inside a cluster of nearly equal eigenvalues), so the comparison uses what is invariant:
    the pairwise cosines of a 2 000-row sample, the row norms (Mahalanobis distances), and the covariance of the result.
  * the PLAIN loop (embed_fast, src/embedding.rs:106-136) for the full 40 iterations: every row — hub rows included, since
    spmm.hip's hub_inorder_kernel adds them edge by edge like the reference — is BIT-EQUAL to the oracle's after every
    iteration.  The oracle's 40 iterations are pinned as hashes (tests/golden/plain_loop_hashes_C2.json, written by
    tests/golden/make_plain_loop_hashes.py from the oracle on the GPU box); the test re-runs the oracle live for the first
    iterations (array_equal) and compares the GPU's hashes with the record at 1, 2, 3, 5, 10, 20 and 40.  Only with
    CLEORA_F_HUB_SEGMENTS does the loop drift (hub rows summed in segments): held to 5e-5, as before.

Tolerances (stated; the measured values of the last GPU run are merged into gpurun_out/r06_parity_at_scale.json — started from
the committed profiles/r06_parity_at_scale.json, so a partial run never drops a key — and quoted in DESIGN.md §4):
  whitened loop, 4 iterations:  max |cos_gpu - cos_oracle| <= 1e-4, relative row-norm difference <= 1e-4,
                                max |cov(E_gpu) - I| <= 1e-3 over all rows      (measured round 3: 1.3e-6, 6.1e-7)
  plain loop, 40 iterations:    E_gpu == E_oracle bit for bit; with CLEORA_F_HUB_SEGMENTS max |E_gpu - E_oracle| <= 5e-5 on
                                unit-norm rows (measured round 3: 2.4e-5; 49 hub rows, longest 14 170 edges)
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


RECORD = "r06_parity_at_scale.json"


def _record(key, values):
    """Measured values for DESIGN.md / profiles/ (gpurun_out/ is merged back from the GPU box).  The file starts from the
    committed profiles/ copy, so a run of a subset of these tests keeps every other key (round 4 lost three that way)."""
    path = os.path.join(ROOT, "gpurun_out", RECORD)
    base = os.path.join(ROOT, "profiles", RECORD)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        src = path if os.path.exists(path) else base
        data = json.load(open(src)) if os.path.exists(src) else {}
        data[key] = values
        json.dump(data, open(path, "w"), indent=1)
    except (OSError, ValueError):
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


class _OracleWhitenedLoop:
    """The oracle's default loop (oracle.whiten.embed_slow over oracle.spmm) on the C2 graph, advanced ONE iteration at a time and
    shared by the tests of this module: each iteration costs 20 GB of CPU gathers + an fp64 Gram of 1M x 256, so nobody repeats one."""

    def __init__(self, host, x0):
        self.host, self.x, self.done, self.seconds, self.snap = host, x0, 0, 0.0, {}
        self.threads = oracle.max_threads()

    def advance_to(self, iters, keep=(), budget_s=None):
        import time
        while self.done < iters:
            t0 = time.perf_counter()
            self.x, _ = ow.embed_slow(lambda v: oracle.spmm(self.host["rowptr"], self.host["col"], self.host["val"], v, self.threads), self.x, 1, whiten=True)
            self.done += 1
            dt = time.perf_counter() - t0
            self.seconds += dt
            if self.done in keep:
                self.snap[self.done] = self.x
            if budget_s is not None and self.seconds + dt > budget_s:
                break
        return self.x, self.done


@pytest.fixture(scope="module")
def oracle_whitened(c2):
    n, nnz, graph, host, hashes = c2
    return _OracleWhitenedLoop(host, oracle.init(hashes, 256, 0))


def test_whitened_loop_at_c2_size_against_the_oracle_loop(c2, oracle_whitened):
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
    oracle_whitened.advance_to(iters, keep=(iters,))
    want = oracle_whitened.snap[iters]

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


def test_whitened_loop_record_at_c2_size_is_what_the_live_oracle_computes_and_the_gpu_matches_it(c2, oracle_whitened):
    """the default loop at BASELINE's FULL sizes (config 3: 10M x 256, config 5: 2M x 1024) is pinned by
    records of the oracle's loop (tests/golden/whitened_loop_{C3,C5}.npz, written by tests/golden/make_whitened_loop_record.py on the
    GPU box; bench.py compares the GPU with them in every run: `whitened.checks.vs_oracle_record`).  Here the same machinery at
    config 2's size, where the oracle also runs LIVE: (1) the record equals what the live oracle loop computes (invariants to 1e-5:
    the oracle itself is only reproducible to the order of its BLAS's summation), (2) the GPU's default loop and its
    reference-order loop match the record within the stated tolerances — cosines and row norms 1e-4, spectrum 1e-4 of the largest
    eigenvalue, |cov - I| over all rows 1e-3."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_whitened_loop_record as wrec
    from tests.golden.make_plain_loop_hashes import hash_array
    n, nnz, graph, host, hashes = c2
    d = 256
    rec = wrec.load_record("C2")
    assert rec is not None, "tests/golden/whitened_loop_C2.npz is missing"
    m = rec["meta"]
    here = "-".join(hash_array(np.ascontiguousarray(host[k]).reshape(-1, 1)) for k in ("rowptr", "col", "val"))
    x0 = oracle.init(hashes, d, 0)
    if (m["n"], m["nnz"], m["d"]) != (n, nnz, d) or m["graph"] != here or m["x0"] != hash_array(x0):
        pytest.skip("the golden record describes another graph (generator output differs on this torch build)")
    iters = int(m["iterations"])
    # (1) the live oracle (the module's shared loop: iterations 1-4 are also what the test above uses)
    oracle_whitened.advance_to(iters, keep=(iters,))
    want = oracle_whitened.snap[iters]
    live = wrec.compare(rec, want[rec["rows"]], np.linalg.norm(want[rec["norm_rows"]].astype(np.float64), axis=1))
    assert live["max_abs_cosine_diff"] <= 1e-5 and live["max_rel_row_norm_diff"] <= 1e-5, live
    # (2) the GPU against the record
    L = _hip.lib()
    x0_dev = torch.from_numpy(x0).to("cuda:0")
    res = {"default_loop": wrec.gpu_invariants(L, graph, x0_dev, n, d, rec),
           "reference_order_loop": wrec.gpu_invariants(L, graph, x0_dev, n, d, rec, threshold=1e-30)}
    _record("whitened_loop_record_c2", {"record_vs_live_oracle": {k: live[k] for k in ("max_abs_cosine_diff", "max_rel_row_norm_diff")},
                                        "gpu_vs_record": {k: {a: b for a, b in v.items() if a != "tolerances"} for k, v in res.items()}})
    for name, r in res.items():
        assert r["finite"] and r["within_tolerance"], (name, r)
        assert r["max_abs_cov_minus_identity_all_rows"] <= 1e-3, (name, r)


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


def test_default_loop_for_forty_iterations_against_the_reference_order(c2, oracle_whitened):
    """the shipped default loop is a RE-ORDERING of the reference's operations (SpMM commuted past
    the projection, Cholesky whitening and split-bf16 Gram in the intermediate iterations, normalisation in the projection's
    epilogue); its equivalence holds in exact arithmetic, and round 3 showed it for 3-4 iterations only.  Here: the reference
    default of 40 iterations (pycleora/__init__.py:12-13), at BASELINE config 2's size, at 10, 20 and 40 iterations, against

      (a) the REFERENCE'S ORDER on the same GPU — SpMM, L2 norm, f64 Gram, eigensolver, projection, every iteration
          (pycleora/__init__.py:109-117) — which `cleora_embed_dev` runs when a convergence threshold is set (a threshold that
          is never met: 1e-30) and which test_whitened_loop_at_c2_size_against_the_oracle_loop's sibling below pins to the
          oracle loop;
      (b) oracle.whiten.embed_slow (numpy fp64 statistics, LAPACK eigh: the reference's own arithmetic) for 7 iterations,
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
    # (b) the oracle's loop, 10 iterations (continuing the module's shared loop: iterations 1-4 were run by the test above), time-boxed
    t0, threads = time.perf_counter(), oracle_whitened.threads
    x, oracle_iters = oracle_whitened.advance_to(7, budget_s=150.0)        # (each oracle iteration is ~3 s of all host cores: 7, not 10 — the record test pins 4, the GPU's own orders are compared at 10 / 20 / 40)
    got = gpu(oracle_iters, 0.0)
    cg, ng = _invariants(got, rows)
    co, no = _invariants(x, rows)
    results["oracle"] = {"iterations": oracle_iters, "oracle_seconds": round(oracle_whitened.seconds, 1), "oracle_threads": threads,
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


def _golden(config):
    path = os.path.join(ROOT, "tests", "golden", f"plain_loop_hashes_{config}.json")
    return json.load(open(path)) if os.path.exists(path) else None


def test_forty_iterations_of_the_plain_loop_at_c2_size_are_bit_equal(c2):
    """40 iterations of embed_fast on the GPU against the oracle's (oracle.spmm + oracle.l2_normalize per iteration — the
    arithmetic of src/embedding.rs:106-136, rw = 0): array_equal.  The oracle runs live for the first LIVE iterations; its
    iterates at 1, 2, 3, 5, 10, 20, 40 are pinned as hashes by tests/golden/make_plain_loop_hashes.py (the same oracle, run
    once on the GPU box), and the record's hashes of the graph and of E_0 must match the ones built here — if they do not (another
    generator), the oracle runs all 40 iterations live instead."""
    from tests.golden.make_plain_loop_hashes import graph_hash, hash_array
    import torch
    n, nnz, graph, host, hashes = c2
    d, LIVE = 256, 3
    L = _hip.lib()
    x0 = oracle.init(hashes, d, 0)
    threads = oracle.max_threads()
    deg = np.diff(host["rowptr"].astype(np.int64))
    hub = deg > graph.info().hub_threshold
    gold = _golden("C2")
    here = "-".join(hash_array(np.ascontiguousarray(host[k]).reshape(-1, 1)) for k in ("rowptr", "col", "val"))
    pinned = bool(gold) and gold["graph"] == here and gold["x0"] == hash_array(x0) and gold["d"] == d and gold["hash"] == "xxh3_128"
    checkpoints = sorted(int(k) for k in gold["iterations"]) if pinned else [1, 2, 3, 5, 10, 20, 40]

    def gpu(iters, flags=0):
        dx = _hip.DevArray.from_host(x0)
        ran = ctypes.c_uint64(0)
        _hip.check(L.cleora_embed_dev(graph.handle, dx.ptr, _hip.LEFT, d, iters, 0.0, 0.0, flags, ctypes.byref(ran)))
        assert ran.value == iters
        return dx.to_host()

    x, results = x0, {}
    live = LIVE if pinned else checkpoints[-1]
    for it in range(1, live + 1):
        x = oracle.l2_normalize(oracle.spmm(host["rowptr"], host["col"], host["val"], x, threads), threads)
        if it in checkpoints or it <= LIVE:
            got = gpu(it)
            np.testing.assert_array_equal(got, x, err_msg=f"iteration {it}")
            results[it] = {"bit_equal_to_live_oracle": True}
            if pinned and str(it) in gold["iterations"]:
                assert hash_array(x) == gold["iterations"][str(it)], f"the golden record disagrees with the live oracle at iteration {it}"
    if pinned:
        for it in checkpoints:
            h = hash_array(gpu(it))
            assert h == gold["iterations"][str(it)], f"GPU iterate after {it} iterations differs from the oracle's (golden record)"
            results.setdefault(it, {})["hash_equal_to_oracle_record"] = True
    # the segmented hub sum drifts (within the stated tolerance): what round 3 / 4 measured for the default, now opt-in
    # (against the in-order GPU iterate of the same 10 iterations, which the checks above tie to the oracle's)
    seg_diff = float(np.abs(gpu(10, _hip.F_HUB_SEGMENTS).astype(np.float64) - gpu(10).astype(np.float64)).max())
    _record("plain_loop_c2", {"n": n, "nnz": nnz, "d": d, "hub_rows": int(hub.sum()), "longest_row": int(deg.max()), "oracle_threads": threads,
                              "pinned_by_golden_record": pinned, "live_oracle_iterations": live, "compared_at_iterations": {str(k): v for k, v in sorted(results.items())},
                              "bit_equal_through_iteration": max(results), "hub_segments_flag_max_abs_diff_after_10": seg_diff})
    assert max(results) == 40
    assert seg_diff <= 5e-5, seg_diff


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
