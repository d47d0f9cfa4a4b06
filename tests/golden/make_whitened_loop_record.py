"""Golden record of the ORACLE's default (whitened) loop — pycleora.embed(graph, d, k) with whiten=True:
propagate, `_normalize(..., "l2")`, `whiten_embeddings` every iteration (pycleora/__init__.py:109-117, 130-164, 942-946) —
on the bench's synthetic workloads at BASELINE's full sizes (config 3: 10M x 256; config 5: 2M x 1024).

Why a record: one oracle iteration at config 3 is 200 GB of CPU gathers, a numpy pass over 10 GB for the norm, an fp64
covariance of 10M x 256 in 50 000-row chunks (1.3 Tflop) and a 1.3 Tflop f32 projection: minutes of all host cores, once —
not in every bench run.  PCA whitening is defined up to the sign of every column (and up to a rotation inside a cluster of
nearly equal eigenvalues), so the loop is not pinned by hashes as the plain loop is (make_plain_loop_hashes.py) but by what is
invariant under those: for SAMPLE fixed rows the rows themselves (their pairwise cosines and norms are invariant; 2 000 rows at
d = 256 (1 000 at config 2), 768 at d = 1024 to keep the fixture at 1.3-3.5 MB), the row norms of 25 000 further rows, the descending eigenvalue
spectrum of the covariance the LAST iteration whitened, and max |cov(result) - I| over all rows.

Runs on the GPU box (the graphs come from the bench's generators, cleora_amd/synth.py on the GPU / the C++ host builder):

    gpurun -- 'python tests/golden/make_whitened_loop_record.py --config C3 --iters 4'        (C5: --iters 3; C2: --iters 4)

and writes gpurun_out/whitened_loop_<config>.npz, committed as tests/golden/whitened_loop_<config>.npz.  After the oracle it
runs the GPU loop (`cleora_embed_dev + CLEORA_F_WHITEN`) once and prints the comparison, so that a record is never committed
blind.  The oracle side touches nothing of the HIP library: oracle.spmm (oracle/cleora_oracle.c) + oracle/whiten.py (numpy).

`compare()` / `gpu_invariants()` below are the checker bench.py (`whitened.checks.vs_oracle_record`) and
tests/test_gpu_parity_at_scale.py share.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from make_plain_loop_hashes import graph_hash, hash_array  # noqa: E402

SAMPLE_SEED, NORM_SEED, NORM_ROWS = 11, 12, 25_000
TOL = {"cosine": 1e-4, "row_norm_rel": 1e-4, "spectrum_rel_to_largest": 1e-4}


def sample_rows(n, d):
    k = min(n, (2000 if n > 2_000_000 else 1000) if d <= 256 else 768)      # a 1.3 - 3.5 MB fixture
    return np.sort(np.random.default_rng(SAMPLE_SEED).choice(n, k, replace=False))


def norm_rows(n):
    return np.sort(np.random.default_rng(NORM_SEED).choice(n, min(n, NORM_ROWS), replace=False))


def record_path(config):
    return os.path.join(ROOT, "tests", "golden", f"whitened_loop_{config}.npz")


def load_record(config, path=None):
    path = path or record_path(config)
    if not os.path.exists(path):
        return None
    z = np.load(path, allow_pickle=False)
    rec = {k: z[k] for k in z.files if k != "meta"}
    rec["meta"] = json.loads(str(z["meta"]))
    return rec


def _cosines(rows_f32):
    s = np.asarray(rows_f32, dtype=np.float64)
    s = s / np.linalg.norm(s, axis=1, keepdims=True)
    return s @ s.T


def compare(rec, sample, norms, spectrum=None):
    """GPU-side invariants against the record: `sample` = the result's rows rec['rows'] (f32), `norms` = f64 norms of the result's
    rows rec['norm_rows'], `spectrum` = descending eigenvalues of the covariance the last iteration whitened (or None)."""
    out = {"iterations": int(rec["meta"]["iterations"]), "sample_rows": int(rec["rows"].shape[0]), "norm_rows": int(rec["norm_rows"].shape[0]),
           "max_abs_cosine_diff": float(np.abs(_cosines(sample) - _cosines(rec["sample"])).max()),
           "max_rel_row_norm_diff": float((np.abs(np.asarray(norms, np.float64) - rec["norms"]) / rec["norms"]).max())}
    ok = out["max_abs_cosine_diff"] <= TOL["cosine"] and out["max_rel_row_norm_diff"] <= TOL["row_norm_rel"]
    if spectrum is not None:
        lam, ref = np.asarray(spectrum, np.float64), rec["spectrum"]
        out["max_spectrum_diff_rel_to_largest"] = float(np.abs(lam - ref).max() / ref[0])
        out["max_spectrum_rel_diff_per_eigenvalue"] = float((np.abs(lam - ref) / np.abs(ref)).max())
        out["spectrum_largest_smallest"] = [float(ref[0]), float(ref[-1])]
        ok = ok and out["max_spectrum_diff_rel_to_largest"] <= TOL["spectrum_rel_to_largest"]
    out["tolerances"] = TOL
    out["within_tolerance"] = bool(ok)
    return out


def covariance_all_rows(x, n, rows_per_pass=1_000_000):
    """f64 covariance of ALL n rows of a device f32 matrix, by chunked torch matmuls (rocBLAS dgemm — not a kernel of this
    repository): two passes, mean first, like pycleora/__init__.py:136-143."""
    import torch
    d = x.shape[1]
    mean = torch.zeros(d, dtype=torch.float64, device=x.device)
    for r0 in range(0, n, rows_per_pass):
        mean += x[r0:min(n, r0 + rows_per_pass)].double().sum(0)
    mean /= n
    cov = torch.zeros((d, d), dtype=torch.float64, device=x.device)
    for r0 in range(0, n, rows_per_pass):
        blk = x[r0:min(n, r0 + rows_per_pass)].double() - mean
        cov += blk.T @ blk
    cov /= (n - 1)
    return cov


def gpu_invariants(L, graph, x0_dev, n, d, rec, flags_extra=0, threshold=0.0):
    """Runs the product's default loop for the record's iteration count on the GPU and returns compare()'s result plus the
    all-rows covariance check.  x0_dev: torch f32 [>= n, d] holding E_0 (left untouched)."""
    import torch
    from cleora_amd import _hip
    iters = int(rec["meta"]["iterations"])
    dev = x0_dev.device
    spectrum = None
    x = x0_dev[:n].clone()
    if "spectrum" in rec:
        # the covariance the LAST iteration whitens: Y_k = l2(A E_{k-1}); E_{k-1} from the same loop with one iteration less (its
        # last iteration is a PCA whitening too, so it equals the oracle's E_{k-1} up to an orthogonal column transform, under
        # which SpMM, the row norm and the spectrum of the covariance are invariant)
        if iters > 1:
            _hip.check(L.cleora_embed_dev(graph.handle, x.data_ptr(), _hip.LEFT, d, iters - 1, 0.0, threshold, _hip.F_WHITEN | flags_extra, None))
        y = torch.empty_like(x)
        _hip.check(L.cleora_propagate_dev(graph.handle, _hip.LEFT, x.data_ptr(), d, d, y.data_ptr(), d, _hip.F_L2NORM, 0.0, None, None, None,
                                          torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        spectrum = torch.linalg.eigvalsh(covariance_all_rows(y, n)).flip(0).cpu().numpy()
        del y
        x.copy_(x0_dev[:n])
    _hip.check(L.cleora_embed_dev(graph.handle, x.data_ptr(), _hip.LEFT, d, iters, 0.0, threshold, _hip.F_WHITEN | flags_extra, None))
    torch.cuda.synchronize()
    rows = torch.from_numpy(rec["rows"].astype(np.int64)).to(dev)
    nrows = torch.from_numpy(rec["norm_rows"].astype(np.int64)).to(dev)
    sample = x[rows].cpu().numpy()
    norms = x[nrows].double().norm(dim=1).cpu().numpy()
    out = compare(rec, sample, norms, spectrum)
    cov = covariance_all_rows(x, n)
    out["max_abs_cov_minus_identity_all_rows"] = float((cov - torch.eye(d, dtype=torch.float64, device=dev)).abs().max())
    out["max_abs_cov_minus_identity_all_rows_oracle"] = float(rec["meta"]["max_abs_cov_minus_identity_all_rows"])
    out["finite"] = bool(torch.isfinite(x).all())
    return out


def oracle_loop(rowptr, col, val, x, iters, threads, log=print):
    """pycleora.embed()'s whiten=True loop with the oracle's pieces, buffers reused (30 GB of host memory at config 3 instead of 60).
    Row-wise steps run in 50 000-row chunks — the reference's own chunk (pycleora/__init__.py:134); per-row arithmetic does not
    depend on the chunking.  Returns (E_k, spectrum of the covariance whitened by the last iteration)."""
    import oracle
    from oracle import whiten as ow
    n, d = x.shape
    spectrum = None
    t0 = time.perf_counter()
    for it in range(1, iters + 1):
        y = oracle.spmm(rowptr, col, val, x, threads)                     # src/embedding.rs:52-86
        del x
        for r0 in range(0, n, ow.CHUNK):                                  # _normalize(emb, "l2"), pycleora/__init__.py:942-946
            y[r0:r0 + ow.CHUNK] = ow.normalize_l2(y[r0:r0 + ow.CHUNK])
        mean, cov = ow.whiten_stats(y)                                    # :136-143
        transform, lam = ow.whiten_transform(cov)                         # :145-156
        spectrum = lam
        mean_f32 = mean.astype(np.float32)
        for r0 in range(0, n, ow.CHUNK):                                  # :157-163 (in place: a chunk is read before it is written)
            y[r0:r0 + ow.CHUNK] = np.dot(y[r0:r0 + ow.CHUNK] - mean_f32, transform)
        x = y
        log(f"oracle iteration {it}: {time.perf_counter() - t0:.1f} s; spectrum {lam[0]:.3e} .. {lam[-1]:.3e}")
    return x, spectrum


def main():
    import torch
    import bench
    import oracle
    from oracle import whiten as ow
    from cleora_amd import _hip
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3", choices=sorted(bench.CONFIGS))
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--hyperedges", type=int, default=0)
    ap.add_argument("--products", type=int, default=0)
    ap.add_argument("--nodes", type=int, default=0)
    ap.add_argument("--pairs", type=int, default=0)
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g, hashes, label, cfg = bench.make_workload(args, dev, 0, 1, False)
    d = args.dim or cfg["dim"]
    n, nnz = g["n"], g["nnz"]
    meta = {"config": args.config, "label": label, "n": n, "nnz": nnz, "d": d, "iterations": args.iters, "hash": "xxh3_128", "graph": graph_hash(g),
            "oracle": "oracle.spmm (oracle/cleora_oracle.c) + oracle/whiten.py normalize_l2 / whiten_stats / whiten_transform / projection (numpy "
                      f"{np.__version__}, fp64 statistics, LAPACK eigh): pycleora/__init__.py:109-117,130-164",
            "generated_by": "tests/golden/make_whitened_loop_record.py on the GPU box (graph from the bench's generator)"}
    rowptr = g["rowptr"].cpu().numpy().astype(np.uint64)
    col = g["col"].cpu().numpy().view(np.uint32)
    val = g["val_left"].cpu().numpy()
    x0 = oracle.init(hashes.cpu().numpy().view(np.uint64), d, 0)
    meta["x0"] = hash_array(x0)
    threads = oracle.max_threads()
    t0 = time.perf_counter()
    x, spectrum = oracle_loop(rowptr, col, val, x0.copy(), args.iters, threads, log=lambda s: print(s, flush=True))
    meta["oracle_seconds"] = round(time.perf_counter() - t0, 1)
    meta["oracle_threads"] = threads
    rows, nrows = sample_rows(n, d), norm_rows(n)
    _, cov = ow.whiten_stats(x)
    meta["max_abs_cov_minus_identity_all_rows"] = float(np.abs(cov - np.eye(d)).max())
    rec = {"rows": rows.astype(np.int32), "sample": x[rows].astype(np.float32), "norm_rows": nrows.astype(np.int32),
           "norms": np.linalg.norm(x[nrows].astype(np.float64), axis=1), "spectrum": np.asarray(spectrum, np.float64)}
    out = args.out or os.path.join(ROOT, "gpurun_out", f"whitened_loop_{args.config}.npz")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez(out, meta=np.array(json.dumps(meta)), **rec)
    print("wrote", out, meta, flush=True)
    del x, cov
    # never commit a record blind: the product's loop against it, right away
    rec["meta"] = meta
    graph = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None, 0,
                                   keepalive=(g["rowptr"], g["col"], g["val_left"]))
    x0_dev = torch.from_numpy(x0).to(dev)
    res = {"default_loop": gpu_invariants(_hip.lib(), graph, x0_dev, n, d, rec),
           "reference_order_loop": gpu_invariants(_hip.lib(), graph, x0_dev, n, d, rec, threshold=1e-30)}
    print(json.dumps(res), flush=True)
    json.dump({"meta": meta, "gpu_vs_record": res}, open(out.replace(".npz", "_gpu_check.json"), "w"), indent=1)
    graph.close()


if __name__ == "__main__":
    main()
