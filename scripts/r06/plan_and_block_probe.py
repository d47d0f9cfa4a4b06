"""Round 6 (VERDICT round 5, next #7 and #4c): what ONE rank of the 8-way row partition holds and runs at BASELINE configs 3, 4 (size
stand-in) and 5 — as executable checks, on the real degree sequence of the bench's graphs.

  plan     cleora_sharded_plan (P = 8, 4 steps, rows / nnz balance) + cleora_embed_sharded_bytes: device bytes per rank (asserted
           <= 288 GB), heaviest / lightest block, heaviest / lightest rank, the longest row in any block (the in-order chain).
  blocks   (config 3 and 5; --blocks) the SpMM + fused L2 of ONE rank's blocks timed ALONE on the GPU — which is what that rank's GPU
           runs between two all-gathers, measurable on a one-GPU box: the rank that owns the longest row, its four blocks as four
           graph handles, in-order hub launch (4 / 2 lanes) against the segmented hub sum.

    python scripts/r06/plan_and_block_probe.py C3|C4s|C5 [--blocks]
"""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cleora_amd import _hip, sharded  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
do_blocks = "--blocks" in sys.argv
dev = torch.device("cuda:0")
L = _hip.lib()
S = torch.cuda.current_stream().cuda_stream
P, K = 8, 4
HBM = 288e9


class A:
    config, nodes, pairs, hyperedges, products, dim = cfg, 0, 0, 0, 0, 0


g, hashes, label, c = bench.make_workload(A, dev, 0, 1, False)
n, nnz, d = g["n"], g["nnz"], c["dim"]
rowptr = g["rowptr"].cpu().numpy().astype(np.uint64)
deg = np.diff(rowptr.astype(np.int64))
out = {"config": cfg, "n": n, "nnz": nnz, "d": d, "world": P, "steps": K, "longest_row": int(deg.max()), "plans": {}}
for balance in ("rows", "nnz"):
    bounds, n_pad, mode = sharded.plan_rows(n, rowptr, P, K, balance)
    b = np.minimum(np.asarray(bounds, dtype=np.int64), n)
    block_nnz = (rowptr[b[1:]] - rowptr[b[:-1]]).astype(np.int64)
    block_rows = np.diff(np.asarray(bounds, dtype=np.int64))
    rank_nnz = np.array([block_nnz[r::P].sum() for r in range(P)])
    rank_rows = np.array([block_rows[r::P].sum() for r in range(P)])
    longest_in_block = [int(deg[b[j]:b[j + 1]].max()) if b[j + 1] > b[j] else 0 for j in range(P * K)]
    per_rank = []
    for r in range(P):
        replica = n_pad * d * 4
        extra = int(L.cleora_embed_sharded_bytes(n_pad, int(rank_rows[r]), n, P, d, _hip.F_WHITEN))
        csr = (int(rank_rows[r]) + K) * 8 + int(rank_nnz[r]) * (4 + 4 + 4 + 4)      # rowptr, col, two value streams, the gather policy's private col copy
        per_rank.append({"rank": r, "rows": int(rank_rows[r]), "nnz": int(rank_nnz[r]), "replica_GB": replica / 1e9,
                         "whitened_loop_extra_GB": extra / 1e9, "csr_GB": csr / 1e9, "total_GB": (replica + extra + csr) / 1e9})
    worst = max(p["total_GB"] for p in per_rank)
    assert worst * 1e9 <= HBM, f"{cfg} {balance}: a rank would hold {worst:.1f} GB"
    out["plans"][balance] = {
        "mode": mode, "n_pad": int(n_pad), "block_nnz_heaviest_lightest_mean": [int(block_nnz.max()), int(block_nnz.min()), float(block_nnz.mean())],
        "rank_nnz_heaviest_lightest_mean": [int(rank_nnz.max()), int(rank_nnz.min()), float(rank_nnz.mean())],
        "rank_imbalance": float(rank_nnz.max() / rank_nnz.mean()), "longest_row_in_any_block": max(longest_in_block),
        "block_with_the_longest_row": int(np.argmax(longest_in_block)), "per_rank": per_rank, "max_total_GB_per_rank": worst, "fits_288_GB": bool(worst * 1e9 <= HBM),
        # the in-order chain against the rank's own SpMM time (the estimate hub_lanes() uses: 52 ns per edge at 4 lanes; 6.4 TB/s)
        "estimated_ms": {"rank_spmm_heaviest": float(rank_nnz.max() * d * 4 / 6.4e12 * 1e3), "longest_chain_4_lanes": max(longest_in_block) * 52e-6,
                         "longest_chain_2_lanes": max(longest_in_block) * 52e-6 * 12.3 / 16.7}}
print(json.dumps({k: v for k, v in out.items() if k != "plans"} | {"plans": {b: {k: v for k, v in p.items() if k != "per_rank"} for b, p in out["plans"].items()}}), flush=True)

if do_blocks:
    # the blocks of the rank owning the longest row, each as its own graph handle over the full iterate (n_cols = n)
    bounds, n_pad, mode = sharded.plan_rows(n, rowptr, P, K, "auto")
    b = np.minimum(np.asarray(bounds, dtype=np.int64), n)
    j_long = int(np.argmax([int(deg[b[j]:b[j + 1]].max()) if b[j + 1] > b[j] else 0 for j in range(P * K)]))
    rank = j_long % P
    x = torch.empty((n, d), dtype=torch.float32, device=dev)
    y = torch.empty((n, d), dtype=torch.float32, device=dev)
    _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, x.data_ptr(), d, S))
    graphs = []
    for k in range(K):
        j = k * P + rank
        r0, r1 = int(b[j]), int(b[j + 1])
        e0, e1 = int(rowptr[r0]), int(rowptr[r1])
        rp = (g["rowptr"][r0:r1 + 1] - e0).contiguous()
        col = g["col"][e0:e1].contiguous()
        val = g["val_left"][e0:e1].contiguous()
        gr = _hip.Graph.from_device(r1 - r0, n, e1 - e0, rp.data_ptr(), col.data_ptr(), val.data_ptr(), None, 0, keepalive=(rp, col, val))
        graphs.append((gr, r0, r1, e1 - e0))

    def run(flags, iters=10, warm=4):
        def once():
            for gr, r0, r1, _ in graphs:
                _hip.check(L.cleora_propagate_dev(gr.handle, 0, x.data_ptr(), d, d, y[r0:r1].data_ptr(), d, _hip.F_L2NORM | flags, 0.0, x[r0:r1].data_ptr(), None, None, S))
        for _ in range(warm):
            once()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            once()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    res = {"rank": rank, "mode": mode, "blocks": [{"rows": r1 - r0, "nnz": e, "longest_row": int(deg[r0:r1].max()), "n_inorder_rows": int(gr.info().n_inorder_rows)} for gr, r0, r1, e in graphs],
           "ms_per_iteration_of_the_ranks_four_blocks": {}}
    for rep in range(2):
        NEVER = 2 ** 64 - 1
        for name, lanes, flags, chain in (("default", 0, 0, 0), ("chain_all_hub_rows", 0, 0, 1), ("chain_8k", 0, 0, 8192), ("no_chain_lanes4", 4, 0, NEVER),
                                          ("no_chain_lanes2", 2, 0, NEVER), ("segments", 0, _hip.F_HUB_SEGMENTS, NEVER)):
            for gr, *_ in graphs:
                gr.set_hub_lanes(lanes)
                gr.set_hub_chain_min(chain)
            res["ms_per_iteration_of_the_ranks_four_blocks"].setdefault(name, []).append(round(run(flags), 3))
    res["whole_graph_spmm_over_8_ms"] = round(nnz * d * 4 / 6.4e12 * 1e3 / P, 3)
    out["rank_blocks_alone_on_the_gpu"] = res
    print(json.dumps(res), flush=True)
    for gr, *_ in graphs:
        gr.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"r06_plan_{cfg}.json"), "w"), indent=1)
