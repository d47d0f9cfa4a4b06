"""Golden record of the ORACLE's plain loop (embed_fast: SpMM + L2 norm per iteration, src/embedding.rs:106-136) on the bench's
synthetic workloads, as hashes of the iterate after chosen iterations.

Why hashes: since the hub rows are summed in the reference's order (spmm.hip hub_inorder_kernel) the GPU loop is BIT-EQUAL to
the oracle's at every iteration, so a 128-bit hash of the iterate is a complete comparison — and the oracle's 40 iterations
(20 GB of random gathers each at config 2, 200 GB at config 3: minutes of all host cores) need not be repeated by every test
run and every bench line.  The graphs come from the bench's own generators (cleora_amd/synth.py on the GPU, the C++ host
builder for config 5), so this script runs on the GPU box:

    gpurun -- 'python tests/golden/make_plain_loop_hashes.py --config C2 --iters 40'      (also C3; C5 with fewer iterations)

and writes gpurun_out/plain_loop_hashes_<config>.json, which is then committed as tests/golden/plain_loop_hashes_<config>.json.
The record carries hashes of the graph and of E_0 as well: a consumer that builds a different graph (generator changed, other
torch build) sees the mismatch and must fall back to running the oracle itself.

Nothing here touches the HIP library: the loop is oracle.spmm_aos_l2_inplace (oracle/cleora_oracle.c, the reference's AoS edge
layout, one accumulator per row in stored order, separate L2 pass), all host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

HASH = "xxh3_128"


def hash_array(a, rows_per_chunk=1 << 20):
    """xxh3-128 of the C-contiguous bytes of `a` (numpy array or torch tensor on any device), fed in row chunks."""
    import xxhash
    h = xxhash.xxh3_128()
    n = a.shape[0]
    for r0 in range(0, n, rows_per_chunk):
        blk = a[r0:r0 + rows_per_chunk]
        if not isinstance(blk, np.ndarray):
            blk = blk.contiguous().cpu().numpy()
        h.update(memoryview(np.ascontiguousarray(blk)).cast("B"))
    return h.hexdigest()


def graph_hash(g):
    """One hash over the CSR the loops read: rowptr (u64), col (u32), left values (f32)."""
    return "-".join(hash_array(g[k].view(-1, 1)) for k in ("rowptr", "col", "val_left"))


def main():
    import torch
    import bench
    import oracle
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2", choices=sorted(bench.CONFIGS))
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--at", default="1,2,3,5,10,20,40", help="iterations whose iterate is hashed")
    ap.add_argument("--hyperedges", type=int, default=0)
    ap.add_argument("--products", type=int, default=0)
    ap.add_argument("--nodes", type=int, default=0)
    ap.add_argument("--pairs", type=int, default=0)
    ap.add_argument("--dim", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g, hashes, label, cfg = bench.make_workload(args, dev, 0, 1, False)
    d = args.dim or cfg["dim"]
    n, nnz = g["n"], g["nnz"]
    rec = {"config": args.config, "label": label, "n": n, "nnz": nnz, "d": d, "hash": HASH, "graph": graph_hash(g),
           "oracle": "oracle.spmm_aos_l2_inplace (oracle/cleora_oracle.c): AoS edges, one accumulator per row in stored order, separate L2 pass",
           "generated_by": "tests/golden/make_plain_loop_hashes.py on the GPU box (graph from the bench's generator)"}
    rowptr = g["rowptr"].cpu().numpy().astype(np.uint64)
    edges = np.empty(nnz, dtype=oracle.EDGE_DTYPE)
    edges["col"] = g["col"].cpu().numpy().view(np.uint32)
    edges["left"] = g["val_left"].cpu().numpy()
    edges["sym"] = g["val_sym"].cpu().numpy()
    x = oracle.init(hashes.cpu().numpy().view(np.uint64), d, 0)
    del g
    torch.cuda.empty_cache()
    rec["x0"] = hash_array(x)
    y = np.empty_like(x)
    threads = oracle.max_threads()
    at = sorted({int(t) for t in args.at.split(",") if int(t) <= args.iters} | {args.iters})
    rec["iterations"] = {}
    t0 = time.perf_counter()
    for it in range(1, args.iters + 1):
        oracle.spmm_aos_l2_inplace(rowptr, edges, False, x, y, threads)
        x, y = y, x
        if it in at:
            rec["iterations"][str(it)] = hash_array(x)
            print(f"iteration {it}: {rec['iterations'][str(it)]}  ({time.perf_counter() - t0:.1f} s)", flush=True)
    rec["oracle_seconds"] = round(time.perf_counter() - t0, 1)
    rec["oracle_threads"] = threads
    out = os.path.join(ROOT, "gpurun_out", f"plain_loop_hashes_{args.config}.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rec, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
