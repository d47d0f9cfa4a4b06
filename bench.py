#!/usr/bin/env python3
"""bench.py — the Markov-propagation hot path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 40 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one propagate iteration of BASELINE.json's metric workload: CSR x dense SpMM with
the L2 normalisation fused into its epilogue (src/embedding.rs:106-136, one loop body of
embed_full) on the synthetic power-law graph |V| = 10M, |E| = nnz ~ 200M, d = 256.  For N > 1
(cleora_amd/sharded.py) the default is the COLUMN partition — each rank owns d/N columns of X
and the whole CSR, and the only collective per iteration is an all-reduce of the n row
sums-of-squares; `--partition row` selects north_star's literal layout (row blocks + in-place
all-gather of the 10 GB iterate), which is xGMI-bound.  The graph, X and the CSR
are resident in HBM before the timed region.  Total work is fixed as N grows ("strong" scaling,
as BASELINE.json quotes the same graph at 1/2/4/8 GPUs).

Prints ONE JSON line on rank 0.  `value` = nnz * d * steps / seconds (edge*dim/s, whole job);
`roofline` is for the dominant kernel (spmm_rows_kernel) from HIP events recorded inside the
timed region on the launch stream; `cpu_baseline` is the oracle's reference-order CPU port timed
on this box's host cores (N = 1, rank 0 only) — a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from cleora_amd import _hip, sharded, synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming ceiling)


def algorithmic_bytes(nnz, n_rows_written, n_rowptr, d):
    """SURVEY.md §8(d) gather model: col + one value stream, rowptr, one gathered X row per
    edge, Y written once (the L2 norm is fused)."""
    return nnz * 8 + (n_rowptr + 1) * 8 + nnz * d * 4 + n_rows_written * d * 4


def cpu_baseline(g, x_dev, n, d, budget_s=12.0):
    """Reference-order CPU port (oracle AoS SpMM + separate L2 pass, all host cores) on whole
    iterations of the same graph and X; bounded to about `budget_s` seconds."""
    import oracle
    rowptr = g["rowptr"].cpu().numpy().astype(np.uint64)
    edges = np.empty(g["nnz"], dtype=oracle.EDGE_DTYPE)
    edges["col"] = g["col"].cpu().numpy().view(np.uint32)
    edges["left"] = g["val_left"].cpu().numpy()
    edges["sym"] = g["val_sym"].cpu().numpy()
    x = x_dev[:n].cpu().numpy()
    y = np.empty_like(x)
    threads = oracle.max_threads()
    oracle.spmm_aos_l2_inplace(rowptr, edges, False, x, y, threads)  # page-touch + pool spin-up
    count, t0 = 0, time.perf_counter()
    while True:
        oracle.spmm_aos_l2_inplace(rowptr, edges, False, x, y, threads)
        x, y = y, x
        count += 1
        el = time.perf_counter() - t0
        if el >= budget_s or count >= 5:
            break
    out = {"value": g["nnz"] * d * count / el, "unit": "edge*dim/s", "cores": threads,
           "kind": "port", "iterations_per_sec": count / el,
           "sample": f"{count} full iteration(s) of the same graph and X (SpMM, reference AoS edge "
                     f"layout, dynamic row schedule + separate L2 pass), {el:.1f} s"}
    # independent line: single-thread scipy CSR @ dense on a contiguous block of rows
    try:
        import scipy.sparse as sp
        r0, rows = n // 3, min(n - n // 3, 400_000)
        e0, e1 = int(rowptr[r0]), int(rowptr[r0 + rows])
        a = sp.csr_matrix((edges["left"][e0:e1], edges["col"][e0:e1].astype(np.int64),
                           (rowptr[r0:r0 + rows + 1] - rowptr[r0]).astype(np.int64)), shape=(rows, n))
        t0 = time.perf_counter()
        a @ x
        el = time.perf_counter() - t0
        out["scipy_single_thread"] = {"value": (e1 - e0) * d / el, "unit": "edge*dim/s", "cores": 1,
                                      "sample": f"scipy.sparse CSR @ dense, rows [{r0}, {r0 + rows}) = {e1 - e0} edges, {el:.1f} s"}
    except Exception as ex:  # scipy is optional; the port above is the baseline
        out["scipy_single_thread"] = {"error": str(ex)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--pairs", type=int, default=95_000_000)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--overlap-steps", type=int, default=0,
                    help="row blocks per rank per iteration (gather of block k overlaps SpMM of k+1); "
                         "0 = 1 on one GPU, 4 otherwise")
    ap.add_argument("--partition", default="auto", choices=["auto", "row", "column"],
                    help="multi-GPU partition: 'column' = each rank owns d/N columns of X (one all-reduce of "
                         "n floats per iteration); 'row' = row blocks + in-place all-gather of X (north_star's "
                         "literal layout, 10 GB per iteration over xGMI); auto = column for N > 1")
    ap.add_argument("--placement-candidates", type=int, default=8,
                    help="before the timed region, try up to this many allocations as the partner buffer of X "
                         "and keep the fastest ping-pong pair (1 = no tuning).  The same kernel runs 34.4-39.9 ms "
                         "depending on WHICH two allocations hold X and Y (DESIGN.md §3.1, placement sensitivity).")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # developer switches for exercising the N > 1 code path on a ONE-GPU box (every rank on cuda:0, gloo
    # instead of RCCL, which refuses two ranks on one device); numbers from such a run mean nothing
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help=argparse.SUPPRESS)
    ap.add_argument("--share-gpu", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # a run takes 1-3 minutes; if a collective ever deadlocks, leave a traceback of every thread and exit
    # instead of hanging the launcher
    import faulthandler
    faulthandler.dump_traceback_later(1500, exit=True)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; cleora_amd has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # backend "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")
    partition = args.partition
    if partition == "auto":
        partition = "column" if world > 1 else "row"
    if partition == "column" and args.dim % (4 * world) != 0:
        partition = "row"

    d = args.dim
    steps_per_iter = args.overlap_steps or (1 if world == 1 else 4)
    g = synth.power_law_graph(args.nodes, args.pairs, 2, dev)  # same seed on every rank
    n, nnz = g["n"], g["nnz"]
    deg = torch.diff(g["rowptr"])
    hashes = synth.entity_hashes(n, 0, dev)
    L = _hip.lib()
    backend = sharded.HipBackend(dev)
    launch_bytes = []
    if partition == "row":
        sg = sharded.ShardedGraph(n, g["rowptr"], g["col"], g["val_left"], None, rank, world,
                                  steps_per_iter, backend)
        blocks = sg.blocks
        # algorithmic bytes of this rank's dominant-kernel launches (rows the rows-kernel owns)
        for k in range(steps_per_iter):
            r0 = min((k * world + rank) * sg.block, n)
            r1 = min(r0 + sg.block, n)
            dk = deg[r0:r1]
            # the launch gathers for ALL edges of the block (hub segments are its first work items)
            # and writes every row except the hub rows (hub_finish_kernel writes those)
            n_hub = int((dk > sg.blocks[k].info().hub_threshold).sum())
            launch_bytes.append(algorithmic_bytes(int(dk.sum()), sg.block - n_hub, sg.block, d))
        x = torch.zeros((sg.n_pad, d), dtype=torch.float32, device=dev)
        x_next = torch.zeros_like(x)
        _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, x.data_ptr(), d,
                                     torch.cuda.current_stream().cuda_stream))
        kernel_name = "spmm_rows_kernel<64,1,4,true>"
        par = (f"row-block-cyclic x{world}, {steps_per_iter} block(s)/rank/iter"
               + (", in-place RCCL all-gather overlapped with the next block" if world > 1 else ""))

        def iterate():
            nonlocal x, x_next
            sg.propagate(_hip.LEFT, x, x_next)
            x, x_next = x_next, x
    else:
        cg = sharded.ColumnShardedGraph(n, g["rowptr"], g["col"], g["val_left"], None, d, rank, world, backend,
                                        steps=steps_per_iter)
        blocks = cg.blocks
        dl = cg.dl
        for blk, (r0, r1) in zip(cg.blocks, cg.row_blocks):
            dk = deg[r0:r1]
            n_hub = int((dk > blk.info().hub_threshold).sum())
            launch_bytes.append(algorithmic_bytes(int(dk.sum()), (r1 - r0) - n_hub, r1 - r0, dl))
        x = torch.empty((n, dl), dtype=torch.float32, device=dev)
        x_next = torch.empty_like(x)
        rowsq = torch.zeros(n, dtype=torch.float32, device=dev)
        # columns [c0, c0 + dl) of the deterministic init: init_value depends on hash + col + seed only
        _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, dl, cg.c0, x.data_ptr(), dl,
                                     torch.cuda.current_stream().cuda_stream))
        g_lanes = max(8, min(64, 1 << (max(dl // 4, 1) - 1).bit_length()))
        kernel_name = f"spmm_rows_kernel<{g_lanes},{max(1, dl // 256)},4,true>"
        par = (f"column partition x{world}: every rank owns {dl} of {d} columns and the whole CSR; "
               f"RCCL all-reduce of the n f32 row sums-of-squares per iteration in {steps_per_iter} row block(s) "
               f"(block k's reduce overlaps block k+1's SpMM), no exchange of X")

        def iterate():
            nonlocal x, x_next
            cg.propagate(_hip.LEFT, x, x_next, rowsq)
            x, x_next = x_next, x
    keep_full = g if (world == 1 and rank == 0 and not args.no_cpu_baseline) else None
    del deg, hashes
    if partition == "row":
        del g
    torch.cuda.empty_cache()

    # ---- placement tuning (outside the timed region) ---------------------------------------------
    # The same launch runs 34.4-39.9 ms depending on WHICH two allocations hold X and Y: a pair is
    # slow when both buffers fall into the same (unknown, physical) placement class — unaffected by
    # offsets inside the allocations (scripts/alloc_probe*.py).  So X stays where it is
    # and candidate partners are allocated one by one (with spacer allocations in between to move the
    # allocator along) until one is clearly faster than the slowest seen, or the budget is used.
    placement = None
    ncand = max(1, args.placement_candidates)
    if ncand > 1:
        base, src = x, x.clone()

        def step_time(a, b):
            nonlocal x, x_next
            a.copy_(src)
            x, x_next = a, b
            iterate()                       # warm (iterate swaps x / x_next)
            x, x_next = a, b
            a.copy_(src)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            iterate()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1)

        tried, spacers = [], []
        # with several ranks every iterate() contains a collective, so all ranks must run the SAME
        # number of trial steps: a fixed count and no data-dependent early exit
        if world > 1:
            ncand = min(ncand, 4)
        for k in range(ncand):
            cand = x_next if k == 0 else torch.zeros_like(base)
            tried.append((step_time(base, cand) + step_time(cand, base), cand))
            lo, hi = min(t for t, _ in tried), max(t for t, _ in tried)
            if world == 1 and k >= 1 and lo < 0.95 * hi:
                break
            spacers.append(torch.empty(int((0.6 + 0.83 * (k + 1)) * 2 ** 30), dtype=torch.uint8, device=dev))
        best_t, best = min(tried, key=lambda p: p[0])
        x, x_next = base, best
        x.copy_(src)
        placement = {"partners_tried": len(tried), "pair_ms_tried": [round(t / 2, 3) for t, _ in tried],
                     "chosen_pair_ms": round(best_t / 2, 3)}
        del tried, spacers, src, best, base
        torch.cuda.empty_cache()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        iterate()
    for blk in blocks:
        blk.set_timing(True)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        iterate()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    # dominant kernel: average launch duration from the HIP events recorded in the timed region
    rows_ms, calls, other_ms = 0.0, 0, 0.0
    for blk in blocks:
        ms, c = blk.get_timing()
        blk.set_timing(False)
        rows_ms += ms[1]
        other_ms += ms[0] + ms[2]
        calls += c
    avg_ms = rows_ms / max(calls, 1)
    avg_bytes = sum(launch_bytes) / len(launch_bytes)
    achieved = avg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    finite = bool(torch.isfinite(x[:n]).all())
    sumsq = x[:n].double().pow(2).sum(1)
    if partition == "column" and world > 1:
        dist.all_reduce(sumsq)
    norm_err = float((sumsq.sqrt() - 1).abs().max())

    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                rec = json.load(open(tpath))
                if rec.get("n") == n and rec.get("nnz") == nnz and rec.get("d") == d and world == 1:
                    traffic = rec.get("bytes_per_launch")
            except Exception:
                traffic = None
        hot_rows = int(blocks[0].info().hot_rows)      # gather cache policy (cleora_graph_set_hot_cache), 0 = inactive
        kernel_name = kernel_name[:-1] + (",true>" if hot_rows else ",false>")
        out = {
            "metric": "propagate edges*dim/sec (SpMM + fused L2 norm"
                      + ("" if world == 1 else (" + all-gather of X" if partition == "row" else " + all-reduce of row norms"))
                      + "), |V|=10M |E|=200M d=256",
            "value": nnz * d * args.steps / elapsed,
            "unit": "edge*dim/s",
            "iterations_per_sec": args.steps / elapsed,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic power-law graph, reflexive column semantics "
                                   f"(BASELINE config 3): n={n}, nnz={nnz}, d={d}, left Markov",
                       "n": n, "nnz": nnz, "d": d,
                       "parallelism": par, "partition": partition, "seed": 2},
            "roofline": {"bound": "hbm", "kernel": kernel_name,
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": avg_bytes, "avg_launch_ms": avg_ms,
                         "launches": calls, "gather_cache_policy_hot_rows": hot_rows,
                         "hub_kernels_ms_per_launch": other_ms / max(calls, 1),
                         "frac_of_measured_copy_ceiling_6290": achieved / 6290.0,
                         # SURVEY.md §8(d) secondary model: every operand once (lower bound on traffic)
                         "compulsory_bytes_per_iteration": nnz * 8 + (n + 1) * 8 + 2 * n * d * 4},
            "checks": {"finite": finite, "max_abs_row_norm_minus_1": norm_err},
            "placement_tuning": placement,
        }
        if keep_full is not None:
            out["cpu_baseline"] = cpu_baseline(keep_full, x, n, d)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()
