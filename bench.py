#!/usr/bin/env python3
"""bench.py — the Markov-propagation hot path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 40 --warmup 3
    python bench.py --gpus N --steps K --warmup W          (WORLD_SIZE unset: bench.py launches its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one propagate iteration of BASELINE.json's metric workload: CSR x dense SpMM with
the L2 normalisation fused into its epilogue (src/embedding.rs:106-136, one loop body of
embed_full) on the synthetic power-law graph |V| = 10M, |E| = nnz ~ 200M, d = 256 (BASELINE
config 3).  The graph, X and the CSR are resident in HBM before the timed region.  Total work is
fixed as N grows ("strong" scaling: BASELINE.json quotes the same graph at 1/2/4/8 GPUs).

N > 1 (csrc/sharded.hip through cleora_amd/sharded.py; kernels AND collectives go through the C ABI — RCCL is bound directly in
csrc/comm.hip; torch.distributed/gloo is only the launcher: it distributes the RCCL id and reduces the
timings).  BOTH partitions are measured, W + K iterations each, and reported in `partitions`:
  row     north_star's layout: row blocks of the CSR, full replicas of X, in-place all-gather of the
          10 GB iterate per iteration over xGMI (block k's gather overlaps block k+1's SpMM);
  column  each rank owns d/N columns of X and the whole CSR; the only collective per iteration is an
          all-reduce of the n row sums-of-squares.
`value` is the ROW partition's (north_star's layout) for every N; the column partition is reported beside it.
Before the big graph is built, every partition runs one iteration on a 100k-row graph and is compared with the same
iteration computed by the rank alone (`selftest`; a broken collective costs seconds, not the run).

N = 1 additionally reports `whitened` — the default pycleora.embed() loop (SpMM + L2, then
whiten_embeddings: pycleora/__init__.py:109-117,130-164) with per-kernel milliseconds from HIP events and
MFMA roofline fractions for the Gram and projection kernels —, `checks` (one GPU iteration against one
oracle iteration on the same graph and X: every row compared) and `cpu_baseline` (the oracle's
reference-order CPU port timed on this box's host cores) — a reported baseline, not the target.

Prints ONE JSON line on rank 0.  `value` = nnz * d * steps / seconds (edge*dim/s, whole job);
`roofline` is for the dominant kernel (spmm_rows_kernel) from HIP events recorded inside the
timed region on the launch stream.
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def self_launch(n_ranks, argv):
    """`python bench.py --gpus N` without torchrun (the way the driver starts the N = 1 run): this process becomes the
    launcher — it starts N copies of itself, one rank per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    set the way torch.distributed.run would set them, passes rank 0's stdout (the ONE JSON line) through, sends the other
    ranks' stdout to stderr, and exits with the first non-zero return code (the remaining ranks are terminated by PID).
    The launcher never touches a GPU."""
    import signal
    import subprocess
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n_ranks)))
    procs = []
    for r in range(n_ranks):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e,
                                      stdout=None if r == 0 else sys.stderr))

    def stop_all(*_):
        for p in procs:
            if p.poll() is None:
                p.terminate()
    signal.signal(signal.SIGTERM, lambda *_: (stop_all(), sys.exit(143)))
    rc, alive = 0, set(range(n_ranks))
    try:
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr, flush=True)
                    stop_all()
            time.sleep(0.05)
    except KeyboardInterrupt:
        stop_all()
        rc = 130
    for p in procs:
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


def _gpus_requested(argv):
    for i, arg in enumerate(argv):
        if arg == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1]) if argv[i + 1].isdigit() else 1
        if arg.startswith("--gpus="):
            return int(arg.split("=", 1)[1]) if arg.split("=", 1)[1].isdigit() else 1
    return 1


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and _gpus_requested(sys.argv[1:]) > 1 and not {"-h", "--help"} & set(sys.argv[1:]):
    # `python bench.py --gpus N` without torchrun: this process only launches the ranks — before numpy / torch are imported (on a box
    # with slow storage that import is tens of seconds, and every rank pays it once more anyway)
    raise SystemExit(self_launch(_gpus_requested(sys.argv[1:]), sys.argv[1:]))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from cleora_amd import _hip, comm as comm_mod, sharded, synth  # noqa: E402

METRIC = "propagate iterations/sec & edges·dim/sec, |V|=10M |E|=200M d=256, 1/2/4/8 GPU"  # BASELINE.json, verbatim
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming ceiling)
F32_MFMA_PEAK_TF = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, = the f32 vector rate
BF16_MFMA_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 (v_mfma_f32_32x32x16_bf16 measured 2495 TF)
F64_MFMA_PEAK_TF = 78.6     # AMD's MI355X datasheet figure for FP64 matrix (the guide lists none); the measured
                            # ceiling of v_mfma_f64_16x16x4_f64 on this chip is in docs/history.md §3.5 (scripts/mfma_peak.hip)
KERNEL_SOURCES = ("cleora_amd/csrc/spmm.hip", "cleora_amd/csrc/row_epilogue.h", "cleora_amd/csrc/hot.hip", "cleora_amd/csrc/common.h")


def algorithmic_bytes(nnz, n_rows_written, n_rowptr, d):
    """SURVEY.md §8(d) gather model: col + one value stream, rowptr, one gathered X row per
    edge, Y written once (the L2 norm is fused)."""
    return nnz * 8 + (n_rowptr + 1) * 8 + nnz * d * 4 + n_rows_written * d * 4


def kernel_source_stamp():
    """Identifies the build the committed PMC traffic figure was measured on."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def rows_bit_equal(dev, host, chunk=1 << 20):
    """Number of rows of the device matrix whose bits equal the host matrix's."""
    same = 0
    n = host.shape[0]                                     # (the device matrix may carry padding rows below)
    for r0 in range(0, n, chunk):
        got = dev[r0:min(n, r0 + chunk)].cpu().numpy()
        same += int((got.view(np.uint32) == host[r0:r0 + chunk].view(np.uint32)).all(axis=1).sum())
    return same


def host_cpu_model():
    """The host CPU the `cpu_baseline` ran on (the core count alone does not name the box class)."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_and_checks(g, x_dev, y_dev, n, d, hub_threshold, gpu_iterate=None, budget_s=12.0):
    """Reference-order CPU port (oracle AoS SpMM + separate L2 pass, all host cores) on whole iterations of the
    same graph and X, bounded to about `budget_s` seconds; its FIRST iteration doubles as the parity check:
    y_dev is the GPU's iteration from the same x_dev — EVERY row must be bit-equal, hub rows included (they are added in
    the reference's order by hub_inorder_kernel).  gpu_iterate(k): the GPU continues k iterations from y_dev and returns
    the iterate — compared with the oracle's after the same number of iterations."""
    import oracle
    rowptr = g["rowptr"].cpu().numpy().astype(np.uint64)
    edges = np.empty(g["nnz"], dtype=oracle.EDGE_DTYPE)
    edges["col"] = g["col"].cpu().numpy().view(np.uint32)
    edges["left"] = g["val_left"].cpu().numpy()
    edges["sym"] = g["val_sym"].cpu().numpy()
    x = x_dev[:n].cpu().numpy()
    y = np.empty_like(x)
    threads = oracle.max_threads()
    oracle.spmm_aos_l2_inplace(rowptr, edges, False, x, y, threads)  # page-touch + pool spin-up; also the check
    deg = np.diff(rowptr.astype(np.int64))
    hub = deg > hub_threshold
    equal_rows, hub_equal, chunk = 0, 0, 1 << 20
    for r0 in range(0, n, chunk):
        r1 = min(n, r0 + chunk)
        got = y_dev[r0:r1].cpu().numpy()
        same = (got.view(np.uint32) == y[r0:r1].view(np.uint32)).all(axis=1)
        equal_rows += int(same.sum())
        hub_equal += int(same[hub[r0:r1]].sum())
    checks = {"oracle_rows_compared": int(n), "oracle_rows_bit_equal": equal_rows, "hub_rows": int(hub.sum()),
              "hub_rows_bit_equal": hub_equal, "longest_row_edges": int(deg.max())}
    x, y = y, x                                          # x = the oracle's iterate after 1 iteration
    count, t0 = 0, time.perf_counter()
    while True:
        oracle.spmm_aos_l2_inplace(rowptr, edges, False, x, y, threads)
        x, y = y, x
        count += 1
        el = time.perf_counter() - t0
        if el >= budget_s or count >= 5:
            break
    if gpu_iterate is not None:
        # the baseline's iterations double as a multi-iteration check: the GPU continues from its own first iterate
        checks["oracle_iterations_compared"] = 1 + count
        checks["rows_bit_equal_after_those_iterations"] = rows_bit_equal(gpu_iterate(count), x)
    out = {"value": g["nnz"] * d * count / el, "unit": "edge*dim/s", "cores": threads, "host_cpu": host_cpu_model(),
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
    return out, checks


def golden_loop_check(config, g, hashes, a, b, iterate, n, d, L):
    """The plain loop from E_0 against the ORACLE's loop on the same workload, pinned as hashes of the iterate
    (tests/golden/plain_loop_hashes_<config>.json, written by tests/golden/make_plain_loop_hashes.py: the oracle run once on the
    GPU box, 40 iterations at C2 / C3).  Equal hashes = bit-equal iterates.  Measured live in this run; a workload the record
    does not describe (other generator output, other d) is reported as such, never skipped silently."""
    path = os.path.join(ROOT, "tests", "golden", f"plain_loop_hashes_{config}.json")
    if not os.path.exists(path):
        return {"error": f"no golden record tests/golden/plain_loop_hashes_{config}.json for this config"}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        from make_plain_loop_hashes import graph_hash, hash_array
        rec = json.load(open(path))
        if (rec["n"], rec["nnz"], rec["d"]) != (n, g["nnz"], d) or rec["graph"] != graph_hash(g):
            return {"error": "the golden record describes another graph (generator output differs): re-run tests/golden/make_plain_loop_hashes.py"}
        stream = torch.cuda.current_stream().cuda_stream
        _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, a.data_ptr(), d, stream))
        torch.cuda.synchronize()
        if hash_array(a[:n]) != rec["x0"]:
            return {"error": "E_0 differs from the golden record's"}
        want = {int(k): v for k, v in rec["iterations"].items()}
        at = [k for k in (1, 10, max(want)) if k in want]
        out, p, q, done = {}, a, b, 0
        for k in sorted(set(at)):
            for _ in range(k - done):
                iterate(p, q)
                p, q = q, p
            done = k
            torch.cuda.synchronize()
            out[str(k)] = hash_array(p[:n]) == want[k]
        return {"source": f"tests/golden/plain_loop_hashes_{config}.json (oracle loop, {rec['oracle_threads']} host threads, {rec['oracle_seconds']} s; "
                          f"{rec['hash']} of the iterate); the GPU side measured in this run",
                "iterate_bit_equal_to_oracle_after_iterations": out, "all_equal": all(out.values())}
    except Exception as ex:                                   # noqa: BLE001
        return {"error": f"{type(ex).__name__}: {ex}"[:300]}


class _CsrHost:
    """What SparseMatrix needs of a host graph when the CSR already exists (the bench's synthetic graphs never were text lines)."""

    def __init__(self, arr):
        self._arr = arr

    def arrays(self):
        return self._arr

    def sizes(self):
        return int(self._arr["rowptr"].shape[0] - 1), int(self._arr["col"].shape[0])


def end_to_end(g, hashes, n, d, iterations, L):
    """SURVEY.md 8(d) "report end-to-end separately": the wall clock of the DROP-IN calls on this workload, host arrays in, host array
    out — `SparseMatrix.embed_fast(d, 40)` (src/lib.rs:320-364) and `pycleora.embed(graph, d, 40)` with its default whiten=True
    (pycleora/__init__.py:51-127; cleora_amd.embed.embed is what accelerate() binds there) — and what they are made of: the one-time CSR
    upload (+ hub schedule), the iteration loop (cleora_last_embed_loop_ms), and everything else of the call (allocation of the iterates,
    E_0, hot-row marking on the third launch, the placement decision, the 10 GB download through the pinned pipeline)."""
    from cleora_amd import embed as dev_embed
    from cleora_amd.pycleora import SparseMatrix
    arr = {"rowptr": g["rowptr"].cpu().numpy().astype(np.uint64), "col": g["col"].cpu().numpy().view(np.uint32),
           "val_left": g["val_left"].cpu().numpy(), "val_sym": g["val_sym"].cpu().numpy(),
           "hashes": hashes.cpu().numpy().view(np.uint64)}
    sm = SparseMatrix._wrap(_CsrHost(arr))
    out = {"iterations": iterations, "workload": f"n={n}, nnz={g['nnz']}, d={d}; host CSR and host result (pageable numpy memory)"}
    t0 = time.perf_counter()
    sm._graph()
    out["graph_upload_s"] = round(time.perf_counter() - t0, 3)
    for name, call in (("embed_fast", lambda: sm.embed_fast(d, iterations)),
                       ("embed_default_whitened", lambda: dev_embed.embed(sm, d, iterations))):
        try:
            t0 = time.perf_counter()
            res = call()
            wall = time.perf_counter() - t0
            loop = L.cleora_last_embed_loop_ms() / 1e3
            out[name] = {"wall_s": round(wall, 3), "loop_s": round(loop, 3), "outside_the_loop_s": round(wall - loop, 3),
                         "iterations_per_sec_end_to_end": round(iterations / wall, 2), "finite": bool(np.isfinite(res[:: max(1, n // 4096)]).all())}
            del res
        except Exception as ex:                               # noqa: BLE001 - an extra beside the headline
            out[name] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    sm._dev_graph.close()
    return out


def cpu_baseline_row_block(g, x_dev, n, d, rows=500_000, budget_s=20.0):
    """cpu_baseline for graphs whose iterate does not fit the host comfortably: the oracle's SpMM + L2 over a contiguous
    block of `rows` OUTPUT rows of the same graph.  The gathered X rows are remapped to a compact array (only the rows the block
    touches are copied from the device), so the arithmetic and the access pattern per edge are the reference's; the rate is
    per edge and extrapolates to the whole graph."""
    import oracle
    r0 = n // 3
    rows = min(rows, n - r0)
    rp = g["rowptr"][r0:r0 + rows + 1]
    e0, e1 = int(rp[0]), int(rp[-1])
    cols = g["col"][e0:e1].long()
    ucols, inv = torch.unique(cols, return_inverse=True)
    x = x_dev[ucols].cpu().numpy()
    edges = np.empty(e1 - e0, dtype=oracle.EDGE_DTYPE)
    edges["col"] = inv.cpu().numpy().astype(np.uint32)
    edges["left"] = g["val_left"][e0:e1].cpu().numpy()
    edges["sym"] = 0.0
    rowptr = (rp - e0).cpu().numpy().astype(np.uint64)
    y = np.empty((rows, d), np.float32)
    threads = oracle.max_threads()
    oracle.spmm_aos_l2_inplace(rowptr, edges, False, x, y, threads)           # page touch
    count, t0 = 0, time.perf_counter()
    while True:
        oracle.spmm_aos_l2_inplace(rowptr, edges, False, x, y, threads)
        count += 1
        el = time.perf_counter() - t0
        if el >= budget_s or count >= 5:
            break
    return {"value": (e1 - e0) * d * count / el, "unit": "edge*dim/s", "cores": threads, "host_cpu": host_cpu_model(), "kind": "port",
            "sample": f"{count} pass(es) over output rows [{r0}, {r0 + rows}) = {e1 - e0} edges gathering {int(ucols.numel())} distinct X rows "
                      f"(reference AoS edge layout, SpMM + L2), {el:.1f} s; the whole graph is {g['nnz']} edges"}


def sampled_row_check(g, x_dev, y_dev, n, d, hub_threshold, rows=4096, seed=11, hub_rows=24):
    """Parity check that scales to graphs whose iterate does not fit a host-side oracle run (config 4's size: X is 114 GB):
    `rows` random rows AND the `hub_rows` longest rows (hub rows: the in-order hub launch) of the GPU's iteration
    y = l2_normalise(A x), recomputed by the oracle (oracle.spmm_aos_l2_inplace: acc += v * x[c] edge by edge in stored order with
    separate f32 multiply and add, the sum of squares in index order, v * (1 / max(sqrt(s), 1e-10)): src/embedding.rs:76-102) on a
    compact sub-problem — exactly the X rows those edges touch, gathered on the device and copied once.  Every compared row must be
    bit-equal."""
    import oracle
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed)
    rp = g["rowptr"]
    deg_all = torch.diff(rp)
    longest = torch.topk(deg_all, min(hub_rows, n)).indices.cpu()
    cand = torch.cat([torch.randint(0, n, (rows * 2,), generator=gen), longest]).unique()
    deg = deg_all[cand.to(rp.device)].cpu()
    is_long = torch.isin(cand, longest)
    keep = (deg > 0) & (is_long | (torch.cumsum((~is_long).long(), 0) <= rows))
    pick = cand[keep]
    pick_d = pick.to(rp.device)
    beg, end = rp[pick_d], rp[pick_d + 1]
    cnt = (end - beg)
    sub_rp = torch.zeros(len(pick) + 1, dtype=torch.int64, device=rp.device)
    sub_rp[1:] = torch.cumsum(cnt, 0)
    # edge indices of the picked rows, in order
    total = int(sub_rp[-1])
    row_of = torch.repeat_interleave(torch.arange(len(pick), device=rp.device), cnt)
    idx = beg[row_of] + (torch.arange(total, device=rp.device) - sub_rp[row_of])
    cols = g["col"][idx].long()
    ucols, inv = torch.unique(cols, return_inverse=True)
    xs = x_dev[ucols].cpu().numpy()                       # only the rows these edges gather
    edges = np.empty(total, dtype=oracle.EDGE_DTYPE)
    edges["col"] = inv.cpu().numpy().astype(np.uint32)
    edges["left"] = g["val_left"][idx].cpu().numpy()
    edges["sym"] = 0.0
    want = np.empty((len(pick), d), np.float32)
    oracle.spmm_aos_l2_inplace(sub_rp.cpu().numpy().astype(np.uint64), edges, False, xs, want, oracle.max_threads())
    got = y_dev[pick_d].cpu().numpy()
    same = (got.view(np.uint32) == want.view(np.uint32)).all(axis=1)
    hub = (cnt.cpu().numpy() > hub_threshold)
    return {"sampled_rows_compared": int(len(pick)), "sampled_rows_bit_equal": int(same.sum()),
            "sampled_hub_rows_compared": int(hub.sum()), "sampled_hub_rows_bit_equal": int(same[hub].sum()),
            "longest_row_edges": int(cnt.max()), "sampled_rows_edges": total,
            "note": "the oracle's iteration (reference order) on a compact sub-problem of randomly chosen rows plus the longest rows (the full "
                    "oracle iteration is run when the iterate fits the host: see oracle_rows_compared)"}


def c_abi_communicator(local_rank, dev, rank, world, transport="rccl"):
    """The C-ABI communicator, checked with one small all-reduce before anything depends on it.  Agreement points over the gloo
    launcher group — after the creation and after the probe — so that the ranks always take the same branch.  transport "rccl":
    RCCL over xGMI (csrc/comm.hip); if it cannot be created or gives a wrong sum on ANY rank, every rank moves to the library's
    second transport, the peer-direct hipIpc one (csrc/peer.hip, "local") — still the C ABI, and config.collectives says which; if
    that fails too every rank stops (exit code != 0: nothing is measured over torch.distributed's collectives — VERDICT round 3,
    weak #6).  (What this cannot catch: a rank that dies before ncclCommInitRank leaves the others waiting inside it; the
    watchdog ends that run.)"""
    def agree(err):
        ok = torch.tensor([0 if err else 1], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok) == 1:
            return None
        errs = [None] * world
        dist.all_gather_object(errs, err)
        return next(e for e in errs if e)

    def attempt(local):
        err, comm = "", None
        try:
            comm = comm_mod.RcclComm.from_torch_distributed(local_rank, local=local)
        except Exception as e:                    # noqa: BLE001 - reported below, on every rank
            err = f"{type(e).__name__}: {e}"
        first = agree(err)
        if first is None:
            try:
                probe = torch.full((1024,), float(rank + 1), device=dev)
                comm.allreduce(probe)
                if dev.type == "cuda":
                    torch.cuda.synchronize()
                if not bool((probe == world * (world + 1) / 2).all()):
                    err = f"all-reduce probe gave {float(probe[0])}, expected {world * (world + 1) / 2}"
            except Exception as e:                # noqa: BLE001
                err = f"{type(e).__name__}: {e}"
            first = agree(err)
        if first is not None and comm is not None:
            try:
                comm.close()
            except Exception:                     # noqa: BLE001
                pass
            comm = None
        return comm, first

    why = None
    if transport == "rccl":
        comm, why = attempt(False)
        if comm is not None:
            return comm, "RCCL via the C ABI (csrc/comm.hip)"
        if rank == 0:
            print(f"bench.py: the RCCL communicator is unusable ({why}); trying the peer-direct transport of the C ABI", file=sys.stderr, flush=True)
    comm, why2 = attempt(True)
    if comm is not None:
        return comm, ("peer-direct hipIpc transport via the C ABI (csrc/peer.hip)"
                      + (f" — the RCCL communicator was unusable: {why[:200]}" if why else " (--backend local)"))
    raise SystemExit(f"bench.py: no C-ABI communicator on rank {rank}: RCCL: {why}; peer-direct: {why2}; nothing was measured")


class native_stdout_to_stderr:
    """gloo announces its connections on the process's C stdout ("[Gloo] Rank 0 is connected to ..."): while the process groups
    and communicators are being created, file descriptor 1 points at stderr, so that stdout carries the ONE JSON line only."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def launch_check(rank, world):
    """--launch-check: what the launcher must provide, without a GPU — the ranks find each other over gloo and agree on a sum;
    rank 0 prints one JSON line (tests/test_bench_launch.py runs it in the build container)."""
    with native_stdout_to_stderr():
        dist.init_process_group("gloo")
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
    ok = int(t) == world * (world + 1) // 2
    if rank == 0:
        print(json.dumps({"launch_check": ok, "world": world, "sum": int(t), "local_ranks": "0.." + str(world - 1)}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


class Launcher:
    """torch.distributed as the launcher only: barrier and max-over-ranks of host timings (gloo, CPU tensors)."""

    def __init__(self, world):
        self.world = world

    def barrier(self):
        if self.world > 1:
            dist.barrier()

    def max(self, v):
        if self.world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)


def register_replicas(comm, *bufs):
    """Both replicas of the iterate mapped by every rank for the peer-direct all-gather (collective).  On a LOCAL communicator that
    transport is all there is: a failure ends the run.  On an RCCL communicator it is the optional third algorithm: if the
    mapping fails on this node (the C side fails on every rank together, csrc/peer.hip exchange_and_map) the algorithm leaves the
    pick and RCCL's two remain."""
    if not isinstance(comm, comm_mod.RcclComm) or not (comm.local or getattr(comm, "peer_enabled", False)):
        return
    err, done = "", []
    try:
        for b in bufs:
            comm.register(b)
            done.append(b)
    except Exception as e:                        # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    if comm.local:
        if err:
            raise SystemExit(f"bench.py: the peer-direct transport could not map a replica ({err}); nothing was measured")
        return
    ok = torch.tensor([0 if err else 1], dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok) != 1:
        for b in done:
            comm.unregister(b)
        comm.peer_enabled = False
        comm.peer_note = err[:200] or "the mapping failed on another rank"
        print(f"bench.py: peer-direct all-gather excluded: {comm.peer_note}", file=sys.stderr, flush=True)


def unregister_replicas(comm, *bufs):
    if isinstance(comm, comm_mod.RcclComm):
        for b in bufs:
            comm.unregister(b)                    # a no-op for a buffer that was never registered


def partition_selftest(dev, rank, world, comm, backend, L, nodes=100_000, pairs=950_000, d=64, column=True):
    """N > 1, before anything big is built: ONE iteration (SpMM + L2 norm) of every partition on a 100k-row power-law graph,
    compared on every rank with the same iteration computed by that rank alone through the single-GPU path.  The row
    partition must reproduce it bit for bit (same kernel, same edges per row, all-gather of finished rows); the column
    partition to 2e-6 on unit rows (the row norm is a sum of per-slice partial sums).  Any rank failing stops all ranks."""
    t0 = time.perf_counter()
    g = synth.power_law_graph(nodes, pairs, 7, dev)
    n, nnz = g["n"], g["nnz"]
    hashes = synth.entity_hashes(n, 3, dev)
    stream = torch.cuda.current_stream().cuda_stream
    single = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(), None,
                                    dev.index or 0, keepalive=g)
    x = torch.empty((n, d), dtype=torch.float32, device=dev)
    _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, x.data_ptr(), d, stream))
    want = torch.empty_like(x)
    _hip.check(L.cleora_propagate_dev(single.handle, _hip.LEFT, x.data_ptr(), d, d, want.data_ptr(), d, _hip.F_L2NORM, 0.0, None,
                                      None, None, stream))
    out = {"graph": f"power-law n={n} nnz={nnz} d={d}"}
    sg = sharded.DeviceShardedGraph(n, g["rowptr"], g["col"], g["val_left"], None, comm, 2, "auto", dev.index or 0)
    xr = torch.zeros((sg.n_pad, d), dtype=torch.float32, device=dev)
    xr[:n] = x
    yr = torch.zeros_like(xr)
    torch.cuda.synchronize()                              # the fills must have RUN before a peer may store into the replicas
    register_replicas(comm, xr, yr)
    out["row_max_abs_diff"], out["row_bit_equal"], algos = 0.0, True, [_hip.ALLGATHER_PEER] if comm.local else [_hip.ALLGATHER_RING, _hip.ALLGATHER_P2P]
    if not comm.local and getattr(comm, "peer_enabled", False):
        algos.append(_hip.ALLGATHER_PEER)
    names = {_hip.ALLGATHER_RING: "rccl_allgather", _hip.ALLGATHER_P2P: "p2p_mesh", _hip.ALLGATHER_PEER: "peer_direct"}
    for algo in algos:                                    # every all-gather algorithm the timed run may pick
        comm.set_allgather(algo)
        # the zero fill of this rank must have RUN on every rank before any rank's first block is pushed into the peers' copies: a
        # quicker peer's rows would be wiped by a fill still queued here (seen with 8 ranks sharing one GPU, round 5: three of eight
        # ranks lost rows; the timed loops never fill a buffer the peers write)
        yr.zero_()
        torch.cuda.synchronize()
        dist.barrier()
        sg.propagate(_hip.LEFT, xr, yr)
        torch.cuda.synchronize()
        diff, equal, note = float((yr[:n] - want).abs().max()), bool(torch.equal(yr[:n], want)), ""
        try:
            comm.check()                                  # a device-side wait that ran out of its budget (peer-direct transport)
        except RuntimeError as e:
            equal, note = False, str(e)[:200]
        flag = torch.tensor([1 if equal else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag) != 1 and algo == _hip.ALLGATHER_PEER and not comm.local:
            # the optional third algorithm misbehaves on this node: it leaves the pick (on every rank alike); RCCL's two remain
            comm.peer_enabled = False
            out["peer_direct_excluded"] = note or f"one iteration differed from the single-rank result by {diff:.3g} on some rank"
            continue
        out.setdefault("algorithms_checked", []).append(names[algo])
        out["row_max_abs_diff"] = max(out["row_max_abs_diff"], diff)
        out["row_bit_equal"] = out["row_bit_equal"] and equal
    comm.set_allgather(algos[0])
    unregister_replicas(comm, xr, yr)
    sg.close()
    if column and d % (4 * world) == 0:
        # the column partition through the C ABI (csrc/colsharded.hip): the rows' sums of squares travel from rank to rank in the
        # reference's order, so the slice must equal the single-rank result bit for bit
        cg = sharded.DeviceColShardedGraph(n, g["rowptr"], g["col"], g["val_left"], None, d, comm, 2, dev.index or 0)
        xc = x[:, cg.c0:cg.c0 + cg.dl].contiguous()
        yc = torch.zeros_like(xc)
        cg.propagate(_hip.LEFT, xc, yc, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        out["column_max_abs_diff"] = float((yc - want[:, cg.c0:cg.c0 + cg.dl]).abs().max())
        out["column_bit_equal"] = bool(torch.equal(yc.view(torch.int32), want[:, cg.c0:cg.c0 + cg.dl].contiguous().view(torch.int32)))
        cg.close()
    ok = out["row_max_abs_diff"] <= 1e-6 and out.get("column_max_abs_diff", 0.0) <= 2e-6 and bool(torch.isfinite(yr).all())
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    out["seconds"] = round(time.perf_counter() - t0, 2)
    single.close()
    if int(flag) != 1:
        raise SystemExit(f"bench.py: partition self-test FAILED on rank {rank}: {out} (some rank disagrees with its own "
                         f"single-GPU iteration: collectives or partition logic are broken; nothing was measured)")
    return out


CONFIGS = {
    # BASELINE.json configs; C3 is the one the metric is quoted on and the default the driver runs
    "C3": {"kind": "power_law", "nodes": 10_000_000, "pairs": 95_000_000, "dim": 256,
           "label": "synthetic power-law graph, reflexive column semantics (BASELINE config 3)"},
    "C2": {"kind": "bipartite", "nodes": 1_000_000, "pairs": 10_000_000, "dim": 256,
           "label": "synthetic bipartite graph, two plain columns (BASELINE config 2)"},
    "C4s": {"kind": "power_law", "nodes": 111_000_000, "pairs": 800_000_000, "dim": 256,
            "label": "ogbn-papers100M stand-in on ONE GPU: synthetic power-law graph of its size, |V| = 111M, ~1.6e9 stored edges "
                     "(BASELINE config 4; the dataset itself is not in the image)"},
    "C5": {"kind": "hypergraph", "hyperedges": 5_000_000, "products": 2_000_000, "dim": 1024,
           "label": "hypergraph complex::reflexive::product, clique expansion by the host builder (BASELINE config 5)"},
}


def hypergraph_lines(n_lines, n_products, seed):
    """5M-line scale without 40M Python strings: the text buffer of `complex::reflexive::product` lines is assembled with
    numpy — tokens `pNNNNNNN` (fixed width), arities uniform on 2..14 (mean 8), product ids Zipf-like (floor(P u^3), randomly
    permuted like synth.power_law_graph).  Returns (bytes, offsets u64[n_lines + 1], total tokens)."""
    rng = np.random.default_rng(seed)
    arity = rng.integers(2, 15, n_lines, dtype=np.int64)
    total = int(arity.sum())
    u = rng.random(total)
    perm = rng.permutation(n_products)
    ids = perm[np.minimum((u * u * u * n_products).astype(np.int64), n_products - 1)]
    del u
    tok = np.empty((total, 9), dtype=np.uint8)
    tok[:, 0] = ord("p")
    rem = ids.copy()
    for c in range(7, 0, -1):
        tok[:, c] = (rem % 10 + ord("0")).astype(np.uint8)
        rem //= 10
    tok[:, 8] = ord(" ")          # the separator; the one behind a line's last token is removed by the builder's trim
    offsets = np.zeros(n_lines + 1, dtype=np.uint64)
    np.cumsum(arity * 9, out=offsets[1:].view(np.int64))
    return tok.tobytes(), offsets, total


def make_workload(args, dev, rank, world, share_gpu):
    """The graph of --config as device CSR tensors (the same dict synth.* returns) + entity hashes + a description."""
    cfg = dict(CONFIGS[args.config])
    if args.nodes:
        cfg["nodes"] = args.nodes
    if args.pairs:
        cfg["pairs"] = args.pairs
    if cfg["kind"] == "hypergraph":
        import ctypes
        from cleora_amd import _host
        if args.hyperedges:
            cfg["hyperedges"] = args.hyperedges
        if args.products:
            cfg["products"] = args.products
        t0 = time.perf_counter()
        data, offsets, tokens = hypergraph_lines(cfg["hyperedges"], cfg["products"], 5)
        t1 = time.perf_counter()
        h = _host.vp()
        rc = _host.lib().cleora_host_build_from_lines(data, offsets.ctypes.data_as(_host.vp), cfg["hyperedges"],
                                                      b"complex::reflexive::product", 16, ctypes.byref(h))
        if rc != 0:
            raise SystemExit("host builder: " + _host.last_error())
        hg = _host.HostGraph(h)
        arr = hg.arrays()
        t2 = time.perf_counter()
        del data
        n, nnz = int(arr["rowptr"].shape[0] - 1), int(arr["col"].shape[0])
        g = {"n": n, "nnz": nnz, "rowptr": torch.from_numpy(arr["rowptr"].view(np.int64)).to(dev),
             "col": torch.from_numpy(arr["col"].view(np.int32)).to(dev), "val_left": torch.from_numpy(arr["val_left"]).to(dev),
             "val_sym": torch.from_numpy(arr["val_sym"]).to(dev)}
        hashes = torch.from_numpy(arr["hashes"].view(np.int64)).to(dev)
        hg.close()
        desc = (f"{cfg['label']}: {cfg['hyperedges']} hyperedges, {tokens} tokens (arity 2..14, mean {tokens / cfg['hyperedges']:.2f}), "
                f"Zipf-like over {cfg['products']} products; built by cleora_host_build_from_lines in {t2 - t1:.1f} s "
                f"(text assembled in {t1 - t0:.1f} s)")
        return g, hashes, desc, cfg

    def gen():
        if cfg["kind"] == "bipartite":
            return synth.bipartite_graph(cfg["nodes"] // 2, cfg["nodes"] // 2, cfg["pairs"], 1, dev)
        return synth.power_law_graph(cfg["nodes"], cfg["pairs"], 2, dev)
    if share_gpu and world > 1:
        # developer mode: FOUR processes building the graph at once on ONE GPU (torch sort / unique / mask-index) stalled
        # for minutes before any collective ran (round 1's "hang"; tracebacks in gpurun_out/r02a/share4.log) — take turns
        for r in range(world):
            if r == rank:
                g = gen()
                torch.cuda.synchronize()
            dist.barrier()
    else:
        g = gen()                             # same seed on every rank
    return g, synth.entity_hashes(g["n"], 0, dev), cfg["label"], cfg


def placed_pair(block, rows, d, dev, args):
    """The two iterate buffers, placed by the LIBRARY's search (cleora_alloc_iterates — the same call the product's
    loops make: cleora_amd/embed.py, cleora_embed), as torch views.  Outside the timed region."""
    if args.no_placement:
        return (torch.empty((rows, d), dtype=torch.float32, device=dev), torch.empty((rows, d), dtype=torch.float32, device=dev),
                {"library_search": False})
    (a, b), ms = _hip.DevArray.iterates(block, rows, d, 2)
    return (torch.as_tensor(a, device=dev), torch.as_tensor(b, device=dev),
            {"library_search": True, "untuned_launch_ms": round(ms[0], 3), "chosen_launch_ms": round(ms[1], 3),
             "note": "cleora_alloc_iterates: median of three SpMM launches per candidate partner buffer (at most four), the best taken; "
                     "untuned = the first (plain) pair.  The embed loops skip the search when the iterations cannot repay it "
                     "(cleora_alloc_iterates_for: ~100 iterations at this size)"})


def run_partition(part, args, g, hashes, deg, dev, rank, world, comm, launcher, backend, L):
    """W + K iterations of one partition; returns (result dict, x, x_next, iterate, blocks, n_pad)."""
    n, nnz, d = g["n"], g["nnz"], args.dim
    steps_per_iter = args.overlap_steps or (1 if world == 1 else 4)
    launch_bytes = []
    stream = torch.cuda.current_stream().cuda_stream
    sg = None
    if part == "row":
        # csrc/sharded.hip through the C ABI: cleora_sharded_create + cleora_sharded_propagate_dev (the loop body of cleora_embed_sharded)
        ccomm = comm if isinstance(comm, comm_mod.RcclComm) else None
        sg = sharded.DeviceShardedGraph(n, g["rowptr"], g["col"], g["val_left"], None, ccomm, steps_per_iter, args.balance, dev.index or 0)
        blocks = [sg.block(k)[0] for k in range(steps_per_iter)]
        for k in range(steps_per_iter):
            b0, b1 = sg.my_rows[k]
            r0, r1 = min(b0, n), min(b1, n)
            dk = deg[r0:r1]
            # one SpMM = the main launch + the in-order hub launch beside it on the side stream (spmm.hip): together they gather
            # for ALL edges of the block and write every row once; the time below is the span of the pair on the launch stream
            launch_bytes.append(algorithmic_bytes(int(dk.sum()), b1 - b0, b1 - b0, d))
        x, x_next, placement = placed_pair(blocks[0], sg.n_pad, d, dev, args)
        x.zero_()
        x_next.zero_()
        _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, x.data_ptr(), d, stream))
        if ccomm is not None:
            torch.cuda.synchronize()                   # the zero fills must have RUN before a peer may store into the replicas
            register_replicas(ccomm, x, x_next)        # peer-direct all-gather: both replicas mapped by every rank
        dl = d
        par = (f"row-block-cyclic x{world} ({sg.balance}-balanced), {steps_per_iter} block(s)/rank/iter, csrc/sharded.hip through the C ABI"
               + (", in-place all-gather of X (C ABI) overlapped with the next block" if world > 1 else ""))

        def iterate(a, b):
            sg.propagate(_hip.LEFT, a, b)
    else:
        cg = sharded.DeviceColShardedGraph(n, g["rowptr"], g["col"], g["val_left"], None, d, comm if world > 1 else None, steps_per_iter, dev.index or 0)
        sg = cg                                        # (closed by the caller like the row partition's handle)
        blocks, dl = cg.blocks, cg.dl
        for blk, (r0, r1) in zip(cg.blocks, cg.row_blocks):
            dk = deg[r0:r1]
            launch_bytes.append(algorithmic_bytes(int(dk.sum()), r1 - r0, r1 - r0, dl))
        x, x_next, placement = placed_pair(blocks[0], n, dl, dev, args)
        # columns [c0, c0 + dl) of the deterministic init: init_value depends on hash + col + seed only
        _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, dl, cg.c0, x.data_ptr(), dl, stream))
        par = (f"column partition x{world} through the C ABI (csrc/colsharded.hip): every rank owns {dl} of {d} columns and the whole CSR; the rows' "
               f"sums of squares travel from rank to rank in the reference's order ({world} broadcasts of n / {steps_per_iter} floats per row block, block k's "
               f"chain beside block k+1's SpMM): bit-equal to one GPU, no exchange of X")

        def iterate(a, b):
            cg.propagate(_hip.LEFT, a, b, stream=stream)

    def sync():
        torch.cuda.synchronize()
        launcher.barrier()
        torch.cuda.synchronize()

    extra = {}
    # all-gather algorithm (row partition, N > 1): RCCL's ring all-gather, the send/recv mesh, the peer-direct stores — two timed
    # iterations each, BEFORE the timed region; every rank sees the max-reduced times, so they agree on the pick
    if part == "row" and world > 1 and isinstance(comm, comm_mod.RcclComm):
        candidates = [("peer_direct", _hip.ALLGATHER_PEER)] if comm.local else [("rccl_allgather", _hip.ALLGATHER_RING), ("p2p_mesh", _hip.ALLGATHER_P2P)]
        if not comm.local and getattr(comm, "peer_enabled", False):
            candidates.append(("peer_direct", _hip.ALLGATHER_PEER))
        if getattr(comm, "peer_note", None):
            extra["peer_direct_excluded"] = comm.peer_note
        tried = {}
        for name, algo in candidates:
            comm.set_allgather(algo)
            iterate(x, x_next)
            sync()
            t0 = time.perf_counter()
            iterate(x_next, x)
            iterate(x, x_next)
            sync()
            ms = launcher.max(time.perf_counter() - t0) / 2 * 1e3
            bad = 0.0
            try:
                comm.check()
            except RuntimeError:
                bad = 1.0
            if launcher.max(bad) > 0:
                extra.setdefault("allgather_excluded", []).append(name)
                continue
            tried[name] = ms
        if not tried:
            raise SystemExit("bench.py: no all-gather algorithm of the C-ABI communicator worked on every rank; nothing was measured")
        best = min(tried, key=tried.get)
        comm.set_allgather(dict(candidates)[best])
        extra["allgather_ms_per_iter_tried"] = {k: round(v, 3) for k, v in tried.items()}
        extra["allgather"] = best
        _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, x.data_ptr(), d, stream))

    a, b = x, x_next
    for _ in range(args.warmup):
        iterate(a, b)
        a, b = b, a
    if sg is not None:
        sg.set_timing(True)                            # the blocks' kernels and the all-gathers on the communication stream
    else:
        for blk in blocks:
            blk.set_timing(True)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        iterate(a, b)
        a, b = b, a
    sync()
    elapsed = launcher.max(time.perf_counter() - t0)

    rows_ms, calls, other_ms = 0.0, 0, 0.0
    for blk in blocks:
        ms, c = blk.get_timing()
        rows_ms += ms[1]
        other_ms += ms[0] + ms[2]
        calls += c
    if sg is not None:
        (_, gather_ms), _ = sg.get_timing()            # (the blocks' own records were read above: the SpMM part comes back as 0 here)
        sg.set_timing(False)
        if world > 1 and part == "row":
            # VERDICT round 3, next #1c: what an iteration is made of on the slowest rank, beside the stated ceiling
            spmm_iter = launcher.max((rows_ms + other_ms) / args.steps)
            gather_iter = launcher.max(gather_ms / args.steps)
            step_ms = elapsed / args.steps * 1e3
            recv_gb = (world - 1) / world * n * d * 4 / 1e9
            extra["per_iteration_ms"] = {"spmm_kernels_max_rank": round(spmm_iter, 3), "allgather_on_comm_stream_max_rank": round(gather_iter, 3),
                                         "exposed_beyond_spmm": round(step_ms - spmm_iter, 3), "wall": round(step_ms, 3)}
            lo, hi = recv_gb / ((world - 1) * 0.1536) , recv_gb / ((world - 1) * 0.0768)     # ms at 153.6 / 76.8 GB/s per link and direction
            extra["ceiling"] = {"received_GB_per_rank_per_iteration": round(recv_gb, 3), "links": world - 1,
                                "gather_ms_at_153.6_GBps_per_link": round(lo, 2), "gather_ms_at_76.8_GBps_per_link": round(hi, 2),
                                "iterations_per_sec_if_gather_bound": [round(1e3 / max(hi, spmm_iter), 1), round(1e3 / max(lo, spmm_iter), 1)],
                                "note": "one xGMI link per peer (7 x ~153 GB/s per GPU by the environment's figure; 76.8 if that is bidirectional); "
                                        "block k's gather overlaps block k+1's SpMM, so an iteration is >= max(SpMM, gather)"}
    else:
        for blk in blocks:
            blk.set_timing(False)
    # span of one SpMM on the launch stream: [fork of the hub launch | main kernel | wait for the hub launch's epilogue]
    avg_ms = (rows_ms + other_ms) / max(calls, 1)
    avg_bytes = sum(launch_bytes) / len(launch_bytes)
    achieved = avg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    finite, sumsq = finite_and_row_sumsq(a, n)
    if part == "column" and world > 1:
        comm.allreduce(sumsq)
        torch.cuda.synchronize()
    norm_err = float((sumsq.sqrt() - 1).abs().max())
    hot_rows = int(blocks[0].info().hot_rows)      # gather cache policy (cleora_graph_set_hot_cache), 0 = inactive
    g_lanes = max(8, min(64, 1 << (max(dl // 4, 1) - 1).bit_length()))
    n_long = sum(int(blk.info().n_hub_rows) for blk in blocks)
    n_io = sum(int(blk.info().n_inorder_rows) for blk in blocks)
    kernel_name = (f"spmm_rows_kernel<{g_lanes},{max(1, dl // 256)},4,true,{'true' if hot_rows else 'false'}>"
                   f"  (G lanes per row, float4 chunks per lane, W, FULL, gather cache policy; its first work items: the {n_long - n_io} rows of "
                   f"{blocks[0].info().hub_threshold} < edges <= {blocks[0].info().hub_inorder_min}, longest first)"
                   + (f" + hub_inorder_kernel<48,L> beside it on a side stream ({n_io} longer rows cut into column slabs, reference order)" if n_io else ""))
    res = {
        "value": nnz * d * args.steps / elapsed, "iterations_per_sec": args.steps / elapsed,
        "ms_per_step": elapsed / args.steps * 1e3, "parallelism": par,
        "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                     "algorithmic_bytes_per_launch": avg_bytes, "avg_launch_ms": avg_ms, "launches": calls,
                     "gather_cache_policy_hot_rows": hot_rows,
                     "main_kernel_ms_per_launch": rows_ms / max(calls, 1),
                     "fork_and_join_of_the_hub_launch_ms_per_launch": other_ms / max(calls, 1),   # (the sharded loops join the hub launch at the block's gather or the end of the call: csrc/sharded.hip)
                     "timing": "HIP events on the launch stream around [fork | main kernel | join]: the span of the main launch and the in-order hub launch "
                               "(side stream) together; achieved = the gather model's bytes of ALL edges and rows / that span",
                     "frac_of_measured_copy_ceiling_6290": achieved / 6290.0,
                     # SURVEY.md §8(d) secondary model: every operand once (lower bound on traffic)
                     "compulsory_bytes_per_iteration": nnz * 8 + (n + 1) * 8 + 2 * n * d * 4},
        "checks": {"finite": finite, "max_abs_row_norm_minus_1": norm_err},
        "placement_tuning": placement,
    }
    res.update(extra)
    if sg is not None and isinstance(comm, comm_mod.RcclComm) and world > 1:
        comm.check()                                   # a device-side wait that ran out of its budget is an error, not a number
    return res, a, b, iterate, blocks, sg


def finite_and_row_sumsq(a, n, rows_per_pass=4_000_000):
    """Post-loop checks in row slabs, so that the f64 temporaries stay a few GB next to a 100+ GB iterate (config 4s)."""
    finite, sumsq = True, torch.empty(n, dtype=torch.float64, device=a.device)
    for r0 in range(0, n, rows_per_pass):
        blk = a[r0:min(n, r0 + rows_per_pass)]
        finite = finite and bool(torch.isfinite(blk).all())
        sumsq[r0:r0 + blk.shape[0]] = blk.double().pow(2).sum(1)
    return finite, sumsq


def project_roofline(n, d, ms, split):
    """The projection (X - mu) T, n x d by d x d.  f32-MFMA forms: bound by the f32 matrix cores (2 n d^2 flops against 157.3 TF).
    Split-bf16 form: it executes SIX bf16 MFMA products per f32 product (12 n d^2 bf16 flops against the 2.5 PF dense bf16
    peak) and moves n d 4 bytes in and n d 4 bytes out; the line names whichever bound is the tighter one and carries both."""
    if not ms:
        return None
    if not split:
        a = 2.0 * n * d * d / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "dtype": "f32", "achieved": a, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": a / F32_MFMA_PEAK_TF}
    tf = 12.0 * n * d * d / (ms * 1e-3) / 1e12
    gbps = 2.0 * n * d * 4 / (ms * 1e-3) / 1e9
    mf, hf = tf / BF16_MFMA_PEAK_TF, gbps / HBM_PEAK_GBPS
    out = {"mfma": {"dtype": "bf16 (3-way split f32 operands, 6 products)", "achieved": tf, "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": mf},
           "hbm": {"achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hf, "algorithmic_bytes": 2.0 * n * d * 4},
           "f32_equivalent_tflops": 2.0 * n * d * d / (ms * 1e-3) / 1e12}
    out["bound"] = "mfma" if mf >= hf else "hbm"
    out["frac"] = max(mf, hf)
    return out


def whitened_record_check(config, g, hashes, n, d, graph, L, dev):
    """The DEFAULT loop (cleora_embed_dev + CLEORA_F_WHITEN, what pycleora.embed() runs) at this workload's FULL size against the
    ORACLE's loop (oracle.spmm + oracle/whiten.py: pycleora/__init__.py:109-117,130-164) pinned as tests/golden/whitened_loop_<config>.npz
    — written by tests/golden/make_whitened_loop_record.py from the oracle on the GPU box (4 iterations at config 3, 3 at config 5).
    Invariants of PCA whitening: pairwise cosines and norms of fixed rows, the spectrum of the covariance the last iteration whitened,
    and |cov - I| over all rows.  The GPU side is measured live in this run, in the product's order and in the reference's order."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import make_whitened_loop_record as wrec
        rec = wrec.load_record(config)
        if rec is None:
            return {"error": f"no golden record tests/golden/whitened_loop_{config}.npz for this config"}
        m = rec["meta"]
        if (m["n"], m["nnz"], m["d"]) != (n, g["nnz"], d) or m["graph"] != wrec.graph_hash(g):
            return {"error": "the golden record describes another graph (generator output differs): re-run tests/golden/make_whitened_loop_record.py"}
        x0 = torch.empty((n, d), dtype=torch.float32, device=dev)
        _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, x0.data_ptr(), d, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        if wrec.hash_array(x0) != m["x0"]:
            return {"error": "E_0 differs from the golden record's"}
        out = {"source": f"tests/golden/whitened_loop_{config}.npz (oracle loop: {m['oracle_threads']} host threads, {m['oracle_seconds']} s); the GPU side measured in this run",
               "default_loop": wrec.gpu_invariants(L, graph, x0, n, d, rec),
               "reference_order_loop": wrec.gpu_invariants(L, graph, x0, n, d, rec, threshold=1e-30)}
        out["within_tolerance"] = bool(out["default_loop"]["within_tolerance"] and out["reference_order_loop"]["within_tolerance"])
        return out
    except Exception as ex:                                   # noqa: BLE001
        return {"error": f"{type(ex).__name__}: {ex}"[:300]}


def run_whitened(args, g, x, dev, L, iters, hashes):
    """The default pycleora.embed() loop on one GPU (pycleora/__init__.py:109-117): SpMM + fused L2 norm, then
    whiten_embeddings (cleora_whiten_dev: statistics, f64-MFMA Gram, eigensolver, f32-MFMA projection)."""
    n, nnz, d = g["n"], g["nnz"], args.dim
    gr = _hip.Graph.from_device(n, n, nnz, g["rowptr"].data_ptr(), g["col"].data_ptr(), g["val_left"].data_ptr(),
                                None, dev.index or 0, keepalive=(g["rowptr"], g["col"], g["val_left"]))
    # SpMM output in the first buffer, the two whitened iterates tuned against it (as cleora_amd/embed.py does)
    (m_, p_, n_), place_ms = _hip.DevArray.iterates(gr, n, d, 3)
    mid, prev, nxt = (torch.as_tensor(b, device=dev) for b in (m_, p_, n_))
    prev.copy_(x[:n])
    ws = torch.empty(L.cleora_whiten_workspace(n, d), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def iterate():
        nonlocal prev, nxt
        _hip.check(L.cleora_propagate_dev(gr.handle, _hip.LEFT, prev.data_ptr(), d, d, mid.data_ptr(), d,
                                          _hip.F_L2NORM, 0.0, None, None, None, stream))
        _hip.check(L.cleora_whiten_dev(mid.data_ptr(), d, n, d, d, nxt.data_ptr(), d, ws.data_ptr(), None, stream))
        prev, nxt = nxt, prev

    for _ in range(3):          # the gather cache policy arms on the third launch
        iterate()
    torch.cuda.synchronize()
    gr.set_timing(True)
    _hip.check(L.cleora_whiten_set_timing(1))
    t0 = time.perf_counter()
    for _ in range(iters):
        iterate()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms, c = gr.get_timing()
    gr.set_timing(False)
    wms, wc = (ctypes.c_double * 4)(), ctypes.c_uint64(0)
    _hip.check(L.cleora_whiten_get_timing(ctypes.byref(wms), ctypes.byref(wc)))
    _hip.check(L.cleora_whiten_set_timing(0))
    k = max(wc.value, 1)
    stats_ms, gram_ms, eigh_ms, proj_ms = (wms[i] / k for i in range(4))
    # executed MFMA work of the Gram: block tiles on/above the diagonal, and of a diagonal block tile only its
    # 36 upper 16x16 MFMA tiles of 64 (csrc/whiten.hip)
    tiles = -(-d // 128)
    mfma_tiles = tiles * 36 + (tiles * (tiles - 1) // 2) * 64
    gram_flops = 2.0 * n * mfma_tiles * 256
    proj_flops = 2.0 * n * d * d
    # the Gram form the product's loop takes in its intermediate iterations (f32 matrix cores for d a multiple of 256, csrc/whiten.hip
    # gram32_kernel), timed stand-alone with events on this stream
    m64 = torch.empty(d, dtype=torch.float64, device=dev)
    g64 = torch.empty((d, d), dtype=torch.float64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _hip.check(L.cleora_whiten_stats_dev(mid.data_ptr(), d, n, d, ws.data_ptr(), 1, m64.data_ptr(), g64.data_ptr(), stream))
    e0.record()
    for _ in range(3):
        _hip.check(L.cleora_whiten_stats_dev(mid.data_ptr(), d, n, d, ws.data_ptr(), 1, m64.data_ptr(), g64.data_ptr(), stream))
    e1.record()
    torch.cuda.synchronize()
    stats_inter_ms = e0.elapsed_time(e1) / 3
    # the statistics of the intermediate iterations: the split-bf16 form for d a multiple of 256 (csrc/whiten.hip gram16_kernel: THREE bf16
    # MFMA products per f32 product — 3 x the tile flops against the bf16 matrix peak — plus the diagonal's r1^2 term on the vector
    # unit), the f64 matrix cores for other widths
    split_gram = d % 256 == 0 and d <= 2048 and n >= 4096
    sup = d // 256        # 256-column super-tiles: 36 upper 32x32 tiles per diagonal one, 64 per off-diagonal pair (csrc/whiten.hip)
    gram32_flops = 2.0 * n * (sup * 36 + sup * (sup - 1) // 2 * 64) * 1024 if split_gram else gram_flops
    split_proj = d % 32 == 0
    if split_gram:
        gi_flops, gi_peak, gi_dtype, gi_kernel = 3.0 * gram32_flops, BF16_MFMA_PEAK_TF, "bf16 (two-way split f32 operands, 3 products + diagonal correction)", "gram16_kernel<true> (+ shift, mean, reduce)"
    else:
        gi_flops, gi_peak, gi_dtype, gi_kernel = gram_flops, F64_MFMA_PEAK_TF, "f64", "gram_kernel (f64)"
    gram_intermediate = {"bound": "mfma", "dtype": gi_dtype, "kernel": gi_kernel, "achieved": gi_flops / (stats_inter_ms * 1e-3) / 1e12,
                         "peak": gi_peak, "unit": "TFLOP/s", "frac": gi_flops / (stats_inter_ms * 1e-3) / 1e12 / gi_peak,
                         "executed_flops": gi_flops, "f32_equivalent_tflops": (gi_flops / 3.0 if split_gram else gi_flops) / (stats_inter_ms * 1e-3) / 1e12}
    del m64, g64
    # the projection the product's loop takes in its INTERMEDIATE iterations at d = 256 (csrc/project_f16.hip: bounded operands, three
    # f16 MFMAs per product, the transform resident in registers), timed stand-alone on the SpMM's output with its column means and a
    # dense transform; `form` tells which kernel ran (1: f16; 0: the six-product bf16 form — other widths)
    proj_inter = None
    if d % 32 == 0:
        rsum = torch.empty(n, dtype=torch.float32, device=dev)
        rabs = torch.empty(n, dtype=torch.float32, device=dev)
        _hip.check(L.cleora_csr_rowsums_dev(gr.handle, _hip.LEFT, rsum.data_ptr(), rabs.data_ptr(), stream))
        mu32 = torch.zeros(d, dtype=torch.float32, device=dev)
        for r0 in range(0, n, 1 << 21):
            mu32 += mid[r0:r0 + (1 << 21)].sum(0) / n
        mu32.clamp_(-1.0, 1.0)
        gen = torch.Generator(device=dev)
        gen.manual_seed(7)
        tr = (torch.randn((d, d), generator=gen, device=dev) + 16.0 * torch.eye(d, device=dev)).contiguous()
        nd_, fm_ = ctypes.c_int(0), ctypes.c_int(-1)

        def proj_once():
            _hip.check(L.cleora_project_bounded_dev(mid.data_ptr(), d, n, d, mu32.data_ptr(), tr.data_ptr(), d, nxt.data_ptr(), d,
                                                    rsum.data_ptr(), rabs.data_ptr(), 1, ctypes.byref(nd_), ctypes.byref(fm_), stream))
            if nd_.value == 0:        # several column passes (k > 256): the row pass the loop runs behind the projection (tree-sum form)
                _hip.check(L.cleora_rowops_dev(nxt.data_ptr(), d, n, d, nxt.data_ptr(), d, _hip.F_L2NORM | _hip.F_FASTNORM, 0.0, None, None, None, stream))
        proj_once()
        e0.record()
        for _ in range(5):
            proj_once()
        e1.record()
        torch.cuda.synchronize()
        pms = e0.elapsed_time(e1) / 5
        pbytes = 2.0 * n * d * 4 * (1 if nd_.value else 2)
        names = {1: "project_f16_kernel<1> (csrc/project_f16.hip)",
                 2: "project_split_kernel in its bounded-operand mode (three f16 products, csrc/whiten.hip)" + ("" if nd_.value else " + the stand-alone row normalise"),
                 0: "project_split_kernel (six-product bf16 form)"}
        proj_inter = {"kernel": names.get(fm_.value, str(fm_.value)),
                      "ms": pms, "bound": "hbm", "achieved": pbytes / (pms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                      "frac": pbytes / (pms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes": pbytes,
                      "mfma": {"dtype": "f16 (two-way split operands, 3 products)", "achieved": 6.0 * n * d * d / (pms * 1e-3) / 1e12,
                               "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": 6.0 * n * d * d / (pms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF},
                      "row_norm_max_abs_minus_1": float((nxt[:1 << 20].double().pow(2).sum(1).sqrt() - 1).abs().max())}
        del rsum, rabs, mu32, tr
    # the product's loop for this path (cleora_embed_dev, what cleora_amd.embed.embed() runs): SpMM(t+1) beside Gram / eigh(t)
    del mid, nxt, ws, m_, p_, n_
    torch.cuda.empty_cache()
    xo = x[:n].clone()
    loops = {}
    # untimed two-iteration call first: the first Cholesky / eigensolver call of a process initialises rocSOLVER and rocBLAS
    _hip.check(L.cleora_embed_dev(gr.handle, xo.data_ptr(), _hip.LEFT, d, 2, 0.0, 0.0, _hip.F_WHITEN, None))
    for label, thr in (("overlapped", 0.0), ("sequential", 1e-30)):      # a never-met threshold keeps the reference's order
        xo.copy_(x[:n])
        _hip.check(L.cleora_embed_dev(gr.handle, xo.data_ptr(), _hip.LEFT, d, iters, 0.0, thr, _hip.F_WHITEN, None))
        loops[label] = L.cleora_last_embed_loop_ms() / iters
    # the marginal iteration: the same call with twice the iterations (the first SpMM and the final PCA whitening — f64 Gram,
    # eigensolver — are paid once per call whatever its length)
    xo.copy_(x[:n])
    _hip.check(L.cleora_embed_dev(gr.handle, xo.data_ptr(), _hip.LEFT, d, 2 * iters, 0.0, 0.0, _hip.F_WHITEN, None))
    marginal_ms = (L.cleora_last_embed_loop_ms() - loops["overlapped"] * iters) / iters
    prev = xo
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_whitened_loop_record as wrec
    # ALL rows, f64, chunked torch matmuls (rocBLAS dgemm: not a kernel under test) — a 2M-row sample of config 3 read 3e-3 of pure
    # sampling noise
    cov_err = float((wrec.covariance_all_rows(prev, n) - torch.eye(d, device=dev, dtype=torch.float64)).abs().max())
    finite = finite_and_row_sumsq(prev, prev.shape[0])[0]
    del prev, xo
    torch.cuda.empty_cache()
    vs_record = whitened_record_check(args.config, g, hashes, n, d, gr, L, dev)
    out = {
        "ms_per_iter": loops["overlapped"], "iterations": iters, "iterations_per_sec": 1e3 / loops["overlapped"],
        "marginal_ms_per_iter": marginal_ms,
        "loop": "cleora_embed_dev + CLEORA_F_WHITEN (the SpMM taken before the projection; intermediate iterations: split-bf16 statistics before the "
                "SpMM, Cholesky transform, on the host for d <= 256; last iteration: f64 Gram + eigensolver); "
                "ms_per_iter = loop wall clock / iterations, incl. the final whitening, after an untimed 2-iteration call; "
                "marginal_ms_per_iter = (wall clock of a call with twice the iterations - this call's) / iterations",
        "sequential_ms_per_iter": {"c_loop_reference_order": loops["sequential"], "python_driven_with_stage_events": el / iters * 1e3},
        "placement_launch_ms": {"untuned": round(place_ms[0], 3), "chosen": round(place_ms[1], 3)},
        "kernels_ms": {"spmm_l2": ms[1] / max(c, 1) + (ms[0] + ms[2]) / max(c, 1), "column_statistics": stats_ms,
                       "gram_f64_mfma": gram_ms, "eigensolver_transform": eigh_ms, "project": proj_ms,
                       "statistics_intermediate_form": stats_inter_ms,
                       "project_intermediate_form": proj_inter["ms"] if proj_inter else None},
        "kernels_ms_note": "spmm_l2 .. project: the reference-order loop driven from Python with stage events (every iteration a full PCA whitening: f64 Gram, "
                           "eigensolver, six-product projection); *_intermediate_form: what the default loop runs instead in all but its last iteration",
        "project_intermediate_roofline": proj_inter,
        "project_form": ("split-bf16: every f32 product from six bf16 MFMAs of three-way split operands (csrc/whiten.hip)"
                         if split_proj else "f32 MFMA (tiled kernel: d is not a multiple of 32)"),
        "gram_intermediate_roofline": gram_intermediate,
        "gram_roofline": {"bound": "mfma", "dtype": "f64", "achieved": gram_flops / (gram_ms * 1e-3) / 1e12 if gram_ms else 0.0,
                          "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                          "frac": gram_flops / (gram_ms * 1e-3) / 1e12 / F64_MFMA_PEAK_TF if gram_ms else 0.0,
                          "executed_flops": gram_flops, "full_flops_2nd2": 2.0 * n * d * d},
        "project_roofline": project_roofline(n, d, proj_ms, split_proj),
        "checks": {"finite": finite, "max_abs_cov_minus_identity_all_rows": cov_err,
                   "cov_note": f"covariance of all {n} rows of the loop's result in f64 (chunked torch matmul), stated bound {'2e-3' if d > 256 else '1e-3'}",
                   "vs_oracle_record": vs_record},
    }
    gr.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", choices=sorted(CONFIGS),
                    help="BASELINE.json workload: C3 (default; the one the metric is quoted on), C2, C4s (papers100M-size stand-in "
                         "on one GPU), C5 (hypergraph through the host builder, d = 1024, whitened)")
    ap.add_argument("--nodes", type=int, default=0, help="override the config's node count")
    ap.add_argument("--pairs", type=int, default=0, help="override the config's pair count")
    ap.add_argument("--hyperedges", type=int, default=0, help="C5: override the number of hyperedges")
    ap.add_argument("--products", type=int, default=0, help="C5: override the number of products")
    ap.add_argument("--dim", type=int, default=0, help="override the config's embedding width")
    ap.add_argument("--no-selftest", action="store_true",
                    help="N > 1: skip the one-iteration comparison of every partition with a single-rank result on a small graph")
    ap.add_argument("--overlap-steps", type=int, default=0,
                    help="row blocks per rank per iteration (exchange of block k overlaps SpMM of k+1); "
                         "0 = 1 on one GPU, 4 otherwise")
    ap.add_argument("--partition", default="both", choices=["both", "row", "column"],
                    help="multi-GPU partition(s) to measure: 'row' = row blocks + in-place all-gather of X (north_star's "
                         "layout); 'column' = each rank owns d/N columns of X (one all-reduce of n floats per "
                         "iteration); both (default) measures the two: `value` is the row partition's, the column partition is reported in `partitions`")
    ap.add_argument("--balance", default="auto", choices=["auto", "rows", "nnz"],
                    help="row partition: equal row counts, or balanced on the rowptr prefix sum")
    ap.add_argument("--no-placement", action="store_true",
                    help="plain allocations for the iterates instead of cleora_alloc_iterates (DESIGN.md §2.1)")
    ap.add_argument("--whiten-iters", type=int, default=8, help="iterations of the whitened default loop (N = 1); 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the wall clock of the drop-in calls (embed_fast / embed, 40 iterations, host in / host out)")
    ap.add_argument("--watchdog", type=int, default=1500, help="seconds before every thread's traceback is dumped and the run exits")
    ap.add_argument("--backend", default="rccl", choices=["rccl", "local", "gloo"],
                    help="N > 1: transport of the C-ABI communicator — rccl (RCCL over xGMI, csrc/comm.hip; the peer-direct all-gather is timed "
                         "beside its two algorithms) or local (the peer-direct hipIpc transport alone, csrc/peer.hip; also what --share-gpu "
                         "needs: RCCL refuses two ranks on one device).  'gloo' is accepted as an alias of 'local' (round 3's developer mode)")
    # developer switch for exercising the N > 1 code path on a ONE-GPU box: every rank on cuda:0; numbers mean nothing
    ap.add_argument("--share-gpu", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started the way the driver starts the one-GPU run (`python bench.py --gpus N ...`, no torchrun): launch the ranks here
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # a run takes 1-3 minutes; if a collective ever deadlocks, leave a traceback of every thread and exit
    import faulthandler
    faulthandler.dump_traceback_later(args.watchdog, exit=True)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: either leave WORLD_SIZE unset (bench.py launches its own ranks) "
                         f"or start exactly --gpus ranks with torch.distributed.run")
    if args.launch_check:
        return launch_check(rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; cleora_amd has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    L = _hip.lib()
    launcher = Launcher(world)
    collectives = None
    if world > 1:
        with native_stdout_to_stderr():
            dist.init_process_group("gloo")       # the launcher; the data path's collectives are the C ABI's
            transport = "local" if (args.backend in ("local", "gloo") or args.share_gpu) else "rccl"
            comm, collectives = c_abi_communicator(local_rank, dev, rank, world, transport)
            comm.peer_enabled = False
            if not comm.local:
                # the peer-direct all-gather as a third algorithm beside RCCL's two (collective; every rank must succeed)
                err = ""
                try:
                    comm.enable_peer()
                except Exception as e:            # noqa: BLE001
                    err = f"{type(e).__name__}: {e}"
                ok = torch.tensor([0 if err else 1], dtype=torch.int32)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                comm.peer_enabled = int(ok) == 1
                collectives += "; peer-direct all-gather available" if comm.peer_enabled else f"; peer-direct transport not available ({err[:160] or 'another rank failed'})"
    else:
        comm = comm_mod.LocalComm()

    backend = sharded.HipBackend(dev)
    selftest = None
    if world > 1 and not args.no_selftest:
        # a broken collective or partition must cost seconds, not the run: one iteration of every partition on a small
        # graph against the same iteration computed by this rank alone, BEFORE the big graph is built
        selftest = partition_selftest(dev, rank, world, comm, backend, L, column=args.partition != "row")
    g, hashes, workload_label, cfg = make_workload(args, dev, rank, world, args.share_gpu)
    torch.cuda.empty_cache()                  # the iterates come from hipMalloc (cleora_alloc_iterates), not from torch's cache
    args.dim = args.dim or cfg["dim"]
    d = args.dim
    n, nnz = g["n"], g["nnz"]
    deg = torch.diff(g["rowptr"])
    if world == 1:
        parts = ["row"]
    else:
        parts = ["column", "row"] if args.partition == "both" else [args.partition]
        if d % (4 * world) != 0 and "column" in parts:
            parts.remove("column")
            parts = parts or ["row"]

    results, keep, whitened_sharded, cpu = {}, None, None, None
    for part in parts:
        res, a, b, iterate, blocks, sg = run_partition(part, args, g, hashes, deg, dev, rank, world, comm, launcher,
                                                       backend, L)
        results[part] = res
        if world == 1:
            keep = (a, b, iterate, blocks, sg)
        else:
            if part == "row" and args.whiten_iters > 0 and sg.embed_bytes(d, _hip.F_WHITEN) + 3 * sg.n_pad * d * 4 < torch.cuda.get_device_properties(dev).total_memory * 0.9:
                # the DEFAULT loop over the partition, one call: cleora_embed_sharded + CLEORA_F_WHITEN (wall clock incl. the final PCA whitening).
                # An extra beside the headline: a failure here is reported in the line, it does not cost the measurement above.
                unregister_replicas(comm, a, b)
                err = ""
                try:
                    _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, a.data_ptr(), d, torch.cuda.current_stream().cuda_stream))
                    torch.cuda.synchronize()
                    sg.embed(a, _hip.LEFT, d, 2, 0.0, 0.0, _hip.F_WHITEN)       # untimed: the first eigensolver call of a process
                    _hip.check(L.cleora_init_dev(hashes.data_ptr(), n, d, 0, a.data_ptr(), d, torch.cuda.current_stream().cuda_stream))
                    torch.cuda.synchronize()
                    launcher.barrier()
                    t0 = time.perf_counter()
                    sg.embed(a, _hip.LEFT, d, args.whiten_iters, 0.0, 0.0, _hip.F_WHITEN)
                    el = time.perf_counter() - t0
                except Exception as e:                                          # noqa: BLE001
                    err, el = f"{type(e).__name__}: {e}"[:300], 0.0
                el = launcher.max(el)
                if launcher.max(1.0 if err else 0.0) > 0:
                    whitened_sharded = {"error": err or "another rank failed"}
                else:
                    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
                    import make_whitened_loop_record as wrec
                    cov = wrec.covariance_all_rows(a, n)
                    whitened_sharded = {"ms_per_iter": el / args.whiten_iters * 1e3, "iterations": args.whiten_iters,
                                        "loop": "cleora_embed_sharded + CLEORA_F_WHITEN (csrc/sharded.hip: two replicas + the rank's own rows; statistics "
                                                "all-reduced, one all-gather of the iterate per iteration); wall clock of the call incl. its allocations, "
                                                "the registration of the replicas and the final PCA whitening; its schedule (statistics, then Z = A Y beside the d x d step, "
                                                "projection and gather block by block) was never tuned on multi-GPU hardware",
                                        "device_bytes_beside_the_callers_replica": sg.embed_bytes(d, _hip.F_WHITEN),
                                        "max_abs_cov_minus_identity_all_rows": float((cov - torch.eye(d, device=dev, dtype=torch.float64)).abs().max())}
            elif part == "row":
                unregister_replicas(comm, a, b)
            if part == "row" and rank == 0 and not args.no_cpu_baseline and cpu is None:
                # N > 1: the CPU baseline of the (d) rule beside the line — rank 0's host cores on a bounded row block of the same graph
                # and iterate (the N = 1 protocol for graphs too big for a host-side iteration); the other ranks wait at the barrier below
                try:
                    cpu = cpu_baseline_row_block(g, a, n, d, rows=min(500_000, n // 2), budget_s=12.0)
                    cpu["note"] = "measured by rank 0 while the other ranks wait (N > 1): a row block of the same graph, all host cores"
                except Exception as ex:                                     # noqa: BLE001
                    cpu = {"error": f"{type(ex).__name__}: {ex}"[:300]}
            if sg is not None:
                sg.close()
            del a, b, iterate, blocks, sg
            torch.cuda.empty_cache()
    # `value` is north_star's layout — row partition + all-gather of X — for every N; the column partition is a measured
    # comparison beside it (`partitions`), never the headline
    best = "row" if "row" in results else next(iter(results))
    r = results[best]

    whitened = e2e = None
    if world == 1:
        a, b, iterate, blocks, sg = keep
        if not args.no_cpu_baseline:
            iterate(a, b)                                     # the GPU iteration the oracle is compared with
            torch.cuda.synchronize()
            if n * d * 4 > (48 << 30):
                # the host-side oracle would need two n x d f32 arrays (C4s: 2 x 114 GB): sampled rows in the reference's order
                # instead, and the CPU baseline on a bounded row block of the same graph
                r["checks"].update(sampled_row_check(g, a, b, n, d, blocks[0].info().hub_threshold))
                cpu = cpu_baseline_row_block(g, a, n, d)
            else:
                def gpu_more(k, a=a, b=b):
                    p, q = b, a
                    for _ in range(k):
                        iterate(p, q)
                        p, q = q, p
                    torch.cuda.synchronize()
                    return p
                cpu, checks = cpu_baseline_and_checks(g, a, b, n, d, blocks[0].info().hub_threshold, gpu_more)
                r["checks"].update(checks)
                r["checks"]["plain_loop_vs_oracle_record"] = golden_loop_check(args.config, g, hashes, a, b, iterate, n, d, L)
        x_w = a
        sg.close()                                            # the rank's copy of the CSR
        del b, iterate, blocks, keep, sg
        torch.cuda.empty_cache()
        if args.whiten_iters > 0 and n * d * 4 * 4 < torch.cuda.get_device_properties(dev).total_memory * 0.8:
            whitened = run_whitened(args, g, x_w, dev, L, args.whiten_iters, hashes)
        elif args.whiten_iters > 0:
            whitened = {"skipped": "the whitened loop keeps three iterates and a workspace resident: does not fit one GPU at this size"}
        if not args.no_end_to_end and n * d * 4 * 4 < torch.cuda.get_device_properties(dev).total_memory * 0.7 and n * d * 4 < (48 << 30):
            del x_w
            torch.cuda.empty_cache()
            e2e = end_to_end(g, hashes, n, d, 40, L)

    if rank == 0:
        # PMC traffic of the dominant kernel: a committed measurement, valid only for the kernel build it was taken on
        traffic, tnote = None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if world == 1 and os.path.exists(tpath):
            try:
                rec = json.load(open(tpath))
                if (rec.get("n"), rec.get("nnz"), rec.get("d")) != (n, nnz, d):
                    tnote = "profiles/hbm_traffic.json is for another workload"
                elif rec.get("kernel_source_sha16") != kernel_source_stamp():
                    tnote = "profiles/hbm_traffic.json was measured on another build of the kernel (stale): re-profile"
                else:
                    traffic = rec.get("bytes_per_launch")
                    tnote = (f"rocprofv3 PMC of the same kernel build, committed in profiles/{rec.get('source', 'r02_pmc.json')} — NOT measured in this run "
                             f"(counters perturb the timing; the stamp in profiles/hbm_traffic.json ties it to these kernel sources)")
            except Exception as ex:
                tnote = f"unreadable profiles/hbm_traffic.json: {ex}"
        r["roofline"]["traffic"] = traffic
        r["roofline"]["traffic_note"] = tnote
        # the same launch on the FIRST (plain) allocation pair, before the library's placement search
        pt = r.get("placement_tuning") or {}
        if pt.get("untuned_launch_ms"):
            r["roofline"]["frac_untuned"] = r["roofline"]["algorithmic_bytes_per_launch"] / (pt["untuned_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
            r["roofline"]["frac_untuned_note"] = "median of three launches on the first allocation pair cleora_alloc_iterates drew (what a plain hipMalloc pair runs at)"
        # end-to-end figures of the GPU suite at config 2's size (tests/test_gpu_parity_at_scale.py), quoted from the committed record;
        # a missing file or key is an error in the line, not an empty object (round 4 shipped one)
        ppath = os.path.join(ROOT, "profiles", "r06_parity_at_scale.json")
        try:
            pj = json.load(open(ppath))
            plain, dl = pj["plain_loop_c2"], pj["default_loop_40_iterations_c2"]
            r["checks"]["drift_at_scale"] = {
                "source": "profiles/r06_parity_at_scale.json (tests/test_gpu_parity_at_scale.py at BASELINE config 2's size; NOT measured in this run — "
                          "this run's own multi-iteration checks are rows_bit_equal_after_those_iterations and plain_loop_vs_oracle_record)",
                "plain_loop_bit_equal_to_oracle_through_iteration": plain["bit_equal_through_iteration"],
                "plain_loop_with_CLEORA_F_HUB_SEGMENTS_max_abs_diff_after_10": plain["hub_segments_flag_max_abs_diff_after_10"],
                "default_loop_max_abs_cosine_diff_vs_reference_order": {k: v["max_abs_cosine_diff_2000_rows"] for k, v in dl["vs_reference_order_on_gpu"].items()},
                "default_loop_max_abs_cosine_diff_vs_oracle_loop": {"iterations": dl["vs_oracle_loop"]["iterations"], "value": dl["vs_oracle_loop"]["max_abs_cosine_diff_2000_rows"]},
                "default_loop_stated_tolerance": 1e-4}
        except Exception as ex:                               # noqa: BLE001
            r["checks"]["drift_at_scale"] = {"error": f"profiles/r06_parity_at_scale.json missing or incomplete: {type(ex).__name__}: {ex}"}
        out = {
            "metric": METRIC if args.config == "C3" else f"propagate iterations/sec & edges·dim/sec, BASELINE config {args.config}", "value": r["value"], "unit": "edge*dim/s",
            "iterations_per_sec": r["iterations_per_sec"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{workload_label}: n={n}, nnz={nnz}, d={d}, left Markov; one step = SpMM + fused L2 norm"
                                   + ("" if world == 1 else " + the partition's exchange step"),
                       "name": args.config, "n": n, "nnz": nnz, "d": d, "parallelism": r["parallelism"], "partition": best,
                       "seed": 2, "collectives": collectives},
            "roofline": r["roofline"], "checks": r["checks"], "placement_tuning": r["placement_tuning"],
            "cpu_baseline": cpu,
        }
        if world > 1:
            out["partitions"] = {p: {k: v for k, v in results[p].items() if k not in ("placement_tuning",)} for p in results}
            out["selftest"] = selftest
            for k in ("allgather", "allgather_ms_per_iter_tried", "per_iteration_ms", "ceiling"):
                if k in r:
                    out["config"][k] = r[k]
            out["config"]["ranks"] = world
            out["expected"] = ("row partition: every rank must RECEIVE (P-1)/P of the n x d iterate per iteration over its xGMI links (config.ceiling), so P = 2 — one link, "
                               "half the iterate — is expected at or BELOW one GPU's rate at this size, P = 4 about level with it, P = 8 at 2-3.9x (DESIGN.md 6); "
                               "the column partition beside it (partitions.column) has no such term")
            if whitened_sharded is not None:
                out["whitened_sharded"] = whitened_sharded
        if whitened is not None:
            out["whitened"] = whitened
        if e2e is not None:
            out["end_to_end"] = e2e
        print(json.dumps(out), flush=True)
    comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()
