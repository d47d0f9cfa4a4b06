"""Row-partitioned Markov propagation: one process per GPU, RCCL all-gather over xGMI.

The reference is single-process (rayon over rows, src/embedding.rs:59-63); this is the
multi-GPU design BASELINE.json:north_star asks for, not a port of anything:

  * output rows are independent, each needs arbitrary rows of the PREVIOUS iterate, so every
    rank keeps a full replica of X (n x d f32: 10 GB at |V| = 10M, d = 256 — small next to
    288 GB of HBM) and owns a set of row blocks of the CSR and of the next iterate;
  * rows are dealt out BLOCK-CYCLICALLY: with P ranks and K steps per iteration the row space is
    cut into P*K contiguous blocks (equal row counts, or balanced on the rowptr prefix sum when the
    ids are ordered by degree: `row_bounds`) and rank r owns blocks {k*P + r}.  Step k of an
    iteration computes block (k, r) on every rank and then all-gathers exactly the contiguous
    row range of blocks [k*P, (k+1)*P) of the next replica, IN PLACE (the rank's own slot of the
    output is the input: cleora_allgatherv_f32_dev) — no staging copies, natural row order;
  * the collective of step k runs on the communicator's stream while the SpMM of step k+1
    runs on the compute stream, so only the last step's gather is exposed (the all-gather, not
    the SpMM, is the critical path at 8 GPUs: SURVEY.md §8e);
  * the L2 normalisation is row-local and fused into the SpMM epilogue, so what travels over
    xGMI is the finished next iterate.

The PRODUCT path is `DeviceShardedGraph`: a thin ctypes wrapper over the row-partitioned loops of libcleora_hip.so
(csrc/sharded.hip: cleora_sharded_create / cleora_sharded_propagate_dev / cleora_embed_sharded) — block schedule, stream
ordering, statistics all-reduce and the Cholesky / PCA switch all live behind the C ABI, where a Rust host finds them too
(INTEGRATION.md); it needs no torch.

`ColumnShardedGraph` below is the column (dimension) partition that bench.py measures beside north_star's row partition: every
rank owns d/P columns of X and the whole CSR; arithmetic (`backend`: HipBackend = the C ABI's kernels) and exchange steps (`comm`,
cleora_amd/comm.py) are injected, so the CPU suite runs it over gloo with a numpy backend.
(The row partition's Python MODEL — the executable specification the CPU suite holds against the oracle and, for the plan, against
cleora_sharded_plan — lives with the tests: tests/sharded_model.py.)
"""
import ctypes

import numpy as np

from . import _hip
from . import comm as comm_mod

class _LazyTorch:
    """The model classes below work on torch tensors; the C-ABI wrappers (DeviceShardedGraph, DeviceColShardedGraph, plan_rows) do
    not — and a process that only needs those (every rank of the multi-process C-ABI tests) should not pay for `import torch`
    (seconds warm, a minute or two on a freshly provisioned box).  Imported on first use."""

    def __getattr__(self, name):
        import torch as _torch
        globals()["torch"] = _torch
        return getattr(_torch, name)


torch = _LazyTorch()


class _BorrowedGraph(_hip.Graph):
    """A block of a DeviceShardedGraph: the sharded handle owns it."""

    def close(self):
        self.handle = None


def _dev_ptr(a):
    """Device pointer of a torch tensor / DevArray (None passes through)."""
    if a is None:
        return None
    return a.data_ptr() if hasattr(a, "data_ptr") else a.ptr


class DeviceShardedGraph:
    """This rank's row blocks of a CSR graph on its GPU and the loops over the partition, through the C ABI (csrc/sharded.hip).

    rowptr / col / val_left / val_sym: the WHOLE graph (identical on every rank) as numpy host arrays, or as device arrays
    (torch tensors / _hip.DevArray) — only the rank's slices are copied.  comm: an RcclComm (RCCL, or local=True for the
    peer-direct transport) or None for a world of one.  Replicas are (n_pad, d) f32, rows >= n zero."""

    def __init__(self, n, rowptr, col, val_left, val_sym=None, comm=None, steps=1, balance="auto", device=0):
        L = self.L = _hip.lib()
        self.comm = comm
        mode = {"auto": _hip.BALANCE_AUTO, "rows": _hip.BALANCE_ROWS, "nnz": _hip.BALANCE_NNZ}.get(balance)
        if mode is None:
            raise ValueError("balance must be 'auto', 'rows' or 'nnz'")
        on_device = not isinstance(rowptr, np.ndarray)
        if on_device:
            args = [_dev_ptr(rowptr), _dev_ptr(col), _dev_ptr(val_left), _dev_ptr(val_sym)]
            nnz = int(col.numel() if hasattr(col, "numel") else col.shape[0])
        else:
            rowptr = np.ascontiguousarray(rowptr, dtype=np.uint64)
            col = np.ascontiguousarray(col, dtype=np.uint32)
            val_left = np.ascontiguousarray(val_left, dtype=np.float32)
            val_sym = None if val_sym is None else np.ascontiguousarray(val_sym, dtype=np.float32)
            args = [_hip.ptr(rowptr), _hip.ptr(col), _hip.ptr(val_left), _hip.ptr(val_sym)]
            nnz = int(col.shape[0])
        h = _hip.vp()
        _hip.check(L.cleora_sharded_create(comm.handle if comm is not None else None, int(device), int(n), nnz, *args,
                                           1 if on_device else 0, int(steps), mode, ctypes.byref(h)))
        self.handle = h
        info = _hip.ShardedInfo()
        _hip.check(L.cleora_sharded_get_info(h, ctypes.byref(info)))
        self.n, self.n_pad, self.local_rows, self.local_nnz = info.n, info.n_pad, info.local_rows, info.local_nnz
        self.steps, self.rank, self.world = info.steps, info.rank, info.world
        self.balance = {_hip.BALANCE_ROWS: "rows", _hip.BALANCE_NNZ: "nnz"}[info.balance]
        b = np.zeros(self.world * self.steps + 1, dtype=np.uint64)
        _hip.check(L.cleora_sharded_bounds(h, _hip.ptr(b)))
        self.bounds = [int(v) for v in b]
        self.my_rows = [(self.bounds[k * self.world + self.rank], self.bounds[k * self.world + self.rank + 1]) for k in range(self.steps)]

    def block(self, k):
        """(block k as a _hip.Graph that does NOT own its handle — info, timing, cleora_alloc_iterates —, row_begin, row_end)."""
        g, b0, b1 = _hip.vp(), _hip.c_u64(0), _hip.c_u64(0)
        _hip.check(self.L.cleora_sharded_block(self.handle, int(k), ctypes.byref(g), ctypes.byref(b0), ctypes.byref(b1)))
        return _BorrowedGraph(g, self), b0.value, b1.value

    def propagate(self, kind, x, x_next, flags=_hip.F_L2NORM, rw=0.0, row_sqdiff=None, gather=True, stream=None, d=None):
        """One iteration; x / x_next: (n_pad, d) replicas (torch tensors or DevArrays)."""
        d = int(d if d is not None else x.shape[1])
        if stream is None and torch is not None and hasattr(x, "data_ptr"):
            stream = torch.cuda.current_stream(x.device).cuda_stream
        _hip.check(self.L.cleora_sharded_propagate_dev(self.handle, int(kind), _dev_ptr(x), _dev_ptr(x_next), d, int(flags), float(rw),
                                                       _dev_ptr(row_sqdiff), 1 if gather else 0, _hip.vp(stream) if stream else None))

    def embed(self, x, kind, d, iterations, residual_weight=0.0, convergence_threshold=0.0, flags=_hip.F_L2NORM):
        """The loops of cleora_embed_sharded on the replica x (E_0 in, result out).  Returns the iterations run."""
        ran = _hip.c_u64(0)
        _hip.check(self.L.cleora_embed_sharded(self.handle, _dev_ptr(x), int(kind), int(d), int(iterations), float(residual_weight),
                                               float(convergence_threshold), int(flags), ctypes.byref(ran)))
        return ran.value

    def embed_bytes(self, d, flags=0):
        return int(self.L.cleora_embed_sharded_bytes(self.n_pad, self.local_rows, self.n, self.world, int(d), int(flags)))

    def set_timing(self, enable):
        _hip.check(self.L.cleora_sharded_set_timing(self.handle, 1 if enable else 0))

    def get_timing(self):
        """((spmm ms, all-gather ms) summed, propagate calls) since the last query."""
        ms, calls = (ctypes.c_double * 2)(), _hip.c_u64(0)
        _hip.check(self.L.cleora_sharded_get_timing(self.handle, ctypes.byref(ms), ctypes.byref(calls)))
        return (ms[0], ms[1]), calls.value

    def close(self):
        if getattr(self, "handle", None):
            self.L.cleora_sharded_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plan_rows(n, rowptr, world, steps, balance="auto"):
    """cleora_sharded_plan (pure host arithmetic, no GPU): (bounds list, n_pad, 'rows' | 'nnz')."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.uint64)
    mode = {"auto": _hip.BALANCE_AUTO, "rows": _hip.BALANCE_ROWS, "nnz": _hip.BALANCE_NNZ}[balance]
    b = np.zeros(world * steps + 1, dtype=np.uint64)
    n_pad, got = _hip.c_u64(0), _hip.c_int(0)
    _hip.check(_hip.lib().cleora_sharded_plan(int(n), _hip.ptr(rowptr), int(world), int(steps), mode, _hip.ptr(b), ctypes.byref(n_pad),
                                              ctypes.byref(got)))
    return [int(v) for v in b], n_pad.value, {_hip.BALANCE_ROWS: "rows", _hip.BALANCE_NNZ: "nnz"}[got.value]


class DeviceColShardedGraph:
    """This rank's COLUMN slice of the propagation through the C ABI (csrc/colsharded.hip): the rank owns columns
    [rank d/P, (rank + 1) d/P) of the iterate and the whole CSR; the row L2 norm travels from rank to rank as a running sum
    (CLEORA_F_ROWSQ_CONT), so propagate and the plain loop are bit-equal to the one-GPU calls.  rowptr / col / val_*: the WHOLE graph
    as numpy host arrays (copied) or as device arrays (torch tensors / _hip.DevArray: col / val_* are VIEWED and must outlive the handle)."""

    def __init__(self, n, rowptr, col, val_left, val_sym, d_total, comm=None, steps=1, device=0):
        L = self.L = _hip.lib()
        self.comm = comm
        on_device = not isinstance(rowptr, np.ndarray)
        if on_device:
            self._keep = (rowptr, col, val_left, val_sym)
            args = [_dev_ptr(rowptr), _dev_ptr(col), _dev_ptr(val_left), _dev_ptr(val_sym)]
            nnz = int(col.numel() if hasattr(col, "numel") else col.shape[0])
        else:
            rowptr = np.ascontiguousarray(rowptr, dtype=np.uint64)
            col = np.ascontiguousarray(col, dtype=np.uint32)
            val_left = np.ascontiguousarray(val_left, dtype=np.float32)
            val_sym = None if val_sym is None else np.ascontiguousarray(val_sym, dtype=np.float32)
            args = [_hip.ptr(rowptr), _hip.ptr(col), _hip.ptr(val_left), _hip.ptr(val_sym)]
            nnz = int(col.shape[0])
        h = _hip.vp()
        _hip.check(L.cleora_colsharded_create(comm.handle if comm is not None else None, int(device), int(n), nnz, *args,
                                              1 if on_device else 0, int(d_total), int(steps), ctypes.byref(h)))
        self.handle = h
        info = _hip.ColShardedInfo()
        _hip.check(L.cleora_colsharded_get_info(h, ctypes.byref(info)))
        self.n, self.nnz, self.d, self.dl, self.c0 = info.n, info.nnz, info.d_total, info.d_local, info.col_begin
        self.steps, self.rank, self.world = info.steps, info.rank, info.world
        self.blocks, self.row_blocks = [], []
        for k in range(self.steps):
            g, b0, b1 = _hip.vp(), _hip.c_u64(0), _hip.c_u64(0)
            _hip.check(L.cleora_colsharded_block(h, k, ctypes.byref(g), ctypes.byref(b0), ctypes.byref(b1)))
            self.blocks.append(_BorrowedGraph(g, self))
            self.row_blocks.append((b0.value, b1.value))

    def propagate(self, kind, x_local, x_next_local, flags=_hip.F_L2NORM, rw=0.0, row_sqdiff=None, stream=None):
        """One iteration on the (n, d/P) slices (torch tensors or DevArrays, contiguous)."""
        _hip.check(self.L.cleora_colsharded_propagate_dev(self.handle, int(kind), _dev_ptr(x_local), _dev_ptr(x_next_local), int(flags), float(rw),
                                                          _dev_ptr(row_sqdiff), stream))

    def set_timing(self, enable):
        """Kernel timing of the blocks (cleora_graph_set_timing); the hand-offs of the row sums are not timed separately."""
        for blk in self.blocks:
            blk.set_timing(enable)

    def get_timing(self):
        return (0.0, 0.0), 0            # (DeviceShardedGraph's shape: the blocks' own records hold the SpMM part)

    def embed(self, x_local, kind, iterations, residual_weight=0.0, convergence_threshold=0.0, flags=0):
        ran = _hip.c_u64(0)
        _hip.check(self.L.cleora_embed_colsharded(self.handle, _dev_ptr(x_local), int(kind), int(iterations), float(residual_weight),
                                                  float(convergence_threshold), int(flags), ctypes.byref(ran)))
        return ran.value

    def close(self):
        if self.handle:
            self.L.cleora_colsharded_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:            # noqa: BLE001
            pass


class HipBackend:
    """Per-block SpMM through the C ABI on the current torch stream."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.lib = _hip.lib()

    def make_block(self, rowptr, col, val_left, val_sym, n_cols, hub_threshold=0, hub_segment=0):
        keep = (rowptr, col, val_left, val_sym)
        return _hip.Graph.from_device(
            rowptr.numel() - 1, n_cols, col.numel(), rowptr.data_ptr(), col.data_ptr(),
            val_left.data_ptr(), val_sym.data_ptr() if val_sym is not None else None,
            self.device.index or 0, hub_threshold, hub_segment, keepalive=keep)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def propagate(self, block, kind, x, y, flags, rw=0.0, x_self=None, row_sqdiff=None, row_sumsq=None):
        d = x.shape[1]
        _hip.check(self.lib.cleora_propagate_dev(
            block.handle, kind, x.data_ptr(), x.stride(0), d, y.data_ptr(), y.stride(0), flags, rw,
            x_self.data_ptr() if x_self is not None else None,
            row_sqdiff.data_ptr() if row_sqdiff is not None else None,
            row_sumsq.data_ptr() if row_sumsq is not None else None, self._stream()))

    def rowops(self, x, y, flags, rw=0.0, x_self=None, row_sqdiff=None, row_sumsq=None):
        n, d = x.shape
        _hip.check(self.lib.cleora_rowops_dev(
            x.data_ptr(), x.stride(0), n, d, y.data_ptr(), y.stride(0), flags, rw,
            x_self.data_ptr() if x_self is not None else None,
            row_sqdiff.data_ptr() if row_sqdiff is not None else None,
            row_sumsq.data_ptr() if row_sumsq is not None else None, self._stream()))

    # whitening pieces on a contiguous row range (pycleora/__init__.py:136-163)
    def colsum(self, x):
        n, d = x.shape
        ws = torch.empty(self.lib.cleora_colsum_workspace(n, d), dtype=torch.float64, device=x.device)
        out = torch.empty(d, dtype=torch.float64, device=x.device)
        _hip.check(self.lib.cleora_colsum_dev(x.data_ptr(), x.stride(0), n, d, ws.data_ptr(),
                                              out.data_ptr(), self._stream()))
        return out

    def gram(self, x, mean):
        n, d = x.shape
        ws = torch.empty(self.lib.cleora_gram_workspace(n, d), dtype=torch.float64, device=x.device)
        out = torch.empty((d, d), dtype=torch.float64, device=x.device)
        _hip.check(self.lib.cleora_centered_gram_dev(x.data_ptr(), x.stride(0), n, d, mean.data_ptr(),
                                                     ws.data_ptr(), out.data_ptr(), self._stream()))
        return out

    def whiten_transform(self, gram, n, kdim):
        d = gram.shape[0]
        ws = torch.empty(self.lib.cleora_eigh_workspace(d), dtype=torch.uint8, device=gram.device)
        out = torch.empty((d, kdim), dtype=torch.float32, device=gram.device)
        _hip.check(self.lib.cleora_whiten_transform_dev(gram.data_ptr(), n, d, kdim, out.data_ptr(), None,
                                                        ws.data_ptr(), self._stream()))
        return out

    def project(self, x, mean32, transform, out):
        n, d = x.shape
        _hip.check(self.lib.cleora_project_dev(x.data_ptr(), x.stride(0), n, d, mean32.data_ptr(),
                                               transform.data_ptr(), transform.shape[1], out.data_ptr(),
                                               out.stride(0), self._stream()))

    # pieces of the reorganised whitened loop (csrc/abi.hip embed_whitened_overlapped) on a row range
    def rowsum(self, block, kind, out):
        """out[r] = sum of the stored values of row r of the block: s = A 1."""
        _hip.check(self.lib.cleora_csr_rowsum_dev(block.handle, kind, out.data_ptr(), self._stream()))

    def whiten_transform_any(self, gram, n):
        """(transform d x d, form): the Cholesky whitening when the reference's eigenvalue clamp is provably inactive
        (form 1), else the PCA form (form 0).  Waits for the stream (cleora_whiten_transform_any_dev)."""
        import ctypes
        d = gram.shape[0]
        ws = torch.empty(self.lib.cleora_eigh_workspace(d), dtype=torch.uint8, device=gram.device)
        out = torch.empty((d, d), dtype=torch.float32, device=gram.device)
        form = ctypes.c_int(0)
        _hip.check(self.lib.cleora_whiten_transform_any_dev(gram.data_ptr(), n, d, out.data_ptr(), ws.data_ptr(),
                                                            self._stream(), ctypes.byref(form)))
        return out, form.value

    def project_general(self, x, mean32, transform, out, rowscale=None, x2=None, alpha=1.0, beta=0.0, norm=0):
        """out = normalise((alpha (x - rowscale mean) + beta (x2 - mean)) @ transform); returns whether the kernel
        normalised the rows itself (else the caller runs rowops)."""
        import ctypes
        n, d = x.shape
        done = ctypes.c_int(0)
        _hip.check(self.lib.cleora_project_general_dev(
            x.data_ptr(), x.stride(0), n, d, mean32.data_ptr(), transform.data_ptr(), transform.shape[1], out.data_ptr(),
            out.stride(0), rowscale.data_ptr() if rowscale is not None else None,
            x2.data_ptr() if x2 is not None else None, x2.stride(0) if x2 is not None else 0, alpha, beta, norm,
            ctypes.byref(done), self._stream()))
        return bool(done.value)


class ColumnShardedGraph:
    """Column (dimension) partition: every rank holds the WHOLE CSR and d/P columns of every
    embedding row.

    "Cleora operates on dimensions independently" (reference README.md:361): the SpMM of a column
    slice needs no data from other slices, so nothing of the n x d iterate ever crosses xGMI.  Only
    the L2 norm couples the columns: each rank computes its part of every row's sum of squares in the
    SpMM epilogue (CLEORA_F_ROWSQ), one all-reduce of n floats (40 MB at |V| = 10M, against the
    10 GB all-gather of the row partition) completes them, and a row-scale pass (CLEORA_F_SCALE)
    finishes the iteration.  Each element of A @ X is still the reference's in-order f32 sum; the
    row norm is a sum of per-slice partial sums (last-ulp differences from the single-GPU result).

    Cost per rank and iteration at P ranks: gathers of nnz * (d/P) * 4 B (rows of 1024/P bytes),
    the full col/val streams (nnz * 8 B), n * (d/P) * 4 * 3 B for Y and the scale pass.
    Measured on one MI355X at the C3 graph (what a rank of an 8/4/2-way run executes):
    d/P = 32 -> 5.2 ms, 64 -> 10.0 ms, 128 -> 19.4 ms, against 35.6 ms for d = 256.
    """

    def __init__(self, n, rowptr, col, val_left, val_sym, d, rank, world, backend,
                 hub_threshold=0, hub_segment=0, comm=None, steps=1, group=None):
        if d % world != 0:
            raise ValueError(f"feature_dim {d} must be divisible by the number of ranks {world}")
        self.n, self.d, self.rank, self.world = n, d, rank, world
        self.dl = d // world
        self.c0 = rank * self.dl
        self.backend = backend
        self.comm = comm if comm is not None else comm_mod.default_comm(group)
        self.nnz = int(col.numel())
        # `steps` row blocks per iteration: the all-reduce of block k's row sums overlaps the SpMM
        # of block k+1 (the blocks are zero-copy views of the one CSR every rank holds)
        self.steps = max(1, min(int(steps), max(1, n)))
        rp64 = rowptr.to(torch.int64)
        bounds = [n * k // self.steps for k in range(self.steps + 1)]
        self.row_blocks, self.blocks = [], []
        for k in range(self.steps):
            r0, r1 = bounds[k], bounds[k + 1]
            e0, e1 = int(rp64[r0]), int(rp64[r1])
            brp = (rp64[r0:r1 + 1] - e0).contiguous()
            self.blocks.append(backend.make_block(brp, col[e0:e1], val_left[e0:e1],
                                                  val_sym[e0:e1] if val_sym is not None else None, n,
                                                  hub_threshold, hub_segment))
            self.row_blocks.append((r0, r1))
        self.block = self.blocks[0]
        # row chunks used when the whitening step switches to a row layout (equal splits, padded)
        self.rows_per = -(-n // world)
        self.n_pad = self.rows_per * world

    def propagate(self, kind, x, x_next, rowsq, flags=_hip.F_L2NORM, rw=0.0, row_sqdiff=None, exact_norm=False):
        """x, x_next: (n, d/P) column slices; rowsq: f32[n] scratch.  One iteration.
        exact_norm: the rows' sums of squares travel from rank to rank as running sums (CLEORA_F_ROWSQ_CONT; one broadcast per rank and
        row block) instead of being all-reduced: the reference's summation order, bit-equal to one GPU — what csrc/colsharded.hip runs."""
        if (flags & _hip.F_L1NORM) and self.world > 1:
            # each rank would divide its slice by its PARTIAL sum of |.|: wrong.  (The L2 norm has the ROWSQ / SCALE pair.)
            raise ValueError("the column partition supports the L2 normalisation only: use the row partition for 'l1'")
        norm = flags & _hip.F_L2NORM
        if self.world == 1:   # nothing to reduce: the fused single-pass epilogue
            for blk, (r0, r1) in zip(self.blocks, self.row_blocks):
                self.backend.propagate(blk, kind, x, x_next[r0:r1], flags, rw, x[r0:r1],
                                       row_sqdiff[r0:r1] if row_sqdiff is not None else None)
            return
        if exact_norm and norm:
            first = flags & ~(_hip.F_L2NORM | _hip.F_SQDIFF)
            for blk, (r0, r1) in zip(self.blocks, self.row_blocks):
                self.backend.propagate(blk, kind, x, x_next[r0:r1], first, rw, x[r0:r1], None, None)
                for p in range(self.world):
                    if p == self.rank:
                        self.backend.rowops(x_next[r0:r1], x_next[r0:r1], _hip.F_ROWSQ | (_hip.F_ROWSQ_CONT if p else 0), 0.0, None, None, rowsq[r0:r1])
                    self.comm.broadcast(rowsq[r0:r1], p)
            self.comm.join()
            second = _hip.F_SCALE | (flags & _hip.F_SQDIFF)
            for r0, r1 in self.row_blocks:
                self.backend.rowops(x_next[r0:r1], x_next[r0:r1], second, 0.0, x[r0:r1] if (flags & _hip.F_SQDIFF) else None,
                                    row_sqdiff[r0:r1] if row_sqdiff is not None else None, rowsq[r0:r1])
            return
        first = (flags & ~(_hip.F_L2NORM | _hip.F_SQDIFF)) | (_hip.F_ROWSQ if norm else 0)
        for blk, (r0, r1) in zip(self.blocks, self.row_blocks):
            self.backend.propagate(blk, kind, x, x_next[r0:r1], first, rw, x[r0:r1], None,
                                   rowsq[r0:r1] if norm else None)
            if norm:
                self.comm.allreduce_async(rowsq[r0:r1])       # beside the next block's SpMM
        self.comm.join()
        second = (_hip.F_SCALE if norm else 0) | (flags & _hip.F_SQDIFF)
        if second:
            for r0, r1 in self.row_blocks:
                self.backend.rowops(x_next[r0:r1], x_next[r0:r1], second, 0.0,
                                    x[r0:r1] if (flags & _hip.F_SQDIFF) else None,
                                    row_sqdiff[r0:r1] if row_sqdiff is not None else None,
                                    rowsq[r0:r1] if norm else None)

    def whiten(self, y_local, out_local, any_whitening=False):
        """whiten_embeddings (pycleora/__init__.py:130-164) for a column-partitioned matrix.
        any_whitening: an intermediate iteration of the L2-normalised loop — the Cholesky transform where the reference's
        eigenvalue clamp is provably inactive (cleora_whiten_transform_any_dev), the eigensolver only otherwise.
        The Gram matrix couples every pair of columns, so the step runs in a ROW layout:
        all-to-all (each rank receives all d columns of its n/P rows: (P-1)/P^2 of the matrix
        per rank, 8x less than an all-gather at P = 8) -> row-local f64 column sums / centred Gram,
        all-reduced (d and d*d doubles) -> eigh on the device, replicated -> row-local projection ->
        all-to-all back to columns.  y_local, out_local: (n_pad, d/P), rows >= n zero."""
        P, rp, dl, d = self.world, self.rows_per, self.dl, self.d
        if P > 1:
            recv = torch.empty((P, rp, dl), dtype=y_local.dtype, device=y_local.device)
            self.comm.alltoall(y_local[: self.n_pad].contiguous(), recv)
            rows = recv.permute(1, 0, 2).reshape(rp, d).contiguous()      # my rows, all columns
        else:
            rows = y_local[: self.n_pad]
        nv = max(0, min(rp, self.n - self.rank * rp))
        cs = self.backend.colsum(rows[:nv]) if nv else torch.zeros(d, dtype=torch.float64, device=rows.device)
        self.comm.allreduce(cs)
        mean = cs / float(self.n)
        gram = self.backend.gram(rows[:nv], mean) if nv else torch.zeros((d, d), dtype=torch.float64, device=rows.device)
        self.comm.allreduce(gram)
        # every rank holds the same all-reduced Gram: eigh is replicated (deterministic routine),
        # but the transform is still broadcast from rank 0 so the ranks cannot drift apart
        transform = self.backend.whiten_transform_any(gram, self.n)[0] if any_whitening else self.backend.whiten_transform(gram, self.n, d)
        self.comm.broadcast(transform, 0)
        proj = torch.zeros((rp, d), dtype=torch.float32, device=rows.device)
        if nv:
            self.backend.project(rows[:nv], mean.to(torch.float32), transform, proj[:nv])
        if P > 1:
            send = proj.view(rp, P, dl).permute(1, 0, 2).contiguous()       # column block j -> rank j
            if out_local.shape[0] == self.n_pad and out_local.is_contiguous():
                self.comm.alltoall(send, out_local)
            else:
                tmp = torch.empty((self.n_pad, dl), dtype=out_local.dtype, device=out_local.device)
                self.comm.alltoall(send, tmp)
                out_local[: self.n_pad].copy_(tmp)
        else:
            out_local[: self.n_pad].copy_(proj)

    def gather_columns(self, x_local):
        """(rows, d) on every rank from the (rows, d/P) slices (not part of the iteration)."""
        if self.world == 1:
            return x_local
        rows = x_local.shape[0]
        parts = torch.empty((self.world * rows, self.dl), dtype=x_local.dtype, device=x_local.device)
        parts[self.rank * rows:(self.rank + 1) * rows].copy_(x_local)
        self.comm.allgather_rows(parts, [r * rows for r in range(self.world + 1)])
        self.comm.join()
        return parts.view(self.world, rows, self.dl).permute(1, 0, 2).reshape(rows, self.d).contiguous()

    def sqdiff_total(self, row_sqdiff):
        t = row_sqdiff.sum(dtype=torch.float64).reshape(1)
        self.comm.allreduce(t)
        return float(t)


def embed_column_sharded(cg, kind, x0_local, iterations, residual_weight=0.0,
                         convergence_threshold=0.0, flags=_hip.F_L2NORM, whiten=False, exact_norm=False):
    """embed_full / embed_full_with_convergence over a ColumnShardedGraph; whiten=True runs the
    default embed() loop (normalise, then whiten, every iteration) and needs x0_local padded to
    cg.n_pad rows.  x0_local: this rank's (n or n_pad, d/P) columns of the initial matrix.
    Returns (x_local, iterations_run)."""
    x = x0_local
    x_next = torch.zeros_like(x0_local)
    rowsq = torch.zeros(cg.n, dtype=torch.float32, device=x.device)
    if whiten:
        if x0_local.shape[0] < cg.n_pad:
            raise ValueError(f"whiten=True needs {cg.n_pad} (padded) rows, got {x0_local.shape[0]}")
        y = torch.zeros_like(x0_local)
        for it in range(iterations):
            # the Python loop blends for ANY rw > 0 (pycleora/__init__.py:111-115); intermediate iterations of the
            # L2-normalised loop may take any whitening (Cholesky), the last one the reference's PCA form
            cg.propagate(kind, x[: cg.n], y[: cg.n], rowsq, flags | _hip.F_RESIDUAL | _hip.F_BLEND_ANY, residual_weight)
            cg.whiten(y, x_next, any_whitening=(flags == _hip.F_L2NORM and it + 1 < iterations and whiten != "sequential"))
            x, x_next = x_next, x
        return x, iterations
    check = convergence_threshold > 0
    flags = flags | _hip.F_RESIDUAL
    sq = torch.zeros(cg.n, dtype=torch.float64, device=x.device) if check else None
    ran = iterations
    total = float(cg.n) * cg.d
    for it in range(iterations):
        test = check and it > 0
        cg.propagate(kind, x, x_next, rowsq, flags | (_hip.F_SQDIFF if test else 0), residual_weight,
                     sq if test else None, exact_norm=exact_norm)
        x, x_next = x_next, x
        if test:
            rmse = (cg.sqdiff_total(sq) / total) ** 0.5
            if rmse < convergence_threshold:
                ran = it + 1
                break
    return x, ran
