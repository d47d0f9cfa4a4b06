"""Row-partitioned Markov propagation: one process per GPU, RCCL all-gather over xGMI.

The reference is single-process (rayon over rows, src/embedding.rs:59-63); this is the
multi-GPU design BASELINE.json:north_star asks for, not a port of anything:

  * output rows are independent, each needs arbitrary rows of the PREVIOUS iterate, so every
    rank keeps a full replica of X (n x d f32: 10 GB at |V| = 10M, d = 256 — small next to
    288 GB of HBM) and owns a set of row blocks of the CSR and of the next iterate;
  * rows are dealt out BLOCK-CYCLICALLY: with P ranks and K steps per iteration the padded row
    space is cut into P*K blocks of B rows and rank r owns blocks {k*P + r}.  Step k of an
    iteration computes block (k, r) on every rank and then all-gathers exactly the contiguous
    row range [k*P*B, (k+1)*P*B) of the next replica, IN PLACE (`all_gather_into_tensor` with
    the input being the rank's own slot of the output) — no staging copies, natural row order;
  * the collective of step k runs on the process group's stream while the SpMM of step k+1
    runs on the compute stream, so only the last step's gather is exposed (the all-gather, not
    the SpMM, is the critical path at 8 GPUs: SURVEY.md §8e);
  * the L2 normalisation is row-local and fused into the SpMM epilogue, so what travels over
    xGMI is the finished next iterate.

`backend` does the per-block arithmetic.  HipBackend is the product path (libcleora_hip.so);
tests inject a CPU backend so the partition / collective logic runs under gloo without a GPU.
"""
import torch
import torch.distributed as dist

from . import _hip


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


def transform_from_gram(backend, gram, n, kdim):
    """cov = gram/(n-1) -> eigh -> descending -> V / sqrt(max(lambda, 1e-10)) as f32, d x kdim
    (pycleora/__init__.py:143-156).  On the HIP backend this stays on the device
    (cleora_whiten_transform_dev: rocSOLVER dsyevd); injected CPU backends use numpy's LAPACK."""
    if hasattr(backend, "whiten_transform"):
        return backend.whiten_transform(gram, n, kdim)
    import numpy as np
    w, v = np.linalg.eigh(gram.cpu().numpy() * (1.0 / (n - 1)))
    idx = np.argsort(w)[::-1][:kdim]                                       # :147-153
    scale = 1.0 / np.sqrt(np.maximum(w[idx], 1e-10))                       # :155
    return torch.from_numpy(np.ascontiguousarray((v[:, idx] * scale).astype(np.float32))).to(gram.device)


def block_size(n, world, steps):
    """Rows per block: the padded row count is block * world * steps (block a multiple of 4 so
    every block of a 16-byte-aligned matrix stays 16-byte aligned for any d)."""
    b = -(-n // (world * steps))
    return max(4, -(-b // 4) * 4)


class ShardedGraph:
    """This rank's row blocks of a CSR graph plus the replica bookkeeping."""

    def __init__(self, n, rowptr, col, val_left, val_sym, rank, world, steps, backend,
                 hub_threshold=0, hub_segment=0, group=None):
        self.n, self.rank, self.world, self.steps = n, rank, world, steps
        self.backend, self.group = backend, group
        self.block = block_size(n, world, steps)
        self.n_pad = self.block * world * steps
        self.blocks = []
        self.local_nnz = 0
        rp = rowptr.to(torch.int64)
        for k in range(steps):
            r0 = min((k * world + rank) * self.block, n)
            r1 = min(r0 + self.block, n)
            e0, e1 = int(rp[r0]), int(rp[r1])
            brp = torch.full((self.block + 1,), e1 - e0, dtype=torch.int64, device=rp.device)
            brp[: r1 - r0 + 1] = rp[r0:r1 + 1] - e0     # padded rows are empty
            blk = backend.make_block(brp, col[e0:e1].clone(), val_left[e0:e1].clone(),
                                     val_sym[e0:e1].clone() if val_sym is not None else None,
                                     self.n_pad, hub_threshold, hub_segment)
            self.blocks.append(blk)
            self.local_nnz += e1 - e0

    def rows_of_step(self, k):
        """(first row of this rank's block, first row of the step's gathered range)."""
        g0 = k * self.world * self.block
        return g0 + self.rank * self.block, g0

    def _gather_step(self, buf, k):
        """Enqueue the in-place all-gather of step k's row range of `buf` (async)."""
        mine, g0 = self.rows_of_step(k)
        return dist.all_gather_into_tensor(buf[g0:g0 + self.world * self.block],
                                           buf[mine:mine + self.block], group=self.group, async_op=True)

    def propagate(self, kind, x, x_next, flags=_hip.F_L2NORM, rw=0.0, row_sqdiff=None, gather=True):
        """One iteration: x_next <- rowops(A @ x), replicated on every rank (gather=True) or only
        this rank's row blocks of x_next written (gather=False: whitening follows).
        x, x_next: (n_pad, d) f32 replicas.  Returns after the collectives are enqueued and
        waited on the current stream (no host sync)."""
        works = []
        for k in range(self.steps):
            mine, _ = self.rows_of_step(k)
            y = x_next[mine:mine + self.block]
            xs = x[mine:mine + self.block]
            sq = row_sqdiff[k * self.block:(k + 1) * self.block] if row_sqdiff is not None else None
            self.backend.propagate(self.blocks[k], kind, x, y, flags, rw, xs, sq)
            if self.world > 1 and gather:
                works.append(self._gather_step(x_next, k))
        for w in works:
            w.wait()

    def _valid_rows(self, k):
        mine, _ = self.rows_of_step(k)
        return mine, max(0, min(self.block, self.n - mine))

    def whiten(self, y, out, n_components=None):
        """whiten_embeddings (pycleora/__init__.py:130-164) over the row partition: `y` holds this
        rank's blocks of the matrix to whiten; `out` receives the whitened matrix, replicated.
        Local f64 column sums and centred Gram -> all-reduce (d and d*d doubles) -> transform
        (cleora_whiten_transform_dev, replicated; rank 0's copy is broadcast) -> row-local projection -> in-place all-gather per block."""
        import numpy as np
        d = y.shape[1]
        cs = torch.zeros(d, dtype=torch.float64, device=y.device)
        for k in range(self.steps):
            r0, nv = self._valid_rows(k)
            if nv:
                cs += self.backend.colsum(y[r0:r0 + nv])
        if self.world > 1:
            dist.all_reduce(cs, group=self.group)
        mean = cs / float(self.n)
        gram = torch.zeros((d, d), dtype=torch.float64, device=y.device)
        for k in range(self.steps):
            r0, nv = self._valid_rows(k)
            if nv:
                gram += self.backend.gram(y[r0:r0 + nv], mean)
        if self.world > 1:
            dist.all_reduce(gram, group=self.group)
        kdim = d if n_components is None else min(int(n_components), d)
        # every rank holds the same all-reduced Gram, so the (deterministic) eigensolver is replicated;
        # the transform is still broadcast from rank 0 so that the ranks cannot drift apart
        transform = transform_from_gram(self.backend, gram, self.n, kdim)
        if self.world > 1:
            dist.broadcast(transform, src=0, group=self.group)
        mean32 = mean.to(torch.float32)
        works = []
        for k in range(self.steps):
            r0, nv = self._valid_rows(k)
            if nv:
                self.backend.project(y[r0:r0 + nv], mean32, transform, out[r0:r0 + nv])
            if self.world > 1:
                works.append(self._gather_step(out, k))
        for w_ in works:
            w_.wait()

    def sqdiff_total(self, row_sqdiff):
        """Sum of the per-row squared differences over all ranks (f64)."""
        t = row_sqdiff.sum(dtype=torch.float64).reshape(1)
        if self.world > 1:
            dist.all_reduce(t, group=self.group)
        return float(t)


def embed_sharded(sg, kind, x0, iterations, residual_weight=0.0, convergence_threshold=0.0,
                  flags=_hip.F_L2NORM, whiten=False):
    """embed_full / embed_full_with_convergence (src/embedding.rs:106-188) over a ShardedGraph;
    with whiten=True the default embed() loop of pycleora/__init__.py:109-125 (normalise, then
    whiten, every iteration; no convergence test in that mode here).
    x0: (n_pad, d) replica (rows >= n zero).  Returns (x, iterations_run)."""
    x = x0
    x_next = torch.zeros_like(x0)
    if whiten:
        y = torch.zeros_like(x0)
        for _ in range(iterations):
            sg.propagate(kind, x, y, flags | _hip.F_RESIDUAL, residual_weight, gather=False)
            sg.whiten(y, x_next)
            x, x_next = x_next, x
        return x, iterations
    check = convergence_threshold > 0
    flags = flags | _hip.F_RESIDUAL
    sq = torch.zeros(sg.steps * sg.block, dtype=torch.float64, device=x0.device) if check else None
    ran = iterations
    total = float(sg.n) * x0.shape[1]
    for it in range(iterations):
        test = check and it > 0
        sg.propagate(kind, x, x_next, flags | (_hip.F_SQDIFF if test else 0), residual_weight,
                     sq if test else None)
        x, x_next = x_next, x
        if test:
            rmse = (sg.sqdiff_total(sq) / total) ** 0.5
            if rmse < convergence_threshold:
                ran = it + 1
                break
    return x, ran


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
                 hub_threshold=0, hub_segment=0, group=None, steps=1):
        if d % world != 0:
            raise ValueError(f"feature_dim {d} must be divisible by the number of ranks {world}")
        self.n, self.d, self.rank, self.world = n, d, rank, world
        self.dl = d // world
        self.c0 = rank * self.dl
        self.backend, self.group = backend, group
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

    def propagate(self, kind, x, x_next, rowsq, flags=_hip.F_L2NORM, rw=0.0, row_sqdiff=None):
        """x, x_next: (n, d/P) column slices; rowsq: f32[n] scratch.  One iteration."""
        norm = flags & _hip.F_L2NORM
        if self.world == 1:   # nothing to reduce: the fused single-pass epilogue
            for blk, (r0, r1) in zip(self.blocks, self.row_blocks):
                self.backend.propagate(blk, kind, x, x_next[r0:r1], flags, rw, x[r0:r1],
                                       row_sqdiff[r0:r1] if row_sqdiff is not None else None)
            return
        first = (flags & ~(_hip.F_L2NORM | _hip.F_SQDIFF)) | (_hip.F_ROWSQ if norm else 0)
        works = []
        for blk, (r0, r1) in zip(self.blocks, self.row_blocks):
            self.backend.propagate(blk, kind, x, x_next[r0:r1], first, rw, x[r0:r1], None,
                                   rowsq[r0:r1] if norm else None)
            if norm:
                works.append(dist.all_reduce(rowsq[r0:r1], group=self.group, async_op=True))
        second = (_hip.F_SCALE if norm else 0) | (flags & _hip.F_SQDIFF)
        for k, (r0, r1) in enumerate(self.row_blocks):
            if norm:
                works[k].wait()
            if second:
                self.backend.rowops(x_next[r0:r1], x_next[r0:r1], second, 0.0,
                                    x[r0:r1] if (flags & _hip.F_SQDIFF) else None,
                                    row_sqdiff[r0:r1] if row_sqdiff is not None else None,
                                    rowsq[r0:r1] if norm else None)

    def whiten(self, y_local, out_local):
        """whiten_embeddings (pycleora/__init__.py:130-164) for a column-partitioned matrix.
        The Gram matrix couples every pair of columns, so the step runs in a ROW layout:
        all-to-all (each rank receives all d columns of its n/P rows: (P-1)/P^2 of the matrix
        per rank, 8x less than an all-gather at P = 8) -> row-local f64 column sums / centred Gram,
        all-reduced (d and d*d doubles) -> eigh on the device, replicated -> row-local projection ->
        all-to-all back to columns.  y_local, out_local: (n_pad, d/P), rows >= n zero."""
        P, rp, dl, d = self.world, self.rows_per, self.dl, self.d
        if P > 1:
            recv = torch.empty((P, rp, dl), dtype=y_local.dtype, device=y_local.device)
            dist.all_to_all_single(recv.view(-1), y_local[: self.n_pad].reshape(-1), group=self.group)
            rows = recv.permute(1, 0, 2).reshape(rp, d).contiguous()      # my rows, all columns
        else:
            rows = y_local[: self.n_pad]
        nv = max(0, min(rp, self.n - self.rank * rp))
        cs = self.backend.colsum(rows[:nv]) if nv else torch.zeros(d, dtype=torch.float64, device=rows.device)
        if P > 1:
            dist.all_reduce(cs, group=self.group)
        mean = cs / float(self.n)
        gram = self.backend.gram(rows[:nv], mean) if nv else torch.zeros((d, d), dtype=torch.float64, device=rows.device)
        if P > 1:
            dist.all_reduce(gram, group=self.group)
        # every rank holds the same all-reduced Gram: eigh is replicated (deterministic routine),
        # but the transform is still broadcast from rank 0 so the ranks cannot drift apart
        transform = transform_from_gram(self.backend, gram, self.n, d)
        if P > 1:
            dist.broadcast(transform, src=0, group=self.group)
        proj = torch.zeros((rp, d), dtype=torch.float32, device=rows.device)
        if nv:
            self.backend.project(rows[:nv], mean.to(torch.float32), transform, proj[:nv])
        if P > 1:
            send = proj.view(rp, P, dl).permute(1, 0, 2).contiguous()       # column block j -> rank j
            dist.all_to_all_single(out_local[: self.n_pad].view(-1), send.view(-1), group=self.group)
        else:
            out_local[: self.n_pad].copy_(proj)

    def gather_columns(self, x_local):
        """(n, d) on every rank from the (n, d/P) slices (not part of the iteration)."""
        if self.world == 1:
            return x_local
        parts = [torch.empty_like(x_local) for _ in range(self.world)]
        dist.all_gather(parts, x_local.contiguous(), group=self.group)
        return torch.cat(parts, dim=1)

    def sqdiff_total(self, row_sqdiff):
        t = row_sqdiff.sum(dtype=torch.float64).reshape(1)
        if self.world > 1:
            dist.all_reduce(t, group=self.group)
        return float(t)


def embed_column_sharded(cg, kind, x0_local, iterations, residual_weight=0.0,
                         convergence_threshold=0.0, flags=_hip.F_L2NORM, whiten=False):
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
        for _ in range(iterations):
            cg.propagate(kind, x[: cg.n], y[: cg.n], rowsq, flags | _hip.F_RESIDUAL, residual_weight)
            cg.whiten(y, x_next)
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
                     sq if test else None)
        x, x_next = x_next, x
        if test:
            rmse = (cg.sqdiff_total(sq) / total) ** 0.5
            if rmse < convergence_threshold:
                ran = it + 1
                break
    return x, ran
