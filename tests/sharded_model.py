"""TEST INFRASTRUCTURE: the row partition of csrc/sharded.hip as a Python model — the same block-cyclic schedule, exchange steps
and partitioned whitening with the arithmetic (`backend`) and the collectives (`comm`, cleora_amd/comm.py) injected, so that the
algorithm runs WITHOUT a GPU: the CPU suite drives it over gloo with a numpy backend at world 2 and 4 against the oracle
(tests/test_sharded_cpu.py) and holds the library's plan against `row_bounds` boundary for boundary (tests/test_sharded_plan_cpu.py);
the GPU suite also runs it on the HIP kernels (cleora_amd.sharded.HipBackend) as a second opinion beside the library's own loops.
The PRODUCT path is cleora_amd.sharded.DeviceShardedGraph -> cleora_sharded_create / cleora_embed_sharded (until round 3 these
classes were that path; round 4 moved the loops behind the C ABI).

  * output rows are independent, each needs arbitrary rows of the PREVIOUS iterate, so every rank keeps a full replica of X and
    owns a set of row blocks of the CSR and of the next iterate;
  * rows are dealt out BLOCK-CYCLICALLY: with P ranks and K steps per iteration the row space is cut into P*K contiguous blocks
    (equal row counts, or balanced on the rowptr prefix sum: `row_bounds`) and rank r owns blocks {k*P + r}.  Step k computes block
    (k, r) on every rank and then all-gathers exactly the contiguous row range of blocks [k*P, (k+1)*P) of the next replica, IN PLACE;
  * the collective of step k runs beside the SpMM of step k+1; the L2 normalisation is row-local and fused into the SpMM epilogue.
The reference is single-process (rayon over rows, src/embedding.rs:59-63)."""
import torch

from cleora_amd import _hip
from cleora_amd import comm as comm_mod


def block_size(n, world, steps):
    """Rows per block of the equal-rows split: the padded row count is block * world * steps (block a
    multiple of 4 so every block of a 16-byte-aligned matrix stays 16-byte aligned for any d)."""
    b = -(-n // (world * steps))
    return max(4, -(-b // 4) * 4)


def row_bounds(n, rowptr, world, steps, balance="auto"):
    """Row boundaries of the world * steps contiguous blocks, block j = rows [bounds[j], bounds[j+1]).
    Rank r owns blocks {k * world + r}; step k's exchange covers blocks [k * world, (k+1) * world).

    "rows": equal row counts (a multiple of 4; the row space is padded to block * world * steps, the padding
            rows are empty) — the shards of a step are equal, so the exchange is one ncclAllGather.
    "nnz":  SURVEY.md §8e: split on the prefix sum of the work per row, weight(row) = edges + 1 (one gathered
            X row per edge plus the one Y row written), so that power-law graphs with ORDERED ids do not leave
            one rank with the hubs.  Boundaries are multiples of 4 rows; shards are unequal (all-gather-v).
    "auto": "rows" when its heaviest block is within 3 % of the mean (true for randomly permuted ids), else "nnz".
    Returns (bounds list, n_pad, mode)."""
    nb = world * steps
    rp = rowptr.to(torch.int64)
    block = block_size(n, world, steps)
    eq = [j * block for j in range(nb + 1)]
    if balance not in ("auto", "rows", "nnz"):
        raise ValueError("balance must be 'auto', 'rows' or 'nnz'")
    if balance == "auto" and nb > 1:
        cut = torch.tensor([min(b, n) for b in eq], dtype=torch.int64, device=rp.device)
        w = (rp[cut[1:]] - rp[cut[:-1]]) + (cut[1:] - cut[:-1])
        total = float(rp[n]) + n
        balance = "rows" if float(w.max()) <= 1.03 * total / nb else "nnz"
    if balance != "nnz" or nb == 1:
        return eq, block * nb, "rows"
    n4 = -(-n // 4) * 4
    cum = rp[: n + 1] + torch.arange(n + 1, dtype=torch.int64, device=rp.device)     # work before row r
    total = int(cum[n])
    targets = torch.tensor([total * j // nb for j in range(1, nb)], dtype=torch.int64, device=rp.device)
    cuts = torch.searchsorted(cum, targets).cpu().tolist() if nb > 1 else []
    bounds, prev = [0], 0
    for c in cuts:
        c = min(n4, max(prev, (int(c) + 3) // 4 * 4))
        bounds.append(c)
        prev = c
    bounds.append(n4)
    return bounds, n4, "nnz"


class ShardedGraph:
    """This rank's row blocks of a CSR graph plus the replica bookkeeping."""

    def __init__(self, n, rowptr, col, val_left, val_sym, rank, world, steps, backend,
                 hub_threshold=0, hub_segment=0, comm=None, balance="auto", group=None):
        self.n, self.rank, self.world, self.steps = n, rank, world, steps
        self.backend = backend
        self.comm = comm if comm is not None else comm_mod.default_comm(group)
        self.bounds, self.n_pad, self.balance = row_bounds(n, rowptr, world, steps, balance)
        self.block = self.bounds[1] - self.bounds[0]       # rows of the first block (every block, in "rows" mode)
        self.blocks, self.my_rows, self.sq_offset = [], [], [0]
        self.local_nnz = 0
        rp = rowptr.to(torch.int64)
        for k in range(steps):
            b0, b1 = self.bounds[k * world + rank], self.bounds[k * world + rank + 1]
            r0, r1 = min(b0, n), min(b1, n)
            e0, e1 = int(rp[r0]), int(rp[r1])
            brp = torch.full((b1 - b0 + 1,), e1 - e0, dtype=torch.int64, device=rp.device)
            brp[: r1 - r0 + 1] = rp[r0:r1 + 1] - e0     # padding rows are empty
            blk = backend.make_block(brp, col[e0:e1].clone(), val_left[e0:e1].clone(),
                                     val_sym[e0:e1].clone() if val_sym is not None else None,
                                     self.n_pad, hub_threshold, hub_segment)
            self.blocks.append(blk)
            self.my_rows.append((b0, b1))
            self.sq_offset.append(self.sq_offset[-1] + (b1 - b0))
            self.local_nnz += e1 - e0

    @property
    def local_rows(self):
        return self.sq_offset[-1]

    def step_bounds(self, k):
        """Row boundaries of the world shards of step k's exchange."""
        return self.bounds[k * self.world:(k + 1) * self.world + 1]

    def propagate(self, kind, x, x_next, flags=_hip.F_L2NORM, rw=0.0, row_sqdiff=None, gather=True):
        """One iteration: x_next <- rowops(A @ x), replicated on every rank (gather=True) or only
        this rank's row blocks of x_next written (gather=False: whitening follows).
        x, x_next: (n_pad, d) f32 replicas.  The exchange of block k runs beside the SpMM of block k+1;
        returns with every collective enqueued and ordered before later work on the compute stream
        (no host synchronisation)."""
        for k in range(self.steps):
            b0, b1 = self.my_rows[k]
            sq = row_sqdiff[self.sq_offset[k]:self.sq_offset[k + 1]] if row_sqdiff is not None else None
            self.backend.propagate(self.blocks[k], kind, x, x_next[b0:b1], flags, rw, x[b0:b1], sq)
            if self.world > 1 and gather:
                self.comm.allgather_rows(x_next, self.step_bounds(k))
        self.comm.join()

    def _valid_rows(self, k):
        b0, b1 = self.my_rows[k]
        return b0, max(0, min(b1, self.n) - b0)

    def stats(self, y):
        """(mean f64[d], centred Gram f64[d, d]) of the matrix whose row blocks `y` holds on every rank: local f64 column
        sums and centred Gram (pycleora/__init__.py:136-143), all-reduced (d and d*d doubles)."""
        d = y.shape[1]
        cs = torch.zeros(d, dtype=torch.float64, device=y.device)
        for k in range(self.steps):
            r0, nv = self._valid_rows(k)
            if nv:
                cs += self.backend.colsum(y[r0:r0 + nv])
        self.comm.allreduce(cs)
        mean = cs / float(self.n)
        gram = torch.zeros((d, d), dtype=torch.float64, device=y.device)
        for k in range(self.steps):
            r0, nv = self._valid_rows(k)
            if nv:
                gram += self.backend.gram(y[r0:r0 + nv], mean)
        self.comm.allreduce(gram)
        return mean, gram

    def rowsums(self, kind, like):
        """s = A 1 for this rank's row blocks (one f32 tensor per block, padding rows 0)."""
        out = []
        for k in range(self.steps):
            b0, b1 = self.my_rows[k]
            t = torch.zeros(b1 - b0, dtype=torch.float32, device=like.device)
            self.backend.rowsum(self.blocks[k], kind, t)
            out.append(t)
        return out

    def whiten(self, y, out, n_components=None):
        """whiten_embeddings (pycleora/__init__.py:130-164) over the row partition: `y` holds this
        rank's blocks of the matrix to whiten; `out` receives the whitened matrix, replicated.
        Local f64 column sums and centred Gram -> all-reduce (d and d*d doubles) -> transform
        (cleora_whiten_transform_dev, replicated; rank 0's copy is broadcast) -> row-local projection ->
        in-place all-gather per block."""
        d = y.shape[1]
        mean, gram = self.stats(y)
        kdim = d if n_components is None else min(int(n_components), d)
        # every rank holds the same all-reduced Gram, so the (deterministic) eigensolver is replicated;
        # the transform is still broadcast from rank 0 so that the ranks cannot drift apart
        transform = self.backend.whiten_transform(gram, self.n, kdim)
        self.comm.broadcast(transform, 0)
        mean32 = mean.to(torch.float32)
        for k in range(self.steps):
            r0, nv = self._valid_rows(k)
            if nv:
                self.backend.project(y[r0:r0 + nv], mean32, transform, out[r0:r0 + nv])
            if self.world > 1:
                self.comm.allgather_rows(out, self.step_bounds(k))
        self.comm.join()

    def sqdiff_total(self, row_sqdiff):
        """Sum of the per-row squared differences over all ranks (f64)."""
        t = row_sqdiff.sum(dtype=torch.float64).reshape(1)
        self.comm.allreduce(t)
        return float(t)


def embed_whitened_sharded(sg, kind, x0, iterations, residual_weight=0.0):
    """The default embed() loop (pycleora/__init__.py:109-117, L2 normalisation, no convergence test) over a ShardedGraph in
    the reorganised form of the single-GPU library loop (csrc/abi.hip embed_whitened_overlapped, docs/history.md §3.7-3.8):

        Y_0 = normalise(A E_0 [+ blend]);   per iteration:  Z = A Y (row blocks, no epilogue) | statistics of Y
        -> all-reduce (d + d*d doubles) -> replicated transform: Cholesky form while the reference's eigenvalue clamp is
        provably inactive (any whitening leads to the same final result), else the PCA form
        -> Y' = normalise((alpha (Z - s mu^T) + rw (Y - mu)) T) on the local rows, normalised in the projection's epilogue
        -> in-place all-gather of Y' (block k's gather beside block k+1's projection);   E_T = PCA-whiten(Y_{T-1}).

    One all-gather of the n x d iterate per iteration, as in the plain loop; the eigensolver only in the last iteration.
    x0: (n_pad, d) replica.  Returns the replica of E_T."""
    if iterations <= 0:
        return x0
    n, rw = sg.n, float(residual_weight)
    blend = rw > 0.0
    y = torch.zeros_like(x0)
    y_next = torch.zeros_like(x0)
    z = torch.zeros_like(x0)                       # only this rank's row blocks are ever written
    s = sg.rowsums(kind, x0)
    sg.propagate(kind, x0, y, _hip.F_L2NORM | _hip.F_RESIDUAL | _hip.F_BLEND_ANY, rw)            # Y_0, replicated
    for _ in range(iterations - 1):
        sg.propagate(kind, y, z, 0, 0.0, gather=False)
        mean, gram = sg.stats(y)
        transform, _ = sg.backend.whiten_transform_any(gram, n) if n > 1 else (None, 0)
        if transform is not None:
            sg.comm.broadcast(transform, 0)        # identical on every rank already; keeps the ranks from drifting apart
        mean32 = mean.to(torch.float32)
        for k in range(sg.steps):
            r0, nv = sg._valid_rows(k)
            if nv and transform is not None:
                rows = slice(r0, r0 + nv)
                done = sg.backend.project_general(z[rows], mean32, transform, y_next[rows], rowscale=s[k][:nv],
                                                  x2=y[rows] if blend else None, alpha=1.0 - rw, beta=rw, norm=1)
                if not done:
                    sg.backend.rowops(y_next[rows], y_next[rows], _hip.F_L2NORM)
            elif nv:                               # one entity: whiten_embeddings returns its input (:132-133)
                rows = slice(r0, r0 + nv)
                sg.backend.rowops(z[rows], y_next[rows], _hip.F_L2NORM | _hip.F_RESIDUAL | _hip.F_BLEND_ANY, rw, y[rows])
            if sg.world > 1:
                sg.comm.allgather_rows(y_next, sg.step_bounds(k))
        sg.comm.join()
        y, y_next = y_next, y
    out = y_next
    sg.whiten(y, out)
    return out


def embed_sharded(sg, kind, x0, iterations, residual_weight=0.0, convergence_threshold=0.0,
                  flags=_hip.F_L2NORM, whiten=False):
    """embed_full / embed_full_with_convergence (src/embedding.rs:106-188) over a ShardedGraph;
    with whiten=True the default embed() loop of pycleora/__init__.py:109-125 (normalise, then
    whiten, every iteration; no convergence test in that mode here) — in the reorganised form
    (embed_whitened_sharded) for the L2 normalisation, in the reference's order (whiten="sequential", or any other
    normalisation: the Cholesky intermediate whitening needs the rotation invariance of the L2 norm).
    x0: (n_pad, d) replica (rows >= n zero).  Returns (x, iterations_run)."""
    x = x0
    x_next = torch.zeros_like(x0)
    if whiten:
        if convergence_threshold > 0:
            #  the model's whitened loops run exactly `iterations` iterations; the early stop between whitened iterates
            # (pycleora/__init__.py:122-125) is in the library's loop: DeviceShardedGraph.embed / cleora_embed_sharded
            raise ValueError("the model's whitened loops have no convergence test: use DeviceShardedGraph.embed (cleora_embed_sharded)")
        if whiten != "sequential" and flags == _hip.F_L2NORM:
            return embed_whitened_sharded(sg, kind, x0, iterations, residual_weight), iterations
        y = torch.zeros_like(x0)
        for _ in range(iterations):
            # the Python loop blends for ANY rw > 0 (pycleora/__init__.py:111-115)
            sg.propagate(kind, x, y, flags | _hip.F_RESIDUAL | _hip.F_BLEND_ANY, residual_weight, gather=False)
            sg.whiten(y, x_next)
            x, x_next = x_next, x
        return x, iterations
    check = convergence_threshold > 0
    flags = flags | _hip.F_RESIDUAL
    sq = torch.zeros(sg.local_rows, dtype=torch.float64, device=x0.device) if check else None
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
