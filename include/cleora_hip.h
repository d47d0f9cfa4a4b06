/*
 * cleora_hip.h — C ABI of libcleora_hip.so: the MI355X (gfx950) implementation of
 * pycleora 3.2.1's Markov-propagation hot path.
 *
 * The reference has no C ABI of its own: its boundary is the PyO3 class
 * `pycleora.pycleora.SparseMatrix` (src/lib.rs:84-476) whose methods call
 * `NdArrayMatrix::*` (src/embedding.rs).  BASELINE.json:north_star asks for the Rust
 * host to reach the kernels "through a thin extern-C FFI"; the entry points below
 * are exactly what that FFI binds.  Every function cites the reference interface it
 * replaces (paths relative to the reference checkout).  INTEGRATION.md shows the
 * Rust `extern "C"` block and the call sites in src/embedding.rs / src/lib.rs.
 *
 * Conventions
 *   - plain C, opaque handles, plain pointers and sizes; no C++/torch types.
 *   - every function returns CLEORA_OK (0) or a negative CLEORA_E_* code;
 *     cleora_last_error() returns a thread-local message for the last failure.
 *   - "*_dev" entry points take DEVICE pointers and a hipStream_t (as void*; NULL =
 *     the default stream) and only enqueue work.  The others take HOST pointers,
 *     borrow them for the duration of the call and return with results on the host.
 *   - matrices are row-major f32 with a leading dimension in ELEMENTS (ld >= d).
 *   - threads: every entry point may be called from any thread; calls on one graph handle are
 *     serialised by a mutex inside the handle while they enqueue.  A handle owns device scratch
 *     (the hub rows' sums, its side stream, timing events), so launches on the SAME handle must be ordered on the
 *     device too — use one stream per handle (or the default stream); different handles are independent.
 *   - there is no CPU fallback: without a gfx950 device every compute entry point
 *     fails with CLEORA_E_NODEVICE / CLEORA_E_HIP.
 */
#ifndef CLEORA_HIP_H
#define CLEORA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLEORA_ABI_VERSION 5

#define CLEORA_OK 0
#define CLEORA_E_INVALID (-1)   /* bad argument (shape, null pointer, unknown enum) */
#define CLEORA_E_OOM (-2)       /* device or host allocation failed */
#define CLEORA_E_HIP (-3)       /* a HIP runtime call failed; see cleora_last_error() */
#define CLEORA_E_NODEVICE (-4)  /* no usable GPU */
#define CLEORA_E_RCCL (-5)      /* RCCL could not be loaded or a collective failed; see cleora_last_error() */

/* MarkovType (src/embedding.rs:7-10) */
#define CLEORA_LEFT 0
#define CLEORA_SYMMETRIC 1

/* flags of cleora_propagate_dev / cleora_rowops_dev */
#define CLEORA_F_L2NORM 1u    /* fuse l2_normalize_inplace (src/embedding.rs:88-104) into the epilogue */
#define CLEORA_F_FASTNORM 2u  /* sum of squares by wave butterfly instead of the reference's
                                 sequential order (last-ulp differences; default is sequential) */
#define CLEORA_F_RESIDUAL 4u  /* y = (1-rw)*y + rw*x_self before the norm (src/embedding.rs:121-129) */
#define CLEORA_F_SQDIFF 8u    /* row_sqdiff[r] = sum_j (y[r][j]-x_self[r][j])^2 in f64 (src/embedding.rs:169-176) */
#define CLEORA_F_ROWSQ 16u    /* OUT: row_sumsq[r] = sum_j y[r][j]^2 (f32, the order of src/embedding.rs:94-97); with
                                 L2NORM absent the row is left unscaled — used when a row's columns live on
                                 several GPUs and the sums must be all-reduced before the scale */
#define CLEORA_F_SCALE 32u    /* IN: row_sumsq[r] is the complete sum of squares; y[r] *= 1/max(sqrt(.),1e-10)
                                 (src/embedding.rs:98-102) without recomputing it */
#define CLEORA_F_WHITEN 64u   /* cleora_embed only: whiten_embeddings after the L2 norm of every iteration (pycleora/__init__.py:963-971) */
#define CLEORA_F_L1NORM 128u  /* y[r] /= max(sum_j |y[r][j]|, 1e-10): _normalize(emb, "l1") (pycleora/__init__.py:947-950);
                                 exclusive with L2NORM / ROWSQ / SCALE */
#define CLEORA_F_BLEND_ANY 256u /* with RESIDUAL: blend for ANY rw > 0, like the Python loop of embed() (pycleora/__init__.py:111-115);
                                 without it the blend is gated on 0 < rw < 1 like the Rust loop (src/embedding.rs:116) */
#define CLEORA_F_SQDIFF64 512u /* with SQDIFF: delta = (double)y - (double)x_self, like _compute_rmse (pycleora/__init__.py:974-976);
                                 without it delta is the f32 difference like src/embedding.rs:172 */
#define CLEORA_F_HUB_SEGMENTS 1024u /* rows longer than hub_threshold: sum hub_segment-edge segments on separate wavefronts and add the
                                     * partial sums in a fixed order, instead of the default — every row, hub rows included, added
                                     * edge by edge in stored order like src/embedding.rs:76-83 (bit-equal to the reference).  For a
                                     * pathological hub (10^7+ edges in one row: an in-order chain of as many dependent adds) this
                                     * is the faster form; its hub rows differ from the reference by rounding (<= 2e-6 * sum|terms|) */

#define CLEORA_F_ROWSQ_CONT 2048u /* with ROWSQ: the sum of squares CONTINUES from row_sumsq[r] (in: the sum over the columns to the left,
                                   * out: that sum extended over this call's d columns, the reference's order src/embedding.rs:94-97) — the
                                   * column partition hands a row's running sum from rank to rank so that the L2 norm is the one-GPU sum bit
                                   * for bit (csrc/colsharded.hip) */

typedef struct cleora_graph cleora_graph; /* device-resident CSR shard (struct SparseMatrix, src/sparse_matrix.rs:56-78) */

typedef struct cleora_graph_info {
    uint64_t n_rows, n_cols, nnz;
    uint64_t n_hub_rows;      /* rows longer than hub_threshold ("long" rows) */
    uint64_t n_hub_segments;  /* their hub_segment-edge segments (the CLEORA_F_HUB_SEGMENTS form) */
    uint64_t device_bytes;    /* HBM held by the handle */
    uint64_t hot_rows;        /* rows the gather cache policy currently keeps cacheable; 0 = policy inactive */
    uint32_t hub_threshold, hub_segment;
    int32_t device;
    int32_t has_symmetric;
    uint64_t n_inorder_rows;  /* long rows beyond hub_inorder_min edges: summed by the in-order hub launch (spmm.hip hub_inorder_kernel);
                                 the other long rows are the first work items of the main launch, longest first */
    uint64_t hub_inorder_min;
} cleora_graph_info;

/* ---- library / device ------------------------------------------------------------ */
int cleora_abi_version(void);
const char *cleora_last_error(void);
int cleora_device_count(int *count);
int cleora_set_device(int device);

/* ---- device memory + streams (plumbing for hosts that do not bring their own) ------ */
int cleora_malloc(uint64_t bytes, void **dev_ptr);
int cleora_free(void *dev_ptr);
int cleora_memcpy_h2d(void *dst_dev, const void *src_host, uint64_t bytes, void *stream);
int cleora_memcpy_d2h(void *dst_host, const void *src_dev, uint64_t bytes, void *stream);
int cleora_memcpy_d2d(void *dst_dev, const void *src_dev, uint64_t bytes, void *stream);
int cleora_memset(void *dst_dev, int value, uint64_t bytes, void *stream);
int cleora_stream_sync(void *stream);
/* A non-blocking stream of the current device, and the one ordering primitive the multi-GPU loop needs:
 * everything enqueued on `waiter` after this call runs after everything enqueued on `signaller` before it
 * (an event record + stream-wait; no host synchronisation).  NULL = the default stream. */
int cleora_stream_create(void **stream);
int cleora_stream_destroy(void *stream);
int cleora_stream_wait_stream(void *waiter, void *signaller);

/* ---- graph: the CSR the kernels read ------------------------------------------------
 * Replaces the in-memory `edges`/`slices` of struct SparseMatrix (src/sparse_matrix.rs:56-78):
 * AoS {u32 col, f32 left, f32 sym} + (start,end) pairs become SoA rowptr u64[n_rows+1],
 * col u32[nnz], one f32[nnz] stream per MarkovType (val_sym may be NULL).
 * Rows are the OUTPUT rows of this shard (all rows on one GPU; a row block when the graph
 * is row-partitioned); col indexes the n_cols rows of the full embedding matrix.
 * hub_threshold: rows with more edges are "long" rows — scheduled first (longest first) in the main launch or, beyond a length
 * that grows with the graph (cleora_graph_set_hub_inorder_min), summed by a launch of their own beside the main one, one wavefront per
 * 64-column slab: the reference's order either way.  With CLEORA_F_HUB_SEGMENTS they are summed as hub_segment-edge segments on
 * separate wavefronts instead (0 = defaults 256 / 256).  Copies the arrays; the caller keeps
 * ownership of its buffers. */
int cleora_graph_create(int device, uint64_t n_rows, uint64_t n_cols, uint64_t nnz,
                        const uint64_t *rowptr, const uint32_t *col, const float *val_left,
                        const float *val_sym, uint32_t hub_threshold, uint32_t hub_segment,
                        cleora_graph **out);
/* Same, but rowptr/col/val_* are DEVICE pointers on `device` and are ADOPTED without a copy:
 * they must stay valid until cleora_graph_destroy (which does not free them). */
int cleora_graph_create_dev(int device, uint64_t n_rows, uint64_t n_cols, uint64_t nnz,
                            const uint64_t *rowptr_dev, const uint32_t *col_dev,
                            const float *val_left_dev, const float *val_sym_dev,
                            uint32_t hub_threshold, uint32_t hub_segment, cleora_graph **out);
int cleora_graph_destroy(cleora_graph *g);
int cleora_graph_get_info(const cleora_graph *g, cleora_graph_info *info);

/* Gather cache policy (no reference counterpart; changes no result bit).  The most frequently
 * referenced embedding rows — those with the largest in-degree, up to `hot_bytes` of X — are gathered
 * with the default cache policy and all other rows non-temporally, so the cold stream does not evict
 * the hot set: measured -10 % on the C3 graph.
 * hot_bytes < 0: automatic (768 MiB budget when X is >= 1 GiB; the default), 0: off, > 0: forced budget.
 * Needs n_cols < 2^31 (the mark is bit 31 of a private copy of `col`: +4 B per edge of HBM). */
int cleora_graph_set_hot_cache(cleora_graph *g, int64_t hot_bytes);

/* Per-kernel timing for roofline reporting (no reference counterpart).  While enabled, every
 * cleora_propagate_dev call on this graph brackets its three kernels with HIP events on the
 * launch stream.  cleora_graph_get_timing waits for the recorded events, returns the summed
 * durations in milliseconds — ms[0] the fork of the in-order hub launch onto its side stream, ms[1] spmm_rows (the dominant kernel,
 * with the hub launch beside it), ms[2] the join (with CLEORA_F_HUB_SEGMENTS: hub_finish_kernel) — and the number of calls they
 * cover, then resets the record.  ms[0] + ms[1] + ms[2] = the span of one SpMM on the launch stream. */
/* Which long rows take the in-order hub launch: those with MORE than min_edges edges (default: nnz / 8192 — what one wavefront can
 * gather in a quarter of the launch — clamped to [hub_threshold, 128 x hub_threshold]; values below hub_threshold mean hub_threshold).  Long rows up to min_edges are the FIRST work items of the main launch, longest first, one
 * wavefront each like any row — long enough to be a tail if they started last, short enough for one wavefront; rows beyond it are
 * cut by column over many wavefronts (hub_inorder_kernel).  Either way every row is summed in the reference's order: the setting
 * moves work between two kernels, never a bit of the result.  Waits for the device (the tables are replaced). */
int cleora_graph_set_hub_inorder_min(cleora_graph *g, uint64_t min_edges);
/* The in-order hub launch's shape: lanes per edge — 4 (four edges per 16-byte load instruction, 64-column slabs per wavefront) or
 * 2 (eight edges, 32-column slabs: twice the edges in flight per wavefront, twice the wavefronts per row — a shorter chain for the
 * longest row at twice the vector instructions); 0 = automatic (default): 4 while the estimated chain of the graph's longest row
 * stays under half the main kernel's estimated time, else 2.  Results are bit-identical for either choice. */
int cleora_graph_set_hub_lanes(cleora_graph *g, int lanes);
/* Which rows of the in-order hub launch take the chain kernel (hub_chain_kernel, csrc/spmm.hip: one 8-wave block per (row, 64-column
 * slab) — seven waves gather and multiply, one wave only adds, in the reference's order, src/embedding.rs:76-83; ~5 x faster per
 * edge than one wavefront doing everything): the rows with AT LEAST min_edges edges.  0 = automatic (the rows whose chain would
 * outlast a quarter of the main kernel's estimated time, at most 128 blocks), 1 = every row of the hub launch, UINT64_MAX = none.
 * Moves work between kernels, never a bit of the result. */
int cleora_graph_set_hub_chain_min(cleora_graph *g, uint64_t min_edges);
int cleora_graph_set_timing(cleora_graph *g, int enable);
int cleora_graph_get_timing(cleora_graph *g, double ms[3], uint64_t *calls);

/* Iterate buffers placed for the SpMM (no reference counterpart).  The same launch has been measured up to 12-20 % slower when the
 * buffer it gathers from and the buffer it writes fall into the same physical placement class of the HBM (DESIGN.md
 * §3.1; not visible in virtual addresses; box-dependent).  Allocates `count` buffers of max(n_rows, n_cols) x d floats: bufs[0]
 * first, then each partner by TIMING the real kernel on candidates (count = 2: gathers from bufs[0], writes the candidate; count >= 3,
 * the whitened loop's use: gathers from the candidate, writes bufs[0]; a candidate's time
 * is the median of three launches) until the best is >= 4 % faster than the slowest seen (at most 4 per slot, none once
 * CLEORA_PLACEMENT_BUDGET_MS = 1500 ms of wall clock are spent); the best is taken (the first candidate when the best is within 1 % of it).  cleora_alloc_iterates_for also takes the number of SpMM
 * launches the caller is about to run (0 = unknown) and stops searching when the most it could win — 15 % of a launch per
 * iteration — is less than trying another candidate costs (three launches + freeing the loser, ~30 ms per GB): the embed loops
 * call it that way.  Use bufs[0] as the buffer EVERY SpMM of the loop touches: ping-pong = the pair (bufs[0], bufs[1]); the
 * whitened loop = SpMM output in bufs[0], the two whitened iterates in bufs[1] and bufs[2].  Iterates below 256 MiB are allocated
 * without the search.  Contents are unspecified.  ms (optional, double[2]): the first candidate's median launch time — what a
 * plain allocation pair runs at — and the chosen pair's (never more).  Free every buffer with cleora_free. */
int cleora_alloc_iterates(const cleora_graph *g, uint32_t d, uint32_t count, void **bufs, double *ms);
int cleora_alloc_iterates_for(const cleora_graph *g, uint32_t d, uint32_t count, uint64_t iterations, void **bufs, double *ms);

/* ---- device-pointer hot path ------------------------------------------------------- */

/* NdArrayMatrix::spmm_kernel / multiply_into (src/embedding.rs:41-86) with the optional
 * fused epilogue of embed_full (residual blend :121-129, l2_normalize_inplace :88-104,
 * squared-difference for the RMSE test :169-176).
 *   y[r,:] = sum_{e in row r, stored order} val_e * x[col_e,:]     separate f32 mul / add
 * Rows without edges produce zeros (the reference zero-fills first, :21/:47).
 * x: n_cols x d (ldx); y: n_rows x d (ldy); x_self: the n_rows rows of the previous iterate
 * that correspond to this shard's rows (ld = ldx), needed by RESIDUAL / SQDIFF;
 * row_sqdiff: f64[n_rows] or NULL; row_sumsq: f32[n_rows] (ROWSQ out / SCALE in) or NULL.
 * x and y must not alias. */
int cleora_propagate_dev(const cleora_graph *g, int markov_type, const float *x, uint64_t ldx,
                         uint32_t d, float *y, uint64_t ldy, uint32_t flags,
                         float residual_weight, const float *x_self, double *row_sqdiff,
                         float *row_sumsq, void *stream);

/* The same SpMM with CALLER-SUPPLIED per-edge values (f32[nnz], the graph's edge order) in place of the stored
 * Markov values: the `weighted_adj @ embeddings` step of the reference's re-weighted variants
 * (embed_with_attention, pycleora/__init__.py:269; any host that rescales edges between iterations). */
int cleora_propagate_vals_dev(const cleora_graph *g, const float *edge_vals_dev, const float *x, uint64_t ldx,
                              uint32_t d, float *y, uint64_t ldy, uint32_t flags, float residual_weight,
                              const float *x_self, double *row_sqdiff, float *row_sumsq, void *stream);

/* Attention weights of embed_with_attention (pycleora/__init__.py:241-268) for every stored edge (r, c):
 * score = cos(x_r, x_c) / temperature; softmax over the row's edges; multiplied by the stored Markov value of
 * `markov_type` and renormalised so each row sums to 1 (each denominator clamped at 1e-10 like the reference).
 * edge_vals_out_dev: f32[nnz] in edge order, ready for cleora_propagate_vals_dev.  Square graphs only. */
int cleora_edge_attention_dev(const cleora_graph *g, int markov_type, const float *x, uint64_t ldx, uint32_t d,
                              float temperature, float *edge_vals_out_dev, void *stream);

/* One iteration of embed_with_attention (pycleora/__init__.py:241-270) as ONE pass over the edges: the attention weights of
 * cleora_edge_attention_dev and the weighted sum of cleora_propagate_vals_dev together (softmax accumulated online: every
 * neighbour row is gathered once instead of twice), then the same epilogue flags as cleora_propagate_dev (RESIDUAL / BLEND_ANY,
 * L2NORM or L1NORM, SQDIFF / SQDIFF64; not ROWSQ / SCALE).  Square graphs; rows of d % 4 == 0 floats, d <= 2048, x / y /
 * x_self 16-byte aligned with ld % 4 == 0 (CLEORA_E_INVALID otherwise: run the two calls above instead). */
int cleora_propagate_attention_dev(const cleora_graph *g, int markov_type, const float *x, uint64_t ldx, uint32_t d,
                                   float temperature, float *y, uint64_t ldy, uint32_t flags, float residual_weight,
                                   const float *x_self, double *row_sqdiff, void *stream);

/* Row-wise epilogue alone: NdArrayMatrix::l2_normalize_inplace (src/embedding.rs:88-104) when
 * flags = CLEORA_F_L2NORM; same flags as above.  x may equal y (in place). */
int cleora_rowops_dev(const float *x, uint64_t ldx, uint64_t n, uint32_t d, float *y, uint64_t ldy,
                      uint32_t flags, float residual_weight, const float *x_self,
                      double *row_sqdiff, float *row_sumsq, void *stream);

/* initialize_deterministically_rust + init_value (src/lib.rs:69-81, 478-488) from the cached
 * XXH64 entity hashes (hash_entity, src/entity.rs:109-114).  Bit-exact (integer arithmetic). */
int cleora_init_dev(const uint64_t *entity_hash_dev, uint64_t n, uint32_t d, int64_t seed,
                    float *x, uint64_t ldx, void *stream);

/* Deterministic (fixed-order) sum of a f64 vector into *out_dev; used for the RMSE of
 * embed_full_with_convergence (src/embedding.rs:169-183).  workspace: f64[cleora_reduce_workspace(n)]. */
uint64_t cleora_reduce_workspace(uint64_t n);
int cleora_reduce_sum_f64_dev(const double *v, uint64_t n, double *workspace, double *out_dev,
                              void *stream);

/* ---- whitening (pycleora/__init__.py:130-164) -------------------------------------- */

/* colsum[c] = sum_r x[r,c] in f64 (the mean of :136 is colsum / n).
 * workspace: f64[cleora_colsum_workspace(n, d)]. */
uint64_t cleora_colsum_workspace(uint64_t n, uint32_t d);
int cleora_colsum_dev(const float *x, uint64_t ldx, uint64_t n, uint32_t d, double *workspace,
                      double *colsum_dev, void *stream);

/* gram[i,j] = sum_r (x[r,i]-mean[i]) * (x[r,j]-mean[j]) in f64 on the f64 matrix cores
 * (:138-143 without the 1/(n-1) factor).  gram: d x d row-major, full symmetric matrix.
 * workspace: f64[cleora_gram_workspace(n, d)]. */
uint64_t cleora_gram_workspace(uint64_t n, uint32_t d);
int cleora_centered_gram_dev(const float *x, uint64_t ldx, uint64_t n, uint32_t d,
                             const double *mean_dev, double *workspace, double *gram_dev,
                             void *stream);

/* out[r,:] = (x[r,:] - mean_f32) @ transform   (f32 MFMA; :157-163).
 * transform: d x k row-major f32; out: n x k (ldo). */
int cleora_project_dev(const float *x, uint64_t ldx, uint64_t n, uint32_t d,
                       const float *mean_f32_dev, const float *transform_dev, uint32_t k,
                       float *out, uint64_t ldo, void *stream);

/* The projection of the REORGANISED whitened loop (csrc/abi.hip embed_whitened_overlapped; the SpMM is linear, so it is
 * taken before the projection: A ((Y - 1 mu^T) T) = (A Y - s mu^T) T with s = A 1):
 *     out[r,:] = normalise( (alpha * (x[r,:] - rowscale[r] * mean) + beta * (x2[r,:] - mean)) @ transform )
 * rowscale_dev == NULL: scale 1; x2 == NULL: no second term (alpha is then 1) — the plain cleora_project_dev.
 * norm: 0 none, 1 L2-normalise (src/embedding.rs:98-102), 2 L1-normalise (pycleora/__init__.py:947-950) every output row in
 * the kernel's epilogue when the shape allows it (k <= 256); *norm_done (host, may be NULL) reports whether it was — if not,
 * the caller runs cleora_rowops_dev on `out`. */
int cleora_project_general_dev(const float *x, uint64_t ldx, uint64_t n, uint32_t d, const float *mean_f32_dev,
                               const float *transform_dev, uint32_t k, float *out, uint64_t ldo,
                               const float *rowscale_dev, const float *x2, uint64_t ldx2, float alpha, float beta,
                               int norm, int *norm_done, void *stream);

/* rowsum[r] = sum of the stored values of row r (f32, stored order): s = A 1 of the identity above; 1 for every row of a
 * row-stochastic left Markov matrix up to rounding, anything for symmetric values or trimmed hyperedges. */
int cleora_csr_rowsum_dev(const cleora_graph *g, int markov_type, float *rowsum_dev, void *stream);
/* The same plus rowabs[r] = sum of |values| of row r, rounded up: a bound on |(A Y)[r][j]| for |Y| <= 1 (may be NULL). */
int cleora_csr_rowsums_dev(const cleora_graph *g, int markov_type, float *rowsum_dev, float *rowabs_dev, void *stream);

/* The same projection for BOUNDED operands — what the loop's intermediate iterations have: x = A Y with unit rows Y, so
 * |x[r][j]| <= rowbound[r] (cleora_csr_rowsums_dev's rowabs; NULL: 1) and |mean[j]| <= 1 —
 *     out[r,:] = normalise( (x[r,:] - rowscale[r] * mean) @ transform )
 * At d = k = 256 this takes the f16 matrix cores with the transform resident in registers (csrc/project_f16.hip: operands scaled
 * into the f16 range by per-row / per-column powers of two, each f32 product from three f16 MFMAs of two-way split operands:
 * 2^-22 per product, the error class of the f32 GEMM of pycleora/__init__.py:163); *form (host, may be NULL) reports 1 when it
 * did.  Any other shape with rowbound != NULL and d a multiple of 32 takes the same arithmetic in the general kernel (the transform
 * streamed through LDS: *form == 2; config 5's d = 1024); with rowbound == NULL, or d not a multiple of 32, the call is
 * cleora_project_general_dev (*form == 0).  An operand that violates its bound overflows f16: the result is then inf / NaN, never
 * silently wrong.  norm as above: always applied when *form == 1, as *norm_done says otherwise. */
int cleora_project_bounded_dev(const float *x, uint64_t ldx, uint64_t n, uint32_t d, const float *mean_f32_dev,
                               const float *transform_dev, uint32_t k, float *out, uint64_t ldo, const float *rowscale_dev,
                               const float *rowbound_dev, int norm, int *norm_done, int *form, void *stream);

/* mean64[c] = colsum[c] / n (f64, :136) and mean32[c] = (float)mean64[c] (:159), on the device. */
int cleora_mean_dev(const double *colsum_dev, uint64_t n, uint32_t d, double *mean64_dev,
                    float *mean32_dev, void *stream);

/* From the centred Gram matrix to the whitening transform, on the device (:143-156):
 *   cov = gram / (n-1);  lambda, V = eigh(cov);  descending order;
 *   transform[:, j] = (float)(V[:, j] / sqrt(max(lambda_j, 1e-10)))   for the k leading components.
 * gram: d x d f64 (read only); transform: d x k row-major f32; eigenvalues_dev: f64[d] descending, may be
 * NULL.  workspace: cleora_eigh_workspace(d) BYTES.  The eigenproblem is rocSOLVER's dsyevd (the device
 * counterpart of the LAPACK call behind np.linalg.eigh), bound with dlopen on first use; eigenvector
 * signs are the solver's (the reference's are LAPACK's: whitening is defined up to them). */
uint64_t cleora_eigh_workspace(uint32_t d);
int cleora_whiten_transform_dev(const double *gram_dev, uint64_t n, uint32_t d, uint32_t k,
                                float *transform_dev, double *eigenvalues_dev, void *workspace,
                                void *stream);

/* The statistics half of whiten_embeddings in ONE pass over X (what cleora_whiten_dev and the whitened cleora_embed loop run):
 * the exact f64 mean (:136) and the centred Gram sum_r (x_r - mean)(x_r - mean)^T (:138-143 without the 1/(n-1)), computed
 * around a sampled shift and corrected exactly (csrc/whiten.hip).  intermediate = 0: f64 matrix cores end to end, the form
 * behind every whitening a caller can observe.  intermediate = 1: the form the whitened loop takes for iterations whose
 * whitening only has to BE a whitening (see cleora_whiten_transform_any_dev) — for d a multiple of 256 (<= 2048) and n >= 4096 the
 * Gram comes from the bf16 matrix cores with split f32 operands (the matrix cores sum 32 rows, the vector unit adds those sums in f32
 * over <= 2048 rows, f64 across): six exact bf16 products per f32 product below 2^20 rows (~1e-8 of the f64 Gram), three plus the
 * diagonal's residual term from 2^20 rows on (what they drop falls as sqrt(d) 2^-18 / sqrt(n): 1.9e-8 at n = 10 M, d = 256); other
 * shapes as intermediate = 0.
 * workspace: cleora_whiten_workspace(n, d) BYTES; mean64_dev: f64[d]; gram_dev: f64[d*d].  n >= 2. */
int cleora_whiten_stats_dev(const float *x, uint64_t ldx, uint64_t n, uint32_t d, void *workspace, int intermediate,
                            double *mean64_dev, double *gram_dev, void *stream);

/* The transform for the INTERMEDIATE iterations of the loop E <- whiten(l2_normalise(A E)) (pycleora/__init__.py:109-117):
 * inside that loop any W with W^T cov W = I gives the same final embedding as the reference's PCA form (two whitenings
 * differ by a rotation, which the row-wise L2 norm and the last iteration's PCA whitening remove), so the cheap Cholesky form
 * transform = L^-T (cov = L L^T; d x d row-major f32, upper triangular) is taken — but ONLY when the reference's clamp
 * max(lambda, 1e-10) (:155) is provably inactive: potrf succeeds, the smallest squared pivot is >= 1e-8 and
 * trace(cov^-1) = ||L^-T||_F^2 <= 1e10 (=> lambda_min >= 1e-10).  Otherwise the PCA form of cleora_whiten_transform_dev
 * (k = d) is computed.  *form_out (host, may be NULL): 1 = Cholesky form, 0 = PCA form.  For d <= 256 the factorisation runs
 * on the host (cleora_cholesky_whiten_host below: Gram down, transform up, ~1 ms), beyond on rocSOLVER's potrf + trtri.  gram_dev is taken as
 * EXACT (f64 statistics); the library's own loops add a margin when theirs are the approximate ones (csrc/eigh.hip).
 * Unlike the other *_dev entry points this one WAITS for `stream` (the decision is taken on the host). */
int cleora_whiten_transform_any_dev(const double *gram_dev, uint64_t n, uint32_t d, float *transform_dev,
                                    void *workspace, void *stream, int *form_out);

/* The host half of cleora_whiten_transform_any_dev for d <= 256 (csrc/dxd_host.cpp; plain host memory, no GPU involved):
 * cov = gram / (n - 1) = U^T U on one host core, transform = U^-1 = L^-T as f32 (d x d row-major, upper triangular).
 * Returns 0, or 1 if cov is not safely positive definite (a squared pivot < 1e-8 or trace(cov^-1) > 0.999e10: the caller
 * takes the PCA form, which reproduces the reference's clamp, pycleora/__init__.py:155), < 0 on bad arguments.
 * *trace_inverse_out (may be NULL) = ||transform||_F^2 = sum 1/lambda_i. */
int cleora_cholesky_whiten_host(const double *gram_host, uint64_t n, uint32_t d, float *transform_host, double *trace_inverse_out);

/* whiten_embeddings (pycleora/__init__.py:130-164) on device buffers, one stream, no host round trip:
 * column sums -> mean -> centred Gram -> transform -> projection.  y: n x k (ldy), must not alias x.
 * n_components = 0 (or >= d) keeps all d components.  n == 1 copies the row unchanged (:132-133).
 * workspace: cleora_whiten_workspace(n, d) BYTES.  eigenvalues_dev: f64[d] descending or NULL. */
uint64_t cleora_whiten_workspace(uint64_t n, uint32_t d);
int cleora_whiten_dev(const float *x, uint64_t ldx, uint64_t n, uint32_t d, uint32_t n_components,
                      float *y, uint64_t ldy, void *workspace, double *eigenvalues_dev, void *stream);

/* Per-stage timing of cleora_whiten_dev for roofline reporting (no reference counterpart; process-wide).  While
 * enabled every call brackets its stages with HIP events on the launch stream; cleora_whiten_get_timing waits for
 * them, returns the summed milliseconds — ms[0] column statistics (mean), ms[1] centred Gram (f64 MFMA),
 * ms[2] eigensolver + transform, ms[3] projection (f32 MFMA) — and the number of calls covered, then resets. */
int cleora_whiten_set_timing(int enable);
int cleora_whiten_get_timing(double ms[4], uint64_t *calls);

/* ---- similarity ---------------------------------------------------- */

/* scores[r] = (x[r] . query) / max(||x[r]||, 1e-10): the row-normalise + GEMV of find_most_similar /
 * predict_links (pycleora/__init__.py:636-681, 753-781) in one pass over X.  `query` is expected
 * already normalised (d floats, device). */
int cleora_cosine_scores_dev(const float *x, uint64_t ldx, uint64_t n, uint32_t d,
                             const float *query_dev, float *scores_dev, void *stream);

/* ---- multi-GPU exchange steps: RCCL over xGMI (no reference counterpart: pycleora is single-process) ----
 * BASELINE.json:north_star: "the graph is row-partitioned across the 8 GPUs of one node with an RCCL all-gather of
 * the embedding matrix over xGMI between iterations", reachable "through a thin extern-C FFI".  One communicator
 * per process (= per GPU).  RCCL is bound with dlopen on the first call (CLEORA_RCCL=<path> overrides the name), so
 * single-GPU hosts never load it.  Bootstrap like NCCL: rank 0 calls cleora_comm_unique_id and the HOST distributes
 * the CLEORA_COMM_ID_BYTES bytes to the other ranks by its own means (MPI, a TCP store, a file), then every rank
 * calls cleora_comm_create (collective: returns once all ranks have joined).
 * Every collective works IN PLACE on device memory, only ENQUEUES on `stream`, and must be called by all ranks in
 * the same order; order it against the kernels with cleora_stream_wait_stream. */
#define CLEORA_COMM_ID_BYTES 128
#define CLEORA_ALLGATHER_RING 0  /* ncclAllGather for equal shards (grouped ncclBroadcast for unequal ones) */
#define CLEORA_ALLGATHER_P2P 1   /* grouped ncclSend/ncclRecv: every shard crosses each xGMI link once, directly */
#define CLEORA_ALLGATHER_PEER 2  /* peer-direct stores: a copy kernel writes the shard straight into every peer's registered buffer through
                                    hipIpc mappings, all links at once, a flag per peer (needs cleora_comm_enable_peer + cleora_comm_register) */
typedef struct cleora_comm cleora_comm;
int cleora_comm_unique_id(void *id_out);
int cleora_comm_create(const void *id, int rank, int world, int device, cleora_comm **out);
int cleora_comm_destroy(cleora_comm *c);
int cleora_comm_info(const cleora_comm *c, int *rank, int *world, int *device);
int cleora_comm_set_allgather(cleora_comm *c, int algo);   /* default RING (PEER on a local communicator) */
int cleora_comm_get_allgather(const cleora_comm *c, int *algo);   /* which of the two cleora_allgatherv_f32_dev will take */

/* Peer-direct transport (csrc/peer.hip; one node): the ranks map each other's buffers with hipIpc and exchange data with plain stores
 * over xGMI — SURVEY 8e's "peer-mapped direct stores".  Bootstrap through a POSIX shared-memory segment named after the id.
 *   cleora_comm_enable_peer   adds it to an RCCL communicator (collective, host-synchronous); CLEORA_ALLGATHER_PEER then selects it.
 *   cleora_comm_create_local  a communicator WITHOUT RCCL: all-gather, all-reduce (every rank sums the contributions in rank order:
 *                             bit-identical everywhere) and broadcast over the same mappings.  Unlike RCCL it accepts several ranks on
 *                             one device, so the multi-rank loops run through this ABI on a one-GPU box.  id: CLEORA_COMM_ID_BYTES
 *                             bytes shared by the ranks of ONE communicator (cleora_comm_local_id draws them; distribute like the RCCL id).
 *   cleora_comm_register      every rank passes ITS copy of a buffer (device memory from hipMalloc / cleora_malloc, same size everywhere):
 *                             afterwards cleora_allgatherv_f32_dev on any range inside it may take the peer-direct form (collective,
 *                             host-synchronous; a no-op on a communicator without the peer transport).  cleora_comm_unregister before
 *                             freeing the buffer (collective).
 *   cleora_comm_check         CLEORA_E_RCCL if a device-side wait of this rank ever ran out of its 60 s budget (a peer died): the waits are
 *                             bounded so that a lost rank cannot hang the GPU; synchronise the streams first. */
int cleora_comm_local_id(void *id_out);
int cleora_comm_create_local(const void *id, int rank, int world, int device, cleora_comm **out);
int cleora_comm_enable_peer(cleora_comm *c);
int cleora_comm_register(cleora_comm *c, void *buf_dev, uint64_t bytes);
int cleora_comm_unregister(cleora_comm *c, void *buf_dev);
int cleora_comm_check(cleora_comm *c);
/* The peer transport's data-visibility self-test (csrc/peer.hip peer_selftest; runs by itself inside cleora_comm_enable_peer /
 * cleora_comm_create_local, in milliseconds): every CU reads a 4 KiB-per-rank probe buffer, every rank writes a fresh pattern into its
 * slot, the all-gather under test runs, a checker kernel compares every word with PLAIN loads — the load path of the SpMM's gathers.
 * If the PUSH form (plain stores into the peers' replicas) fails, every rank switches to the PULL form (system-scope loads out of the
 * peers' replicas) and that is tested; if it fails too: CLEORA_E_RCCL, the transport must not be used (an RCCL communicator keeps its
 * RCCL all-gathers: cleora_comm_set_allgather).  Collective, host-synchronous; all ranks end in the same form.
 * flags (fault injection, for tests): CLEORA_SELFTEST_FAIL_PUSH / CLEORA_SELFTEST_FAIL_PULL make the respective checker expect a pattern
 * nobody wrote.  cleora_comm_peer_mode: 0 PUSH, 1 PULL, -1 no peer transport. */
#define CLEORA_SELFTEST_FAIL_PUSH 1u
#define CLEORA_SELFTEST_FAIL_PULL 2u
int cleora_comm_selftest(cleora_comm *c, uint32_t flags);
int cleora_comm_peer_mode(const cleora_comm *c, int *mode);

/* The exchange step of the row partition: buf holds offsets[world] floats (e.g. a row range of the next iterate,
 * contiguous, ld = d); rank r has just written elements [offsets[r], offsets[r+1]) and every rank ends up with all
 * of them.  cleora_allgather_f32_dev is the equal-shard form (offsets[r] = r * elems_per_rank). */
int cleora_allgatherv_f32_dev(cleora_comm *c, float *buf, const uint64_t *offsets, void *stream);
int cleora_allgather_f32_dev(cleora_comm *c, float *buf, uint64_t elems_per_rank, void *stream);
/* Sums over ranks, in place: the row sums of squares of the column partition (f32) and the whitening statistics
 * (f64 column sums and Gram matrix, pycleora/__init__.py:136-143) under either partition. */
int cleora_allreduce_f32_dev(cleora_comm *c, float *buf, uint64_t n, void *stream);
int cleora_allreduce_f64_dev(cleora_comm *c, double *buf, uint64_t n, void *stream);
int cleora_broadcast_dev(cleora_comm *c, void *buf, uint64_t bytes, int root, void *stream);
/* recv[j * elems_per_rank ...] <- rank j's send[me * elems_per_rank ...]: the column <-> row layout switch around the
 * whitening step of the column partition. */
int cleora_alltoall_f32_dev(cleora_comm *c, const float *send, float *recv, uint64_t elems_per_rank, void *stream);

/* ---- the row-partitioned loops (csrc/sharded.hip; no reference counterpart: pycleora is single-process, src/embedding.rs:59-63) ----
 * north_star's multi-GPU layout as calls a Rust host makes through the FFI instead of re-writing the schedule: one process per GPU;
 * every rank keeps full replicas of the iterate (n_pad x d, ld = d; rows >= n zero) and owns row blocks of the CSR.  With P ranks
 * and K steps per iteration the row space is cut into P*K contiguous blocks, rank r owns blocks {k*P + r}; step k computes block
 * (k, r) straight into its slot of the next replica, then the contiguous row range of step k is all-gathered IN PLACE on a
 * communication stream while the SpMM of step k+1 runs (cleora_allgatherv_f32_dev with the communicator's algorithm).
 * comm == NULL: a world of one (no collectives). */
#define CLEORA_BALANCE_AUTO 0   /* ROWS when its heaviest block is within 3 % of the mean work (randomly permuted ids), else NNZ */
#define CLEORA_BALANCE_ROWS 1   /* equal row counts (multiples of 4 rows): the shards of a step are equal, one ncclAllGather */
#define CLEORA_BALANCE_NNZ 2    /* split on the prefix sum of edges + 1 per row: unequal shards, all-gather-v */
typedef struct cleora_sharded cleora_sharded;
typedef struct cleora_sharded_info {
    uint64_t n, n_pad;         /* entities; rows of a replica (n padded to whole blocks; the padding rows are empty) */
    uint64_t local_rows;       /* rows of this rank's blocks (padding included) */
    uint64_t local_nnz, device_bytes;
    uint32_t steps;
    int32_t rank, world, balance, has_symmetric;
} cleora_sharded_info;

/* The row boundaries of the world * steps blocks: bounds_out[world * steps + 1]; block j = rows [bounds[j], bounds[j + 1]).
 * Pure host arithmetic (no GPU): rowptr_host is the whole graph's u64[n + 1]. */
int cleora_sharded_plan(uint64_t n, const uint64_t *rowptr_host, uint32_t world, uint32_t steps, int balance, uint64_t *bounds_out,
                        uint64_t *n_pad_out, int *mode_out);
/* This rank's blocks of the WHOLE graph's CSR (every rank passes the same arrays: the graph build is deterministic).  rowptr / col /
 * val_*: host pointers, or device pointers on `device` when arrays_on_device != 0; only the rank's slices are copied, the caller
 * keeps its arrays.  val_sym may be NULL.  comm: the communicator whose rank / world decide the blocks (NULL: world of one). */
int cleora_sharded_create(cleora_comm *comm, int device, uint64_t n, uint64_t nnz, const uint64_t *rowptr, const uint32_t *col,
                          const float *val_left, const float *val_sym, int arrays_on_device, uint32_t steps, int balance,
                          cleora_sharded **out);
int cleora_sharded_destroy(cleora_sharded *s);
int cleora_sharded_get_info(const cleora_sharded *s, cleora_sharded_info *info);
int cleora_sharded_bounds(const cleora_sharded *s, uint64_t *bounds_out);           /* world * steps + 1 values */
int cleora_sharded_block(const cleora_sharded *s, uint32_t k, cleora_graph **graph, uint64_t *row_begin, uint64_t *row_end);
/* One iteration: x_next <- epilogue(A x) with the flags of cleora_propagate_dev (x_self = the same rows of x), replicated on every
 * rank when gather != 0 (else only this rank's rows of x_next are written).  row_sqdiff_local: f64[local_rows] in block order, or
 * NULL.  Enqueues on `stream` and on the handle's communication stream; returns with every collective ordered before later work on
 * `stream`.  With the peer-direct all-gather both replicas must be registered (cleora_comm_register). */
int cleora_sharded_propagate_dev(cleora_sharded *s, int markov_type, const float *x, float *x_next, uint32_t d, uint32_t flags,
                                 float residual_weight, double *row_sqdiff_local, int gather, void *stream);
/* The stream cleora_embed_sharded runs its loop on (default: the device's null stream).  Several handles driven by the threads of ONE
 * process on ONE device (csrc/multi.hip's one-GPU test form) need a stream each: on a shared null stream one rank's wait for a peer
 * would sit in front of that peer's signal. */
int cleora_sharded_set_stream(cleora_sharded *s, void *stream);
/* Timing for throughput reports: while enabled, ms[0] sums the SpMM kernels of this rank's blocks and ms[1] the all-gathers on the
 * communication stream (HIP events), over `calls` cleora_sharded_propagate_dev calls; get waits for the events and resets. */
int cleora_sharded_set_timing(cleora_sharded *s, int enable);
int cleora_sharded_get_timing(cleora_sharded *s, double ms[2], uint64_t *calls);
/* The loops over the partition, x_replica (n_pad x d): E_0 in, result out, replicated; synchronous.
 *   flags without CLEORA_F_WHITEN: embed_full / embed_full_with_convergence (src/embedding.rs:106-188) — bit-equal to the one-GPU
 *     loop (every row, hub rows included);
 *   with CLEORA_F_WHITEN: the default loop of pycleora.embed() (pycleora/__init__.py:109-117).  L2 norm and no convergence test: the
 *     reorganised form of cleora_embed (SpMM before the projection, Cholesky whitening in the intermediate iterations behind the clamp
 *     guard, normalisation in the projection's epilogue): Z = A Y on the rank's rows beside the statistics of Y (every rank one
 *     contiguous range of the replica, one pass; column sums and Gram all-reduced: d + d*d doubles), transform decided together and
 *     broadcast, one all-gather of the iterate per iteration.  Otherwise the reference's order, statistics in the reference's two passes.
 * Device memory beside the caller's replica: cleora_embed_sharded_bytes — one more replica for the plain loop; one more replica + the
 * rank's own rows + the whitening workspace for the whitened one (config 4 on 8 GPUs: 2 x 113.7 + 14.2 GB of iterates). */
/* Test hook of cleora_embed_sharded's first-use check of the peer-direct all-gather (csrc/sharded.hip verify_peer_gather: on first
 * use the loop's exchange runs once on a real replica with a pattern every rank verifies with plain loads; a failing PUSH form is
 * replaced by PULL on all ranks, a failing PULL form is an error): the next `attempts` (0, 1, 2) checks report a mismatch. */
int cleora_sharded_debug_fail_first_gather(cleora_sharded *s, int attempts);
uint64_t cleora_embed_sharded_bytes(uint64_t n_pad, uint64_t local_rows, uint64_t n, uint32_t world, uint32_t d, uint32_t flags);
int cleora_embed_sharded(cleora_sharded *s, float *x_replica, int markov_type, uint32_t d, uint64_t max_iterations,
                         float residual_weight, float convergence_threshold, uint32_t flags, uint64_t *iterations_run);

/* ---- the column partition (csrc/colsharded.hip; no reference counterpart) --------------------------------------------------------
 * The comparison layout beside the row partition: rank r of P owns columns [r d/P, (r + 1) d/P) of every row of the iterate and the
 * WHOLE CSR ("Cleora operates on dimensions independently", reference README.md:361): no n x d data crosses xGMI.  The row L2 norm
 * (src/embedding.rs:88-104: sum of squares over j = 0 .. d-1 in order) is kept bit-equal to one GPU by handing every row's running
 * sum from rank to rank (CLEORA_F_ROWSQ_CONT): P small broadcasts per row block instead of an all-reduce that would add the slices'
 * sums in another order.  x_local / x_next_local: this rank's n x d/P slice, ld = d/P.  arrays_on_device != 0: col / val_* are
 * DEVICE arrays that must outlive the handle (viewed, not copied); rowptr may then be a device pointer too (copied).
 * comm == NULL: a world of one.  d_total must be divisible by the number of ranks.
 * cleora_colsharded_propagate_dev: flags out of L2NORM, RESIDUAL, BLEND_ANY, SQDIFF, SQDIFF64, HUB_SEGMENTS; row_sqdiff receives this
 * rank's part of every row's squared difference.  cleora_embed_colsharded: embed_full / embed_full_with_convergence
 * (src/embedding.rs:106-188), the iterate bit-equal to the one-GPU loop's.  The L1 norm and the whitened loop need whole rows: take
 * the row partition (cleora_embed_sharded). */
typedef struct cleora_colsharded cleora_colsharded;
typedef struct cleora_colsharded_info {
    uint64_t n, nnz;
    uint32_t d_total, d_local, col_begin, steps;
    int32_t rank, world, has_symmetric, reserved;
} cleora_colsharded_info;
int cleora_colsharded_create(cleora_comm *comm, int device, uint64_t n, uint64_t nnz, const uint64_t *rowptr, const uint32_t *col,
                             const float *val_left, const float *val_sym, int arrays_on_device, uint32_t d_total, uint32_t steps,
                             cleora_colsharded **out);
int cleora_colsharded_destroy(cleora_colsharded *s);
int cleora_colsharded_get_info(const cleora_colsharded *s, cleora_colsharded_info *info);
/* row block k (< steps) of the handle as a graph (borrowed: for cleora_graph_get_info / set_timing / cleora_alloc_iterates) and its rows */
int cleora_colsharded_block(const cleora_colsharded *s, uint32_t k, cleora_graph **graph, uint64_t *row_begin, uint64_t *row_end);
int cleora_colsharded_propagate_dev(cleora_colsharded *s, int markov_type, const float *x_local, float *x_next_local, uint32_t flags,
                                    float residual_weight, double *row_sqdiff, void *stream);
int cleora_embed_colsharded(cleora_colsharded *s, float *x_local, int markov_type, uint64_t max_iterations, float residual_weight,
                            float convergence_threshold, uint32_t flags, uint64_t *iterations_run);

/* ---- one process, P devices (csrc/multi.hip) --------------------------------------------------------------------------------------
 * The row partition above behind ONE handle, for a host whose call is `embed(graph, 256, 40)` in one process (pycleora/__init__.py:
 * 51-127; SparseMatrix::embed_fast, src/lib.rs:320-364): SURVEY.md 8(b) B2's graph handle with (n_devices, device_ids*).  The handle
 * runs one host thread per device for the duration of a call; thread p is rank p of a local communicator (peer-direct stores; between
 * threads of one process the peers' pointers are used directly, hipDeviceEnablePeerAccess across devices) and owns its blocks of the
 * CSR (cleora_sharded_create) and a stream; the loops are cleora_embed_sharded / cleora_sharded_propagate_dev.  Every rank moves its
 * own share over its own PCIe link.  device_ids may repeat (several shards on one GPU: how a one-GPU box tests this path).
 * steps: row blocks per rank and iteration (0 = 4 for P > 1: block k's gather overlaps block k + 1's SpMM).
 * Results: the plain loop and the propagate are bit-equal to the one-GPU calls; the whitened loop agrees within its stated tolerance. */
typedef struct cleora_multi cleora_multi;
typedef struct cleora_multi_info {
    uint64_t n, nnz, n_pad;
    uint32_t world, steps;
    int32_t has_symmetric, reserved;
    int32_t device[64];
    uint64_t local_rows[64], local_nnz[64], device_bytes[64];
} cleora_multi_info;
/* rowptr / col / val_*: HOST arrays of the whole graph (borrowed for the call; every shard copies its slices).  val_sym may be NULL. */
int cleora_multi_create(const int *device_ids, uint32_t n_devices, uint64_t n, uint64_t nnz, const uint64_t *rowptr, const uint32_t *col,
                        const float *val_left, const float *val_sym, uint32_t steps, int balance, cleora_multi **out);
int cleora_multi_destroy(cleora_multi *m);
int cleora_multi_get_info(const cleora_multi *m, cleora_multi_info *info);
/* cleora_embed's contract (below) over the partition: embed_fast / embed_fast_convergence, or the default loop of pycleora.embed() with
 * CLEORA_F_WHITEN.  entity_hash_host (n u64) or x0_host (n x d) as the start; out_host: n x d.  Synchronous. */
int cleora_multi_embed(cleora_multi *m, const uint64_t *entity_hash_host, const float *x0_host, int markov_type, uint32_t d,
                       uint64_t max_iterations, int64_t seed, float residual_weight, float convergence_threshold, uint32_t flags,
                       float *out_host, uint64_t *iterations_run);
/* cleora_propagate's contract: y = A x (SparseMatrix::markov_propagate, src/lib.rs:29-47), x_host and y_host n x d. */
int cleora_multi_propagate(cleora_multi *m, int markov_type, const float *x_host, uint32_t d, float *y_host);

/* The k nearest rows of X by cosine similarity for a batch of query ROWS of X, selected on the device: the
 * `normed @ normed[src]`, the -2 masks and `argsort()[::-1][:top_k]` of predict_links / find_most_similar
 * (pycleora/__init__.py:636-681, 753-781) without a launch, a sync and an n-float download per query.
 *   score(q, r) = (x[r] . x[q]) / (max(||x[r]||, 1e-10) * max(||x[q]||, 1e-10))
 *   exclude_self: score(q, q) = -2;  exclude_existing (needs the square graph of X): -2 for every r with a stored edge
 *   (q, r) or (r, q) (:650-660).  out_index / out_score: [n_queries][k], descending score; ties: the larger row index
 *   first (numpy's argsort()[::-1]); entries with score <= -2 are masked candidates the caller drops (:663-664).
 * Up to 8 queries: one pass over X on the vector units.  More: the batch is a GEMM — X . Q (Q = the normalised query
 * rows, d x q) on the f32 matrix cores, 64 queries per pass over X, then a row scale by 1 / ||x_r||.
 * Selection: from 256 Ki rows on, a threshold from a stratified row sample, ONE pass over the scores that compacts what
 * passes it into a short list (checked on the host: the call synchronises the stream once per batch of queries), the
 * ordering on that list; below that size, or if a list came out shorter than k or longer than its buffer, k rounds of an
 * arg-max over all n scores.  Same result either way.  cleora_topk_last_route(): which one the calling thread's last call took
 * (1 = the short list in every batch, 0 = in none, 2 = in some); cleora_topk_set_route(route) forces one for the calling thread's
 * later calls (0 = automatic, the default; 1 = the rounds; 2 = the short list from 2048 rows on) — for tests and A/B runs.
 * Unlike the other *_dev entry points this one may WAIT for `stream`: the short list's lengths are checked on the host once per
 * batch of queries (64 per pass).
 * workspace: cleora_topk_workspace_for(n, k, n_queries) BYTES (8 n floats + candidates up to 8 queries, 64 n floats
 * beyond); cleora_topk_workspace(n, k) is the size that serves any batch.  1 <= k <= min(n, 1024). */
uint64_t cleora_topk_workspace(uint64_t n, uint32_t k);
uint64_t cleora_topk_workspace_for(uint64_t n, uint32_t k, uint32_t n_queries);
int cleora_topk_cosine_dev(const cleora_graph *g, const float *x, uint64_t ldx, uint64_t n, uint32_t d,
                           const uint32_t *query_rows_dev, uint32_t n_queries, uint32_t k, int exclude_self,
                           int exclude_existing, uint32_t *out_index_dev, float *out_score_dev, void *workspace, void *stream);
int cleora_topk_last_route(void);
int cleora_topk_set_route(int route);

/* ---- host-pointer entry points: what the PyO3 methods call ------------------------- */

/* SparseMatrix::markov_propagate → NdArrayMatrix::multiply (src/lib.rs:29-47, src/embedding.rs:15-39).
 * x: n_cols x d host; y: n_rows x d host (fresh output, like `to_pyarray`). */
int cleora_propagate(const cleora_graph *g, int markov_type, const float *x_host, uint32_t d,
                     float *y_host);

/* SparseMatrix::l2_normalize (src/lib.rs:414-424). */
int cleora_l2_normalize(const float *x_host, uint64_t n, uint32_t d, float *y_host);

/* SparseMatrix::initialize_deterministically (src/lib.rs:242-252). */
int cleora_init(const uint64_t *entity_hash_host, uint64_t n, uint32_t d, int64_t seed,
                float *x_host);

/* whiten_embeddings(embeddings, n_components) on host arrays (pycleora/__init__.py:130-164).
 * y_host: n x k with k = n_components, or d when n_components is 0 or >= d (n <= 1: n x d copy). */
int cleora_whiten(const float *x_host, uint64_t n, uint32_t d, uint32_t n_components, float *y_host);

/* SparseMatrix::embed_fast / embed_fast_convergence (src/lib.rs:320-412) →
 * NdArrayMatrix::embed_full / embed_full_with_convergence (src/embedding.rs:106-188).
 * Device-resident loop: init (or x0_host if non-NULL), `max_iterations` x (SpMM, residual for
 * 0 < rw < 1, L2), optional RMSE early stop (threshold > 0, from iteration 1).
 * The graph must be square (n_rows == n_cols).  out_host: n x d.  iterations_run may be NULL.
 * The iterate buffers are placed for the SpMM (cleora_alloc_iterates; the plain loop searches on its own iterations).
 * With CLEORA_F_WHITEN in `flags` the loop is the default path of pycleora.embed() instead
 * (pycleora/__init__.py:97-127 with _postprocess_iteration :963-971): every iteration is SpMM, residual blend for
 * ANY rw > 0 (:111-115 — unlike the Rust loop's 0 < rw < 1), L2 normalise (L1 with CLEORA_F_L1NORM), THEN
 * whiten_embeddings; the RMSE of the early stop is taken between whitened iterates in f64 (:122-125, :974-976).
 * Without a convergence test nobody sees the intermediate whitened iterates, and the loop is reorganised without changing
 * what it computes (docs/history.md §3.7-3.8): (1) the SpMM is linear, so A ((Y - 1 mu^T) T) is taken as (A Y - (A 1) mu^T) T — the
 * SpMM of the next iteration is enqueued beside the Gram matrix / d x d step of this one, and the projection normalises its
 * output rows in its own epilogue; (2) with the L2 norm, intermediate iterations may use ANY whitening transform (two differ by
 * an orthogonal factor that the linear steps, the rotation-invariant norm and the final PCA whitening remove): Cholesky
 * (potrf + trtri) instead of the eigensolver, falling back to it when the covariance is near-singular; the last iteration is
 * always the PCA form.  Results agree with the sequential order to f32 rounding (measured at |V| = 1M, d = 256: 5e-7 on pairwise
 * cosines after 10, 20 and 40 iterations).  The reference's order is what a convergence threshold selects (the tests pass one that is
 * never met); there are no environment switches. */
int cleora_embed(const cleora_graph *g, const uint64_t *entity_hash_host, const float *x0_host,
                 int markov_type, uint32_t d, uint64_t max_iterations, int64_t seed,
                 float residual_weight, float convergence_threshold, uint32_t flags,
                 float *out_host, uint64_t *iterations_run);

/* The same loops on a DEVICE-resident iterate: x_dev (n x d, ld = d) holds the initial embeddings on entry and the
 * result on return; nothing crosses PCIe.  Synchronous (returns when the result is in place). */
/* Wall-clock milliseconds of the ITERATION LOOP of the last cleora_embed / cleora_embed_dev on the calling thread —
 * without the allocations, the placement search and the copies around it (for throughput reporting). */
double cleora_last_embed_loop_ms(void);
int cleora_embed_dev(const cleora_graph *g, float *x_dev, int markov_type, uint32_t d, uint64_t max_iterations,
                     float residual_weight, float convergence_threshold, uint32_t flags, uint64_t *iterations_run);

#ifdef __cplusplus
}
#endif
#endif /* CLEORA_HIP_H */
