// Internal declarations shared by the translation units of libcleora_hip.so.
// Public interface: include/cleora_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/cleora_hip.h"

namespace cleora {

void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define CL_HIP(call)                                                          \
    do {                                                                      \
        hipError_t _e = (call);                                               \
        if (_e != hipSuccess) return ::cleora::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

#define CL_REQUIRE(cond, msg)                      \
    do {                                           \
        if (!(cond)) {                             \
            ::cleora::set_error(msg);              \
            return CLEORA_E_INVALID;               \
        }                                          \
    } while (0)

// A launch's x extent in work-items (blocks * 256) is a 32-bit field of the dispatch packet: a 1-D
// grid of more than 2^24 256-thread blocks is silently truncated (seen at |V| = 111M: 27.7M blocks).
// Big 1-D problems therefore use a 2-D grid of at most 2^23 x N blocks and linearise it in the kernel.
constexpr unsigned kMaxGridX = 1u << 23;
inline dim3 grid_1d_as_2d(uint64_t blocks) {
    if (blocks <= kMaxGridX) return dim3((unsigned)(blocks ? blocks : 1));
    return dim3(kMaxGridX, (unsigned)((blocks + kMaxGridX - 1) / kMaxGridX));
}
#define CLEORA_LINEAR_BLOCK() ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x)

constexpr uint32_t kDefaultHubThreshold = 256;   // edges; longer rows are scheduled first (C3 on one box: 32.28 ms at 128 / 256, 32.47 at 1024)
constexpr uint32_t kDefaultHubSegment = 256;     // edges per split segment
constexpr uint32_t kInorderMinCap = 128;         // the in-order hub launch takes every row beyond hub_threshold * this many edges (abi.hip build_hub_schedule)

}  // namespace cleora

// Device CSR shard.  Mirrors struct SparseMatrix's `edges` + `slices`
// (src/sparse_matrix.rs:56-78) as SoA streams.
struct cleora_graph {
    int device = 0;
    uint64_t n_rows = 0, n_cols = 0, nnz = 0;
    const uint64_t *rowptr = nullptr;  // [n_rows + 1]
    const uint32_t *col = nullptr;     // [nnz]
    const float *val[2] = {nullptr, nullptr};  // [nnz] per MarkovType
    bool owns_csr = false;

    // rows longer than hub_threshold: split into segments that run on separate waves
    uint32_t hub_threshold = 0, hub_segment = 0;
    uint64_t n_hub_rows = 0, n_hub_segments = 0;
    uint32_t *hub_rows = nullptr;       // [n_hub_rows]       row ids
    uint64_t *hub_seg_first = nullptr;  // [n_hub_rows + 1]   first segment of each hub row
    uint32_t *seg_row = nullptr;        // [n_hub_segments]   row id of the segment
    uint64_t *seg_begin = nullptr;      // [n_hub_segments]   first edge of the segment
    // Reference-order schedule of the long rows (the default; spmm.hip).  Rows of hub_threshold < edges <= inorder_min are the
    // FIRST work items of the main launch, longest first, one wavefront each like any row ("mid" rows: long enough to be a tail if
    // they started last, short enough for one wavefront); rows beyond inorder_min run on the in-order hub launch
    // (hub_inorder_kernel: one wavefront per 64-column slab) on the side stream, whose rows / order / scratch these are:
    std::vector<uint32_t> long_rows;    // host: every row longer than hub_threshold (ascending row id) ...
    std::vector<uint64_t> long_len;     // ... and its edge count: what cleora_graph_set_hub_inorder_min re-partitions
    uint64_t inorder_min = 0;           // rows with MORE edges go to the in-order hub launch
    uint64_t n_mid_rows = 0, n_io_rows = 0;
    uint32_t *mid_rows = nullptr;       // [n_mid_rows]  row ids, longest first
    uint32_t *io_rows = nullptr;        // [n_io_rows]   row ids (ascending): scratch row h belongs to io_rows[h]
    uint32_t *hub_by_len = nullptr;     // [n_io_rows]   indices into io_rows, longest row first
    std::vector<uint64_t> io_len_desc;  // host: the edge counts in hub_by_len's order (what hub_chain_rows() decides on)
    mutable uint64_t hub_chain_min = 0; // rows of the hub launch with at least this many edges take hub_chain_kernel: 0 = automatic
    bool hub_inorder_ok = true;         // false: some row is too long for the kernel's 32-bit (col, val) offsets
    uint64_t hub_longest = 0;           // edges of the longest row
    mutable int hub_lanes = 0;          // lanes per edge of the in-order hub launch: 0 = automatic (spmm.hip hub_lanes), else 4 / 2 / 1
    mutable hipStream_t hub_stream = nullptr;
    mutable hipEvent_t hub_fork = nullptr, hub_join = nullptr;
    uint64_t device_bytes = 0;

    // scratch for the hub rows' sums (in-order: one row each; segmented: one per segment), sized for the largest d seen so far
    mutable std::mutex mu;
    mutable float *hub_partial = nullptr;
    mutable uint64_t hub_partial_elems = 0;

    // hot-column marking for the gather cache policy (hot.hip): col with bit 31 set on the most
    // referenced rows; hot_bytes < 0 = automatic, 0 = off, > 0 = forced byte budget
    mutable int64_t hot_bytes = -1;
    mutable uint32_t *col_hot = nullptr;
    mutable uint64_t hot_rows_target = 0;
    mutable uint32_t *hot_meta = nullptr;   // device: {threshold in-degree, rows marked}
    mutable bool hot_failed = false;        // building the marks failed once (e.g. out of memory): policy stays off
    mutable uint32_t auto_launches = 0;     // automatic mode arms itself on the third eligible launch

    // device staging of the host-pointer entry points (cleora_propagate): kept between calls, grow-only
    mutable std::mutex io_mu;
    mutable void *io_buf[2] = {nullptr, nullptr};
    mutable uint64_t io_bytes[2] = {0, 0};

    // optional per-kernel timing (cleora_graph_set_timing): 4 events per propagate call,
    // recorded on the launch stream: [hub_partial | rows | hub_finish]
    mutable bool timing = false;
    mutable std::vector<hipEvent_t> ev_pool;   // recycled events
    mutable std::vector<hipEvent_t> ev_used;   // 4 per recorded call, in order
};

namespace cleora {

// spmm.hip
int launch_propagate(const cleora_graph *g, int kind, const float *x, uint64_t ldx, uint32_t d,
                     float *y, uint64_t ldy, uint32_t flags, float rw, const float *x_self,
                     double *row_sqdiff, float *row_sumsq, hipStream_t stream,
                     const float *val_override = nullptr,    // per-edge values replacing g->val[kind]
                     hipEvent_t *hub_join_out = nullptr);    // non-NULL: the in-order hub launch is NOT joined into `stream`; *hub_join_out is the
                                                             // event to wait for before the hub rows of y are read (nullptr: nothing pending)
int launch_rowops(const float *x, uint64_t ldx, uint64_t n, uint32_t d, float *y, uint64_t ldy,
                  uint32_t flags, float rw, const float *x_self, double *row_sqdiff,
                  float *row_sumsq, hipStream_t stream, uint64_t ldxs = 0);  // ldxs: leading dimension of x_self (0 = ldx)
// hot.hip
const uint32_t *ensure_hot_cols(const cleora_graph *g, uint32_t d, uint64_t ldx, hipStream_t stream);
uint64_t hot_rows_marked(const cleora_graph *g);
// rowops.hip
int launch_init(const uint64_t *hash, uint64_t n, uint32_t d, int64_t seed, float *x, uint64_t ldx,
                hipStream_t stream);
uint64_t reduce_workspace(uint64_t n);
int launch_reduce_sum(const double *v, uint64_t n, double *ws, double *out, hipStream_t stream);
uint64_t colsum_workspace(uint64_t n, uint32_t d);
int launch_colsum(const float *x, uint64_t ldx, uint64_t n, uint32_t d, double *ws, double *out,
                  hipStream_t stream);
int launch_csr_rowsum(const cleora_graph *g, int kind, float *out, hipStream_t stream, float *abs_out = nullptr);   // abs_out: sum of |values| per row
int launch_cosine(const float *x, uint64_t ldx, uint64_t n, uint32_t d, const float *q, float *scores,
                  hipStream_t stream);
// whiten.hip
uint64_t gram_workspace(uint64_t n, uint32_t d);
int launch_gram(const float *x, uint64_t ldx, uint64_t n, uint32_t d, const double *mean,
                double *ws, double *gram, hipStream_t stream, double *mean_out64 = nullptr,
                float *mean_out32 = nullptr,    // outputs given: `mean` is only a shift, the exact mean is produced
                int blocks_per_cu = 2);
// the split-bf16 form for the intermediate iterations of the whitened loop (d a multiple of 256: gram32_applies)
bool gram32_applies(const float *x, uint64_t ldx, uint64_t n, uint32_t d);
int launch_gram32(const float *x, uint64_t ldx, uint64_t n, uint32_t d, double *shift64, float *shift32, double *ws, double *gram,
                  hipStream_t stream, double *mean_out64, float *mean_out32);
// out = (alpha * (x - rowscale (x) mean) + beta * (x2 - mean)) @ t; rowscale / x2 == nullptr: the plain (x - mean) @ t
int launch_project(const float *x, uint64_t ldx, uint64_t n, uint32_t d, const float *mean,
                   const float *t, uint32_t k, float *out, uint64_t ldo, hipStream_t stream,
                   const float *rowscale = nullptr, const float *x2 = nullptr, uint64_t ldx2 = 0, float alpha = 1.0f,
                   float beta = 0.0f, int norm = 0,           // norm: 1 = L2-, 2 = L1-normalise the output rows in the epilogue ...
                   bool *norm_done = nullptr,
                   const float *rowbound = nullptr,           // != nullptr (and no x2): |x[r][j]| <= rowbound[r], |mean[j]| <= 1 — the split form's
                   bool *bounded_form = nullptr);             //   three-product f16 mode (whiten.hip); *bounded_form tells whether it ran                // ... if the shape allows (reported here); else the caller runs rowops

// project_f16.hip: the projection for BOUNDED operands (|x[r][j]| <= rowbound[r], |mean[j]| <= 1) at d = k = 256 — three f16 MFMAs per
// product, the transform resident in registers; the whitened loop's intermediate iterations
bool project_f16_applies(const float *x, uint64_t ldx, uint64_t n, uint32_t d, uint32_t k, const float *out, uint64_t ldo, const float *x2);
int launch_project_f16(const float *x, uint64_t ldx, uint64_t n, const float *mean, const float *t, float *out, uint64_t ldo,
                       hipStream_t stream, const float *rowscale, const float *rowbound, int norm);

// similarity.hip
uint64_t topk_workspace_bytes(uint64_t n, uint32_t k, uint32_t n_queries);
int launch_topk_cosine(const cleora_graph *g, const float *x, uint64_t ldx, uint64_t n, uint32_t d,
                       const uint32_t *queries_dev, uint32_t n_queries, uint32_t k, int exclude_self, int exclude_edges,
                       uint32_t *out_index, float *out_score, void *workspace, hipStream_t stream);

// stager.hip: pageable host memory <-> device through a pinned ring, at PCIe speed
int staged_h2d(void *dst_dev, const void *src_host, uint64_t bytes, hipStream_t after);
int staged_d2h(void *dst_host, const void *src_dev, uint64_t bytes, hipStream_t after);

// attention.hip
int launch_propagate_attention(const cleora_graph *g, int kind, const float *x, uint64_t ldx, uint32_t d, float temperature,
                               float *y, uint64_t ldy, uint32_t flags, float rw, const float *x_self, double *row_sqdiff,
                               hipStream_t stream);
int launch_edge_attention(const cleora_graph *g, int kind, const float *x, uint64_t ldx, uint32_t d,
                          float temperature, float *vals_out, hipStream_t stream);
// eigh.hip
uint64_t eigh_workspace(uint32_t d);
int launch_mean(const double *colsum, uint64_t n, uint32_t d, double *mean64, float *mean32, hipStream_t stream);
int launch_whiten_transform(const double *gram, uint64_t n, uint32_t d, uint32_t k, float *transform,
                            double *eigenvalues, void *workspace, hipStream_t stream);
// Cholesky form of the transform (eigh.hip): CLEORA_OK, or 1 when the covariance is not provably >= 1e-10 I (take the PCA form)
int launch_whiten_transform_cholesky(const double *gram, uint64_t n, uint32_t d, float *transform, void *workspace,
                                     hipStream_t stream, bool approximate_gram = false);
uint64_t whiten_workspace(uint64_t n, uint32_t d);
const int *whiten_info(void *workspace, uint64_t n, uint32_t d);
const int *transform_info(void *eigh_workspace, uint32_t d);
int whiten_set_timing(bool enable);
int whiten_get_timing(double ms[4], uint64_t *calls);
int launch_whiten(const float *x, uint64_t ldx, uint64_t n, uint32_t d, uint32_t k, float *y, uint64_t ldy,
                  void *workspace, double *eigenvalues, hipStream_t stream);
// the two halves of launch_whiten: statistics + eigensolver (leaves mean32 and the d x k transform in the workspace) ...
int launch_whiten_fit(const float *x, uint64_t ldx, uint64_t n, uint32_t d, uint32_t k, void *workspace,
                      double *eigenvalues, hipStream_t stream, int gram_blocks_per_cu = 2);
int launch_whiten_fit_stats(const float *x, uint64_t ldx, uint64_t n, uint32_t d, void *workspace, hipStream_t stream,
                            int gram_blocks_per_cu = 2, bool intermediate = false);
// any_whitening: the caller only needs SOME W with W^T cov W = I (intermediate iterations of the L2-normalised loop):
// Cholesky (potrf + trtri) instead of the eigensolver, falling back to it when the covariance is near-singular
// approximate_gram: the statistics in the workspace are the split-bf16 ones; *need_exact_gram is then set (and nothing solved) when
// the Cholesky form is refused — the caller recomputes the statistics with intermediate = false and calls again
int launch_whiten_fit_solve(uint64_t n, uint32_t d, uint32_t k, void *workspace, double *eigenvalues, hipStream_t stream,
                            bool any_whitening = false, bool approximate_gram = false, bool *need_exact_gram = nullptr);
// ... and their location, for a projection launched separately (launch_project)
int whiten_fit_copy_stats(void *workspace, uint64_t n, uint32_t d, double *mean64, double *gram, hipStream_t stream);
void whiten_fit_result(void *workspace, uint64_t n, uint32_t d, const float **mean32, const float **transform);

}  // namespace cleora
