// abi.hip — the extern "C" surface of libcleora_hip.so (include/cleora_hip.h):
// argument checking, graph handles, and the host-pointer entry points that the
// reference's PyO3 methods (src/lib.rs) would call through FFI.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "common.h"

namespace cleora {

static thread_local std::string g_last_error;
static thread_local double g_last_loop_ms = 0.0;   // iteration loop of the last cleora_embed* on this thread

void set_error(const std::string &msg) { g_last_error = msg; }

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    g_last_error = std::string(what) + " failed: " + hipGetErrorString(e) + " (" + file + ":" +
                   std::to_string(line) + ")";
    if (e == hipErrorOutOfMemory) return CLEORA_E_OOM;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return CLEORA_E_NODEVICE;
    return CLEORA_E_HIP;
}

namespace {

inline hipStream_t S(void *s) { return reinterpret_cast<hipStream_t>(s); }

// RAII device buffer for the host-pointer entry points.
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    int alloc(uint64_t bytes) {
        CL_HIP(hipMalloc(&p, bytes ? bytes : 1));
        return CLEORA_OK;
    }
    template <class T> T *as() { return static_cast<T *>(p); }
};

int require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error(std::string("no HIP device available (") + hipGetErrorString(e) +
                  "); libcleora_hip has no CPU fallback");
        return CLEORA_E_NODEVICE;
    }
    return CLEORA_OK;
}

// The reference-order schedule of the long rows (common.h): mid rows -> first items of the main launch, longest first; rows beyond
// inorder_min -> the in-order hub launch, longest first.  Rebuilt by cleora_graph_set_hub_inorder_min (host lists, no rowptr needed).
int build_inorder_tables(cleora_graph *g) {
    std::vector<uint32_t> mid, io;
    std::vector<uint64_t> mid_len, io_len;
    for (size_t i = 0; i < g->long_rows.size(); ++i) {
        if (g->long_len[i] > g->inorder_min) { io.push_back(g->long_rows[i]); io_len.push_back(g->long_len[i]); }
        else { mid.push_back(g->long_rows[i]); mid_len.push_back(g->long_len[i]); }
    }
    // longest first (ties by row id: deterministic)
    std::vector<uint32_t> mid_order(mid.size()), io_order(io.size());
    for (size_t i = 0; i < mid_order.size(); ++i) mid_order[i] = (uint32_t)i;
    for (size_t i = 0; i < io_order.size(); ++i) io_order[i] = (uint32_t)i;
    std::sort(mid_order.begin(), mid_order.end(), [&](uint32_t x, uint32_t y) { return mid_len[x] != mid_len[y] ? mid_len[x] > mid_len[y] : x < y; });
    std::sort(io_order.begin(), io_order.end(), [&](uint32_t x, uint32_t y) { return io_len[x] != io_len[y] ? io_len[x] > io_len[y] : x < y; });
    std::vector<uint32_t> mid_sorted(mid.size());
    for (size_t i = 0; i < mid.size(); ++i) mid_sorted[i] = mid[mid_order[i]];
    (void)hipFree(g->mid_rows);
    (void)hipFree(g->io_rows);
    (void)hipFree(g->hub_by_len);
    g->mid_rows = g->io_rows = g->hub_by_len = nullptr;
    g->n_mid_rows = mid.size();
    g->n_io_rows = io.size();
    g->io_len_desc.resize(io.size());
    for (size_t i = 0; i < io.size(); ++i) g->io_len_desc[i] = io_len[io_order[i]];
    auto up = [&](uint32_t **dst, const std::vector<uint32_t> &v) -> int {
        if (v.empty()) return CLEORA_OK;
        CL_HIP(hipMalloc(reinterpret_cast<void **>(dst), v.size() * sizeof(uint32_t)));
        CL_HIP(hipMemcpy(*dst, v.data(), v.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        return CLEORA_OK;
    };
    int rc;
    if ((rc = up(&g->mid_rows, mid_sorted)) != CLEORA_OK || (rc = up(&g->io_rows, io)) != CLEORA_OK || (rc = up(&g->hub_by_len, io_order)) != CLEORA_OK) return rc;
    return CLEORA_OK;
}

// Finds hub rows from a host copy of rowptr and uploads the split schedule.
int build_hub_schedule(cleora_graph *g, const uint64_t *rowptr_host) {
    std::vector<uint32_t> hub_rows, seg_row;
    std::vector<uint64_t> hub_seg_first, seg_begin;
    for (uint64_t r = 0; r < g->n_rows; ++r) {
        const uint64_t b = rowptr_host[r], e = rowptr_host[r + 1];
        if (e - b > g->hub_threshold) {
            hub_rows.push_back((uint32_t)r);
            hub_seg_first.push_back(seg_row.size());
            for (uint64_t s = b; s < e; s += g->hub_segment) {
                seg_row.push_back((uint32_t)r);
                seg_begin.push_back(s);
            }
        }
    }
    hub_seg_first.push_back(seg_row.size());
    g->long_rows = hub_rows;
    g->long_len.resize(hub_rows.size());
    uint64_t longest = 0;
    for (size_t i = 0; i < hub_rows.size(); ++i) {
        g->long_len[i] = rowptr_host[hub_rows[i] + 1] - rowptr_host[hub_rows[i]];
        longest = std::max(longest, g->long_len[i]);
    }
    g->hub_inorder_ok = longest < (1ull << 30) - 4096;
    g->hub_longest = longest;
    if (g->inorder_min == 0) {
        // A long row that stays in the main launch is ONE wavefront's work: ~3.3 GB/s of gathers beside a saturated memory system, so
        // len * d * 4 B / 3.3 GB/s of time, against ~nnz * d * 4 B / 6.4 TB/s for the whole launch.  Started first it is harmless up to a
        // quarter of that: len <= nnz / 8192 (config 3: 24 k edges, config 5: capped; config 2: 2.4 k; a 6 M-edge block of an 8-way
        // partition: none — every long row goes to the hub launch).  Capped at 128 x hub_threshold = 32 768: beyond it the hub launch's column
        // slabs are the better shape.  Measured on one box (scripts/r05/hub_lanes_probe.py): C5 193.3 ms with every long row on the
        // hub launch, 187.5 at 8 k, 186.6 at 32 k, segments 187.2; C3 32.71 / 32.68 / 32.68 / 32.55.
        const uint64_t by_size = g->nnz / 8192, cap = (uint64_t)g->hub_threshold * kInorderMinCap;
        g->inorder_min = std::max<uint64_t>(g->hub_threshold, std::min(by_size, cap));
    }
    g->n_hub_rows = hub_rows.size();
    g->n_hub_segments = seg_row.size();
    if (g->n_hub_rows == 0) return CLEORA_OK;
    auto up = [&](auto **dst, const auto &v) -> int {
        const size_t bytes = v.size() * sizeof(v[0]);
        CL_HIP(hipMalloc(reinterpret_cast<void **>(dst), bytes));
        CL_HIP(hipMemcpy(*dst, v.data(), bytes, hipMemcpyHostToDevice));
        g->device_bytes += bytes;
        return CLEORA_OK;
    };
    int rc;
    if ((rc = up(&g->hub_rows, hub_rows)) != CLEORA_OK) return rc;
    if ((rc = build_inorder_tables(g)) != CLEORA_OK) return rc;
    if ((rc = up(&g->hub_seg_first, hub_seg_first)) != CLEORA_OK) return rc;
    if ((rc = up(&g->seg_row, seg_row)) != CLEORA_OK) return rc;
    if ((rc = up(&g->seg_begin, seg_begin)) != CLEORA_OK) return rc;
    return CLEORA_OK;
}

int check_csr_host(uint64_t n_rows, uint64_t n_cols, uint64_t nnz, const uint64_t *rowptr,
                   const uint32_t *col) {
    CL_REQUIRE(rowptr[0] == 0 && rowptr[n_rows] == nnz, "rowptr[0] != 0 or rowptr[n_rows] != nnz");
    for (uint64_t r = 0; r < n_rows; ++r)
        CL_REQUIRE(rowptr[r] <= rowptr[r + 1], "rowptr is not non-decreasing");
    if (col)
        for (uint64_t k = 0; k < nnz; ++k) CL_REQUIRE(col[k] < n_cols, "column index out of range");
    return CLEORA_OK;
}

// Device-side bounds check of adopted column indices (cleora_graph_create_dev cannot afford a host
// copy of nnz entries): an out-of-range index would otherwise turn into a wild gather.
__global__ __launch_bounds__(256) void max_u32_kernel(const uint32_t *__restrict__ v, uint64_t n,
                                                      uint32_t *__restrict__ out) {
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        m = v[i] > m ? v[i] : m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t t = (uint32_t)__shfl_xor((int)m, o, 64);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

int check_cols_dev(const uint32_t *col_dev, uint64_t nnz, uint64_t n_cols) {
    if (nnz == 0) return CLEORA_OK;
    uint32_t *d_max = nullptr, h_max = 0;
    CL_HIP(hipMalloc(&d_max, sizeof(uint32_t)));
    hipError_t e = hipMemset(d_max, 0, sizeof(uint32_t));
    if (e == hipSuccess) {
        const uint64_t want = (nnz + 255) / 256;
        hipLaunchKernelGGL(max_u32_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, nullptr, col_dev, nnz, d_max);
        e = hipMemcpy(&h_max, d_max, sizeof(uint32_t), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_max);
    CL_HIP(e);
    CL_REQUIRE((uint64_t)h_max < n_cols, "column index out of range");
    return CLEORA_OK;
}

void free_graph(cleora_graph *g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->owns_csr) {
        (void)hipFree(const_cast<uint64_t *>(g->rowptr));
        (void)hipFree(const_cast<uint32_t *>(g->col));
        (void)hipFree(const_cast<float *>(g->val[0]));
        (void)hipFree(const_cast<float *>(g->val[1]));
    }
    (void)hipFree(g->hub_rows);
    (void)hipFree(g->hub_by_len);
    (void)hipFree(g->mid_rows);
    (void)hipFree(g->io_rows);
    if (g->hub_stream) {
        (void)hipStreamSynchronize(g->hub_stream);
        (void)hipStreamDestroy(g->hub_stream);
        (void)hipEventDestroy(g->hub_fork);
        (void)hipEventDestroy(g->hub_join);
    }
    (void)hipFree(g->hub_seg_first);
    (void)hipFree(g->seg_row);
    (void)hipFree(g->seg_begin);
    (void)hipFree(g->hub_partial);
    (void)hipFree(g->col_hot);
    (void)hipFree(g->hot_meta);
    (void)hipFree(g->io_buf[0]);
    (void)hipFree(g->io_buf[1]);
    for (hipEvent_t e : g->ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : g->ev_used) (void)hipEventDestroy(e);
    delete g;
}

}  // namespace
int topk_last_route();        // similarity.hip: which selection the last top-k call of this thread took
int topk_set_route(int route);
}  // namespace cleora

using namespace cleora;

extern "C" {

int cleora_abi_version(void) { return CLEORA_ABI_VERSION; }

const char *cleora_last_error(void) { return g_last_error.c_str(); }

double cleora_last_embed_loop_ms(void) { return g_last_loop_ms; }

int cleora_device_count(int *count) {
    CL_REQUIRE(count != nullptr, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return CLEORA_OK;
}

int cleora_set_device(int device) {
    CL_HIP(hipSetDevice(device));
    return CLEORA_OK;
}

int cleora_malloc(uint64_t bytes, void **dev_ptr) {
    CL_REQUIRE(dev_ptr != nullptr, "dev_ptr is NULL");
    CL_HIP(hipMalloc(dev_ptr, bytes ? bytes : 1));
    return CLEORA_OK;
}

int cleora_free(void *dev_ptr) {
    if (dev_ptr) CL_HIP(hipFree(dev_ptr));
    return CLEORA_OK;
}

// (synchronous with respect to the host; ordered after the work already enqueued on `stream`; pageable host memory
// goes through the pinned pipeline of stager.hip)
int cleora_memcpy_h2d(void *dst, const void *src, uint64_t bytes, void *stream) {
    CL_REQUIRE((dst != nullptr && src != nullptr) || bytes == 0, "dst / src is NULL");
    return staged_h2d(dst, src, bytes, S(stream));
}

int cleora_memcpy_d2h(void *dst, const void *src, uint64_t bytes, void *stream) {
    CL_REQUIRE((dst != nullptr && src != nullptr) || bytes == 0, "dst / src is NULL");
    return staged_d2h(dst, src, bytes, S(stream));
}

int cleora_memcpy_d2d(void *dst, const void *src, uint64_t bytes, void *stream) {
    CL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, S(stream)));
    return CLEORA_OK;
}

int cleora_memset(void *dst, int value, uint64_t bytes, void *stream) {
    CL_HIP(hipMemsetAsync(dst, value, bytes, S(stream)));
    return CLEORA_OK;
}

int cleora_stream_sync(void *stream) {
    CL_HIP(hipStreamSynchronize(S(stream)));
    return CLEORA_OK;
}

int cleora_graph_create(int device, uint64_t n_rows, uint64_t n_cols, uint64_t nnz,
                        const uint64_t *rowptr, const uint32_t *col, const float *val_left,
                        const float *val_sym, uint32_t hub_threshold, uint32_t hub_segment,
                        cleora_graph **out) {
    CL_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    CL_REQUIRE(rowptr != nullptr, "rowptr is NULL");
    CL_REQUIRE(nnz == 0 || (col != nullptr && val_left != nullptr), "col / val_left is NULL");
    CL_REQUIRE(n_rows < (1ull << 32) && n_cols <= (1ull << 32), "more than 2^32 entities (col is u32)");
    int rc = check_csr_host(n_rows, n_cols, nnz, rowptr, col);
    if (rc != CLEORA_OK) return rc;
    if ((rc = require_device()) != CLEORA_OK) return rc;
    CL_HIP(hipSetDevice(device));

    cleora_graph *g = new (std::nothrow) cleora_graph();
    if (!g) {
        set_error("host allocation failed");
        return CLEORA_E_OOM;
    }
    g->device = device;
    g->n_rows = n_rows;
    g->n_cols = n_cols;
    g->nnz = nnz;
    g->owns_csr = true;
    g->hub_threshold = hub_threshold ? hub_threshold : kDefaultHubThreshold;
    g->hub_segment = hub_segment ? hub_segment : kDefaultHubSegment;

    auto up = [&](auto **dst, const auto *src, uint64_t count) -> int {
        const uint64_t bytes = count * sizeof(*src);
        void *p = nullptr;
        CL_HIP(hipMalloc(&p, bytes ? bytes : 1));
        *dst = static_cast<std::remove_reference_t<decltype(**dst)> *>(p);
        if (bytes) {
            const int r = staged_h2d(p, src, bytes, nullptr);
            if (r != CLEORA_OK) return r;
        }
        g->device_bytes += bytes;
        return CLEORA_OK;
    };
    uint64_t *d_rowptr = nullptr;
    uint32_t *d_col = nullptr;
    float *d_vl = nullptr, *d_vs = nullptr;
    rc = up(&d_rowptr, rowptr, n_rows + 1);
    g->rowptr = d_rowptr;
    if (rc == CLEORA_OK) { rc = up(&d_col, col, nnz); g->col = d_col; }
    if (rc == CLEORA_OK) { rc = up(&d_vl, val_left, nnz); g->val[0] = d_vl; }
    if (rc == CLEORA_OK && val_sym) { rc = up(&d_vs, val_sym, nnz); g->val[1] = d_vs; }
    if (rc == CLEORA_OK) rc = build_hub_schedule(g, rowptr);
    if (rc != CLEORA_OK) {
        free_graph(g);
        return rc;
    }
    *out = g;
    return CLEORA_OK;
}

int cleora_graph_create_dev(int device, uint64_t n_rows, uint64_t n_cols, uint64_t nnz,
                            const uint64_t *rowptr_dev, const uint32_t *col_dev,
                            const float *val_left_dev, const float *val_sym_dev,
                            uint32_t hub_threshold, uint32_t hub_segment, cleora_graph **out) {
    CL_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    CL_REQUIRE(rowptr_dev != nullptr, "rowptr is NULL");
    CL_REQUIRE(nnz == 0 || (col_dev != nullptr && val_left_dev != nullptr), "col / val_left is NULL");
    CL_REQUIRE(n_rows < (1ull << 32) && n_cols <= (1ull << 32), "more than 2^32 entities (col is u32)");
    int rc = require_device();
    if (rc != CLEORA_OK) return rc;
    CL_HIP(hipSetDevice(device));
    std::vector<uint64_t> rp(n_rows + 1);
    CL_HIP(hipMemcpy(rp.data(), rowptr_dev, (n_rows + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if ((rc = check_csr_host(n_rows, n_cols, nnz, rp.data(), nullptr)) != CLEORA_OK) return rc;
    if ((rc = check_cols_dev(col_dev, nnz, n_cols)) != CLEORA_OK) return rc;

    cleora_graph *g = new (std::nothrow) cleora_graph();
    if (!g) {
        set_error("host allocation failed");
        return CLEORA_E_OOM;
    }
    g->device = device;
    g->n_rows = n_rows;
    g->n_cols = n_cols;
    g->nnz = nnz;
    g->owns_csr = false;
    g->rowptr = rowptr_dev;
    g->col = col_dev;
    g->val[0] = val_left_dev;
    g->val[1] = val_sym_dev;
    g->hub_threshold = hub_threshold ? hub_threshold : kDefaultHubThreshold;
    g->hub_segment = hub_segment ? hub_segment : kDefaultHubSegment;
    if ((rc = build_hub_schedule(g, rp.data())) != CLEORA_OK) {
        free_graph(g);
        return rc;
    }
    *out = g;
    return CLEORA_OK;
}

int cleora_graph_destroy(cleora_graph *g) {
    free_graph(g);
    return CLEORA_OK;
}

int cleora_graph_get_info(const cleora_graph *g, cleora_graph_info *info) {
    CL_REQUIRE(g != nullptr && info != nullptr, "graph / info is NULL");
    info->n_rows = g->n_rows;
    info->n_cols = g->n_cols;
    info->nnz = g->nnz;
    info->n_hub_rows = g->n_hub_rows;
    info->n_hub_segments = g->n_hub_segments;
    {
        std::lock_guard<std::mutex> lock(g->mu);
        info->device_bytes = g->device_bytes + (g->col_hot ? g->nnz * sizeof(uint32_t) : 0);
        info->hot_rows = hot_rows_marked(g);
    }
    info->hub_threshold = g->hub_threshold;
    info->hub_segment = g->hub_segment;
    info->n_inorder_rows = g->n_io_rows;
    info->hub_inorder_min = g->inorder_min;
    info->device = g->device;
    info->has_symmetric = g->val[1] != nullptr;
    return CLEORA_OK;
}

int cleora_graph_set_hub_inorder_min(cleora_graph *g, uint64_t min_edges) {
    CL_REQUIRE(g != nullptr, "graph handle is NULL");
    std::lock_guard<std::mutex> lock(g->mu);
    CL_HIP(hipSetDevice(g->device));
    CL_HIP(hipDeviceSynchronize());                       // launches that still read the tables being replaced
    g->inorder_min = min_edges > g->hub_threshold ? min_edges : g->hub_threshold;
    return build_inorder_tables(g);
}

int cleora_graph_set_hub_lanes(cleora_graph *g, int lanes) {
    CL_REQUIRE(g != nullptr, "graph handle is NULL");
    CL_REQUIRE(lanes == 0 || lanes == 2 || lanes == 4, "lanes per edge: 0 (automatic), 4 or 2");
    std::lock_guard<std::mutex> lock(g->mu);
    g->hub_lanes = lanes;
    return CLEORA_OK;
}

int cleora_graph_set_hub_chain_min(cleora_graph *g, uint64_t min_edges) {
    CL_REQUIRE(g != nullptr, "graph handle is NULL");
    std::lock_guard<std::mutex> lock(g->mu);
    g->hub_chain_min = min_edges;
    return CLEORA_OK;
}

int cleora_graph_set_hot_cache(cleora_graph *g, int64_t hot_bytes) {
    CL_REQUIRE(g != nullptr, "graph handle is NULL");
    std::lock_guard<std::mutex> lock(g->mu);
    g->hot_bytes = hot_bytes;
    g->hot_failed = false;
    g->hot_rows_target = 0;   // rebuild (or drop) the marks on the next launch
    if (hot_bytes == 0 && g->col_hot) {
        (void)hipFree(g->col_hot);
        g->col_hot = nullptr;
    }
    return CLEORA_OK;
}

int cleora_graph_set_timing(cleora_graph *g, int enable) {
    CL_REQUIRE(g != nullptr, "graph handle is NULL");
    std::lock_guard<std::mutex> lock(g->mu);
    g->timing = enable != 0;
    return CLEORA_OK;
}

int cleora_graph_get_timing(cleora_graph *g, double ms[3], uint64_t *calls) {
    CL_REQUIRE(g != nullptr && ms != nullptr && calls != nullptr, "graph / ms / calls is NULL");
    std::lock_guard<std::mutex> lock(g->mu);
    ms[0] = ms[1] = ms[2] = 0.0;
    *calls = g->ev_used.size() / 4;
    if (!g->ev_used.empty()) CL_HIP(hipEventSynchronize(g->ev_used.back()));
    for (size_t i = 0; i + 3 < g->ev_used.size(); i += 4) {
        for (int k = 0; k < 3; ++k) {
            float t = 0.f;
            CL_HIP(hipEventElapsedTime(&t, g->ev_used[i + k], g->ev_used[i + k + 1]));
            ms[k] += (double)t;
        }
    }
    g->ev_pool.insert(g->ev_pool.end(), g->ev_used.begin(), g->ev_used.end());
    g->ev_used.clear();
    return CLEORA_OK;
}

int cleora_propagate_dev(const cleora_graph *g, int markov_type, const float *x, uint64_t ldx,
                         uint32_t d, float *y, uint64_t ldy, uint32_t flags,
                         float residual_weight, const float *x_self, double *row_sqdiff,
                         float *row_sumsq, void *stream) {
    return launch_propagate(g, markov_type, x, ldx, d, y, ldy, flags, residual_weight, x_self,
                            row_sqdiff, row_sumsq, S(stream));
}

int cleora_propagate_vals_dev(const cleora_graph *g, const float *edge_vals_dev, const float *x, uint64_t ldx,
                              uint32_t d, float *y, uint64_t ldy, uint32_t flags, float residual_weight,
                              const float *x_self, double *row_sqdiff, float *row_sumsq, void *stream) {
    CL_REQUIRE(edge_vals_dev != nullptr, "edge_vals is NULL");
    return launch_propagate(g, CLEORA_LEFT, x, ldx, d, y, ldy, flags, residual_weight, x_self, row_sqdiff,
                            row_sumsq, S(stream), edge_vals_dev);
}

int cleora_propagate_attention_dev(const cleora_graph *g, int markov_type, const float *x, uint64_t ldx, uint32_t d,
                                   float temperature, float *y, uint64_t ldy, uint32_t flags, float residual_weight,
                                   const float *x_self, double *row_sqdiff, void *stream) {
    return launch_propagate_attention(g, markov_type, x, ldx, d, temperature, y, ldy, flags, residual_weight, x_self, row_sqdiff,
                                      S(stream));
}

int cleora_edge_attention_dev(const cleora_graph *g, int markov_type, const float *x, uint64_t ldx, uint32_t d,
                              float temperature, float *edge_vals_out_dev, void *stream) {
    return launch_edge_attention(g, markov_type, x, ldx, d, temperature, edge_vals_out_dev, S(stream));
}

int cleora_rowops_dev(const float *x, uint64_t ldx, uint64_t n, uint32_t d, float *y, uint64_t ldy,
                      uint32_t flags, float residual_weight, const float *x_self,
                      double *row_sqdiff, float *row_sumsq, void *stream) {
    return launch_rowops(x, ldx, n, d, y, ldy, flags, residual_weight, x_self, row_sqdiff, row_sumsq,
                         S(stream));
}

int cleora_init_dev(const uint64_t *entity_hash_dev, uint64_t n, uint32_t d, int64_t seed,
                    float *x, uint64_t ldx, void *stream) {
    return launch_init(entity_hash_dev, n, d, seed, x, ldx, S(stream));
}

uint64_t cleora_reduce_workspace(uint64_t n) { return reduce_workspace(n); }

int cleora_reduce_sum_f64_dev(const double *v, uint64_t n, double *workspace, double *out_dev,
                              void *stream) {
    return launch_reduce_sum(v, n, workspace, out_dev, S(stream));
}

uint64_t cleora_colsum_workspace(uint64_t n, uint32_t d) { return colsum_workspace(n, d); }

int cleora_colsum_dev(const float *x, uint64_t ldx, uint64_t n, uint32_t d, double *workspace,
                      double *colsum_dev, void *stream) {
    return launch_colsum(x, ldx, n, d, workspace, colsum_dev, S(stream));
}

int cleora_cosine_scores_dev(const float *x, uint64_t ldx, uint64_t n, uint32_t d, const float *query_dev,
                             float *scores_dev, void *stream) {
    return launch_cosine(x, ldx, n, d, query_dev, scores_dev, S(stream));
}

uint64_t cleora_topk_workspace(uint64_t n, uint32_t k) { return topk_workspace_bytes(n, k, ~0u); }   // any batch size

uint64_t cleora_topk_workspace_for(uint64_t n, uint32_t k, uint32_t n_queries) { return topk_workspace_bytes(n, k, n_queries); }

int cleora_topk_cosine_dev(const cleora_graph *g, const float *x, uint64_t ldx, uint64_t n, uint32_t d,
                           const uint32_t *query_rows_dev, uint32_t n_queries, uint32_t k, int exclude_self,
                           int exclude_existing, uint32_t *out_index_dev, float *out_score_dev, void *workspace, void *stream) {
    return launch_topk_cosine(g, x, ldx, n, d, query_rows_dev, n_queries, k, exclude_self, exclude_existing, out_index_dev,
                              out_score_dev, workspace, S(stream));
}

int cleora_topk_last_route(void) { return topk_last_route(); }
int cleora_topk_set_route(int route) {
    CL_REQUIRE(topk_set_route(route) == 0, "route must be 0 (automatic), 1 (selection rounds) or 2 (short list)");
    return CLEORA_OK;
}

uint64_t cleora_gram_workspace(uint64_t n, uint32_t d) { return gram_workspace(n, d); }

int cleora_centered_gram_dev(const float *x, uint64_t ldx, uint64_t n, uint32_t d,
                             const double *mean_dev, double *workspace, double *gram_dev,
                             void *stream) {
    return launch_gram(x, ldx, n, d, mean_dev, workspace, gram_dev, S(stream));
}

int cleora_project_dev(const float *x, uint64_t ldx, uint64_t n, uint32_t d,
                       const float *mean_f32_dev, const float *transform_dev, uint32_t k,
                       float *out, uint64_t ldo, void *stream) {
    return launch_project(x, ldx, n, d, mean_f32_dev, transform_dev, k, out, ldo, S(stream));
}

int cleora_mean_dev(const double *colsum_dev, uint64_t n, uint32_t d, double *mean64_dev,
                    float *mean32_dev, void *stream) {
    return launch_mean(colsum_dev, n, d, mean64_dev, mean32_dev, S(stream));
}

uint64_t cleora_eigh_workspace(uint32_t d) { return eigh_workspace(d); }

int cleora_whiten_transform_dev(const double *gram_dev, uint64_t n, uint32_t d, uint32_t k,
                                float *transform_dev, double *eigenvalues_dev, void *workspace,
                                void *stream) {
    return launch_whiten_transform(gram_dev, n, d, k, transform_dev, eigenvalues_dev, workspace, S(stream));
}

int cleora_project_general_dev(const float *x, uint64_t ldx, uint64_t n, uint32_t d, const float *mean_f32_dev,
                               const float *transform_dev, uint32_t k, float *out, uint64_t ldo,
                               const float *rowscale_dev, const float *x2, uint64_t ldx2, float alpha, float beta,
                               int norm, int *norm_done, void *stream) {
    CL_REQUIRE(norm >= 0 && norm <= 2, "norm must be 0, 1 or 2");
    bool done = false;
    const int rc = launch_project(x, ldx, n, d, mean_f32_dev, transform_dev, k, out, ldo, S(stream), rowscale_dev, x2, ldx2, alpha,
                                  beta, norm, &done);
    if (norm_done) *norm_done = done ? 1 : 0;
    return rc;
}

int cleora_csr_rowsums_dev(const cleora_graph *g, int markov_type, float *rowsum_dev, float *rowabs_dev, void *stream) {
    CL_REQUIRE(g != nullptr && rowsum_dev != nullptr, "graph / rowsum is NULL");
    CL_REQUIRE(markov_type == CLEORA_LEFT || markov_type == CLEORA_SYMMETRIC, "unknown markov_type");
    CL_REQUIRE(g->val[markov_type] != nullptr, "graph has no values for this markov_type");
    return launch_csr_rowsum(g, markov_type, rowsum_dev, S(stream), rowabs_dev);
}

int cleora_project_bounded_dev(const float *x, uint64_t ldx, uint64_t n, uint32_t d, const float *mean_f32_dev,
                               const float *transform_dev, uint32_t k, float *out, uint64_t ldo, const float *rowscale_dev,
                               const float *rowbound_dev, int norm, int *norm_done, int *form, void *stream) {
    CL_REQUIRE(norm >= 0 && norm <= 2, "norm must be 0, 1 or 2");
    CL_REQUIRE(d > 0 && k > 0 && ldx >= d && ldo >= k, "bad d / k / leading dimension");
    CL_REQUIRE(x != nullptr && mean_f32_dev != nullptr && transform_dev != nullptr && out != nullptr, "x / mean / transform / out is NULL");
    CL_REQUIRE((const void *)x != (const void *)out, "x and out must not alias");
    if (form) *form = 0;
    if (n > 0 && project_f16_applies(x, ldx, n, d, k, out, ldo, nullptr)) {
        if (form) *form = 1;
        if (norm_done) *norm_done = norm != 0;
        return launch_project_f16(x, ldx, n, mean_f32_dev, transform_dev, out, ldo, S(stream), rowscale_dev, rowbound_dev, norm);
    }
    // other shapes: with row bounds the split form's three-product f16 mode (*form = 2; d a multiple of 32), else its six bf16 products
    bool done = false, bounded = false;
    const int rc = launch_project(x, ldx, n, d, mean_f32_dev, transform_dev, k, out, ldo, S(stream), rowscale_dev, nullptr, 0, 1.0f, 0.0f,
                                  norm, &done, rowbound_dev, &bounded);
    if (norm_done) *norm_done = done ? 1 : 0;
    if (form && bounded) *form = 2;
    return rc;
}

int cleora_csr_rowsum_dev(const cleora_graph *g, int markov_type, float *rowsum_dev, void *stream) {
    CL_REQUIRE(g != nullptr && rowsum_dev != nullptr, "graph / rowsum is NULL");
    CL_REQUIRE(markov_type == CLEORA_LEFT || markov_type == CLEORA_SYMMETRIC, "unknown markov_type");
    CL_REQUIRE(g->val[markov_type] != nullptr, "graph has no values for this markov_type");
    return launch_csr_rowsum(g, markov_type, rowsum_dev, S(stream));
}

int cleora_whiten_stats_dev(const float *x, uint64_t ldx, uint64_t n, uint32_t d, void *workspace, int intermediate,
                            double *mean64_dev, double *gram_dev, void *stream) {
    CL_REQUIRE(mean64_dev != nullptr && gram_dev != nullptr, "mean / gram is NULL");
    const int rc = launch_whiten_fit_stats(x, ldx, n, d, workspace, S(stream), 2, intermediate != 0);
    if (rc != CLEORA_OK) return rc;
    return whiten_fit_copy_stats(workspace, n, d, mean64_dev, gram_dev, S(stream));
}

int cleora_whiten_transform_any_dev(const double *gram_dev, uint64_t n, uint32_t d, float *transform_dev,
                                    void *workspace, void *stream, int *form_out) {
    int rc = launch_whiten_transform_cholesky(gram_dev, n, d, transform_dev, workspace, S(stream));
    if (rc < 0) return rc;
    if (form_out) *form_out = rc == CLEORA_OK ? 1 : 0;
    if (rc == 1) rc = launch_whiten_transform(gram_dev, n, d, d, transform_dev, nullptr, workspace, S(stream));
    return rc;
}

uint64_t cleora_whiten_workspace(uint64_t n, uint32_t d) { return whiten_workspace(n, d); }

int cleora_whiten_dev(const float *x, uint64_t ldx, uint64_t n, uint32_t d, uint32_t n_components,
                      float *y, uint64_t ldy, void *workspace, double *eigenvalues_dev, void *stream) {
    return launch_whiten(x, ldx, n, d, n_components, y, ldy, workspace, eigenvalues_dev, S(stream));
}

// ---- iterate buffers placed for the SpMM ---------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void fill_pattern_kernel(float *__restrict__ x, uint64_t elems) {
    // finite, sign-mixed values in (-1, 1): the SpMM's speed does not depend on them, NaNs would only look bad
    for (uint64_t i = CLEORA_LINEAR_BLOCK() * 256 + threadIdx.x; i < elems; i += (uint64_t)gridDim.x * gridDim.y * 256) {
        uint32_t h = (uint32_t)i * 2654435761u + (uint32_t)(i >> 32) * 40503u;
        h ^= h >> 15;
        x[i] = (float)((int)(h & 0xffffu) - 32768) * (1.0f / 32768.0f);
    }
}
}  // namespace

namespace {
// iterations_hint: how many SpMM launches the caller is about to run on these buffers (0 = unknown: search)
int alloc_iterates(const cleora_graph *g, uint32_t d, uint32_t count, uint64_t iterations_hint, void **bufs, double *ms) {
    CL_REQUIRE(g != nullptr && bufs != nullptr, "graph / bufs is NULL");
    CL_REQUIRE(d > 0 && count >= 1 && count <= 8, "need d > 0 and 1 <= count <= 8");
    for (uint32_t i = 0; i < count; ++i) bufs[i] = nullptr;
    if (ms) ms[0] = ms[1] = 0.0;
    CL_HIP(hipSetDevice(g->device));
    const uint64_t rows = g->n_rows > g->n_cols ? g->n_rows : g->n_cols;
    const uint64_t bytes = rows * (uint64_t)d * sizeof(float);
    auto fail = [&](int rc) {
        for (uint32_t i = 0; i < count; ++i) { if (bufs[i]) (void)hipFree(bufs[i]); bufs[i] = nullptr; }
        return rc;
    };
    if (hipMalloc(&bufs[0], bytes ? bytes : 1) != hipSuccess) { (void)hipGetLastError(); set_error("out of device memory for the iterates"); return fail(CLEORA_E_OOM); }
    const bool tune = count >= 2 && bytes >= (256ull << 20) && g->nnz > 0 && g->val[0] != nullptr;
    if (!tune) {
        for (uint32_t i = 1; i < count; ++i)
            if (hipMalloc(&bufs[i], bytes ? bytes : 1) != hipSuccess) { (void)hipGetLastError(); set_error("out of device memory for the iterates"); return fail(CLEORA_E_OOM); }
        return CLEORA_OK;
    }
    // The same SpMM launch has been measured up to 12-20 % slower when the buffer it reads and the buffer it writes fall into
    // the same (physical) placement class — on some boxes; on others every pair runs alike.  bufs[0] is fixed;
    // a partner is searched by timing the real kernel.  What round 4's driver run showed (VERDICT weak #5): ONE launch per
    // candidate cannot resolve a 1 % difference from launch noise, and the search itself costs (freeing a rejected 10 GB
    // candidate is ~0.3 s on this driver).  So:
    //   * a candidate's time is the MEDIAN of three launches;
    //   * candidates are drawn (at most four per slot, none once CLEORA_PLACEMENT_BUDGET_MS = 1500 ms of wall clock are spent)
    //     until the best is >= 4 % faster than the slowest seen — the spread of pairs measured on the pool's boxes is 32.3 to
    //     37.1 ms at C3, more a continuum than two classes; all stay allocated until the slot is settled (a freed buffer would
    //     be handed straight back);
    //   * the best is taken, the first candidate when the best is within 1 % of it (chosen <= first by construction);
    //   * when the caller says how many launches follow (the embed loops do), a further candidate is only tried while the
    //     most the search could win — 15 % of a launch, iterations_hint times — exceeds what trying it costs (three
    //     launches + freeing the loser at ~30 ms per GB): at |V| = 10M, d = 256 that needs ~100 iterations, so the
    //     reference's default 40 run on the first pair.
    hipLaunchKernelGGL(fill_pattern_kernel, dim3(8192), dim3(256), 0, nullptr, static_cast<float *>(bufs[0]), rows * (uint64_t)d);
    {   // arm the gather cache policy now (automatic mode waits for the third launch): candidates must be compared alike
        std::lock_guard<std::mutex> lock(g->mu);
        if (g->hot_bytes < 0 && g->auto_launches < 2) g->auto_launches = 2;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    CL_HIP(hipEventCreate(&e0));
    hipError_t ee = hipEventCreate(&e1);
    if (ee != hipSuccess) { (void)hipEventDestroy(e0); fail(0); CL_HIP(ee); }
    // Three or more buffers = the whitened loop's use: the SpMM WRITES bufs[0] and gathers from the partners in turn, so that is the
    // direction a candidate is timed in (a pair is not symmetric: this round's bench showed a triple chosen at 32.5 ms in the other
    // direction run its loop's SpMM at 33.8).  Two buffers ping-pong: both directions occur, the forward one is timed.
    const bool reverse = count >= 3;
    auto launch = [&](void *partner) {
        const float *src = static_cast<const float *>(reverse ? partner : bufs[0]);
        float *dst = static_cast<float *>(reverse ? bufs[0] : partner);
        return launch_propagate(g, CLEORA_LEFT, src, d, d, dst, d, CLEORA_F_L2NORM, 0.f, nullptr, nullptr, nullptr, nullptr);
    };
    auto launch_pair = [&](void *src, void *dst) {
        return launch_propagate(g, CLEORA_LEFT, static_cast<const float *>(src), d, d, static_cast<float *>(dst), d, CLEORA_F_L2NORM, 0.f, nullptr,
                                nullptr, nullptr, nullptr);
    };
    auto median3_pair = [&](void *src, void *dst, float *out_ms) -> int {
        float t[3] = {0.f, 0.f, 0.f};
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, nullptr);
            const int rc = launch_pair(src, dst);
            if (rc != CLEORA_OK) return rc;
            (void)hipEventRecord(e1, nullptr);
            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t[rep], e0, e1) != hipSuccess) { set_error("event timing failed"); return CLEORA_E_HIP; }
        }
        std::sort(t, t + 3);
        *out_ms = t[1];
        return CLEORA_OK;
    };
    auto median3 = [&](void *partner, float *out_ms) -> int {
        float t[3] = {0.f, 0.f, 0.f};
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, nullptr);
            const int rc = launch(partner);
            if (rc != CLEORA_OK) return rc;
            (void)hipEventRecord(e1, nullptr);
            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t[rep], e0, e1) != hipSuccess) { set_error("event timing failed"); return CLEORA_E_HIP; }
        }
        std::sort(t, t + 3);
        *out_ms = t[1];
        return CLEORA_OK;
    };
    double budget_ms = 1500.0;
    if (const char *env = std::getenv("CLEORA_PLACEMENT_BUDGET_MS")) budget_ms = std::atof(env);
    const auto t_search = std::chrono::steady_clock::now();
    const double free_ms = 30.0 * (double)bytes / 1e9;
    int rc = CLEORA_OK;
    bool warmed = false;
    for (uint32_t slot = 1; slot < count && rc == CLEORA_OK; ++slot) {
        void *first = nullptr, *best = nullptr;
        float first_ms = 0.f, best_ms = 0.f, worst_ms = 0.f;
        std::vector<void *> held;                                          // every candidate drawn for this slot
        for (int trial = 0; trial < 4; ++trial) {
            if (trial >= 1) {
                const double spent = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_search).count();
                if (spent > budget_ms) break;
                if (iterations_hint && 0.15 * first_ms * (double)iterations_hint < 3.0 * first_ms + free_ms) break;   // cannot pay for itself
            }
            void *cand = nullptr;
            if (hipMalloc(&cand, bytes) != hipSuccess) { (void)hipGetLastError(); break; }   // no room for another candidate: keep the best so far
            held.push_back(cand);
            if (reverse) hipLaunchKernelGGL(fill_pattern_kernel, dim3(8192), dim3(256), 0, nullptr, static_cast<float *>(cand), rows * (uint64_t)d);
            if (!warmed) {                                                  // hub scratch, hot marks, caches: not part of any timing
                rc = launch(cand);
                warmed = true;
                if (rc != CLEORA_OK) break;
            }
            float t = 0.f;
            rc = median3(cand, &t);
            if (rc != CLEORA_OK) break;
            if (trial == 0) { first = best = cand; first_ms = best_ms = worst_ms = t; continue; }
            if (t < best_ms) { best = cand; best_ms = t; }
            if (t > worst_ms) worst_ms = t;
            if (best_ms < 0.96f * worst_ms) break;                          // both ends of the spread seen: keep the fast one
        }
        // the first (plain) pair stays unless another one is a real gain: 1 % is the noise floor of a median of three launches
        if (best != first && first && !(best_ms < 0.99f * first_ms)) { best = first; best_ms = first_ms; }
        // No spread among the partners of bufs[0]: either every pair is fast, or bufs[0] ITSELF sits badly (seen: four partners within
        // 1 % of 35.7 ms on a box whose other runs were at 32.5).  Then another SOURCE is tried: the candidates already drawn, in
        // pairs among themselves (ping-pong use only: count == 2); a pair >= 4 % faster replaces (bufs[0], partner).
        if (rc == CLEORA_OK && !reverse && count == 2 && held.size() >= 2 && !(best_ms < 0.96f * worst_ms)) {
            void *alt_src = nullptr, *alt_dst = nullptr;
            float alt_ms = best_ms;
            for (size_t si = 0; si < held.size() && si < 2 && rc == CLEORA_OK; ++si) {
                hipLaunchKernelGGL(fill_pattern_kernel, dim3(8192), dim3(256), 0, nullptr, static_cast<float *>(held[si]), rows * (uint64_t)d);
                for (size_t di = 0; di < held.size() && rc == CLEORA_OK; ++di) {
                    if (di == si) continue;
                    const double spent = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_search).count();
                    if (spent > budget_ms || (iterations_hint && 0.15 * first_ms * (double)iterations_hint < 3.0 * first_ms + free_ms)) { si = held.size(); break; }
                    float t = 0.f;
                    rc = median3_pair(held[si], held[di], &t);
                    if (rc == CLEORA_OK && t < alt_ms) { alt_ms = t; alt_src = held[si]; alt_dst = held[di]; }
                    if (alt_ms < 0.96f * best_ms) { si = held.size(); break; }
                }
            }
            if (rc == CLEORA_OK && alt_src && alt_ms < 0.96f * best_ms) {
                (void)hipFree(bufs[0]);
                bufs[0] = alt_src;
                best = alt_dst;
                best_ms = alt_ms;
                held.erase(std::remove(held.begin(), held.end(), alt_src), held.end());      // now bufs[0]: not a loser
            }
        }
        for (void *p : held)
            if (p != best) (void)hipFree(p);
        if (rc != CLEORA_OK) { if (best) (void)hipFree(best); best = nullptr; }
        if (!best && rc == CLEORA_OK) { set_error("out of device memory for the iterates"); rc = CLEORA_E_OOM; }
        bufs[slot] = best;
        if (ms && slot == 1) { ms[0] = first_ms; ms[1] = best_ms; }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != CLEORA_OK) return fail(rc);
    return CLEORA_OK;
}
}  // namespace

int cleora_alloc_iterates(const cleora_graph *g, uint32_t d, uint32_t count, void **bufs, double *ms) {
    return alloc_iterates(g, d, count, 0, bufs, ms);
}

int cleora_alloc_iterates_for(const cleora_graph *g, uint32_t d, uint32_t count, uint64_t iterations, void **bufs, double *ms) {
    return alloc_iterates(g, d, count, iterations, bufs, ms);
}

int cleora_whiten_set_timing(int enable) { return whiten_set_timing(enable != 0); }

int cleora_whiten_get_timing(double ms[4], uint64_t *calls) {
    CL_REQUIRE(ms != nullptr && calls != nullptr, "ms / calls is NULL");
    return whiten_get_timing(ms, calls);
}

// ---- host-pointer entry points ----------------------------------------------------------------

int cleora_propagate(const cleora_graph *g, int markov_type, const float *x_host, uint32_t d,
                     float *y_host) {
    CL_REQUIRE(g != nullptr, "graph handle is NULL");
    CL_REQUIRE(x_host != nullptr && y_host != nullptr, "x / y is NULL");
    CL_REQUIRE(d > 0, "d must be positive");
    CL_HIP(hipSetDevice(g->device));
    // This is what the reference's unmodified embed() calls once per iteration: no allocation per call (the device
    // staging buffers stay with the handle) and both copies through the pinned pipeline of stager.hip.
    std::lock_guard<std::mutex> io(g->io_mu);
    const uint64_t need[2] = {g->n_cols * (uint64_t)d * sizeof(float), g->n_rows * (uint64_t)d * sizeof(float)};
    for (int k = 0; k < 2; ++k) {
        if (g->io_bytes[k] >= need[k] && g->io_buf[k]) continue;
        if (g->io_buf[k]) CL_HIP(hipFree(g->io_buf[k]));
        g->io_buf[k] = nullptr;
        g->io_bytes[k] = 0;
        CL_HIP(hipMalloc(&g->io_buf[k], need[k] ? need[k] : 1));
        g->io_bytes[k] = need[k];
    }
    int rc;
    if ((rc = staged_h2d(g->io_buf[0], x_host, need[0], nullptr)) != CLEORA_OK) return rc;
    rc = launch_propagate(g, markov_type, static_cast<const float *>(g->io_buf[0]), d, d, static_cast<float *>(g->io_buf[1]), d, 0,
                          0.f, nullptr, nullptr, nullptr, nullptr);
    if (rc != CLEORA_OK) return rc;
    return staged_d2h(y_host, g->io_buf[1], need[1], nullptr);
}

int cleora_l2_normalize(const float *x_host, uint64_t n, uint32_t d, float *y_host) {
    CL_REQUIRE(x_host != nullptr && y_host != nullptr, "x / y is NULL");
    CL_REQUIRE(d > 0, "d must be positive");
    int rc = require_device();
    if (rc != CLEORA_OK) return rc;
    DevBuf x;
    const uint64_t bytes = n * (uint64_t)d * sizeof(float);
    if ((rc = x.alloc(bytes)) != CLEORA_OK) return rc;
    if ((rc = staged_h2d(x.p, x_host, bytes, nullptr)) != CLEORA_OK) return rc;
    rc = launch_rowops(x.as<float>(), d, n, d, x.as<float>(), d, CLEORA_F_L2NORM, 0.f, nullptr,
                       nullptr, nullptr, nullptr);
    if (rc != CLEORA_OK) return rc;
    return staged_d2h(y_host, x.p, bytes, nullptr);
}

int cleora_init(const uint64_t *entity_hash_host, uint64_t n, uint32_t d, int64_t seed,
                float *x_host) {
    CL_REQUIRE(entity_hash_host != nullptr && x_host != nullptr, "hash / x is NULL");
    CL_REQUIRE(d > 0, "d must be positive");
    int rc = require_device();
    if (rc != CLEORA_OK) return rc;
    DevBuf h, x;
    const uint64_t bytes = n * (uint64_t)d * sizeof(float);
    if ((rc = h.alloc(n * sizeof(uint64_t))) != CLEORA_OK || (rc = x.alloc(bytes)) != CLEORA_OK) return rc;
    CL_HIP(hipMemcpy(h.p, entity_hash_host, n * sizeof(uint64_t), hipMemcpyHostToDevice));
    rc = launch_init(h.as<uint64_t>(), n, d, seed, x.as<float>(), d, nullptr);
    if (rc != CLEORA_OK) return rc;
    return staged_d2h(x_host, x.p, bytes, nullptr);
}

int cleora_whiten(const float *x_host, uint64_t n, uint32_t d, uint32_t n_components, float *y_host) {
    CL_REQUIRE(d > 0, "d must be positive");
    if (n == 0) return CLEORA_OK;
    CL_REQUIRE(x_host != nullptr && y_host != nullptr, "x / y is NULL");
    int rc = require_device();
    if (rc != CLEORA_OK) return rc;
    const uint32_t k = (n == 1 || n_components == 0 || n_components > d) ? d : n_components;
    DevBuf x, y, ws;
    if ((rc = x.alloc(n * (uint64_t)d * sizeof(float))) != CLEORA_OK ||
        (rc = y.alloc(n * (uint64_t)k * sizeof(float))) != CLEORA_OK ||
        (rc = ws.alloc(whiten_workspace(n, d))) != CLEORA_OK)
        return rc;
    if ((rc = staged_h2d(x.p, x_host, n * (uint64_t)d * sizeof(float), nullptr)) != CLEORA_OK) return rc;
    if ((rc = launch_whiten(x.as<float>(), d, n, d, k, y.as<float>(), k, ws.p, nullptr, nullptr)) != CLEORA_OK) return rc;
    if ((rc = staged_d2h(y_host, y.p, n * (uint64_t)k * sizeof(float), nullptr)) != CLEORA_OK) return rc;
    if (n > 1) {
        int info = 0;
        CL_HIP(hipMemcpy(&info, whiten_info(ws.p, n, d), sizeof(int), hipMemcpyDeviceToHost));
        if (info != 0) {
            set_error("the eigensolver did not converge (dsyevd info = " + std::to_string(info) + ")");
            return CLEORA_E_HIP;
        }
    }
    return CLEORA_OK;
}

namespace {
// The default loop of pycleora.embed() — E <- whiten(normalise(A E)), pycleora/__init__.py:109-117 — with the SpMM of
// iteration t+1 running BESIDE the statistics and the eigensolver of iteration t.
//
// In the reference's order every step waits for the previous one: SpMM (HBM-bound) -> normalise -> Gram (MFMA-bound)
// -> eigh (latency-bound) -> projection.  The SpMM is linear, so it can be taken before the projection:
//     E' = (Y - 1 mu^T) T            the whitened iterate (Y = the normalised rows, mu / T their mean / transform)
//     A E' = (A Y - s mu^T) T        with s = A 1 (the stored row sums; 1 for a row-stochastic left Markov matrix)
// i.e. Z = A Y needs only Y and runs on its own stream while Gram(Y) and eigh produce mu and T on another; the
// projection then takes the operand  alpha (Z - s mu^T) + rw (Y - 1 mu^T)  (the residual blend, :114-115, is linear too)
// and its output rows are normalised: that is Y of the next iteration.  E itself is only formed at the end.
// Same operations as the reference, one of them moved across a linear step: results agree to f32 rounding (the f32
// GEMM sees Z instead of A's input; tests carry the same tolerances as for the sequential loop).  Used when nobody
// needs the intermediate whitened iterates (no convergence test here; the Python driver keeps callbacks sequential).
//   b0 = the SpMM's destination in every launch (cleora_alloc_iterates' bufs[0]); b1 / b2 hold Y in turn.
int embed_whitened_overlapped(const cleora_graph *g, float *b0, float *b1, float *b2, int markov_type, uint32_t d,
                              uint64_t iterations, float rw, uint32_t flags, float **result) {
    const uint64_t n = g->n_rows;
    const uint32_t norm = (flags & CLEORA_F_L1NORM) ? CLEORA_F_L1NORM : CLEORA_F_L2NORM;
    const uint32_t fast = flags & (CLEORA_F_FASTNORM | CLEORA_F_HUB_SEGMENTS);
    const bool blend = rw > 0.0f;                                          // the Python loop: any rw > 0 (:111-115)
    const bool any_whitening = norm == CLEORA_F_L2NORM;                   // rotation invariance needs the L2 norm
    int rc;
    if (iterations == 0) { *result = b0; return CLEORA_OK; }
    DevBuf ws, rowsum, rowabs;
    if ((rc = ws.alloc(whiten_workspace(n, d))) != CLEORA_OK) return rc;
    if ((rc = rowsum.alloc(n * sizeof(float))) != CLEORA_OK) return rc;
    // intermediate projections at d = 256 without a blend: the f16 form (project_f16.hip) — its operand Z = A Y is bounded row by
    // row by the sum of |values| because Y's rows are normalised (L2 or L1: |Y| <= 1 either way)
    const bool f16_project = !blend && n > 1 && project_f16_applies(b0, d, n, d, d, b1, d, nullptr);
    // (other widths without a blend: the same bounds select the split form's three-product mode, whiten.hip)
    const bool bounded = !blend && n > 1;
    if (bounded && (rc = rowabs.alloc(n * sizeof(float))) != CLEORA_OK) return rc;
    struct Streams {
        hipStream_t a = nullptr, b = nullptr;
        hipEvent_t ya = nullptr, fb = nullptr, gs = nullptr;
        ~Streams() {
            if (a) (void)hipStreamDestroy(a);
            if (b) (void)hipStreamDestroy(b);
            if (ya) (void)hipEventDestroy(ya);
            if (fb) (void)hipEventDestroy(fb);
            if (gs) (void)hipEventDestroy(gs);
        }
    } st;
    // Stream a: SpMM and projection; stream b (higher priority: its few hundred resident blocks are dispatched at once, the
    // SpMM's millions of short blocks fill what is left): statistics and the d x d step.
    // What round 3's measurements settled (docs/history.md §3.8; the switches they were taken with are gone from the library):
    //   * the loop is the SUM of its kernels — a SIMD that holds a busy matrix-core wave gives the SpMM waves beside it next
    //     to nothing, and whatever shares the chip with the SpMM waits 10-16 us per memory request;
    //   * so when the statistics take the split-bf16 form (8 waves per CU, every matrix pipe busy) they run strictly BEFORE
    //     the SpMM (C3 51.1 -> 49.7 ms per iteration, C2 5.92 -> 5.58 in round 3).  At d >= 512 the overlapped order used to win
    //     (config 5: 249.0 against 253.3 ms with round 3's statistics); with the three-product form a kernel trace of config 5 shows
    //     the statistics stretched over the whole SpMM, the SpMM 10 ms longer for it (what the statistics take alone) and the
    //     d x d step exposed behind both: statistics first measures 4-5 ms less per iteration (round 6: 36.9 against 41.1 ms above
    //     the SpMM), so it is the order at every width;
    //   * the f64 statistics (one block per CU: at two they take ~410 of the 512 registers of every SIMD) stay beside the SpMM;
    //   * the d x d step runs on the host for d <= 256, beside the SpMM.
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    const bool split_stats = any_whitening && n > 1 && gram32_applies(b1, d, n, d);
    const bool stats_before_spmm = split_stats;
    CL_HIP(hipStreamCreateWithPriority(&st.a, hipStreamNonBlocking, prio_lo));
    CL_HIP(hipStreamCreateWithPriority(&st.b, hipStreamNonBlocking, prio_hi));
    CL_HIP(hipEventCreateWithFlags(&st.ya, hipEventDisableTiming));
    CL_HIP(hipEventCreateWithFlags(&st.fb, hipEventDisableTiming));
    CL_HIP(hipEventCreateWithFlags(&st.gs, hipEventDisableTiming));
    CL_HIP(hipDeviceSynchronize());                                        // b0 (E_0) was filled on the null stream
    const auto t_loop = std::chrono::steady_clock::now();
    if ((rc = launch_csr_rowsum(g, markov_type, rowsum.as<float>(), st.a, bounded ? rowabs.as<float>() : nullptr)) != CLEORA_OK) return rc;
    // Y_0 = normalise(A E_0 [+ blend]): the ordinary fused launch
    if ((rc = launch_propagate(g, markov_type, b0, d, d, b1, d, norm | fast | CLEORA_F_RESIDUAL | CLEORA_F_BLEND_ANY, rw, b0,
                               nullptr, nullptr, st.a)) != CLEORA_OK)
        return rc;
    float *y = b1, *ynext = b2;
    const float *mean32, *transform;
    whiten_fit_result(ws.p, n, d, &mean32, &transform);
    for (uint64_t it = 0; it + 1 < iterations; ++it) {
        CL_HIP(hipEventRecord(st.ya, st.a));                               // Y is complete
        CL_HIP(hipStreamWaitEvent(st.b, st.ya, 0));
        // Order of the launches matters twice.  (1) The Gram blocks (few, ~206 registers a wave) must be handed to the
        // dispatcher BEFORE the SpMM's millions of small blocks: behind them they starve, the SpMM refills every hole.
        // (2) rocSOLVER synchronises with the host inside dsyevd: whatever is launched after it starts only then.
        // So: statistics (stream b) -> SpMM (stream a) -> d x d step (stream b, the host blocks here while the SpMM runs).
        if (n > 1 && (rc = launch_whiten_fit_stats(y, d, n, d, ws.p, st.b, 1, any_whitening)) != CLEORA_OK) return rc;
        if (stats_before_spmm) {                   // the SpMM only once the statistics kernels are through
            CL_HIP(hipEventRecord(st.gs, st.b));
            CL_HIP(hipStreamWaitEvent(st.a, st.gs, 0));
        }
        if ((rc = launch_propagate(g, markov_type, y, d, d, b0, d, 0, 0.f, nullptr, nullptr, nullptr, st.a)) != CLEORA_OK) return rc;
        // intermediate iterations of the L2-normalised loop may take ANY whitening transform (eigh.hip): Cholesky — unless the
        // guard refuses it on statistics that are only ~1e-8 accurate: then this iteration's statistics are taken again in f64
        // and the PCA form follows (rare: a near-singular covariance; e.g. d > rank)
        if (n > 1) {
            bool need_exact = false;
            if ((rc = launch_whiten_fit_solve(n, d, d, ws.p, nullptr, st.b, any_whitening, split_stats, &need_exact)) != CLEORA_OK) return rc;
            if (need_exact) {
                if ((rc = launch_whiten_fit_stats(y, d, n, d, ws.p, st.b, 1, false)) != CLEORA_OK) return rc;
                if ((rc = launch_whiten_fit_solve(n, d, d, ws.p, nullptr, st.b, false)) != CLEORA_OK) return rc;
            }
        }
        CL_HIP(hipEventRecord(st.fb, st.b));
        CL_HIP(hipStreamWaitEvent(st.a, st.fb, 0));
        if (n > 1) {
            // P = (alpha (Z - s mu^T) + rw (Y - mu)) T, then the row normalisation: Y of the next iteration
            bool normed = false;
            if (f16_project) {
                rc = launch_project_f16(b0, d, n, mean32, transform, ynext, d, st.a, rowsum.as<float>(), rowabs.as<float>(),
                                        norm == CLEORA_F_L1NORM ? 2 : 1);
                normed = true;
            } else {
                rc = launch_project(b0, d, n, d, mean32, transform, d, ynext, d, st.a, rowsum.as<float>(), blend ? y : nullptr, d,
                                    1.0f - rw, rw, norm == CLEORA_F_L1NORM ? 2 : 1, &normed, bounded ? rowabs.as<float>() : nullptr);
            }
            if (rc != CLEORA_OK) return rc;
            // (a projection of several column passes — k > 256 — cannot normalise in its epilogue: one more pass over Y.  Like the
            // epilogues' sums it need not keep the reference's summation order inside the loop: the tree-sum form, 2.7 instead of 4.0 ms
            // at config 5's shape; the general exact-order kernel took 11.5 ms there until round 6 gave l2_exact16_kernel a d = 1024 case)
            if (!normed && (rc = launch_rowops(ynext, d, n, d, ynext, d, norm | fast | CLEORA_F_FASTNORM, 0.f, nullptr, nullptr, nullptr, st.a)) != CLEORA_OK) return rc;
        } else {
            // one entity: whiten_embeddings returns its input (:132-133), so E' = Y and the next Y = normalise(A Y [+ blend])
            if ((rc = launch_rowops(b0, d, n, d, ynext, d, norm | fast | CLEORA_F_RESIDUAL | CLEORA_F_BLEND_ANY, rw, y, nullptr, nullptr, st.a)) != CLEORA_OK)
                return rc;
        }
        std::swap(y, ynext);
    }
    // E_T = whiten(Y_{T-1})
    if ((rc = launch_whiten(y, d, n, d, d, b0, d, ws.p, nullptr, st.a)) != CLEORA_OK) return rc;
    CL_HIP(hipStreamSynchronize(st.a));
    CL_HIP(hipStreamSynchronize(st.b));
    g_last_loop_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop).count();
    if (n > 1) {
        int info = 0;
        CL_HIP(hipMemcpy(&info, whiten_info(ws.p, n, d), sizeof(int), hipMemcpyDeviceToHost));
        if (info != 0) {
            set_error("the eigensolver did not converge (dsyevd info = " + std::to_string(info) + ")");
            return CLEORA_E_HIP;
        }
    }
    *result = b0;
    return CLEORA_OK;
}

// The default path of pycleora.embed() (pycleora/__init__.py:97-127): propagate, residual, _normalize,
// whiten_embeddings, f64 RMSE between whitened iterates.  Three buffers rotate: prev -> (SpMM + L2) -> mid
// -> (whiten) -> next.
int embed_whitened(const cleora_graph *g, float *a, float *b, float *c, int markov_type, uint32_t d,
                   uint64_t max_iterations, float rw, float threshold, uint32_t flags, float **result,
                   uint64_t *iterations_run) {
    const uint64_t n = g->n_rows;
    const bool check = threshold > 0.0f;
    DevBuf ws, sq, rws, total;
    int rc;
    if ((rc = ws.alloc(whiten_workspace(n, d))) != CLEORA_OK) return rc;
    if (check && ((rc = sq.alloc(n * sizeof(double))) != CLEORA_OK ||
                  (rc = rws.alloc(reduce_workspace(n) * sizeof(double))) != CLEORA_OK ||
                  (rc = total.alloc(sizeof(double))) != CLEORA_OK))
        return rc;
    float *prev = a, *mid = b, *next = c;
    uint64_t actual = max_iterations;
    CL_HIP(hipDeviceSynchronize());
    const auto t_loop = std::chrono::steady_clock::now();
    // the Python loop blends for any rw > 0 (pycleora/__init__.py:111-115) and normalises with `normalization`
    const uint32_t base = ((flags & CLEORA_F_L1NORM) ? CLEORA_F_L1NORM : CLEORA_F_L2NORM) | CLEORA_F_RESIDUAL |
                          CLEORA_F_BLEND_ANY | (flags & (CLEORA_F_FASTNORM | CLEORA_F_HUB_SEGMENTS));
    for (uint64_t it = 0; it < max_iterations; ++it) {
        if ((rc = launch_propagate(g, markov_type, prev, d, d, mid, d, base, rw, prev, nullptr, nullptr, nullptr)) != CLEORA_OK)
            return rc;
        if ((rc = launch_whiten(mid, d, n, d, d, next, d, ws.p, nullptr, nullptr)) != CLEORA_OK) return rc;
        if (check && it > 0) {                                            // :122-125
            if ((rc = launch_rowops(next, d, n, d, next, d, CLEORA_F_SQDIFF | CLEORA_F_SQDIFF64, 0.f, prev, sq.as<double>(), nullptr, nullptr)) != CLEORA_OK ||
                (rc = launch_reduce_sum(sq.as<double>(), n, rws.as<double>(), total.as<double>(), nullptr)) != CLEORA_OK)
                return rc;
            double sum = 0.0;
            CL_HIP(hipMemcpy(&sum, total.p, sizeof(double), hipMemcpyDeviceToHost));
            if (sqrt(sum / (double)(n * (uint64_t)d)) < (double)threshold) {   // _compute_rmse, :974-976
                std::swap(prev, next);
                actual = it + 1;
                break;
            }
        }
        std::swap(prev, next);
    }
    CL_HIP(hipDeviceSynchronize());
    g_last_loop_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop).count();
    if (n > 1 && max_iterations > 0) {
        int info = 0;
        CL_HIP(hipMemcpy(&info, whiten_info(ws.p, n, d), sizeof(int), hipMemcpyDeviceToHost));
        if (info != 0) {
            set_error("the eigensolver did not converge (dsyevd info = " + std::to_string(info) + ")");
            return CLEORA_E_HIP;
        }
    }
    *result = prev;
    if (iterations_run) *iterations_run = actual;
    return CLEORA_OK;
}
}  // namespace

}  // extern "C"

// The loops behind cleora_embed (host arrays) and cleora_embed_dev (x_dev: E_0 in, result out, device memory).
static int embed_impl(const cleora_graph *g, const uint64_t *entity_hash_host, const float *x0_host, float *x_dev,
                      int markov_type, uint32_t d, uint64_t max_iterations, int64_t seed,
                      float residual_weight, float convergence_threshold, uint32_t flags,
                      float *out_host, uint64_t *iterations_run) {
    CL_REQUIRE(g != nullptr, "graph handle is NULL");
    CL_REQUIRE(g->n_rows == g->n_cols, "cleora_embed needs the whole (square) graph on one device");
    CL_REQUIRE(out_host != nullptr || x_dev != nullptr, "out is NULL");
    CL_REQUIRE(entity_hash_host != nullptr || x0_host != nullptr || x_dev != nullptr, "need entity hashes or x0");
    CL_REQUIRE(d > 0, "d must be positive");
    CL_HIP(hipSetDevice(g->device));
    auto emit = [&](const float *result, uint64_t nbytes) -> int {
        if (x_dev) {
            CL_HIP(hipMemcpy(x_dev, result, nbytes, hipMemcpyDeviceToDevice));
            return CLEORA_OK;
        }
        return staged_d2h(out_host, result, nbytes, nullptr);
    };
    const uint64_t n = g->n_rows;
    const uint64_t bytes = n * (uint64_t)d * sizeof(float);
    const bool check = convergence_threshold > 0.0f;  // embedding.rs:150
    DevBuf a, b, h, sq, ws, total;
    int rc;
    const bool whitened = (flags & CLEORA_F_WHITEN) != 0;
    DevBuf c;
    if (whitened) {
        // three buffers placed for the SpMM: it always writes `b` (mid) and reads `a` / `c` in turn
        void *bufs[3] = {nullptr, nullptr, nullptr};
        if ((rc = alloc_iterates(g, d, 3, max_iterations, bufs, nullptr)) != CLEORA_OK) return rc;   // (the search only where the iterations can repay it)
        b.p = bufs[0];
        a.p = bufs[1];
        c.p = bufs[2];
    } else if ((rc = a.alloc(bytes)) != CLEORA_OK || (rc = b.alloc(bytes)) != CLEORA_OK) {
        return rc;
    }
    // nobody looks at the intermediate whitened iterates when there is no convergence test: SpMM(t+1) beside Gram /
    // eigh(t) (embed_whitened_overlapped).  That loop wants E_0 in the SpMM-side buffer bufs[0] = `b`.
    const bool overlapped = whitened && !check;
    float *e_init = overlapped ? b.as<float>() : a.as<float>();
    if (x_dev) {
        CL_HIP(hipMemcpy(e_init, x_dev, bytes, hipMemcpyDeviceToDevice));
    } else if (x0_host) {
        if ((rc = staged_h2d(e_init, x0_host, bytes, nullptr)) != CLEORA_OK) return rc;
    } else {
        if ((rc = h.alloc(n * sizeof(uint64_t))) != CLEORA_OK) return rc;
        CL_HIP(hipMemcpy(h.p, entity_hash_host, n * sizeof(uint64_t), hipMemcpyHostToDevice));
        if ((rc = launch_init(h.as<uint64_t>(), n, d, seed, e_init, d, nullptr)) != CLEORA_OK) return rc;
    }
    if (whitened) {
        float *result = nullptr;
        if (overlapped) {
            rc = embed_whitened_overlapped(g, b.as<float>(), a.as<float>(), c.as<float>(), markov_type, d, max_iterations,
                                           residual_weight, flags, &result);
            if (rc == CLEORA_OK && iterations_run) *iterations_run = max_iterations;
        } else {
            rc = embed_whitened(g, a.as<float>(), b.as<float>(), c.as<float>(), markov_type, d, max_iterations,
                                residual_weight, convergence_threshold, flags, &result, iterations_run);
        }
        if (rc != CLEORA_OK) return rc;
        return emit(result, bytes);
    }
    if (check) {
        if ((rc = sq.alloc(n * sizeof(double))) != CLEORA_OK ||
            (rc = ws.alloc(reduce_workspace(n) * sizeof(double))) != CLEORA_OK ||
            (rc = total.alloc(sizeof(double))) != CLEORA_OK)
            return rc;
    }
    // Placement tuning on the job's own iterations (large iterates only): the SpMM runs up to 12 %
    // slower when the two ping-pong allocations fall into the same placement class.
    // Buffer `a` stays; the partner buffer is re-drawn every two iterations (a -> c, c -> a, timed
    // with events) until a pair is >= 5 % faster than the slowest pair seen, 4 partners were tried, or the remaining
    // iterations could no longer repay a rejected candidate; the best partner is kept.  Every trial iteration is a real
    // iteration: no work is repeated.
    const bool tune = bytes >= (256ull << 20) && max_iterations >= 8;
    struct Trial { void *buf; float ms; };
    std::vector<Trial> trials;
    DevBuf extra[2];      // candidate partners: at most the best one so far plus the one on trial stay allocated
    int n_tried = 1;      // `b` is the first candidate
    struct Ev {
        hipEvent_t e = nullptr;
        ~Ev() { if (e) (void)hipEventDestroy(e); }
    } e0, e1;
    if (tune) {
        CL_HIP(hipEventCreate(&e0.e));
        CL_HIP(hipEventCreate(&e1.e));
    }
    const hipEvent_t ev0 = e0.e, ev1 = e1.e;
    bool tuning = tune;
    float pair_ms = 0.f;
    float *fixed = a.as<float>();      // holds the iterate after every odd number of trial iterations
    float *partner = b.as<float>();
    float *src = fixed, *dst = partner;
    uint64_t actual = max_iterations;
    const uint32_t base = CLEORA_F_L2NORM | CLEORA_F_RESIDUAL | (flags & (CLEORA_F_FASTNORM | CLEORA_F_HUB_SEGMENTS));
    CL_HIP(hipDeviceSynchronize());
    const auto t_loop = std::chrono::steady_clock::now();
    for (uint64_t it = 0; it < max_iterations; ++it) {
        const bool test = check && it > 0;  // embedding.rs:169
        if (tuning) CL_HIP(hipEventRecord(ev0, nullptr));
        rc = launch_propagate(g, markov_type, src, d, d, dst, d, base | (test ? CLEORA_F_SQDIFF : 0u),
                              residual_weight, src, test ? sq.as<double>() : nullptr, nullptr, nullptr);
        if (rc != CLEORA_OK) return rc;
        if (tuning) {
            CL_HIP(hipEventRecord(ev1, nullptr));
            CL_HIP(hipEventSynchronize(ev1));
            float ms = 0.f;
            CL_HIP(hipEventElapsedTime(&ms, ev0, ev1));
            pair_ms += ms;
        }
        std::swap(src, dst);
        if (test) {
            if ((rc = launch_reduce_sum(sq.as<double>(), n, ws.as<double>(), total.as<double>(), nullptr)) != CLEORA_OK)
                return rc;
            double sum = 0.0;
            CL_HIP(hipMemcpy(&sum, total.p, sizeof(double), hipMemcpyDeviceToHost));
            // rmse = sqrt(diff / (n*d)) < threshold                       (embedding.rs:177-178)
            const float rmse = sqrtf((float)(sum / (double)(n * (uint64_t)d)));
            if (rmse < convergence_threshold) {
                actual = it + 1;
                break;
            }
        }
        if (tuning && (it & 1)) {  // a full a -> partner -> a round trip has been timed; iterate is in `fixed`
            trials.push_back({partner, pair_ms});
            pair_ms = 0.f;
            float lo = trials[0].ms, hi = trials[0].ms;
            size_t best = 0;
            for (size_t k = 1; k < trials.size(); ++k) {
                if (trials[k].ms < lo) { lo = trials[k].ms; best = k; }
                if (trials[k].ms > hi) hi = trials[k].ms;
            }
            const bool found = trials.size() >= 2 && lo < 0.95f * hi;
            // another candidate only while the most it could win on the remaining iterations (15 % of a launch each) exceeds
            // what rejecting one costs (hipFree: ~30 ms per GB on this driver): |V| = 10M, d = 256 needs ~100 iterations
            const double could_win = 0.15 * (double)(hi * 0.5f) * (double)(max_iterations - it - 1), reject_cost = 30.0 * (double)bytes / 1e9;
            bool more = !(found || n_tried == 4 || it + 8 > max_iterations || could_win < reject_cost);
            if (more) {
                // draw the next candidate while the rejected ones still hold their memory (a freed buffer would be
                // handed straight back: same placement), then free every candidate but the best so far
                DevBuf cand;
                const int alloc_rc = cand.alloc(bytes);
                if (alloc_rc == CLEORA_OK) {
                    ++n_tried;
                } else {
                    // out of memory for another candidate (typical at C4 scale): keep the best so far.  The failed
                    // hipMalloc leaves a sticky hipErrorOutOfMemory that the next launch's hipGetLastError() would
                    // report as ITS failure: clear it.
                    (void)hipGetLastError();
                    more = false;
                }
                const Trial keep = trials[best];
                for (DevBuf &e : extra)
                    if (e.p && e.p != keep.buf) e.release();
                trials.assign(1, keep);
                if (more) {
                    for (DevBuf &e : extra)
                        if (!e.p) { std::swap(e.p, cand.p); partner = e.as<float>(); break; }
                }
            }
            if (!more) {
                tuning = false;
                partner = static_cast<float *>(trials[best < trials.size() ? best : 0].buf);
            }
            src = fixed;
            dst = partner;
        }
    }
    CL_HIP(hipDeviceSynchronize());
    g_last_loop_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop).count();
    if (iterations_run) *iterations_run = actual;
    return emit(src, bytes);
}

extern "C" {

int cleora_embed(const cleora_graph *g, const uint64_t *entity_hash_host, const float *x0_host,
                 int markov_type, uint32_t d, uint64_t max_iterations, int64_t seed,
                 float residual_weight, float convergence_threshold, uint32_t flags,
                 float *out_host, uint64_t *iterations_run) {
    CL_REQUIRE(out_host != nullptr, "out is NULL");
    CL_REQUIRE(entity_hash_host != nullptr || x0_host != nullptr, "need entity hashes or x0");
    return embed_impl(g, entity_hash_host, x0_host, nullptr, markov_type, d, max_iterations, seed, residual_weight,
                      convergence_threshold, flags, out_host, iterations_run);
}

int cleora_embed_dev(const cleora_graph *g, float *x_dev, int markov_type, uint32_t d, uint64_t max_iterations,
                     float residual_weight, float convergence_threshold, uint32_t flags, uint64_t *iterations_run) {
    CL_REQUIRE(x_dev != nullptr, "x is NULL");
    return embed_impl(g, nullptr, nullptr, x_dev, markov_type, d, max_iterations, 0, residual_weight, convergence_threshold,
                      flags, nullptr, iterations_run);
}

}  // extern "C"
