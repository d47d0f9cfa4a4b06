// hot.hip — "hot column" marking for the gather cache policy.
//
// Measured on MI355X (scripts/exp_hot.py, C3 graph, d = 256): gathering the COLD embedding rows with
// the non-temporal policy (`buffer_load ... nt`) and the most frequently referenced rows with the default
// policy shortens the SpMM from 34.3 to 31.0 ms (-10 %): the streaming cold gathers no longer evict the
// hot set from L2 / Infinity Cache.  The optimum is flat between 0.25M and 2M hot rows (0.25 - 2 GB of
// X); `nt` on every row, or on the hot rows only, gains nothing; sc0 / sc1 do nothing.
//
// The mark is bit 31 of a private copy of the column indices (entities < 2^31), so the kernels read
// exactly the same bytes as before.  The hot set = the K columns with the largest in-degree, K chosen
// from a byte budget and the row width.
#include <vector>

#include "common.h"

namespace cleora {
namespace {

constexpr int kDegBins = 4096;

__global__ __launch_bounds__(256) void indegree_kernel(const uint32_t *__restrict__ col, uint64_t nnz,
                                                       uint32_t *__restrict__ indeg) {
    for (uint64_t i = CLEORA_LINEAR_BLOCK() * 256 + threadIdx.x; i < nnz;
         i += (uint64_t)gridDim.x * gridDim.y * 256)
        atomicAdd(&indeg[col[i]], 1u);
}

// Block-private histogram in LDS (in-degrees cluster on a few small values: global atomics on those bins
// serialised to 23 ms at n = 10M; this is ~0.1 ms), then one global atomic per non-empty bin per block.
__global__ __launch_bounds__(256) void degree_hist_kernel(const uint32_t *__restrict__ indeg, uint64_t n,
                                                          uint32_t *__restrict__ bins) {
    __shared__ uint32_t sm[kDegBins];
    for (int b = threadIdx.x; b < kDegBins; b += 256) sm[b] = 0;
    __syncthreads();
    for (uint64_t i = CLEORA_LINEAR_BLOCK() * 256 + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * gridDim.y * 256) {
        const uint32_t v = indeg[i];
        atomicAdd(&sm[v < kDegBins - 1 ? v : kDegBins - 1], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kDegBins; b += 256)
        if (sm[b]) atomicAdd(&bins[b], sm[b]);
}

// meta[0] = smallest in-degree that is still "hot", meta[1] = rows at or above it: whole degree classes from the top
// down to the budget `want` (the first class is always taken).  One thread: 4096 bins.
__global__ void select_threshold_kernel(const uint32_t *__restrict__ bins, uint64_t want, uint32_t *__restrict__ meta) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t threshold = kDegBins;
    uint64_t cum = 0;
    for (int b = kDegBins - 1; b >= 1; --b) {
        if (cum + bins[b] > want && cum > 0) break;
        cum += bins[b];
        threshold = (uint32_t)b;
        if (cum >= want) break;
    }
    meta[0] = threshold;
    meta[1] = (uint32_t)(cum > 0xffffffffull ? 0xffffffffull : cum);
}

__global__ __launch_bounds__(256) void mark_kernel(const uint32_t *__restrict__ col, uint64_t nnz,
                                                   const uint32_t *__restrict__ indeg, const uint32_t *__restrict__ meta,
                                                   uint32_t *__restrict__ out) {
    const uint32_t threshold = meta[0];
    for (uint64_t i = CLEORA_LINEAR_BLOCK() * 256 + threadIdx.x; i < nnz;
         i += (uint64_t)gridDim.x * gridDim.y * 256) {
        const uint32_t c = col[i];
        out[i] = c | (indeg[c] >= threshold ? 0x80000000u : 0u);
    }
}

}  // namespace

namespace {
const uint32_t *build_hot_cols(const cleora_graph *g, uint64_t want, hipStream_t stream);
}

// Returns the marked column array for rows of `d` floats (building or rebuilding it if needed), or
// nullptr when the policy does not apply.  Called with g->mu held.  Everything it does is ENQUEUED on `stream`
// (stream-ordered allocations, kernels; the threshold is chosen on the device): a `*_dev` entry point never
// synchronises, allocates with hipMalloc or copies to the host because of it.

const uint32_t *ensure_hot_cols(const cleora_graph *g, uint32_t d, uint64_t ldx, hipStream_t stream) {
    if (g->hot_bytes == 0 || g->hot_failed || g->nnz == 0 || g->n_cols >= (1ull << 31)) return nullptr;
    const uint64_t row_bytes = (uint64_t)d * sizeof(float);
    const uint64_t x_bytes = g->n_cols * ldx * sizeof(float);
    uint64_t budget;
    if (g->hot_bytes < 0) {                           // auto: only when X is far larger than the caches ...
        if (x_bytes < (1ull << 30)) return nullptr;
        // ... and a row fills a whole wavefront's load: on the 64 / 32-column slices of the column partition
        // (P = 4 / 8) the policy measured +2 / +7 % (scripts/column_probe.py); sub-wave groups also need X to
        // fit one 4 GiB buffer descriptor, which the 128-column slice of C3 does not.
        if (d < 256) return nullptr;
        // ... and the graph is evidently being iterated: the marks cost ~13 ms to build at C3 scale
        // (one pass of random atomics over col) and return ~3 ms per launch.
        if (g->auto_launches < 2) { ++g->auto_launches; return nullptr; }
        budget = 768ull << 20;
    } else {
        budget = (uint64_t)g->hot_bytes;
    }
    uint64_t want = budget / row_bytes;
    if (want > g->n_cols / 2) want = g->n_cols / 2;
    if (want == 0) return nullptr;
    if (g->col_hot && g->hot_rows_target == want) return g->col_hot;
    const uint32_t *marked = build_hot_cols(g, want, stream);
    if (!marked) {
        // the policy is an optimisation: a failure here (typically out of memory for the marked copy) must not
        // fail the launch or be retried on every call — clear the sticky HIP error and switch the policy off
        (void)hipGetLastError();
        g->hot_failed = true;
    }
    return marked;
}

namespace {
const uint32_t *build_hot_cols(const cleora_graph *g, uint64_t want, hipStream_t stream) {
    if (hipSetDevice(g->device) != hipSuccess) return nullptr;
    // a rebuild overwrites col_hot in place, ordered on `stream` like every launch of this handle (one stream per
    // handle: include/cleora_hip.h)
    uint32_t *indeg = nullptr, *bins = nullptr;
    if (hipMallocAsync(reinterpret_cast<void **>(&indeg), g->n_cols * sizeof(uint32_t), stream) != hipSuccess) return nullptr;
    if (hipMallocAsync(reinterpret_cast<void **>(&bins), kDegBins * sizeof(uint32_t), stream) != hipSuccess) {
        (void)hipFreeAsync(indeg, stream);
        return nullptr;
    }
    if (!g->hot_meta && hipMallocAsync(reinterpret_cast<void **>(&g->hot_meta), 2 * sizeof(uint32_t), stream) != hipSuccess) g->hot_meta = nullptr;
    if (g->hot_meta && !g->col_hot && hipMallocAsync(reinterpret_cast<void **>(&g->col_hot), g->nnz * sizeof(uint32_t), stream) != hipSuccess)
        g->col_hot = nullptr;
    if (!g->hot_meta || !g->col_hot) {
        (void)hipFreeAsync(indeg, stream);
        (void)hipFreeAsync(bins, stream);
        return nullptr;
    }
    (void)hipMemsetAsync(indeg, 0, g->n_cols * sizeof(uint32_t), stream);
    (void)hipMemsetAsync(bins, 0, kDegBins * sizeof(uint32_t), stream);
    const dim3 grid(4096);
    hipLaunchKernelGGL(indegree_kernel, grid, dim3(256), 0, stream, g->col, g->nnz, indeg);
    hipLaunchKernelGGL(degree_hist_kernel, grid, dim3(256), 0, stream, indeg, g->n_cols, bins);
    hipLaunchKernelGGL(select_threshold_kernel, dim3(1), dim3(64), 0, stream, bins, want, g->hot_meta);
    hipLaunchKernelGGL(mark_kernel, grid, dim3(256), 0, stream, g->col, g->nnz, indeg, g->hot_meta, g->col_hot);
    const bool ok = hipGetLastError() == hipSuccess;
    (void)hipFreeAsync(indeg, stream);
    (void)hipFreeAsync(bins, stream);
    if (!ok) return nullptr;
    g->hot_rows_target = want;
    return g->col_hot;
}
}  // namespace

// rows currently marked hot (a host query: waits for the build if it is still in flight)
uint64_t hot_rows_marked(const cleora_graph *g) {
    if (!g->col_hot || !g->hot_meta || !g->hot_rows_target) return 0;
    uint32_t meta[2] = {0, 0};
    // the build may have been enqueued on a non-blocking stream, which a plain hipMemcpy does not wait for
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(meta, g->hot_meta, sizeof(meta), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return meta[1];
}

}  // namespace cleora
