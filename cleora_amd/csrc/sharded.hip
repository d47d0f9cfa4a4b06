// sharded.hip — the ROW-PARTITIONED propagation loops under the C ABI: one process per GPU, full replicas of the iterate,
// each rank owns row blocks of the CSR, one in-place all-gather of the next iterate per iteration (comm.hip / peer.hip).
//
// The reference is single-process (rayon over rows, src/embedding.rs:59-63); this is the multi-GPU layout
// BASELINE.json:north_star asks for ("the graph is row-partitioned across the 8 GPUs of one node with an RCCL all-gather of
// the embedding matrix over xGMI between iterations ... Host code stays in Rust, calling the kernels through a thin extern-C
// FFI").  Until round 3 the block schedule, stream ordering and the partitioned whitening lived in Python
// and in an example; a Rust host would have had to re-write them.  Here they are ONE call each (the Python loop lives on
// as the model the CPU suite runs over gloo: tests/sharded_model.py):
//
//   cleora_sharded_plan       (pure host arithmetic) the row boundaries of the world * steps contiguous blocks
//   cleora_sharded_create     this rank's row blocks of the CSR as device graphs
//   cleora_sharded_propagate_dev   one iteration: x_next <- epilogue(A x), replicated (block k's gather beside block k+1's SpMM)
//   cleora_embed_sharded      the loops: embed_full / embed_full_with_convergence (src/embedding.rs:106-188) and — with
//                             CLEORA_F_WHITEN — the default embed() loop (pycleora/__init__.py:109-117) in the reorganised form
//                             of the single-GPU loop (abi.hip embed_whitened_overlapped; docs/history.md §3.7-3.8) or in the reference's order
//
// Layout.  With P ranks and K steps per iteration the row space is cut into P*K contiguous blocks (equal row counts, or
// balanced on the rowptr prefix sum for graphs whose ids are ordered by degree); rank r owns blocks {k*P + r}.  Step k
// computes block (k, r) on every rank straight into its slot of the next replica and then all-gathers the contiguous row
// range of blocks [k*P, (k+1)*P) IN PLACE on the communication stream while the SpMM of step k+1 runs.
// Memory plan of the whitened loop (missing #5): TWO full replicas (the caller's, which holds E_0 and is
// recycled as soon as Y_0 exists, and one more) + Z for the rank's OWN rows only + the whitening workspace — at BASELINE
// config 4 on 8 GPUs (n = 111 M, d = 256): 2 x 113.7 GB + 14.2 GB + ~6.6 GB of CSR (with the gather policy's private col) = ~248 GB of the 288 GB
// (cleora_embed_sharded_bytes; tests/test_sharded_plan_cpu.py holds the arithmetic).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

#include "comm_internal.h"

struct cleora_sharded {
    cleora_comm *comm = nullptr;                 // nullptr: a world of one
    int rank = 0, world = 1, device = 0;
    uint64_t n = 0, n_pad = 0, local_rows = 0, local_nnz = 0, device_bytes = 0;
    uint32_t steps = 1;
    int balance = CLEORA_BALANCE_ROWS;
    bool has_sym = false;
    std::vector<uint64_t> bounds;                // world * steps + 1 row boundaries
    struct Block {
        cleora_graph *g = nullptr;
        uint64_t b0 = 0, b1 = 0, valid = 0, first_local = 0;   // rows [b0, b1) of the padded row space; valid = rows below n; offset among the rank's rows
        void *owned[4] = {nullptr, nullptr, nullptr, nullptr};  // device arrays adopted by g (device-input path)
    };
    std::vector<Block> blocks;
    hipStream_t comm_stream = nullptr, side_stream = nullptr;
    hipStream_t loop_stream = nullptr;           // what cleora_embed_sharded runs on (cleora_sharded_set_stream; default: the device's null stream)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_side = nullptr;
    std::vector<uint64_t> offsets;               // scratch for the all-gather-v call
    int debug_fail_first_gather = 0;             // tests: how many attempts of verify_peer_gather report a mismatch
    bool peer_verified = false;                  // the first-use check of the peer-direct all-gather on a real replica has passed (verify_peer_gather)
    bool pending = false;                        // collectives on comm_stream the compute stream has not joined yet
    // timing (cleora_sharded_set_timing): event pairs around every all-gather on the communication stream
    bool timing = false;
    std::vector<hipEvent_t> ev_pool, ev_used;
    uint64_t timed_calls = 0;
    std::mutex mu;
};

namespace cleora {
namespace {

struct DevMem {
    void *p = nullptr;
    ~DevMem() { if (p) (void)hipFree(p); }
    int alloc(uint64_t bytes) {
        CL_HIP(hipMalloc(&p, bytes ? bytes : 1));
        return CLEORA_OK;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

inline hipStream_t S(void *s) { return reinterpret_cast<hipStream_t>(s); }

uint64_t block_size(uint64_t n, uint64_t world, uint64_t steps) {
    const uint64_t nb = world * steps;
    uint64_t b = (n + nb - 1) / nb;
    b = (b + 3) / 4 * 4;                          // a multiple of 4 rows: every block of a 16-byte-aligned matrix stays aligned for any d
    return b < 4 ? 4 : b;
}

// the row boundaries (tests/sharded_model.py row_bounds is the same arithmetic in Python: the two are compared in the CPU suite)
int plan_rows(uint64_t n, const uint64_t *rowptr, uint32_t world, uint32_t steps, int balance, std::vector<uint64_t> &bounds,
              uint64_t *n_pad, int *mode) {
    CL_REQUIRE(world >= 1 && steps >= 1, "world and steps must be positive");
    CL_REQUIRE(balance == CLEORA_BALANCE_AUTO || balance == CLEORA_BALANCE_ROWS || balance == CLEORA_BALANCE_NNZ, "unknown balance mode");
    CL_REQUIRE(rowptr != nullptr || n == 0, "rowptr is NULL");
    const uint64_t nb = (uint64_t)world * steps, block = block_size(n, world, steps);
    bounds.assign(nb + 1, 0);
    for (uint64_t j = 0; j <= nb; ++j) bounds[j] = j * block;
    if (balance == CLEORA_BALANCE_AUTO && nb > 1 && n > 0) {
        // "rows" when its heaviest block is within 3 % of the mean work (true for randomly permuted ids), else "nnz";
        // weight(row) = edges + 1: one gathered X row per edge plus the one Y row written
        double heaviest = 0.0;
        for (uint64_t j = 0; j < nb; ++j) {
            const uint64_t c0 = std::min(bounds[j], n), c1 = std::min(bounds[j + 1], n);
            heaviest = std::max(heaviest, (double)((rowptr[c1] - rowptr[c0]) + (c1 - c0)));
        }
        const double total = (double)rowptr[n] + (double)n;
        balance = heaviest <= 1.03 * total / (double)nb ? CLEORA_BALANCE_ROWS : CLEORA_BALANCE_NNZ;
    }
    if (balance != CLEORA_BALANCE_NNZ || nb == 1 || n == 0) {
        *n_pad = block * nb;
        *mode = CLEORA_BALANCE_ROWS;
        return CLEORA_OK;
    }
    // split on the prefix sum of the work per row; boundaries are multiples of 4 rows; shards are unequal (all-gather-v)
    const uint64_t n4 = (n + 3) / 4 * 4;
    const uint64_t total = rowptr[n] + n;                          // cum(r) = rowptr[r] + r = work before row r
    uint64_t prev = 0;
    for (uint64_t j = 1; j < nb; ++j) {
        const uint64_t target = (uint64_t)((unsigned __int128)total * j / nb);
        uint64_t lo = 0, hi = n + 1;                               // first r in [0, n] with cum(r) >= target (r = n + 1 if none)
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo) / 2;
            if (rowptr[mid] + mid < target) lo = mid + 1; else hi = mid;
        }
        uint64_t c = (lo + 3) / 4 * 4;
        c = std::min(n4, std::max(prev, c));
        bounds[j] = c;
        prev = c;
    }
    bounds[0] = 0;
    bounds[nb] = n4;
    *n_pad = n4;
    *mode = CLEORA_BALANCE_NNZ;
    return CLEORA_OK;
}

void free_sharded(cleora_sharded *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (auto &b : s->blocks) {
        if (b.g) (void)cleora_graph_destroy(b.g);
        for (void *p : b.owned)
            if (p) (void)hipFree(p);
    }
    for (hipEvent_t e : s->ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : s->ev_used) (void)hipEventDestroy(e);
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->ev_join) (void)hipEventDestroy(s->ev_join);
    if (s->ev_side) (void)hipEventDestroy(s->ev_side);
    if (s->comm_stream) (void)hipStreamDestroy(s->comm_stream);
    if (s->side_stream) (void)hipStreamDestroy(s->side_stream);
    delete s;
}

hipEvent_t timing_event(cleora_sharded *s) {
    hipEvent_t e = nullptr;
    if (!s->ev_pool.empty()) { e = s->ev_pool.back(); s->ev_pool.pop_back(); }
    else if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    s->ev_used.push_back(e);
    return e;
}

// the exchange of step k: rows [bounds[kP], bounds[(k+1)P]) of `buf` (a replica, ld = d), on the communication stream,
// after everything enqueued on `stream` so far
int gather_step(cleora_sharded *s, float *buf, uint32_t d, uint32_t k, hipStream_t stream, hipEvent_t hub_join = nullptr) {
    if (s->world == 1) {
        if (hub_join) CL_HIP(hipStreamWaitEvent(stream, hub_join, 0));
        return CLEORA_OK;
    }
    const int P = s->world;
    s->offsets.resize((size_t)P + 1);
    for (int r = 0; r <= P; ++r) s->offsets[r] = s->bounds[(size_t)k * P + r] * (uint64_t)d;
    CL_HIP(hipEventRecord(s->ev_fork, stream));
    CL_HIP(hipStreamWaitEvent(s->comm_stream, s->ev_fork, 0));
    if (hub_join) CL_HIP(hipStreamWaitEvent(s->comm_stream, hub_join, 0));   // the block's hub rows (in-order launch on its side stream)
    hipEvent_t t0 = s->timing ? timing_event(s) : nullptr;
    if (t0) CL_HIP(hipEventRecord(t0, s->comm_stream));
    const int rc = cleora_allgatherv_f32_dev(s->comm, buf, s->offsets.data(), s->comm_stream);
    if (rc != CLEORA_OK) return rc;
    if (t0) {
        hipEvent_t t1 = timing_event(s);
        if (t1) CL_HIP(hipEventRecord(t1, s->comm_stream));
    }
    s->pending = true;
    return CLEORA_OK;
}

int join(cleora_sharded *s, hipStream_t stream) {
    if (!s->pending) return CLEORA_OK;
    CL_HIP(hipEventRecord(s->ev_join, s->comm_stream));
    CL_HIP(hipStreamWaitEvent(stream, s->ev_join, 0));
    s->pending = false;
    return CLEORA_OK;
}

// one iteration (see cleora_sharded_propagate_dev); y_local != nullptr: block k's rows go to y_local + first_local * d instead of x_next
int propagate_blocks(cleora_sharded *s, int kind, const float *x, float *x_next, float *y_local, uint32_t d, uint32_t flags, float rw,
                     double *row_sqdiff, bool gather, hipStream_t stream) {
    // A block's in-order hub launch (its longest row is a chain of dependent adds: milliseconds) is joined where its rows are
    // needed — the block's gather, or the end of the call — not before the NEXT block's launch: with P ranks a block's main kernel is
    // 1/P-th of an iteration, and waiting for every block's longest chain in turn would put the chains on the critical path.
    std::vector<hipEvent_t> open_joins;
    for (uint32_t k = 0; k < s->steps; ++k) {
        const auto &b = s->blocks[k];
        float *out = y_local ? y_local + b.first_local * d : x_next + b.b0 * (uint64_t)d;
        hipEvent_t hub_join = nullptr;
        const int rc = launch_propagate(b.g, kind, x, d, d, out, d, flags, rw, x + b.b0 * (uint64_t)d, row_sqdiff ? row_sqdiff + b.first_local : nullptr,
                                        nullptr, stream, nullptr, &hub_join);
        if (rc != CLEORA_OK) return rc;
        if (gather && !y_local && s->world > 1) {
            const int rg = gather_step(s, x_next, d, k, stream, hub_join);
            if (rg != CLEORA_OK) return rg;
        } else if (hub_join) {
            open_joins.push_back(hub_join);
        }
    }
    for (hipEvent_t e : open_joins) CL_HIP(hipStreamWaitEvent(stream, e, 0));
    return join(s, stream);
}

int allreduce_f64(cleora_sharded *s, double *buf, uint64_t n, hipStream_t stream) {
    if (s->world == 1) return CLEORA_OK;
    return cleora_allreduce_f64_dev(s->comm, buf, n, stream);
}

__global__ __launch_bounds__(256) void add_f64_kernel(double *__restrict__ dst, const double *__restrict__ src, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

// S[c] = rows * mean[c]: a rank's contribution to the global column sums
__global__ __launch_bounds__(256) void scale_mean_kernel(const double *__restrict__ mean, double rows, uint32_t d, double *__restrict__ out) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c < d) out[c] = rows * mean[c];
}

// G (centred at the rank's own mean) += rows (mean_r - mu)(mean_r - mu)^T: centred at the global mean (exact identity)
__global__ __launch_bounds__(256) void recentre_gram_kernel(double *__restrict__ gram, const double *__restrict__ mean_r,
                                                            const double *__restrict__ mu, double rows, uint32_t d) {
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (uint64_t)d * d) return;
    const uint32_t i = (uint32_t)(e / d), j = (uint32_t)(e % d);
    const double a = mean_r[i] - mu[i], b = mean_r[j] - mu[j];
    gram[e] += rows * (i <= j ? a * b : b * a);               // the same product for (i, j) and (j, i): symmetric to the bit
}

// Whitening statistics of a REPLICATED matrix y (every rank holds all n rows): each rank takes one contiguous range of rows in ONE
// pass (launch_whiten_fit_stats), the (mean, centred Gram) pairs are combined exactly — column sums and Gram all-reduced.
// Leaves mean64 / mean32 / gram (d x d f64, not yet divided by n - 1) in `st`.
struct StatBufs {
    DevMem ws, mean_r, gram, colsum, mean64, mean32, transform, eigh, verdict;
    uint64_t stat_rows = 0;
    int alloc(uint64_t rows_for_stats, uint32_t d) {
        stat_rows = rows_for_stats;
        int rc;
        if ((rc = ws.alloc(whiten_workspace(std::max<uint64_t>(rows_for_stats, 2), d))) != CLEORA_OK) return rc;
        if ((rc = mean_r.alloc((uint64_t)d * 8)) != CLEORA_OK || (rc = gram.alloc((uint64_t)d * d * 8)) != CLEORA_OK ||
            (rc = colsum.alloc((uint64_t)d * 8)) != CLEORA_OK || (rc = mean64.alloc((uint64_t)d * 8)) != CLEORA_OK ||
            (rc = mean32.alloc((uint64_t)d * 4)) != CLEORA_OK || (rc = transform.alloc((uint64_t)d * d * 4)) != CLEORA_OK ||
            (rc = eigh.alloc(eigh_workspace(d))) != CLEORA_OK || (rc = verdict.alloc(64)) != CLEORA_OK)
            return rc;
        return CLEORA_OK;
    }
};

void stat_range(const cleora_sharded *s, uint64_t *r0, uint64_t *rows) {
    // rank r: rows [n r / P, n (r + 1) / P) — unless that would leave a rank with a single row (the one-pass statistics need two):
    // then rank 0 takes everything and the others contribute zeros
    const uint64_t P = (uint64_t)s->world, n = s->n;
    if (n < 2 * P) {
        *r0 = 0;
        *rows = s->rank == 0 ? n : 0;
        return;
    }
    *r0 = n * (uint64_t)s->rank / P;
    *rows = n * ((uint64_t)s->rank + 1) / P - *r0;
}

int replicated_stats(cleora_sharded *s, const float *y, uint32_t d, StatBufs &st, bool intermediate, hipStream_t stream) {
    uint64_t r0, rows;
    stat_range(s, &r0, &rows);
    int rc;
    if (rows >= 2) {
        if ((rc = launch_whiten_fit_stats(y + r0 * (uint64_t)d, d, rows, d, st.ws.p, stream, 2, intermediate)) != CLEORA_OK) return rc;
        if ((rc = whiten_fit_copy_stats(st.ws.p, rows, d, st.mean_r.as<double>(), st.gram.as<double>(), stream)) != CLEORA_OK) return rc;
    } else {
        CL_HIP(hipMemsetAsync(st.mean_r.p, 0, (uint64_t)d * 8, stream));
        CL_HIP(hipMemsetAsync(st.gram.p, 0, (uint64_t)d * d * 8, stream));
    }
    if (s->world == 1) {                                           // nothing to combine
        CL_HIP(hipMemcpyAsync(st.mean64.p, st.mean_r.p, (uint64_t)d * 8, hipMemcpyDeviceToDevice, stream));
        hipLaunchKernelGGL(scale_mean_kernel, dim3((d + 255) / 256), dim3(256), 0, stream, st.mean_r.as<double>(), (double)rows, d, st.colsum.as<double>());
        return launch_mean(st.colsum.as<double>(), s->n, d, st.mean64.as<double>(), st.mean32.as<float>(), stream);
    }
    hipLaunchKernelGGL(scale_mean_kernel, dim3((d + 255) / 256), dim3(256), 0, stream, st.mean_r.as<double>(), (double)rows, d, st.colsum.as<double>());
    if ((rc = allreduce_f64(s, st.colsum.as<double>(), d, stream)) != CLEORA_OK) return rc;
    if ((rc = launch_mean(st.colsum.as<double>(), s->n, d, st.mean64.as<double>(), st.mean32.as<float>(), stream)) != CLEORA_OK) return rc;
    const uint64_t elems = (uint64_t)d * d;
    hipLaunchKernelGGL(recentre_gram_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, stream, st.gram.as<double>(), st.mean_r.as<double>(),
                       st.mean64.as<double>(), (double)rows, d);
    CL_HIP(hipGetLastError());
    return allreduce_f64(s, st.gram.as<double>(), elems, stream);
}

// Whitening statistics of a matrix of which every rank holds only ITS row blocks (y_local: the rank's rows back to back):
// the reference's two passes (pycleora/__init__.py:136-143) — f64 column sums, all-reduce, f64 Gram centred at the global mean, all-reduce.
int partitioned_stats(cleora_sharded *s, const float *y_local, uint32_t d, StatBufs &st, hipStream_t stream) {
    int rc;
    DevMem tmp, cws;
    uint64_t longest = 1;
    for (auto &b : s->blocks) longest = std::max(longest, b.valid);
    if ((rc = tmp.alloc((uint64_t)d * d * 8)) != CLEORA_OK || (rc = cws.alloc(colsum_workspace(longest, d) * 8)) != CLEORA_OK) return rc;
    DevMem gws;
    if ((rc = gws.alloc(gram_workspace(longest, d) * 8)) != CLEORA_OK) return rc;
    CL_HIP(hipMemsetAsync(st.colsum.p, 0, (uint64_t)d * 8, stream));
    for (auto &b : s->blocks) {
        if (!b.valid) continue;
        if ((rc = launch_colsum(y_local + b.first_local * (uint64_t)d, d, b.valid, d, cws.as<double>(), tmp.as<double>(), stream)) != CLEORA_OK) return rc;
        hipLaunchKernelGGL(add_f64_kernel, dim3((d + 255) / 256), dim3(256), 0, stream, st.colsum.as<double>(), tmp.as<double>(), (uint64_t)d);
    }
    if ((rc = allreduce_f64(s, st.colsum.as<double>(), d, stream)) != CLEORA_OK) return rc;
    if ((rc = launch_mean(st.colsum.as<double>(), s->n, d, st.mean64.as<double>(), st.mean32.as<float>(), stream)) != CLEORA_OK) return rc;
    const uint64_t elems = (uint64_t)d * d;
    CL_HIP(hipMemsetAsync(st.gram.p, 0, elems * 8, stream));
    for (auto &b : s->blocks) {
        if (!b.valid) continue;
        if ((rc = launch_gram(y_local + b.first_local * (uint64_t)d, d, b.valid, d, st.mean64.as<double>(), gws.as<double>(), tmp.as<double>(), stream)) != CLEORA_OK) return rc;
        hipLaunchKernelGGL(add_f64_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, stream, st.gram.as<double>(), tmp.as<double>(), elems);
    }
    CL_HIP(hipGetLastError());
    rc = allreduce_f64(s, st.gram.as<double>(), elems, stream);
    CL_HIP(hipStreamSynchronize(stream));                           // the temporaries die with this call
    return rc;
}

// The transform from the all-reduced Gram, identical on every rank: the decision Cholesky / PCA is taken TOGETHER (a rank that
// disagreed would leave the others inside a collective), and the result is broadcast from rank 0 so the ranks cannot drift apart.
// any_whitening: an intermediate iteration of the L2-normalised loop.  *refused (may be NULL): the Cholesky form was refused on
// statistics that are only approximate — the caller recomputes them in f64 and calls again with any_whitening = false.
int replicated_transform(cleora_sharded *s, uint32_t d, StatBufs &st, bool any_whitening, bool approximate, bool *refused, hipStream_t stream) {
    if (refused) *refused = false;
    int rc = 1;
    if (any_whitening) {
        rc = launch_whiten_transform_cholesky(st.gram.as<double>(), s->n, d, st.transform.as<float>(), st.eigh.p, stream, approximate);
        if (rc < 0) return rc;
        if (s->world > 1) {                                          // 1 on any rank -> 1 on all
            float v = (float)rc;
            CL_HIP(hipMemcpyAsync(st.verdict.p, &v, sizeof v, hipMemcpyHostToDevice, stream));
            int ra = cleora_allreduce_f32_dev(s->comm, st.verdict.as<float>(), 1, stream);
            if (ra != CLEORA_OK) return ra;
            CL_HIP(hipMemcpyAsync(&v, st.verdict.p, sizeof v, hipMemcpyDeviceToHost, stream));
            CL_HIP(hipStreamSynchronize(stream));
            rc = v > 0.5f ? 1 : 0;
        }
    }
    if (rc == 1 && any_whitening && approximate) {
        CL_REQUIRE(refused != nullptr, "internal: approximate statistics need a caller that can recompute them");
        *refused = true;
        return CLEORA_OK;
    }
    if (rc == 1 && (rc = launch_whiten_transform(st.gram.as<double>(), s->n, d, d, st.transform.as<float>(), nullptr, st.eigh.p, stream)) != CLEORA_OK)
        return rc;
    if (s->world > 1) return cleora_broadcast_dev(s->comm, st.transform.p, (uint64_t)d * d * 4, 0, stream);
    return CLEORA_OK;
}

}  // namespace
}  // namespace cleora

using namespace cleora;

extern "C" {

int cleora_sharded_plan(uint64_t n, const uint64_t *rowptr_host, uint32_t world, uint32_t steps, int balance, uint64_t *bounds_out,
                        uint64_t *n_pad_out, int *mode_out) {
    CL_REQUIRE(bounds_out != nullptr, "bounds_out is NULL");
    std::vector<uint64_t> b;
    uint64_t n_pad = 0;
    int mode = 0;
    const int rc = plan_rows(n, rowptr_host, world, steps, balance, b, &n_pad, &mode);
    if (rc != CLEORA_OK) return rc;
    std::memcpy(bounds_out, b.data(), b.size() * sizeof(uint64_t));
    if (n_pad_out) *n_pad_out = n_pad;
    if (mode_out) *mode_out = mode;
    return CLEORA_OK;
}

int cleora_sharded_create(cleora_comm *comm, int device, uint64_t n, uint64_t nnz, const uint64_t *rowptr, const uint32_t *col,
                          const float *val_left, const float *val_sym, int arrays_on_device, uint32_t steps, int balance,
                          cleora_sharded **out) {
    CL_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    CL_REQUIRE(rowptr != nullptr, "rowptr is NULL");
    CL_REQUIRE(nnz == 0 || (col != nullptr && val_left != nullptr), "col / val_left is NULL");
    CL_REQUIRE(steps >= 1 && steps <= 1024, "steps per iteration: 1 .. 1024");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device available; libcleora_hip has no CPU fallback");
        return CLEORA_E_NODEVICE;
    }
    CL_HIP(hipSetDevice(device));
    cleora_sharded *s = new (std::nothrow) cleora_sharded();
    if (!s) { set_error("host allocation failed"); return CLEORA_E_OOM; }
    s->comm = comm;
    s->device = device;
    if (comm) { s->rank = comm->rank; s->world = comm->world; }
    s->n = n;
    s->steps = steps;
    s->has_sym = val_sym != nullptr;
    auto fail = [&](int rc) { free_sharded(s); return rc; };
    // the row pointers on the host: the plan reads them at the cut points (and, balanced on work, searches them)
    std::vector<uint64_t> rp_copy;
    const uint64_t *rp = rowptr;
    if (arrays_on_device) {
        rp_copy.resize(n + 1);
        if (hipMemcpy(rp_copy.data(), rowptr, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); set_error("copying rowptr to the host failed"); return fail(CLEORA_E_HIP); }
        rp = rp_copy.data();
    }
    if (rp[n] != nnz) { set_error("rowptr[n] != nnz"); return fail(CLEORA_E_INVALID); }
    int rc = plan_rows(n, rp, (uint32_t)s->world, steps, balance, s->bounds, &s->n_pad, &s->balance);
    if (rc != CLEORA_OK) return fail(rc);
    if (s->n_pad >= (1ull << 32)) { set_error("more than 2^32 rows (col is u32)"); return fail(CLEORA_E_INVALID); }
    s->blocks.resize(steps);
    uint64_t first_local = 0;
    std::vector<uint64_t> brp;
    for (uint32_t k = 0; k < steps; ++k) {
        auto &b = s->blocks[k];
        b.b0 = s->bounds[(size_t)k * s->world + s->rank];
        b.b1 = s->bounds[(size_t)k * s->world + s->rank + 1];
        const uint64_t r0 = std::min(b.b0, n), r1 = std::min(b.b1, n);
        b.valid = r1 - r0;
        b.first_local = first_local;
        first_local += b.b1 - b.b0;
        const uint64_t e0 = rp[r0], e1 = rp[r1], rows = b.b1 - b.b0;
        brp.assign(rows + 1, e1 - e0);                              // padding rows are empty
        for (uint64_t i = 0; i <= r1 - r0; ++i) brp[i] = rp[r0 + i] - e0;
        s->local_nnz += e1 - e0;
        if (!arrays_on_device) {
            rc = cleora_graph_create(device, rows, s->n_pad, e1 - e0, brp.data(), col + e0, val_left + e0, val_sym ? val_sym + e0 : nullptr, 0, 0, &b.g);
        } else {
            const uint64_t m = e1 - e0;
            hipError_t e = hipMalloc(&b.owned[0], (rows + 1) * sizeof(uint64_t));
            if (e == hipSuccess) e = hipMalloc(&b.owned[1], (m ? m : 1) * sizeof(uint32_t));
            if (e == hipSuccess) e = hipMalloc(&b.owned[2], (m ? m : 1) * sizeof(float));
            if (e == hipSuccess && val_sym) e = hipMalloc(&b.owned[3], (m ? m : 1) * sizeof(float));
            if (e == hipSuccess) e = hipMemcpy(b.owned[0], brp.data(), (rows + 1) * sizeof(uint64_t), hipMemcpyHostToDevice);
            if (e == hipSuccess && m) e = hipMemcpy(b.owned[1], col + e0, m * sizeof(uint32_t), hipMemcpyDeviceToDevice);
            if (e == hipSuccess && m) e = hipMemcpy(b.owned[2], val_left + e0, m * sizeof(float), hipMemcpyDeviceToDevice);
            if (e == hipSuccess && m && val_sym) e = hipMemcpy(b.owned[3], val_sym + e0, m * sizeof(float), hipMemcpyDeviceToDevice);
            if (e != hipSuccess) { (void)hipGetLastError(); set_error(std::string("building a row block on the device failed: ") + hipGetErrorString(e)); return fail(e == hipErrorOutOfMemory ? CLEORA_E_OOM : CLEORA_E_HIP); }
            rc = cleora_graph_create_dev(device, rows, s->n_pad, m, static_cast<const uint64_t *>(b.owned[0]), static_cast<const uint32_t *>(b.owned[1]),
                                         static_cast<const float *>(b.owned[2]), static_cast<const float *>(b.owned[3]), 0, 0, &b.g);
            s->device_bytes += (rows + 1) * 8 + m * (val_sym ? 12 : 8);
        }
        if (rc != CLEORA_OK) return fail(rc);
        cleora_graph_info gi;
        if (cleora_graph_get_info(b.g, &gi) == CLEORA_OK) s->device_bytes += arrays_on_device ? 0 : gi.device_bytes;
    }
    s->local_rows = first_local;
    hipError_t e = hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->side_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_side, hipEventDisableTiming);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("creating streams / events failed"); return fail(CLEORA_E_HIP); }
    *out = s;
    return CLEORA_OK;
}

int cleora_sharded_destroy(cleora_sharded *s) {
    free_sharded(s);
    return CLEORA_OK;
}

int cleora_sharded_get_info(const cleora_sharded *s, cleora_sharded_info *info) {
    CL_REQUIRE(s != nullptr && info != nullptr, "handle / info is NULL");
    info->n = s->n;
    info->n_pad = s->n_pad;
    info->local_rows = s->local_rows;
    info->local_nnz = s->local_nnz;
    info->device_bytes = s->device_bytes;
    info->steps = s->steps;
    info->rank = s->rank;
    info->world = s->world;
    info->balance = s->balance;
    info->has_symmetric = s->has_sym ? 1 : 0;
    return CLEORA_OK;
}

int cleora_sharded_bounds(const cleora_sharded *s, uint64_t *bounds_out) {
    CL_REQUIRE(s != nullptr && bounds_out != nullptr, "handle / bounds is NULL");
    std::memcpy(bounds_out, s->bounds.data(), s->bounds.size() * sizeof(uint64_t));
    return CLEORA_OK;
}

int cleora_sharded_block(const cleora_sharded *s, uint32_t k, cleora_graph **graph, uint64_t *row_begin, uint64_t *row_end) {
    CL_REQUIRE(s != nullptr && k < s->steps, "handle is NULL / no such block");
    if (graph) *graph = s->blocks[k].g;
    if (row_begin) *row_begin = s->blocks[k].b0;
    if (row_end) *row_end = s->blocks[k].b1;
    return CLEORA_OK;
}

int cleora_sharded_set_stream(cleora_sharded *s, void *stream) {
    CL_REQUIRE(s != nullptr, "handle is NULL");
    std::lock_guard<std::mutex> lock(s->mu);
    s->loop_stream = S(stream);
    return CLEORA_OK;
}

int cleora_sharded_set_timing(cleora_sharded *s, int enable) {
    CL_REQUIRE(s != nullptr, "handle is NULL");
    std::lock_guard<std::mutex> lock(s->mu);
    for (auto &b : s->blocks) {
        const int rc = cleora_graph_set_timing(b.g, enable);
        if (rc != CLEORA_OK) return rc;
    }
    s->timing = enable != 0;
    for (hipEvent_t e : s->ev_used) s->ev_pool.push_back(e);
    s->ev_used.clear();
    s->timed_calls = 0;
    return CLEORA_OK;
}

int cleora_sharded_get_timing(cleora_sharded *s, double ms[2], uint64_t *calls) {
    CL_REQUIRE(s != nullptr && ms != nullptr, "handle / ms is NULL");
    std::lock_guard<std::mutex> lock(s->mu);
    CL_HIP(hipSetDevice(s->device));
    ms[0] = ms[1] = 0.0;
    for (auto &b : s->blocks) {
        double k[3];
        uint64_t c = 0;
        const int rc = cleora_graph_get_timing(b.g, k, &c);
        if (rc != CLEORA_OK) return rc;
        ms[0] += k[0] + k[1] + k[2];
    }
    for (size_t i = 0; i + 1 < s->ev_used.size(); i += 2) {
        CL_HIP(hipEventSynchronize(s->ev_used[i + 1]));
        float t = 0.f;
        CL_HIP(hipEventElapsedTime(&t, s->ev_used[i], s->ev_used[i + 1]));
        ms[1] += t;
    }
    for (hipEvent_t e : s->ev_used) s->ev_pool.push_back(e);
    s->ev_used.clear();
    if (calls) *calls = s->timed_calls;
    s->timed_calls = 0;
    return CLEORA_OK;
}

int cleora_sharded_propagate_dev(cleora_sharded *s, int markov_type, const float *x, float *x_next, uint32_t d, uint32_t flags,
                                 float residual_weight, double *row_sqdiff_local, int gather, void *stream) {
    CL_REQUIRE(s != nullptr, "handle is NULL");
    CL_REQUIRE(x != nullptr && x_next != nullptr && x != x_next, "x / x_next is NULL or they alias");
    CL_REQUIRE(d > 0, "d must be positive");
    CL_REQUIRE(markov_type == CLEORA_LEFT || (markov_type == CLEORA_SYMMETRIC && s->has_sym), "no values for this markov_type");
    std::lock_guard<std::mutex> lock(s->mu);
    CL_HIP(hipSetDevice(s->device));
    if (s->timing) ++s->timed_calls;
    return propagate_blocks(s, markov_type, x, x_next, nullptr, d, flags, residual_weight, row_sqdiff_local, gather != 0, S(stream));
}

namespace {
__device__ __forceinline__ float verify_pattern(uint64_t i) {
    return __builtin_bit_cast(float, 0x3f800000u | (uint32_t)((i * 2654435761ull + (i >> 21)) & 0x007fffffu));     // in [1, 2)
}
__global__ __launch_bounds__(256) void verify_fill_kernel(float *buf, uint64_t first, uint64_t count) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256) buf[first + i] = verify_pattern(first + i);
}
// plain loads (the SpMM's load path) of the WHOLE replica against the pattern; *bad (f64: it travels through the all-reduce) += mismatches
__global__ __launch_bounds__(256) void verify_check_kernel(const float *buf, uint64_t count, double *bad) {
    uint32_t mism = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256)
        mism += __builtin_bit_cast(uint32_t, buf[i]) != __builtin_bit_cast(uint32_t, verify_pattern(i));
    if (mism) atomicAdd(bad, (double)mism);
}
}  // namespace

// First use of the peer-direct all-gather by this handle: the same exchange the loops make — every step's row range, this rank's
// blocks, the communicator's all-gather on the communication stream — on one of the REAL registered replicas (`scratch_replica`: the
// loop's second buffer, not yet in use), with a pattern every rank can check everywhere with plain loads.  cleora_comm_enable_peer's
// self-test vouches for a 4 KiB probe buffer; this vouches for the 10-100 GB allocation the SpMM will gather from.  The ranks decide
// together (all-reduced mismatch count): a failing PUSH form is replaced by PULL on every rank and checked again; if that fails too the
// call returns an error instead of results assembled from stale rows.  `flags` of cleora_comm_selftest can be injected through
// cleora_sharded_debug_fail_first_gather (tests).  Leaves the replica zero-filled, as it found it.
int verify_peer_gather(cleora_sharded *s, float *scratch_replica, uint32_t d, hipStream_t stream) {
    if (s->peer_verified || !s->comm || s->world == 1 || peer_mode(s->comm) < 0) return CLEORA_OK;
    if (!(s->comm->allgather_algo == CLEORA_ALLGATHER_PEER || !s->comm->comm)) return CLEORA_OK;
    const uint64_t total = s->n_pad * (uint64_t)d;
    DevMem bad;
    int rc;
    if ((rc = bad.alloc(8)) != CLEORA_OK) return rc;
    for (int attempt = 0; attempt < 2; ++attempt) {
        CL_HIP(hipMemsetAsync(bad.p, 0, 8, stream));
        for (uint32_t k = 0; k < s->steps; ++k) {
            const auto &b = s->blocks[k];
            const uint64_t first = b.b0 * (uint64_t)d, count = (b.b1 - b.b0) * (uint64_t)d;
            if (count) hipLaunchKernelGGL(verify_fill_kernel, dim3(1024), dim3(256), 0, stream, scratch_replica, first, count);
            if ((rc = gather_step(s, scratch_replica, d, k, stream)) != CLEORA_OK) return rc;
        }
        if ((rc = join(s, stream)) != CLEORA_OK) return rc;
        hipLaunchKernelGGL(verify_check_kernel, dim3(4096), dim3(256), 0, stream, scratch_replica, total, bad.as<double>());
        CL_HIP(hipGetLastError());
        if (s->debug_fail_first_gather > attempt) hipLaunchKernelGGL(verify_check_kernel, dim3(1), dim3(256), 0, stream, scratch_replica + 1, 256, bad.as<double>());   // (injected: an off-by-one view never matches)
        if ((rc = allreduce_f64(s, bad.as<double>(), 1, stream)) != CLEORA_OK) return rc;
        double mism = 0.0;
        CL_HIP(hipMemcpyAsync(&mism, bad.p, 8, hipMemcpyDeviceToHost, stream));
        CL_HIP(hipStreamSynchronize(stream));
        if ((rc = cleora_comm_check(s->comm)) != CLEORA_OK) return rc;
        if (mism == 0.0) {
            CL_HIP(hipMemsetAsync(scratch_replica, 0, total * sizeof(float), stream));
            CL_HIP(hipStreamSynchronize(stream));
            if ((rc = peer_host_barrier(s->comm)) != CLEORA_OK) return rc;        // nobody stores into a replica a slower rank is still clearing
            s->peer_verified = true;
            return CLEORA_OK;
        }
        if (peer_mode(s->comm) == 1 || attempt == 1) break;
        peer_set_mode(s->comm, 1);                                               // every rank saw the same (all-reduced) count: all switch
        if ((rc = peer_host_barrier(s->comm)) != CLEORA_OK) return rc;
    }
    set_error("the peer-direct all-gather left stale rows in a replica (first-use check of cleora_embed_sharded, PUSH and PULL forms): "
              "refusing to run the loop on this transport; take an RCCL communicator's all-gather (cleora_comm_set_allgather)");
    return CLEORA_E_RCCL;
}

int cleora_sharded_debug_fail_first_gather(cleora_sharded *s, int attempts) {
    CL_REQUIRE(s != nullptr, "handle is NULL");
    CL_REQUIRE(attempts >= 0 && attempts <= 2, "attempts: 0 (off), 1 (the PUSH form's check fails), 2 (both fail)");
    std::lock_guard<std::mutex> lock(s->mu);
    s->debug_fail_first_gather = attempts;
    s->peer_verified = false;
    return CLEORA_OK;
}

uint64_t cleora_embed_sharded_bytes(uint64_t n_pad, uint64_t local_rows, uint64_t n, uint32_t world, uint32_t d, uint32_t flags) {
    const uint64_t replica = n_pad * (uint64_t)d * 4, local = local_rows * (uint64_t)d * 4;
    if (!(flags & CLEORA_F_WHITEN)) return replica;                 // the partner of the ping-pong pair
    const uint64_t stat_rows = world ? (n + world - 1) / world : n;
    return replica + local + whiten_workspace(stat_rows < 2 ? 2 : stat_rows, d) + (uint64_t)d * d * 12 + eigh_workspace(d) + (uint64_t)d * 32 + 64;
}

// The loops.  x_replica: n_pad x d (ld = d), E_0 in rows [0, n) — identical on every rank —, rows >= n zero; the result comes back in
// the same buffer, replicated.
int cleora_embed_sharded(cleora_sharded *s, float *x_replica, int markov_type, uint32_t d, uint64_t max_iterations,
                         float residual_weight, float convergence_threshold, uint32_t flags, uint64_t *iterations_run) {
    CL_REQUIRE(s != nullptr && x_replica != nullptr, "handle / x is NULL");
    CL_REQUIRE(d > 0, "d must be positive");
    CL_REQUIRE(markov_type == CLEORA_LEFT || (markov_type == CLEORA_SYMMETRIC && s->has_sym), "no values for this markov_type");
    std::lock_guard<std::mutex> lock(s->mu);
    CL_HIP(hipSetDevice(s->device));
    if (iterations_run) *iterations_run = 0;
    if (max_iterations == 0) return CLEORA_OK;
    const uint64_t n = s->n, replica_bytes = s->n_pad * (uint64_t)d * 4;
    const bool whitened = (flags & CLEORA_F_WHITEN) != 0, check = convergence_threshold > 0.0f;
    const uint32_t norm = (flags & CLEORA_F_L1NORM) ? CLEORA_F_L1NORM : CLEORA_F_L2NORM, fast = flags & (CLEORA_F_FASTNORM | CLEORA_F_HUB_SEGMENTS);
    hipStream_t stream = s->loop_stream;                            // default: the null stream of the device
    int rc;
    DevMem other;
    if ((rc = other.alloc(replica_bytes)) != CLEORA_OK) return rc;
    CL_HIP(hipMemsetAsync(other.p, 0, replica_bytes, stream));     // padding rows stay zero in both replicas
    // The memset must have RUN before any peer may store into `other` (peer-direct all-gather): the registration's host barriers
    // order the ranks' host threads only, so drain the stream first — a quicker peer's rows would otherwise be zeroed afterwards
    // (ADVICE round 4).  The same wait covers whatever of the caller's still produces E_0 on this stream.
    CL_HIP(hipStreamSynchronize(stream));
    struct Registered {                                             // peer-direct transport: both replicas mapped by every rank
        cleora_comm *c; void *a, *b; bool on = false;
        ~Registered() { if (on) { (void)cleora_comm_unregister(c, a); (void)cleora_comm_unregister(c, b); } }
    } reg{s->comm, x_replica, other.p};
    // (only where the gathers will use it: an RCCL communicator whose algorithm is one of RCCL's needs no mapping, and a node
    // where the mapping fails must not lose RCCL's loops with it)
    if (s->comm && s->world > 1 && (s->comm->allgather_algo == CLEORA_ALLGATHER_PEER || !s->comm->comm)) {
        if ((rc = cleora_comm_register(s->comm, x_replica, replica_bytes)) != CLEORA_OK) return rc;
        if ((rc = cleora_comm_register(s->comm, other.p, replica_bytes)) != CLEORA_OK) { (void)cleora_comm_unregister(s->comm, x_replica); return rc; }
        reg.on = true;
        if ((rc = verify_peer_gather(s, other.as<float>(), d, stream)) != CLEORA_OK) return rc;
    }
    float *result = nullptr;
    uint64_t ran = max_iterations;
    void *solver_ws = nullptr;                                      // the whitened loops: where the eigensolver leaves its convergence flag
    auto finish = [&]() -> int {
        CL_HIP(hipStreamSynchronize(stream));
        CL_HIP(hipStreamSynchronize(s->comm_stream));
        if (s->comm) { const int rk = cleora_comm_check(s->comm); if (rk != CLEORA_OK) return rk; }
        if (solver_ws) {                                            // like the one-GPU loops (abi.hip embed_whitened*): non-convergence is an error
            int info = 0;
            CL_HIP(hipMemcpy(&info, transform_info(solver_ws, d), sizeof info, hipMemcpyDeviceToHost));
            if (info != 0) { set_error("the eigensolver did not converge (rocsolver_dsyevd info = " + std::to_string(info) + ")"); return CLEORA_E_HIP; }
        }
        if (result != x_replica) CL_HIP(hipMemcpy(x_replica, result, replica_bytes, hipMemcpyDeviceToDevice));
        if (iterations_run) *iterations_run = ran;
        return CLEORA_OK;
    };

    if (!whitened) {
        // embed_full / embed_full_with_convergence (src/embedding.rs:106-188): SpMM, residual for 0 < rw < 1, L2, swap; RMSE from iteration 1
        DevMem sq, rws, total;
        if (check && ((rc = sq.alloc(std::max<uint64_t>(s->local_rows, 1) * 8)) != CLEORA_OK || (rc = rws.alloc(reduce_workspace(std::max<uint64_t>(s->local_rows, 1)) * 8)) != CLEORA_OK ||
                      (rc = total.alloc(8)) != CLEORA_OK))
            return rc;
        const uint32_t base = CLEORA_F_L2NORM | CLEORA_F_RESIDUAL | fast;
        float *src = x_replica, *dst = other.as<float>();
        for (uint64_t it = 0; it < max_iterations; ++it) {
            const bool test = check && it > 0;                      // embedding.rs:169
            if ((rc = propagate_blocks(s, markov_type, src, dst, nullptr, d, base | (test ? CLEORA_F_SQDIFF : 0u), residual_weight,
                                       test ? sq.as<double>() : nullptr, true, stream)) != CLEORA_OK)
                return rc;
            std::swap(src, dst);
            if (test) {
                if ((rc = launch_reduce_sum(sq.as<double>(), s->local_rows, rws.as<double>(), total.as<double>(), stream)) != CLEORA_OK) return rc;
                if ((rc = allreduce_f64(s, total.as<double>(), 1, stream)) != CLEORA_OK) return rc;
                double sum = 0.0;
                CL_HIP(hipMemcpyAsync(&sum, total.p, 8, hipMemcpyDeviceToHost, stream));
                CL_HIP(hipStreamSynchronize(stream));
                const float rmse = sqrtf((float)(sum / (double)(n * (uint64_t)d)));      // embedding.rs:177-178
                if (rmse < convergence_threshold) { ran = it + 1; break; }
            }
        }
        result = src;
        return finish();
    }

    CL_REQUIRE(n >= 2, "the partitioned whitened loop needs at least two entities");
    StatBufs st;
    uint64_t sr0, srows;
    stat_range(s, &sr0, &srows);
    if ((rc = st.alloc(srows, d)) != CLEORA_OK) return rc;
    solver_ws = st.eigh.p;
    CL_HIP(hipMemsetAsync(const_cast<int *>(transform_info(solver_ws, d)), 0, sizeof(int), stream));
    DevMem local;                                                   // this rank's rows only: Z = A Y (reorganised loop) or the normalised rows (reference order)
    if ((rc = local.alloc(std::max<uint64_t>(s->local_rows, 1) * (uint64_t)d * 4)) != CLEORA_OK) return rc;
    const float rw = residual_weight;
    // project the rank's rows of `in_local` (or of the replica `in_rep`) into the replica `out`, one block at a time, and gather
    // (rowabs != nullptr: the operand is bounded row by row — the loop form without a blend at d = 256 takes the f16 projection,
    // project_f16.hip, like the one-GPU loop: abi.hip)
    auto project_blocks = [&](const float *in_local, const float *in_rep, float *out, const float *x2_rep, const float *rowsum, bool loop_form,
                              int norm_mode, const float *rowabs = nullptr) -> int {
        for (uint32_t k = 0; k < s->steps; ++k) {
            const auto &b = s->blocks[k];
            if (b.valid) {
                const float *in = in_local ? in_local + b.first_local * (uint64_t)d : in_rep + b.b0 * (uint64_t)d;
                float *o = out + b.b0 * (uint64_t)d;
                bool normed = false;
                int r2;
                if (loop_form && rowabs && !x2_rep && norm_mode == 1 && project_f16_applies(in, d, b.valid, d, d, o, d, nullptr)) {
                    r2 = launch_project_f16(in, d, b.valid, st.mean32.as<float>(), st.transform.as<float>(), o, d, stream, rowsum + b.first_local,
                                            rowabs + b.first_local, 1);
                    normed = true;
                } else {
                    r2 = launch_project(in, d, b.valid, d, st.mean32.as<float>(), st.transform.as<float>(), d, o, d, stream,
                                        loop_form ? rowsum + b.first_local : nullptr, x2_rep ? x2_rep + b.b0 * (uint64_t)d : nullptr, d,
                                        loop_form ? 1.0f - rw : 1.0f, loop_form ? rw : 0.0f, norm_mode, &normed,
                                        loop_form && rowabs && !x2_rep ? rowabs + b.first_local : nullptr);
                }
                if (r2 != CLEORA_OK) return r2;
                if (norm_mode && !normed && (r2 = launch_rowops(o, d, b.valid, d, o, d, norm | fast | (loop_form ? CLEORA_F_FASTNORM : 0u), 0.f, nullptr, nullptr, nullptr, stream)) != CLEORA_OK) return r2;
            }
            const int rg = gather_step(s, out, d, k, stream);
            if (rg != CLEORA_OK) return rg;
        }
        return join(s, stream);
    };

    const bool reorganised = norm == CLEORA_F_L2NORM && !check;     // nobody sees the intermediate whitened iterates; rotation invariance needs the L2 norm
    if (reorganised) {
        // Y_0 = normalise(A E_0 [+ blend]) replicated; per iteration Z = A Y on the rank's rows | statistics of Y (contiguous row
        // ranges of the replica: one pass each) -> all-reduce -> transform (Cholesky while the clamp guard allows) -> Y' =
        // normalise((alpha (Z - s mu^T) + rw (Y - mu)) T) on the rank's rows, gathered block by block; E_T = PCA-whiten(Y_{T-1}).
        DevMem rowsum, rowabs;                                      // A 1 and sum |a| per row (the f16 projection's row bounds)
        if ((rc = rowsum.alloc(std::max<uint64_t>(s->local_rows, 1) * 4)) != CLEORA_OK) return rc;
        if ((rc = rowabs.alloc(std::max<uint64_t>(s->local_rows, 1) * 4)) != CLEORA_OK) return rc;
        for (auto &b : s->blocks)
            if ((rc = launch_csr_rowsum(b.g, markov_type, rowsum.as<float>() + b.first_local, stream, rowabs.as<float>() + b.first_local)) != CLEORA_OK) return rc;
        float *y = other.as<float>(), *ynext = x_replica;
        if ((rc = propagate_blocks(s, markov_type, x_replica, y, nullptr, d, CLEORA_F_L2NORM | fast | CLEORA_F_RESIDUAL | CLEORA_F_BLEND_ANY, rw, nullptr, true,
                                   stream)) != CLEORA_OK)
            return rc;
        const bool blend = rw > 0.0f;
        // ONE answer for all ranks (their statistics ranges differ by a row: around the split form's 4 096-row threshold a per-rank
        // answer would send some ranks back into replicated_stats — two all-reduces — and the others on to the broadcast: ADVICE
        // round 4): "approximate" as soon as ANY rank's range may have taken the split-bf16 form
        const uint64_t Pw = (uint64_t)s->world;
        const bool split_stats = n >= 2 * Pw && gram32_applies(y, d, (n + Pw - 1) / Pw, d);
        for (uint64_t it = 0; it + 1 < max_iterations; ++it) {
            // the statistics first (alone on the chip: docs/history.md §3.8), then Z = A Y on a second stream beside the all-reduces and the
            // d x d step (the host blocks there), the projection when both are through
            if ((rc = replicated_stats(s, y, d, st, true, stream)) != CLEORA_OK) return rc;
            CL_HIP(hipEventRecord(s->ev_side, stream));
            CL_HIP(hipStreamWaitEvent(s->side_stream, s->ev_side, 0));
            if ((rc = propagate_blocks(s, markov_type, y, nullptr, local.as<float>(), d, 0, 0.f, nullptr, false, s->side_stream)) != CLEORA_OK) return rc;
            bool refused = false;
            if ((rc = replicated_transform(s, d, st, true, split_stats, &refused, stream)) != CLEORA_OK) return rc;
            if (refused) {
                if ((rc = replicated_stats(s, y, d, st, false, stream)) != CLEORA_OK) return rc;
                if ((rc = replicated_transform(s, d, st, false, false, nullptr, stream)) != CLEORA_OK) return rc;
            }
            CL_HIP(hipEventRecord(s->ev_side, s->side_stream));
            CL_HIP(hipStreamWaitEvent(stream, s->ev_side, 0));
            if ((rc = project_blocks(local.as<float>(), nullptr, ynext, blend ? y : nullptr, rowsum.as<float>(), true, 1, blend ? nullptr : rowabs.as<float>())) != CLEORA_OK) return rc;
            std::swap(y, ynext);
        }
        // E_T = whiten(Y_{T-1}): f64 statistics, the PCA form
        if ((rc = replicated_stats(s, y, d, st, false, stream)) != CLEORA_OK) return rc;
        if ((rc = replicated_transform(s, d, st, false, false, nullptr, stream)) != CLEORA_OK) return rc;
        if ((rc = project_blocks(nullptr, y, ynext, nullptr, nullptr, false, 0)) != CLEORA_OK) return rc;
        result = ynext;
        return finish();
    }

    // the reference's order (pycleora/__init__.py:109-125): propagate, blend for ANY rw > 0, normalise, whiten — every iteration;
    // f64 RMSE between whitened iterates for the early stop (:122-125, :974-976)
    DevMem sq, rws, total;
    if (check && ((rc = sq.alloc(std::max<uint64_t>(s->local_rows, 1) * 8)) != CLEORA_OK || (rc = rws.alloc(reduce_workspace(std::max<uint64_t>(s->local_rows, 1)) * 8)) != CLEORA_OK ||
                  (rc = total.alloc(8)) != CLEORA_OK))
        return rc;
    DevMem part;                                                    // (allocated once: hipFree inside the loop is a device-wide wait)
    if (check && (rc = part.alloc(8)) != CLEORA_OK) return rc;
    float *prev = x_replica, *next = other.as<float>();
    for (uint64_t it = 0; it < max_iterations; ++it) {
        if ((rc = propagate_blocks(s, markov_type, prev, nullptr, local.as<float>(), d, norm | fast | CLEORA_F_RESIDUAL | CLEORA_F_BLEND_ANY, rw, nullptr, false,
                                   stream)) != CLEORA_OK)
            return rc;
        if ((rc = partitioned_stats(s, local.as<float>(), d, st, stream)) != CLEORA_OK) return rc;
        if ((rc = replicated_transform(s, d, st, false, false, nullptr, stream)) != CLEORA_OK) return rc;
        if ((rc = project_blocks(local.as<float>(), nullptr, next, nullptr, nullptr, false, 0)) != CLEORA_OK) return rc;
        if (check && it > 0) {
            for (auto &b : s->blocks) {
                if (!b.valid) continue;
                if ((rc = launch_rowops(next + b.b0 * (uint64_t)d, d, b.valid, d, next + b.b0 * (uint64_t)d, d, CLEORA_F_SQDIFF | CLEORA_F_SQDIFF64, 0.f,
                                        prev + b.b0 * (uint64_t)d, sq.as<double>() + b.first_local, nullptr, stream)) != CLEORA_OK)
                    return rc;
            }
            // padding rows of a block carry no valid difference: their slots were never written — sum the valid prefix of every block
            double sum = 0.0;
            CL_HIP(hipMemsetAsync(total.p, 0, 8, stream));
            for (auto &b : s->blocks) {
                if (!b.valid) continue;
                if ((rc = launch_reduce_sum(sq.as<double>() + b.first_local, b.valid, rws.as<double>(), part.as<double>(), stream)) != CLEORA_OK) return rc;
                hipLaunchKernelGGL(add_f64_kernel, dim3(1), dim3(256), 0, stream, total.as<double>(), part.as<double>(), (uint64_t)1);
            }
            if ((rc = allreduce_f64(s, total.as<double>(), 1, stream)) != CLEORA_OK) return rc;
            CL_HIP(hipMemcpyAsync(&sum, total.p, 8, hipMemcpyDeviceToHost, stream));
            CL_HIP(hipStreamSynchronize(stream));
            if (sqrt(sum / (double)(n * (uint64_t)d)) < (double)convergence_threshold) {      // _compute_rmse, :974-976
                std::swap(prev, next);
                ran = it + 1;
                break;
            }
        }
        std::swap(prev, next);
    }
    result = prev;
    return finish();
}

}  // extern "C"
