// colsharded.hip — the COLUMN partition of the Markov propagation through the C ABI: the comparison layout beside the row
// partition north_star mandates (csrc/sharded.hip), for the "measured comparison" SURVEY.md §7 asks for — both sides through the same
// kind of entry point.  The reference has no distributed code (src/embedding.rs:59-63 is rayon over rows).
//
//   rank r of P owns columns [r d/P, (r + 1) d/P) of every row of the iterate and the WHOLE CSR.  "Cleora operates on dimensions
//   independently" (reference README.md:361): the SpMM of a column slice needs nothing from the other slices — no n x d data ever
//   crosses xGMI.  Only the row L2 norm couples the columns (src/embedding.rs:88-104: sum_sq over j = 0 .. d-1 IN ORDER).
//   An all-reduce of per-slice partial sums would add them in another order (last-ulp differences: what the Python model of
//   round 2-5 did).  Here the running sum travels: rank 0 sums its columns from 0, rank 1 CONTINUES from rank 0's value
//   (CLEORA_F_ROWSQ | CLEORA_F_ROWSQ_CONT), ... rank P-1 ends with the one-GPU sum, bit for bit, and broadcasts it; every rank
//   scales its slice (CLEORA_F_SCALE).  P hand-offs per row block, each a broadcast of the block's n / K floats on the
//   communication stream — block k's chain runs beside block k + 1's SpMM.
//   Per rank and iteration: gathers of nnz d/P 4 B, the full (col, val) streams, three passes over its n x d/P slice, P K small
//   broadcasts.  Results: propagate and the plain loop (embed_full / embed_full_with_convergence, src/embedding.rs:106-188) are
//   bit-equal to the one-GPU calls; the whitened loop is not offered here (the Gram couples all columns: the row partition's job).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "comm_internal.h"

struct cleora_colsharded {
    cleora_comm *comm = nullptr;             // nullptr: a world of one
    int rank = 0, world = 1, device = 0;
    uint64_t n = 0, nnz = 0;
    uint32_t d_total = 0, d_local = 0, steps = 1;
    bool has_sym = false;
    struct Block { cleora_graph *g = nullptr; uint64_t r0 = 0, r1 = 0; void *rowptr_dev = nullptr; };   // rowptr_dev: the block's own row pointers (device-array form)
    std::vector<Block> blocks;               // row blocks of the ONE CSR every rank holds (views when the arrays are the caller's)
    float *rowsq = nullptr;                  // f32[n]: the rows' running sums of squares
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_block = nullptr, ev_join = nullptr;
    std::mutex mu;
};

using namespace cleora;

namespace {
inline hipStream_t S(void *s) { return static_cast<hipStream_t>(s); }

struct DevMem {
    void *p = nullptr;
    ~DevMem() { if (p) (void)hipFree(p); }
    int alloc(uint64_t bytes) {
        if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { (void)hipGetLastError(); p = nullptr; set_error("out of device memory"); return CLEORA_E_OOM; }
        return CLEORA_OK;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

void free_colsharded(cleora_colsharded *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    for (auto &b : s->blocks) {
        if (b.g) (void)cleora_graph_destroy(b.g);
        if (b.rowptr_dev) (void)hipFree(b.rowptr_dev);
    }
    if (s->rowsq) (void)hipFree(s->rowsq);
    if (s->comm_stream) (void)hipStreamDestroy(s->comm_stream);
    if (s->ev_block) (void)hipEventDestroy(s->ev_block);
    if (s->ev_join) (void)hipEventDestroy(s->ev_join);
    delete s;
}

// one iteration on this rank's column slice: y = epilogue(A x) with flags of cleora_propagate_dev out of {L2NORM, RESIDUAL, BLEND_ANY,
// SQDIFF, SQDIFF64, HUB_SEGMENTS}; row_sqdiff: this rank's part of every row's squared difference (the caller adds the ranks')
int propagate_cols(cleora_colsharded *s, int kind, const float *x, float *y, uint32_t flags, float rw, double *row_sqdiff, hipStream_t stream) {
    const uint32_t dl = s->d_local;
    const uint32_t hub = flags & CLEORA_F_HUB_SEGMENTS;
    int rc;
    if (s->world == 1) {                       // nothing to hand over: the fused epilogue
        for (auto &b : s->blocks) {
            if (b.r1 == b.r0) continue;
            if ((rc = launch_propagate(b.g, kind, x, dl, dl, y + b.r0 * (uint64_t)dl, dl, flags, rw, x + b.r0 * (uint64_t)dl,
                                       row_sqdiff ? row_sqdiff + b.r0 : nullptr, nullptr, stream)) != CLEORA_OK)
                return rc;
        }
        return CLEORA_OK;
    }
    const bool norm = (flags & CLEORA_F_L2NORM) != 0;
    const uint32_t first = (flags & (CLEORA_F_RESIDUAL | CLEORA_F_BLEND_ANY)) | hub;      // the SpMM with the blend, unnormalised
    for (auto &b : s->blocks) {
        const uint64_t rows = b.r1 - b.r0;
        if (rows) {
            if ((rc = launch_propagate(b.g, kind, x, dl, dl, y + b.r0 * (uint64_t)dl, dl, first, rw, x + b.r0 * (uint64_t)dl, nullptr, nullptr, stream)) != CLEORA_OK)
                return rc;
        }
        if (!norm) continue;
        // the block's chain on the communication stream, beside the next block's SpMM: rank p extends the running sums over its
        // columns and hands them on (every rank receives them: one collective per hand-off, the last one IS the result)
        CL_HIP(hipEventRecord(s->ev_block, stream));
        CL_HIP(hipStreamWaitEvent(s->comm_stream, s->ev_block, 0));
        for (int p = 0; p < s->world; ++p) {
            if (p == s->rank && rows) {
                float *yb = y + b.r0 * (uint64_t)dl;
                if ((rc = launch_rowops(yb, dl, rows, dl, yb, dl, CLEORA_F_ROWSQ | (p ? CLEORA_F_ROWSQ_CONT : 0u), 0.f, nullptr, nullptr, s->rowsq + b.r0,
                                        s->comm_stream)) != CLEORA_OK)
                    return rc;
            }
            if (rows && (rc = cleora_broadcast_dev(s->comm, s->rowsq + b.r0, rows * sizeof(float), p, s->comm_stream)) != CLEORA_OK) return rc;
        }
    }
    if (norm) {
        CL_HIP(hipEventRecord(s->ev_join, s->comm_stream));
        CL_HIP(hipStreamWaitEvent(stream, s->ev_join, 0));
    }
    const uint32_t second = (norm ? CLEORA_F_SCALE : 0u) | (flags & (CLEORA_F_SQDIFF | CLEORA_F_SQDIFF64));
    if (second) {
        for (auto &b : s->blocks) {
            const uint64_t rows = b.r1 - b.r0;
            if (!rows) continue;
            float *yb = y + b.r0 * (uint64_t)dl;
            if ((rc = launch_rowops(yb, dl, rows, dl, yb, dl, second, 0.f, (flags & CLEORA_F_SQDIFF) ? x + b.r0 * (uint64_t)dl : nullptr,
                                    row_sqdiff ? row_sqdiff + b.r0 : nullptr, norm ? s->rowsq + b.r0 : nullptr, stream)) != CLEORA_OK)
                return rc;
        }
    }
    return CLEORA_OK;
}
}  // namespace

extern "C" {

int cleora_colsharded_create(cleora_comm *comm, int device, uint64_t n, uint64_t nnz, const uint64_t *rowptr, const uint32_t *col,
                             const float *val_left, const float *val_sym, int arrays_on_device, uint32_t d_total, uint32_t steps,
                             cleora_colsharded **out) {
    CL_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    CL_REQUIRE(rowptr != nullptr && (nnz == 0 || (col != nullptr && val_left != nullptr)), "rowptr / col / val_left is NULL");
    CL_REQUIRE(d_total > 0, "d must be positive");
    int rank = 0, world = 1;
    if (comm) {
        int dev = 0;
        const int rc = cleora_comm_info(comm, &rank, &world, &dev);
        if (rc != CLEORA_OK) return rc;
        CL_REQUIRE(dev == device, "the communicator lives on another device");
    }
    CL_REQUIRE(d_total % (uint32_t)world == 0, "the width must be divisible by the number of ranks");
    CL_REQUIRE(n < (1ull << 32), "more than 2^32 rows (col is u32)");
    CL_HIP(hipSetDevice(device));
    cleora_colsharded *s = new (std::nothrow) cleora_colsharded();
    if (!s) { set_error("host allocation failed"); return CLEORA_E_OOM; }
    s->comm = world > 1 ? comm : nullptr;
    s->rank = rank;
    s->world = world;
    s->device = device;
    s->n = n;
    s->nnz = nnz;
    s->d_total = d_total;
    s->d_local = d_total / (uint32_t)world;
    s->steps = std::max<uint32_t>(1, std::min<uint64_t>(steps ? steps : 1, std::max<uint64_t>(n, 1)));
    s->has_sym = val_sym != nullptr;
    auto fail = [&](int rc) { free_colsharded(s); return rc; };
    // the row pointers on the host: the blocks are cut on them
    std::vector<uint64_t> rp_copy;
    const uint64_t *rp = rowptr;
    if (arrays_on_device) {
        rp_copy.resize(n + 1);
        if (hipMemcpy(rp_copy.data(), rowptr, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); set_error("copying rowptr to the host failed"); return fail(CLEORA_E_HIP); }
        rp = rp_copy.data();
    }
    if (rp[n] != nnz) { set_error("rowptr[n] != nnz"); return fail(CLEORA_E_INVALID); }
    s->blocks.resize(s->steps);
    std::vector<uint64_t> brp;
    for (uint32_t k = 0; k < s->steps; ++k) {
        auto &b = s->blocks[k];
        b.r0 = n * k / s->steps;
        b.r1 = n * (k + 1) / s->steps;
        const uint64_t e0 = rp[b.r0], e1 = rp[b.r1], rows = b.r1 - b.r0;
        brp.resize(rows + 1);
        for (uint64_t i = 0; i <= rows; ++i) brp[i] = rp[b.r0 + i] - e0;
        int rc;
        if (!arrays_on_device) {
            rc = cleora_graph_create(device, rows, n, e1 - e0, brp.data(), col + e0, val_left + e0, val_sym ? val_sym + e0 : nullptr, 0, 0, &b.g);
        } else {
            // (col / val slices are views of the caller's device arrays, which must outlive the handle — every rank needs the whole CSR,
            // a copy per handle would double it; the block's row pointers are the handle's own)
            if (hipMalloc(&b.rowptr_dev, (rows + 1) * sizeof(uint64_t)) != hipSuccess) { (void)hipGetLastError(); set_error("out of device memory"); return fail(CLEORA_E_OOM); }
            if (hipMemcpy(b.rowptr_dev, brp.data(), (rows + 1) * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); set_error("uploading a block's row pointers failed"); return fail(CLEORA_E_HIP); }
            rc = cleora_graph_create_dev(device, rows, n, e1 - e0, static_cast<const uint64_t *>(b.rowptr_dev), col + e0, val_left + e0, val_sym ? val_sym + e0 : nullptr, 0, 0, &b.g);
        }
        if (rc != CLEORA_OK) return fail(rc);
    }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&s->rowsq), std::max<uint64_t>(n, 1) * sizeof(float));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_block, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("creating the column partition's streams / scratch failed"); return fail(e == hipErrorOutOfMemory ? CLEORA_E_OOM : CLEORA_E_HIP); }
    *out = s;
    return CLEORA_OK;
}

int cleora_colsharded_destroy(cleora_colsharded *s) {
    free_colsharded(s);
    return CLEORA_OK;
}

int cleora_colsharded_get_info(const cleora_colsharded *s, cleora_colsharded_info *info) {
    CL_REQUIRE(s != nullptr && info != nullptr, "handle / info is NULL");
    info->n = s->n;
    info->nnz = s->nnz;
    info->d_total = s->d_total;
    info->d_local = s->d_local;
    info->col_begin = s->d_local * (uint32_t)s->rank;
    info->steps = s->steps;
    info->rank = s->rank;
    info->world = s->world;
    info->has_symmetric = s->has_sym ? 1 : 0;
    info->reserved = 0;
    return CLEORA_OK;
}

int cleora_colsharded_block(const cleora_colsharded *s, uint32_t k, cleora_graph **graph, uint64_t *row_begin, uint64_t *row_end) {
    CL_REQUIRE(s != nullptr && k < s->steps, "handle is NULL / no such block");
    if (graph) *graph = s->blocks[k].g;
    if (row_begin) *row_begin = s->blocks[k].r0;
    if (row_end) *row_end = s->blocks[k].r1;
    return CLEORA_OK;
}

int cleora_colsharded_propagate_dev(cleora_colsharded *s, int markov_type, const float *x_local, float *x_next_local, uint32_t flags,
                                    float residual_weight, double *row_sqdiff, void *stream) {
    CL_REQUIRE(s != nullptr, "handle is NULL");
    CL_REQUIRE(x_local != nullptr && x_next_local != nullptr && x_local != x_next_local, "x / x_next is NULL or they alias");
    CL_REQUIRE(markov_type == CLEORA_LEFT || (markov_type == CLEORA_SYMMETRIC && s->has_sym), "no values for this markov_type");
    CL_REQUIRE(!(flags & ~(CLEORA_F_L2NORM | CLEORA_F_RESIDUAL | CLEORA_F_BLEND_ANY | CLEORA_F_SQDIFF | CLEORA_F_SQDIFF64 | CLEORA_F_HUB_SEGMENTS)),
               "the column partition takes L2NORM / RESIDUAL / BLEND_ANY / SQDIFF / SQDIFF64 / HUB_SEGMENTS (the L1 norm and the whitened loop need whole rows: row partition)");
    CL_REQUIRE(!(flags & CLEORA_F_SQDIFF) || row_sqdiff != nullptr, "row_sqdiff is NULL");
    std::lock_guard<std::mutex> lock(s->mu);
    CL_HIP(hipSetDevice(s->device));
    return propagate_cols(s, markov_type, x_local, x_next_local, flags, residual_weight, row_sqdiff, S(stream));
}

// embed_full / embed_full_with_convergence (src/embedding.rs:106-188) on the rank's column slice: x_local (n x d/P, ld = d/P) holds its
// columns of E_0 and receives its columns of the result.  The iterate is bit-equal to the one-GPU loop's; the convergence test sums the
// ranks' parts of the squared difference (f64), so a threshold the RMSE meets to the last bit may stop one iteration apart.
int cleora_embed_colsharded(cleora_colsharded *s, float *x_local, int markov_type, uint64_t max_iterations, float residual_weight,
                            float convergence_threshold, uint32_t flags, uint64_t *iterations_run) {
    CL_REQUIRE(s != nullptr && x_local != nullptr, "handle / x is NULL");
    CL_REQUIRE(markov_type == CLEORA_LEFT || (markov_type == CLEORA_SYMMETRIC && s->has_sym), "no values for this markov_type");
    CL_REQUIRE(!(flags & ~(CLEORA_F_FASTNORM | CLEORA_F_HUB_SEGMENTS)), "the column partition runs the plain loop (flags: HUB_SEGMENTS)");
    std::lock_guard<std::mutex> lock(s->mu);
    CL_HIP(hipSetDevice(s->device));
    if (iterations_run) *iterations_run = 0;
    if (max_iterations == 0 || s->n == 0) return CLEORA_OK;
    const uint64_t bytes = s->n * (uint64_t)s->d_local * sizeof(float);
    const bool check = convergence_threshold > 0.0f;
    DevMem other, sq, rws, total;
    int rc;
    if ((rc = other.alloc(bytes)) != CLEORA_OK) return rc;
    if (check && ((rc = sq.alloc(s->n * 8)) != CLEORA_OK || (rc = rws.alloc(reduce_workspace(s->n) * 8)) != CLEORA_OK || (rc = total.alloc(8)) != CLEORA_OK)) return rc;
    const uint32_t base = CLEORA_F_L2NORM | CLEORA_F_RESIDUAL | (flags & CLEORA_F_HUB_SEGMENTS);
    float *src = x_local, *dst = other.as<float>();
    uint64_t ran = max_iterations;
    for (uint64_t it = 0; it < max_iterations; ++it) {
        const bool test = check && it > 0;                          // embedding.rs:169
        if ((rc = propagate_cols(s, markov_type, src, dst, base | (test ? CLEORA_F_SQDIFF : 0u), residual_weight, test ? sq.as<double>() : nullptr, nullptr)) != CLEORA_OK)
            return rc;
        std::swap(src, dst);
        if (test) {
            if ((rc = launch_reduce_sum(sq.as<double>(), s->n, rws.as<double>(), total.as<double>(), nullptr)) != CLEORA_OK) return rc;
            if (s->world > 1 && (rc = cleora_allreduce_f64_dev(s->comm, total.as<double>(), 1, nullptr)) != CLEORA_OK) return rc;
            double sum = 0.0;
            CL_HIP(hipMemcpy(&sum, total.p, 8, hipMemcpyDeviceToHost));
            const float rmse = sqrtf((float)(sum / (double)(s->n * (uint64_t)s->d_total)));      // embedding.rs:177-178
            if (rmse < convergence_threshold) { ran = it + 1; break; }
        }
    }
    CL_HIP(hipDeviceSynchronize());
    if (s->comm) { if ((rc = cleora_comm_check(s->comm)) != CLEORA_OK) return rc; }
    if (src != x_local) CL_HIP(hipMemcpy(x_local, src, bytes, hipMemcpyDeviceToDevice));
    if (iterations_run) *iterations_run = ran;
    return CLEORA_OK;
}

}  // extern "C"
