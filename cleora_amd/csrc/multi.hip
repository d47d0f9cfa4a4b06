// multi.hip — ONE process, P devices: the row partition of sharded.hip behind a single handle, for hosts whose call is
// `embed(graph, 256, 40)` in one process (pycleora/__init__.py:51-127; SparseMatrix::embed_fast, src/lib.rs:320-364) and not a
// torchrun-style job.  BASELINE.json:north_star asks for both at once — "the pycleora SparseMatrix / embed() / left_markov_propagate
// surface is a drop-in" and "the graph is row-partitioned across the 8 GPUs of one node" — and SURVEY.md 8(b) B2 wrote the graph
// handle with (n_devices, device_ids*).
//
// Design: nothing new on the data path.  The handle keeps one host THREAD per device for the duration of a call; thread p is rank p
// of a local communicator (peer.hip: stores through peer mappings; between threads of one process the "mapping" is the peer's own
// pointer, with hipDeviceEnablePeerAccess across devices) and owns a cleora_sharded handle (its blocks of the CSR) and a stream.
// The loops are cleora_embed_sharded / cleora_sharded_propagate_dev — the very calls a one-process-per-GPU host makes — so results
// are the partition's: bit-equal to the one-GPU loops for the plain loop, within the whitened loop's stated tolerance otherwise.
// Host <-> device: every rank uploads E_0 / X to its own replica over its own PCIe link (stager.hip, one pipeline per device) and
// downloads ITS slice of the result into the caller's array — P links in parallel.
// device_ids may repeat: P logical shards on one GPU — how the one-GPU test box exercises this path (SURVEY.md 8e "LoopbackComm").
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <new>
#include <thread>

#include "comm_internal.h"

struct cleora_multi {
    uint32_t world = 0, steps = 1;
    std::vector<int> devices;
    uint64_t n = 0, nnz = 0, n_pad = 0;
    bool has_sym = false;
    std::vector<cleora_comm *> comms;
    std::vector<cleora_sharded *> shards;
    std::vector<hipStream_t> streams;
    std::vector<uint64_t> bounds;          // world * steps + 1 row boundaries (padded row space)
    // per shard: device staging of the host-pointer calls, kept between calls and grow-only like cleora_propagate's (what the
    // reference's unmodified embed() loop calls 40 times; freeing 10 GB costs ~0.3 s on this driver)
    std::vector<void *> io[2];
    std::vector<uint64_t> io_bytes[2];
    std::mutex mu;                         // one call at a time
    // the threads of a call meet here before they enter a collective: a rank that failed on its own (allocation, upload) must not
    // leave the others waiting inside one
    std::mutex gate_mu;
    std::condition_variable gate_cv;
    uint32_t gate_count = 0, gate_gen = 0, gate_absent = 0;      // gate_absent: ranks that left the running call before its gates
    bool gate_fail = false, gate_result = true;
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

struct Borrowed {      // a staging buffer that stays with the handle
    void *p = nullptr;
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// shard p's staging buffer k of at least `bytes` (thread p only)
int staging(cleora_multi *m, uint32_t p, int k, uint64_t bytes, void **out) {
    if (m->io_bytes[k][p] < bytes || !m->io[k][p]) {
        if (m->io[k][p]) CL_HIP(hipFree(m->io[k][p]));
        m->io[k][p] = nullptr;
        m->io_bytes[k][p] = 0;
        CL_HIP(hipMalloc(&m->io[k][p], bytes ? bytes : 1));
        m->io_bytes[k][p] = bytes;
    }
    *out = m->io[k][p];
    return CLEORA_OK;
}

// true when every rank of the call arrived with ok = true
bool gate(cleora_multi *m, bool ok) {
    std::unique_lock<std::mutex> lock(m->gate_mu);
    if (!ok) m->gate_fail = true;
    const uint32_t gen = m->gate_gen;
    if (++m->gate_count + m->gate_absent >= m->world) {       // (ranks that left the call early count as arrived, with a failure)
        m->gate_result = !m->gate_fail && m->gate_absent == 0;
        m->gate_count = 0;
        m->gate_fail = false;
        ++m->gate_gen;
        m->gate_cv.notify_all();
        return m->gate_result;
    }
    m->gate_cv.wait(lock, [&] { return m->gate_gen != gen; });
    return m->gate_result;
}

// fn(rank) on one thread per device; the first failure (lowest rank) becomes the caller's error
int run_all(cleora_multi *m, const std::function<int(uint32_t)> &fn) {
    const uint32_t P = m->world;
    std::vector<int> rc(P, CLEORA_OK);
    std::vector<std::string> err(P);
    auto body = [&](uint32_t p) {
        if (hipSetDevice(m->devices[p]) != hipSuccess) {
            (void)hipGetLastError();
            rc[p] = CLEORA_E_NODEVICE;
            err[p] = "hipSetDevice failed";
            // this rank will meet none of the call's gates: the others must not wait for it (they hung in
            // gate_cv.wait holding the handle's mutex) — it counts as arrived-and-failed at every gate until the call ends
            std::lock_guard<std::mutex> lock(m->gate_mu);
            ++m->gate_absent;
            m->gate_fail = true;
            if (m->gate_count > 0 && m->gate_count + m->gate_absent >= m->world) {
                m->gate_result = false;
                m->gate_count = 0;
                m->gate_fail = false;
                ++m->gate_gen;
                m->gate_cv.notify_all();
            }
            return;
        }
        rc[p] = fn(p);
        if (rc[p] != CLEORA_OK) err[p] = cleora_last_error();
    };
    if (P == 1) {
        body(0);
    } else {
        std::vector<std::thread> threads;
        threads.reserve(P);
        for (uint32_t p = 0; p < P; ++p) threads.emplace_back(body, p);
        for (auto &t : threads) t.join();
    }
    m->gate_absent = 0;
    for (uint32_t p = 0; p < P; ++p)
        if (rc[p] != CLEORA_OK) {
            set_error("device " + std::to_string(m->devices[p]) + " (shard " + std::to_string(p) + " of " + std::to_string(P) + "): " + err[p]);
            return rc[p];
        }
    return CLEORA_OK;
}

void free_multi(cleora_multi *m) {
    if (!m) return;
    // (the communicators' teardown is collective: every rank on its own thread — unless the group never became complete)
    bool complete = true;
    for (cleora_comm *c : m->comms) complete = complete && (c != nullptr || m->world == 1);
    if (!complete)
        for (cleora_comm *c : m->comms) peer_abandon(c);
    (void)run_all(m, [&](uint32_t p) {
        if (p < m->shards.size() && m->shards[p]) (void)cleora_sharded_destroy(m->shards[p]);
        if (p < m->streams.size() && m->streams[p]) { (void)hipStreamSynchronize(m->streams[p]); (void)hipStreamDestroy(m->streams[p]); }
        for (int k = 0; k < 2; ++k)
            if (p < m->io[k].size() && m->io[k][p]) (void)hipFree(m->io[k][p]);
        if (p < m->comms.size() && m->comms[p]) (void)cleora_comm_destroy(m->comms[p]);
        return CLEORA_OK;
    });
    delete m;
}

}  // namespace
}  // namespace cleora

using namespace cleora;

extern "C" {

int cleora_multi_create(const int *device_ids, uint32_t n_devices, uint64_t n, uint64_t nnz, const uint64_t *rowptr, const uint32_t *col,
                        const float *val_left, const float *val_sym, uint32_t steps, int balance, cleora_multi **out) {
    CL_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    CL_REQUIRE(device_ids != nullptr && n_devices >= 1 && n_devices <= 64, "need 1..64 device ids");
    CL_REQUIRE(rowptr != nullptr, "rowptr is NULL");
    CL_REQUIRE(nnz == 0 || (col != nullptr && val_left != nullptr), "col / val_left is NULL");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device available; libcleora_hip has no CPU fallback");
        return CLEORA_E_NODEVICE;
    }
    for (uint32_t p = 0; p < n_devices; ++p) CL_REQUIRE(device_ids[p] >= 0 && device_ids[p] < count, "device id out of range");
    for (uint32_t p = 0; p < n_devices; ++p)
        for (uint32_t q = 0; q < n_devices; ++q) {
            if (device_ids[p] == device_ids[q]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, device_ids[p], device_ids[q]) != hipSuccess || !can) {
                (void)hipGetLastError();
                set_error("device " + std::to_string(device_ids[p]) + " cannot access device " + std::to_string(device_ids[q]) + " (peer access is what the all-gather stores through)");
                return CLEORA_E_INVALID;
            }
        }
    cleora_multi *m = new (std::nothrow) cleora_multi();
    if (!m) { set_error("host allocation failed"); return CLEORA_E_OOM; }
    m->world = n_devices;
    m->steps = steps ? steps : (n_devices == 1 ? 1u : 4u);       // block k's gather overlaps block k + 1's SpMM
    m->devices.assign(device_ids, device_ids + n_devices);
    m->n = n;
    m->nnz = nnz;
    m->has_sym = val_sym != nullptr;
    m->comms.assign(n_devices, nullptr);
    m->shards.assign(n_devices, nullptr);
    m->streams.assign(n_devices, nullptr);
    for (int k = 0; k < 2; ++k) { m->io[k].assign(n_devices, nullptr); m->io_bytes[k].assign(n_devices, 0); }
    unsigned char id[CLEORA_COMM_ID_BYTES];
    int rc = cleora_comm_local_id(id);
    if (rc == CLEORA_OK)
        rc = run_all(m, [&](uint32_t p) {
            int r = CLEORA_OK;
            if (m->world > 1) r = cleora_comm_create_local(id, (int)p, (int)m->world, m->devices[p], &m->comms[p]);    // collective
            bool ok = gate(m, r == CLEORA_OK);
            if (!ok) return r != CLEORA_OK ? r : CLEORA_E_RCCL;
            r = cleora_sharded_create(m->comms[p], m->devices[p], n, nnz, rowptr, col, val_left, val_sym, 0, m->steps, balance, &m->shards[p]);
            if (r == CLEORA_OK && hipStreamCreateWithFlags(&m->streams[p], hipStreamNonBlocking) != hipSuccess) {
                (void)hipGetLastError();
                set_error("hipStreamCreate failed");
                r = CLEORA_E_HIP;
            }
            if (r == CLEORA_OK) r = cleora_sharded_set_stream(m->shards[p], m->streams[p]);
            ok = gate(m, r == CLEORA_OK);
            if (!ok && r == CLEORA_OK) { set_error("another shard failed"); return CLEORA_E_RCCL; }
            return r;
        });
    if (rc == CLEORA_OK) {
        cleora_sharded_info info;
        rc = cleora_sharded_get_info(m->shards[0], &info);
        if (rc == CLEORA_OK) {
            m->n_pad = info.n_pad;
            m->bounds.resize((size_t)m->world * m->steps + 1);
            rc = cleora_sharded_bounds(m->shards[0], m->bounds.data());
        }
    }
    if (rc != CLEORA_OK) {
        const std::string keep = cleora_last_error();
        free_multi(m);
        set_error(keep);
        return rc;
    }
    *out = m;
    return CLEORA_OK;
}

int cleora_multi_destroy(cleora_multi *m) {
    free_multi(m);
    return CLEORA_OK;
}

int cleora_multi_get_info(const cleora_multi *m, cleora_multi_info *info) {
    CL_REQUIRE(m != nullptr && info != nullptr, "handle / info is NULL");
    std::memset(info, 0, sizeof(*info));
    info->n = m->n;
    info->nnz = m->nnz;
    info->n_pad = m->n_pad;
    info->world = m->world;
    info->steps = m->steps;
    info->has_symmetric = m->has_sym ? 1 : 0;
    for (uint32_t p = 0; p < m->world && p < 64; ++p) {
        cleora_sharded_info si;
        const int rc = cleora_sharded_get_info(m->shards[p], &si);
        if (rc != CLEORA_OK) return rc;
        info->device[p] = m->devices[p];
        info->local_rows[p] = si.local_rows;
        info->local_nnz[p] = si.local_nnz;
        info->device_bytes[p] = si.device_bytes;
    }
    return CLEORA_OK;
}

// SparseMatrix::embed_fast / embed_fast_convergence (src/lib.rs:320-412) and, with CLEORA_F_WHITEN, the default loop of
// pycleora.embed() (pycleora/__init__.py:97-127) — cleora_embed's contract over the partition.
int cleora_multi_embed(cleora_multi *m, const uint64_t *entity_hash_host, const float *x0_host, int markov_type, uint32_t d,
                       uint64_t max_iterations, int64_t seed, float residual_weight, float convergence_threshold, uint32_t flags,
                       float *out_host, uint64_t *iterations_run) {
    CL_REQUIRE(m != nullptr, "handle is NULL");
    CL_REQUIRE(out_host != nullptr || m->n == 0, "out_host is NULL");
    CL_REQUIRE(entity_hash_host != nullptr || x0_host != nullptr || m->n == 0, "need entity hashes or initial embeddings");
    CL_REQUIRE(d > 0, "d must be positive");
    CL_REQUIRE(markov_type == CLEORA_LEFT || (markov_type == CLEORA_SYMMETRIC && m->has_sym), "no values for this markov_type");
    if (iterations_run) *iterations_run = 0;
    if (m->n == 0) return CLEORA_OK;
    std::lock_guard<std::mutex> lock(m->mu);
    const uint64_t n = m->n, row_bytes = (uint64_t)d * 4;
    std::vector<uint64_t> ran(m->world, 0);
    const int rc = run_all(m, [&](uint32_t p) {
        DevMem hashes;
        Borrowed x;
        hipStream_t st = m->streams[p];
        int r = staging(m, p, 0, m->n_pad * row_bytes, &x.p);
        if (r == CLEORA_OK && m->n_pad > n && hipMemsetAsync(x.as<char>() + n * row_bytes, 0, (m->n_pad - n) * row_bytes, st) != hipSuccess) {
            (void)hipGetLastError();
            set_error("hipMemsetAsync failed");
            r = CLEORA_E_HIP;
        }
        if (r == CLEORA_OK) {
            if (x0_host) {
                r = staged_h2d(x.p, x0_host, n * row_bytes, st);
            } else {
                r = hashes.alloc(n * 8);
                if (r == CLEORA_OK) r = staged_h2d(hashes.p, entity_hash_host, n * 8, st);
                if (r == CLEORA_OK) r = launch_init(hashes.as<uint64_t>(), n, d, seed, x.as<float>(), d, st);
            }
        }
        if (r == CLEORA_OK && hipStreamSynchronize(st) != hipSuccess) { (void)hipGetLastError(); set_error("hipStreamSynchronize failed"); r = CLEORA_E_HIP; }
        if (!gate(m, r == CLEORA_OK)) {
            if (r == CLEORA_OK) { set_error("another shard failed before the loop"); r = CLEORA_E_RCCL; }
            return r;
        }
        r = cleora_embed_sharded(m->shards[p], x.as<float>(), markov_type, d, max_iterations, residual_weight, convergence_threshold, flags, &ran[p]);
        if (r != CLEORA_OK) return r;
        // the replicas are identical: every rank brings ITS slice of the rows home over its own link
        const uint64_t r0 = n * p / m->world, r1 = n * (p + 1) / m->world;
        return staged_d2h(out_host + r0 * d, x.as<float>() + r0 * d, (r1 - r0) * row_bytes, st);
    });
    if (rc == CLEORA_OK && iterations_run) *iterations_run = ran[0];
    return rc;
}

// SparseMatrix::markov_propagate (src/lib.rs:29-47): y = A x, host in / host out, every rank its own rows.
int cleora_multi_propagate(cleora_multi *m, int markov_type, const float *x_host, uint32_t d, float *y_host) {
    CL_REQUIRE(m != nullptr, "handle is NULL");
    CL_REQUIRE((x_host != nullptr && y_host != nullptr) || m->n == 0, "x_host / y_host is NULL");
    CL_REQUIRE(d > 0, "d must be positive");
    CL_REQUIRE(markov_type == CLEORA_LEFT || (markov_type == CLEORA_SYMMETRIC && m->has_sym), "no values for this markov_type");
    if (m->n == 0) return CLEORA_OK;
    std::lock_guard<std::mutex> lock(m->mu);
    const uint64_t n = m->n, row_bytes = (uint64_t)d * 4;
    return run_all(m, [&](uint32_t p) {
        Borrowed x, y;
        hipStream_t st = m->streams[p];
        int r = staging(m, p, 0, m->n_pad * row_bytes, &x.p);
        if (r == CLEORA_OK) r = staging(m, p, 1, m->n_pad * row_bytes, &y.p);
        if (r == CLEORA_OK && m->n_pad > n && hipMemsetAsync(x.as<char>() + n * row_bytes, 0, (m->n_pad - n) * row_bytes, st) != hipSuccess) {
            (void)hipGetLastError();
            set_error("hipMemsetAsync failed");
            r = CLEORA_E_HIP;
        }
        if (r == CLEORA_OK) r = staged_h2d(x.p, x_host, n * row_bytes, st);
        if (r == CLEORA_OK) r = cleora_sharded_propagate_dev(m->shards[p], markov_type, x.as<float>(), y.as<float>(), d, 0, 0.0f, nullptr, 0, st);
        for (uint32_t k = 0; k < m->steps && r == CLEORA_OK; ++k) {
            const uint64_t b0 = std::min(m->bounds[(size_t)k * m->world + p], n), b1 = std::min(m->bounds[(size_t)k * m->world + p + 1], n);
            if (b1 > b0) r = staged_d2h(y_host + b0 * d, y.as<float>() + b0 * d, (b1 - b0) * row_bytes, st);
        }
        return r;
    });
}

}  // extern "C"
