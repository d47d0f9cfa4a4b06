// comm.hip — the multi-GPU exchange steps of the propagation loop under the C ABI: one communicator per
// process (= per GPU), RCCL over xGMI, bound with dlopen on first use like rocSOLVER in eigh.hip (hosts
// that run on one GPU never load it; CLEORA_RCCL=<path> overrides the library name).
//
// The reference has no distributed code (pycleora is single-process: rayon over rows,
// src/embedding.rs:59-63); BASELINE.json:north_star asks for "the graph row-partitioned across the 8 GPUs
// of one node with an RCCL all-gather of the embedding matrix over xGMI between iterations", reachable from
// a non-Python host "through a thin extern-C FFI".  These entry points are that FFI; the partition logic that
// drives them is csrc/sharded.hip (cleora_sharded_* / cleora_embed_sharded; INTEGRATION.md has the Rust declarations).
//
// Every collective is IN PLACE on device memory and only ENQUEUES on the caller's stream (NCCL semantics):
// order it against the SpMM with cleora_stream_wait_stream.
//
// Second transport (round 4): PEER-DIRECT exchange over hipIpc mappings, for the ranks of ONE node (north_star's layout).
// Every rank maps the peers' registered buffers (cleora_comm_register) and a small uncached mailbox; the all-gather is then a
// copy kernel that stores this rank's finished rows straight into every peer's replica ("peer-mapped direct
// stores"), followed by a release store of a sequence number into each peer's mailbox; a consumer waits with a one-wave kernel
// that polls its own mailbox (bounded: a peer that never arrives raises an error instead of hanging the GPU).  No staging
// buffers, no ring: every shard crosses each xGMI link once, all links at the same time.  The node-local bootstrap (handles,
// host barrier) is a POSIX shared-memory segment named after the communicator's id.
//   * on an RCCL communicator it is an additional all-gather algorithm (CLEORA_ALLGATHER_PEER) beside the ring / send-recv ones;
//   * a LOCAL communicator (cleora_comm_create_local) has no RCCL underneath at all: all-gather, all-reduce (every rank sums the
//     peers' contributions in rank order: bit-identical everywhere) and broadcast over the same mappings.  RCCL refuses two ranks
//     on one device, hipIpc does not — so the multi-rank loops run through the C ABI on a one-GPU box too (tests).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.h"

#include "comm_internal.h"

namespace cleora {
namespace {

struct Rccl {
    void *lib = nullptr;
    std::string error;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclBroadcast) broadcast = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *env = std::getenv("CLEORA_RCCL");
        const char *names[] = {env, "librccl.so.1", "librccl.so"};
        for (const char *name : names) {
            if (!name || !*name) continue;
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
            r.error = dlerror();
        }
        if (!r.lib) return;
        bool ok = true;
        auto sym = [&](auto &fn, const char *name) {
            fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(r.lib, name));
            if (!fn) { ok = false; r.error = std::string("symbol not found: ") + name; }
        };
        sym(r.get_unique_id, "ncclGetUniqueId");
        sym(r.comm_init_rank, "ncclCommInitRank");
        sym(r.comm_destroy, "ncclCommDestroy");
        sym(r.error_string, "ncclGetErrorString");
        sym(r.all_gather, "ncclAllGather");
        sym(r.all_reduce, "ncclAllReduce");
        sym(r.broadcast, "ncclBroadcast");
        sym(r.send, "ncclSend");
        sym(r.recv, "ncclRecv");
        sym(r.group_start, "ncclGroupStart");
        sym(r.group_end, "ncclGroupEnd");
        if (!ok) r.lib = nullptr;
    });
    return r;
}

int need_rccl(Rccl **out) {
    Rccl &r = rccl();
    if (!r.lib) {
        set_error("multi-GPU entry points need RCCL (dlopen failed: " + r.error + "); set CLEORA_RCCL to its path");
        return CLEORA_E_RCCL;
    }
    *out = &r;
    return CLEORA_OK;
}

int rccl_fail(Rccl &r, ncclResult_t e, const char *what) {
    set_error(std::string(what) + " failed: " + (r.error_string ? r.error_string(e) : "?"));
    return CLEORA_E_RCCL;
}

#define CL_RCCL(r, call)                                              \
    do {                                                              \
        ncclResult_t _e = (call);                                     \
        if (_e != ncclSuccess) return rccl_fail((r), _e, #call);      \
    } while (0)

inline hipStream_t S(void *s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace
}  // namespace cleora

using namespace cleora;

extern "C" {

int cleora_stream_create(void **stream) {
    CL_REQUIRE(stream != nullptr, "stream is NULL");
    hipStream_t s = nullptr;
    CL_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return CLEORA_OK;
}

int cleora_stream_destroy(void *stream) {
    if (stream) CL_HIP(hipStreamDestroy(S(stream)));
    return CLEORA_OK;
}

int cleora_stream_wait_stream(void *waiter, void *signaller) {
    if (waiter == signaller) return CLEORA_OK;
    hipEvent_t e = nullptr;
    CL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipError_t err = hipEventRecord(e, S(signaller));
    if (err == hipSuccess) err = hipStreamWaitEvent(S(waiter), e, 0);
    (void)hipEventDestroy(e);   // released once the recorded work completes
    CL_HIP(err);
    return CLEORA_OK;
}

int cleora_comm_unique_id(void *id_out) {
    CL_REQUIRE(id_out != nullptr, "id_out is NULL");
    Rccl *r;
    int rc = need_rccl(&r);
    if (rc != CLEORA_OK) return rc;
    static_assert(sizeof(ncclUniqueId) == CLEORA_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    CL_RCCL(*r, r->get_unique_id(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return CLEORA_OK;
}

int cleora_comm_create(const void *id, int rank, int world, int device, cleora_comm **out) {
    CL_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    CL_REQUIRE(id != nullptr, "id is NULL");
    CL_REQUIRE(world >= 1 && rank >= 0 && rank < world, "need 0 <= rank < world");
    Rccl *r;
    int rc = need_rccl(&r);
    if (rc != CLEORA_OK) return rc;
    CL_HIP(hipSetDevice(device));
    cleora_comm *c = new (std::nothrow) cleora_comm();
    if (!c) {
        set_error("host allocation failed");
        return CLEORA_E_OOM;
    }
    c->rank = rank;
    c->world = world;
    c->device = device;
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    std::memcpy(c->id, id, sizeof(c->id));
    const ncclResult_t e = r->comm_init_rank(&c->comm, world, uid, rank);
    if (e != ncclSuccess) {
        delete c;
        return rccl_fail(*r, e, "ncclCommInitRank");
    }
    *out = c;
    return CLEORA_OK;
}

int cleora_comm_local_id(void *id_out) {
    CL_REQUIRE(id_out != nullptr, "id_out is NULL");
    unsigned char *id = static_cast<unsigned char *>(id_out);
    std::memset(id, 0, CLEORA_COMM_ID_BYTES);
    FILE *f = std::fopen("/dev/urandom", "rb");
    size_t got = f ? std::fread(id, 1, 32, f) : 0;
    if (f) std::fclose(f);
    const uint64_t salt[2] = {(uint64_t)getpid(), (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count()};
    std::memcpy(id + 32, salt, sizeof salt);
    (void)got;
    std::memcpy(id + 64, "cleora-local", 12);
    return CLEORA_OK;
}

int cleora_comm_create_local(const void *id, int rank, int world, int device, cleora_comm **out) {
    CL_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    CL_REQUIRE(id != nullptr, "id is NULL");
    CL_REQUIRE(world >= 1 && rank >= 0 && rank < world, "need 0 <= rank < world");
    CL_HIP(hipSetDevice(device));
    cleora_comm *c = new (std::nothrow) cleora_comm();
    if (!c) {
        set_error("host allocation failed");
        return CLEORA_E_OOM;
    }
    c->rank = rank;
    c->world = world;
    c->device = device;
    c->allgather_algo = CLEORA_ALLGATHER_PEER;
    std::memcpy(c->id, id, sizeof(c->id));
    const int rc = peer_enable(c);
    if (rc != CLEORA_OK) {
        delete c;
        return rc;
    }
    *out = c;
    return CLEORA_OK;
}

int cleora_comm_enable_peer(cleora_comm *c) {
    CL_REQUIRE(c != nullptr, "comm is NULL");
    std::lock_guard<std::mutex> lock(c->mu);
    return peer_enable(c);
}

int cleora_comm_register(cleora_comm *c, void *buf_dev, uint64_t bytes) {
    CL_REQUIRE(c != nullptr, "comm is NULL");
    std::lock_guard<std::mutex> lock(c->mu);
    if (!c->peer) return CLEORA_OK;                      // RCCL needs no registration
    return peer_register(c, buf_dev, bytes);
}

int cleora_comm_unregister(cleora_comm *c, void *buf_dev) {
    CL_REQUIRE(c != nullptr, "comm is NULL");
    std::lock_guard<std::mutex> lock(c->mu);
    return peer_unregister(c, buf_dev);
}

int cleora_comm_selftest(cleora_comm *c, uint32_t flags) {
    CL_REQUIRE(c != nullptr, "comm is NULL");
    CL_REQUIRE((flags & ~3u) == 0, "unknown self-test flags");
    std::lock_guard<std::mutex> lock(c->mu);
    if (!c->peer) return CLEORA_OK;                       // RCCL's own collectives: nothing of ours to test
    return peer_selftest(c, flags);
}

int cleora_comm_peer_mode(const cleora_comm *c, int *mode) {
    CL_REQUIRE(c != nullptr && mode != nullptr, "comm / mode is NULL");
    *mode = peer_mode(c);
    return CLEORA_OK;
}

int cleora_comm_check(cleora_comm *c) {
    CL_REQUIRE(c != nullptr, "comm is NULL");
    return peer_check(c);
}

int cleora_comm_destroy(cleora_comm *c) {
    if (!c) return CLEORA_OK;
    peer_destroy(c);
    Rccl &r = rccl();
    if (r.lib && c->comm) {
        (void)hipSetDevice(c->device);
        (void)r.comm_destroy(c->comm);
    }
    delete c;
    return CLEORA_OK;
}

int cleora_comm_info(const cleora_comm *c, int *rank, int *world, int *device) {
    CL_REQUIRE(c != nullptr, "comm is NULL");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (device) *device = c->device;
    return CLEORA_OK;
}

int cleora_comm_set_allgather(cleora_comm *c, int algo) {
    CL_REQUIRE(c != nullptr, "comm is NULL");
    CL_REQUIRE(algo == CLEORA_ALLGATHER_RING || algo == CLEORA_ALLGATHER_P2P || algo == CLEORA_ALLGATHER_PEER, "unknown all-gather algorithm");
    CL_REQUIRE(algo == CLEORA_ALLGATHER_PEER || c->comm != nullptr, "a local communicator has the peer-direct all-gather only");
    CL_REQUIRE(algo != CLEORA_ALLGATHER_PEER || c->peer != nullptr, "enable the peer transport first (cleora_comm_enable_peer)");
    std::lock_guard<std::mutex> lock(c->mu);
    c->allgather_algo = algo;
    return CLEORA_OK;
}

int cleora_comm_get_allgather(const cleora_comm *c, int *algo) {
    CL_REQUIRE(c != nullptr && algo != nullptr, "comm / algo is NULL");
    *algo = c->allgather_algo;
    return CLEORA_OK;
}

int cleora_allgatherv_f32_dev(cleora_comm *c, float *buf, const uint64_t *offsets, void *stream) {
    CL_REQUIRE(c != nullptr, "comm is NULL");
    CL_REQUIRE(buf != nullptr && offsets != nullptr, "buf / offsets is NULL");
    const int P = c->world, me = c->rank;
    bool equal = true;
    for (int r = 0; r < P; ++r) {
        CL_REQUIRE(offsets[r] <= offsets[r + 1], "offsets must be non-decreasing");
        if (offsets[r + 1] - offsets[r] != offsets[1] - offsets[0]) equal = false;
    }
    if (c->allgather_algo == CLEORA_ALLGATHER_PEER || !c->comm) {
        std::lock_guard<std::mutex> lock(c->mu);
        return peer_allgatherv_f32(c, buf, offsets, S(stream));
    }
    // (a world of one still goes through RCCL: a single-GPU box then exercises the same calls)
    Rccl *r;
    int rc = need_rccl(&r);
    if (rc != CLEORA_OK) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    CL_HIP(hipSetDevice(c->device));
    if (c->allgather_algo == CLEORA_ALLGATHER_P2P) {
        // direct exchange over the fully connected xGMI mesh: every rank sends its shard to each peer and
        // receives each peer's shard, one message per link and direction
        // (a failed call inside a group must still close the group, or every later collective on this communicator hangs)
        ncclResult_t first = ncclSuccess;
        auto note = [&first](ncclResult_t e) { if (first == ncclSuccess) first = e; };
        CL_RCCL(*r, r->group_start());
        for (int k = 1; k < P; ++k) {
            const int to = (me + k) % P, from = (me - k + P) % P;
            const uint64_t mine = offsets[me + 1] - offsets[me], theirs = offsets[from + 1] - offsets[from];
            if (mine) note(r->send(buf + offsets[me], mine, ncclFloat, to, c->comm, S(stream)));
            if (theirs) note(r->recv(buf + offsets[from], theirs, ncclFloat, from, c->comm, S(stream)));
        }
        note(r->group_end());
        if (first != ncclSuccess) return rccl_fail(*r, first, "all-gather (send/recv mesh)");
        return CLEORA_OK;
    }
    if (equal) {
        const uint64_t count = offsets[1] - offsets[0];
        if (count)
            CL_RCCL(*r, r->all_gather(buf + offsets[me], buf + offsets[0], count, ncclFloat, c->comm, S(stream)));
        return CLEORA_OK;
    }
    ncclResult_t first = ncclSuccess;
    auto note = [&first](ncclResult_t e) { if (first == ncclSuccess) first = e; };
    CL_RCCL(*r, r->group_start());
    for (int root = 0; root < P; ++root) {
        const uint64_t count = offsets[root + 1] - offsets[root];
        if (count) note(r->broadcast(buf + offsets[root], buf + offsets[root], count, ncclFloat, root, c->comm, S(stream)));
    }
    note(r->group_end());
    if (first != ncclSuccess) return rccl_fail(*r, first, "all-gather-v (grouped broadcasts)");
    return CLEORA_OK;
}

int cleora_allgather_f32_dev(cleora_comm *c, float *buf, uint64_t elems_per_rank, void *stream) {
    CL_REQUIRE(c != nullptr, "comm is NULL");
    CL_REQUIRE(c->world <= 4096, "world too large");
    std::vector<uint64_t> off((size_t)c->world + 1);
    for (int r = 0; r <= c->world; ++r) off[r] = (uint64_t)r * elems_per_rank;
    return cleora_allgatherv_f32_dev(c, buf, off.data(), stream);
}

static int allreduce(cleora_comm *c, void *buf, uint64_t n, ncclDataType_t t, void *stream) {
    CL_REQUIRE(c != nullptr, "comm is NULL");
    CL_REQUIRE(buf != nullptr || n == 0, "buf is NULL");
    if (n == 0) return CLEORA_OK;
    if (!c->comm) {
        std::lock_guard<std::mutex> lock(c->mu);
        return peer_allreduce(c, buf, n, t == ncclDouble, S(stream));
    }
    Rccl *r;
    int rc = need_rccl(&r);
    if (rc != CLEORA_OK) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    CL_HIP(hipSetDevice(c->device));
    CL_RCCL(*r, r->all_reduce(buf, buf, n, t, ncclSum, c->comm, S(stream)));
    return CLEORA_OK;
}

int cleora_allreduce_f32_dev(cleora_comm *c, float *buf, uint64_t n, void *stream) {
    return allreduce(c, buf, n, ncclFloat, stream);
}

int cleora_allreduce_f64_dev(cleora_comm *c, double *buf, uint64_t n, void *stream) {
    return allreduce(c, buf, n, ncclDouble, stream);
}

int cleora_broadcast_dev(cleora_comm *c, void *buf, uint64_t bytes, int root, void *stream) {
    CL_REQUIRE(c != nullptr, "comm is NULL");
    CL_REQUIRE(root >= 0 && root < c->world, "bad root");
    CL_REQUIRE(buf != nullptr || bytes == 0, "buf is NULL");
    if (bytes == 0) return CLEORA_OK;
    if (!c->comm) {
        std::lock_guard<std::mutex> lock(c->mu);
        return peer_broadcast(c, buf, bytes, root, S(stream));
    }
    Rccl *r;
    int rc = need_rccl(&r);
    if (rc != CLEORA_OK) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    CL_HIP(hipSetDevice(c->device));
    CL_RCCL(*r, r->broadcast(buf, buf, bytes, ncclUint8, root, c->comm, S(stream)));
    return CLEORA_OK;
}

int cleora_alltoall_f32_dev(cleora_comm *c, const float *send, float *recv, uint64_t elems_per_rank, void *stream) {
    CL_REQUIRE(c != nullptr, "comm is NULL");
    CL_REQUIRE((send != nullptr && recv != nullptr) || elems_per_rank == 0, "send / recv is NULL");
    CL_REQUIRE(send != recv || c->world == 1, "send and recv must not alias");
    if (elems_per_rank == 0) return CLEORA_OK;
    CL_HIP(hipSetDevice(c->device));
    if (c->world == 1) {
        if (send != recv)
            CL_HIP(hipMemcpyAsync(recv, send, elems_per_rank * sizeof(float), hipMemcpyDeviceToDevice, S(stream)));
        return CLEORA_OK;
    }
    CL_REQUIRE(c->comm != nullptr, "a local communicator has no all-to-all (the column partition's layout switch needs RCCL)");
    Rccl *r;
    int rc = need_rccl(&r);
    if (rc != CLEORA_OK) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    const int P = c->world, me = c->rank;
    CL_HIP(hipMemcpyAsync(recv + (uint64_t)me * elems_per_rank, send + (uint64_t)me * elems_per_rank,
                          elems_per_rank * sizeof(float), hipMemcpyDeviceToDevice, S(stream)));
    ncclResult_t first = ncclSuccess;
    auto note = [&first](ncclResult_t e) { if (first == ncclSuccess) first = e; };
    CL_RCCL(*r, r->group_start());
    for (int k = 1; k < P; ++k) {
        const int to = (me + k) % P, from = (me - k + P) % P;
        note(r->send(send + (uint64_t)to * elems_per_rank, elems_per_rank, ncclFloat, to, c->comm, S(stream)));
        note(r->recv(recv + (uint64_t)from * elems_per_rank, elems_per_rank, ncclFloat, from, c->comm, S(stream)));
    }
    note(r->group_end());
    if (first != ncclSuccess) return rccl_fail(*r, first, "all-to-all");
    return CLEORA_OK;
}

}  // extern "C"
