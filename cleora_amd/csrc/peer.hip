// peer.hip — the PEER-DIRECT transport of the communicator (comm.hip has the RCCL one): the ranks of one node map each other's
// buffers with hipIpc and exchange data with plain stores / loads over xGMI (or inside one GPU when ranks share it).
//
// The reference has no distributed code (pycleora is single-process: rayon over rows, src/embedding.rs:59-63); this is the
// exchange step BASELINE.json:north_star asks for, in the form SURVEY.md 8e names as the mitigation for the all-gather being
// the critical path at 8 GPUs: "peer-mapped direct stores".
//
//   bootstrap   a POSIX shared-memory segment named after the communicator's id: a sense-reversing barrier for the ranks' HOST
//               threads and one 128-byte record per rank for exchanging hipIpc handles.  Node-local by construction.
//   mailbox     per rank, device memory allocated UNCACHED (hipDeviceMallocUncached): flags[channel][rank] = the last
//               sequence number that rank signalled; mapped by every peer.
//   register    a buffer every rank holds (the replicas of the iterate) is exported once and mapped by every peer.
//   all-gather  push_kernel: this rank's shard -> the same offsets of every peer's buffer (blockIdx.y = peer: all links at the
//               same time), then signal_kernel: a system-scope release store of the sequence number into every peer's mailbox
//               (a kernel boundary separates data and flag), then wait_kernel: one wave polls the own mailbox until every peer
//               has signalled — bounded by a wall-clock budget, after which it records an error instead of hanging the GPU.
//               The consumer's next kernel starts with the usual acquire, so it sees the peers' stores.
//   all-reduce  (local communicators) post to the own scratch, signal, wait, then EVERY rank sums the P contributions in rank
//               order through the mappings: bit-identical results on all ranks, no reduction tree.  Scratch halves alternate
//               with the sequence number; a half is reused two operations later, when every peer has provably read it.
//   broadcast   the same through the root's scratch.
// Why a rank may safely store into a peer's replica: rank r writes rows of X_next of iteration t only after its own SpMM of that
// block, which needs all of X of iteration t, i.e. every peer's last signal of iteration t-1 — sent after that peer's last
// kernel reading the buffer that is X_next now.  (sharded.hip keeps that order: gathers of an iteration are joined before the
// next iteration's first kernel.)
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>

#include "comm_internal.h"
#include "shm_barrier.h"

namespace cleora {

constexpr int kMaxWorld = 64;
constexpr int kChannels = 2;                  // 0: all-gather, 1: all-reduce / broadcast
constexpr double kHostBarrierSeconds = 180.0; // a rank that never arrives is an error, not a hang
constexpr uint64_t kWaitTicks = 60ull * 100000000ull;   // wait_kernel's budget: 60 s of the 100 MHz wall clock
constexpr uint64_t kProbeFloats = 1024;                 // self-test: 4 KiB per rank — small enough to stay resident in L1 / L2, where a stale line would sit

struct ShmRecord {                            // what a rank publishes for one exchange
    hipIpcMemHandle_t handle;
    uint64_t offset, bytes;
    int32_t device, ok;                       // ok: 1 = handle + raw pointer, 2 = raw pointer only (the buffer cannot be exported), 0 = nothing
    uint64_t raw;                             // the buffer's address in the publishing PROCESS: ranks that are threads of one process
    int64_t pid;                              //   (csrc/multi.hip: one process, P devices) use it directly — hipIpc cannot open a handle at home
    uint64_t nonce;                           // drawn once per process: two containers sharing /dev/shm may both be pid 1
    uint8_t pad[128 - sizeof(hipIpcMemHandle_t) - 48];
};
static_assert(sizeof(ShmRecord) == 128, "record size");

struct ShmSegment {
    ShmBarrier barrier;
    std::atomic<uint32_t> failed;
    uint32_t world;
    ShmRecord rec[kMaxWorld];
};

struct Mailbox {                              // device memory (uncached), one per rank
    uint64_t flags[kChannels][kMaxWorld];
    uint64_t error;                           // != 0: a wait timed out (channel + 1 in the low byte, the missing rank above it)
    uint64_t pad[7];
};

struct Registration {
    char *local = nullptr;
    uint64_t bytes = 0;
    char *peer[kMaxWorld] = {nullptr};        // the same buffer in rank p's memory, as mapped here (nullptr for the own rank)
    void *mapped_base[kMaxWorld] = {nullptr}; // what hipIpcOpenMemHandle returned (to close)
};

struct PeerPtrs { void *p[kMaxWorld]; };

struct Mapping {                              // an allocation of a peer opened here; shared by every registration inside it
    int peer;
    hipIpcMemHandle_t handle;
    void *base;
    int refs;
};

struct PeerLayer {
    // every rank is a thread of THIS process (csrc/multi.hip): signals are HIP events instead of flags polled by a kernel — the
    // streams of one process share a handful of hardware queues, and one rank's polling kernel in front of the peer's signal on
    // the same queue would wait for its whole budget
    bool inproc = false;
    bool poisoned = false;                    // a host barrier timed out
    hipEvent_t ev[kChannels][2] = {{nullptr, nullptr}, {nullptr, nullptr}};      // [channel][sequence parity]
    PeerLayer *peer_layer[kMaxWorld] = {nullptr};
    std::vector<Mapping> mappings;
    ShmSegment *shm = nullptr;
    size_t shm_bytes = 0;
    uint32_t local_sense = 0;
    Mailbox *mailbox = nullptr;               // own (device pointer)
    Registration mailboxes;                   // every peer's mailbox
    uint64_t seq[kChannels] = {0, 0};
    std::vector<Registration> regs;
    char *scratch = nullptr;                  // all-reduce / broadcast staging, two halves
    uint64_t scratch_half = 0;
    // all-gather form: 0 = PUSH (plain stores into the peers' replicas, all links at once), 1 = PULL (every rank copies the peers'
    // shards out of THEIR replicas with system-scope loads: slower, but it does not depend on this device's caches seeing stores that
    // arrive from outside).  Decided by peer_selftest(), alike on every rank.
    int mode = 0;
    bool selftested = false;
    float *probe = nullptr;                   // self-test buffer: world slots of kProbeFloats (registered)
    uint32_t *probe_bad = nullptr;            // device counter of mismatching words
};

namespace {

uint64_t fnv1a(const unsigned char *p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

// identifies THIS process among the ranks of a communicator: the pid alone is not enough across pid namespaces
uint64_t process_nonce() {
    static const uint64_t nonce = [] {
        uint64_t v = 0;
        if (FILE *f = std::fopen("/dev/urandom", "rb")) {
            if (std::fread(&v, sizeof v, 1, f) != 1) v = 0;
            std::fclose(f);
        }
        if (!v) v = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((uint64_t)getpid() << 32) ^ reinterpret_cast<uintptr_t>(&v);
        return v | 1;
    }();
    return nonce;
}
bool same_process(const ShmRecord &r) { return r.pid == (int64_t)getpid() && r.nonce == process_nonce(); }

int barrier_host(cleora_comm *c) {
    PeerLayer *pl = c->peer;
    // a barrier that timed out leaves its arrival count behind: every later barrier on the segment would pair up wrongly, so the
    // communicator refuses further collectives instead; destroy it and create a new one
    if (pl->poisoned) {
        set_error("peer transport: this communicator lost a rank at an earlier host barrier and cannot be used any more");
        return CLEORA_E_RCCL;
    }
    if (shm_barrier_wait(&pl->shm->barrier, (uint32_t)c->world, &pl->local_sense, kHostBarrierSeconds)) return CLEORA_OK;
    pl->poisoned = true;
    set_error("peer transport: a rank did not reach the host barrier within " + std::to_string((int)kHostBarrierSeconds) + " s");
    return CLEORA_E_RCCL;
}

void close_registration(PeerLayer *pl, Registration &r, int world);

// every rank publishes (ptr, bytes) [ptr may be nullptr: ok = 0], all map all; returns with `out` filled.  Collective.
int exchange_and_map(cleora_comm *c, void *ptr, uint64_t bytes, Registration *out) {
    PeerLayer *pl = c->peer;
    ShmRecord &mine = pl->shm->rec[c->rank];
    std::memset(&mine, 0, sizeof(mine));
    int rc = CLEORA_OK;
    void *base = nullptr;
    size_t range = 0;
    std::string export_error;
    if (ptr) {
        hipError_t e = hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t *>(&base), &range, ptr);
        if (e == hipSuccess) e = hipIpcGetMemHandle(&mine.handle, base);
        mine.raw = (uint64_t)reinterpret_cast<uintptr_t>(ptr);
        mine.pid = (int64_t)getpid();
        mine.nonce = process_nonce();
        mine.bytes = bytes;
        mine.device = c->device;
        if (e != hipSuccess) {
            (void)hipGetLastError();
            export_error = std::string("the buffer cannot be exported with hipIpcGetMemHandle (") + hipGetErrorString(e) +
                           "); it must come from hipMalloc / cleora_malloc, and HSA_ENABLE_IPC_MODE_LEGACY=0 must be set";
            mine.ok = 2;                                    // still reachable by the ranks of this process
        } else {
            mine.offset = (uint64_t)(static_cast<char *>(ptr) - static_cast<char *>(base));
            mine.ok = 1;
        }
    }
    int b = barrier_host(c);
    if (b != CLEORA_OK) return b;
    out->local = static_cast<char *>(ptr);
    out->bytes = bytes;
    for (int p = 0; p < c->world && rc == CLEORA_OK; ++p) {
        if (p == c->rank) continue;
        const ShmRecord &r = pl->shm->rec[p];
        if (!r.ok || r.bytes != bytes) {
            set_error("peer transport: rank " + std::to_string(p) + " did not publish a matching buffer");
            rc = CLEORA_E_INVALID;
            break;
        }
        if (same_process(r)) {
            // a rank of this process (another host thread, maybe another device): its pointer is ours too — direct peer access
            if (r.device != c->device) {
                const hipError_t e = hipDeviceEnablePeerAccess(r.device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                    (void)hipGetLastError();
                    set_error(std::string("peer transport: hipDeviceEnablePeerAccess(") + std::to_string(r.device) + ") from device " + std::to_string(c->device) +
                              " failed (" + hipGetErrorString(e) + ")");
                    rc = CLEORA_E_HIP;
                    break;
                }
                (void)hipGetLastError();
            }
            out->mapped_base[p] = nullptr;                    // nothing to close
            out->peer[p] = reinterpret_cast<char *>((uintptr_t)r.raw);
            continue;
        }
        if (r.ok != 1) {
            set_error("peer transport: rank " + std::to_string(p) + " (another process) could not export its buffer");
            rc = CLEORA_E_HIP;
            break;
        }
        void *mapped = nullptr;
        hipIpcMemHandle_t h = r.handle;
        for (Mapping &m : pl->mappings)                       // two buffers of one allocation (a caching allocator's segment): one mapping
            if (m.peer == p && std::memcmp(&m.handle, &h, sizeof h) == 0) { mapped = m.base; ++m.refs; break; }
        if (!mapped) {
            const hipError_t e = hipIpcOpenMemHandle(&mapped, h, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                set_error(std::string("peer transport: hipIpcOpenMemHandle of rank ") + std::to_string(p) + "'s buffer failed (" + hipGetErrorString(e) + ")");
                rc = CLEORA_E_HIP;
                break;
            }
            pl->mappings.push_back(Mapping{p, h, mapped, 1});
        }
        out->mapped_base[p] = mapped;
        out->peer[p] = static_cast<char *>(mapped) + r.offset;
    }
    if (rc == CLEORA_OK && !export_error.empty())
        for (int p = 0; p < c->world; ++p)
            if (p != c->rank && !same_process(pl->shm->rec[p])) { set_error("peer transport: " + export_error); rc = CLEORA_E_HIP; break; }
    if (rc != CLEORA_OK) pl->shm->failed.store(1, std::memory_order_release);
    b = barrier_host(c);                                    // nobody reuses the records before everybody has read them
    if (b != CLEORA_OK) return b;
    if (pl->shm->failed.load(std::memory_order_acquire)) {
        close_registration(pl, *out, c->world);
        if (rc == CLEORA_OK) { set_error("peer transport: the exchange failed on another rank"); rc = CLEORA_E_RCCL; }
        (void)barrier_host(c);
        if (c->rank == 0) pl->shm->failed.store(0, std::memory_order_release);
        (void)barrier_host(c);
        return rc;
    }
    return CLEORA_OK;
}

void release_mapping(PeerLayer *pl, int peer, void *base) {
    for (size_t k = 0; k < pl->mappings.size(); ++k) {
        Mapping &m = pl->mappings[k];
        if (m.peer != peer || m.base != base) continue;
        if (--m.refs == 0) {
            (void)hipIpcCloseMemHandle(base);
            pl->mappings.erase(pl->mappings.begin() + (long)k);
        }
        return;
    }
}

void close_registration(PeerLayer *pl, Registration &r, int world) {
    for (int p = 0; p < world; ++p)
        if (r.mapped_base[p]) { release_mapping(pl, p, r.mapped_base[p]); r.mapped_base[p] = nullptr; r.peer[p] = nullptr; }
}

// ---- kernels ------------------------------------------------------------------------------------------------------------
// this rank's shard into every peer: blockIdx.y = peer slot (dst.p[] is already offset to the shard and ordered for link stagger)
__global__ __launch_bounds__(256) void push_kernel(const float *__restrict__ src, PeerPtrs dst, uint64_t n_floats, int vec4) {
    float *out = static_cast<float *>(dst.p[blockIdx.y]);
    const uint64_t stride = (uint64_t)gridDim.x * 256, t0 = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (vec4) {
        const uint64_t n4 = n_floats >> 2;
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        float4 *o4 = reinterpret_cast<float4 *>(out);
        for (uint64_t i = t0; i < n4; i += stride) o4[i] = s4[i];
        for (uint64_t i = (n4 << 2) + t0; i < n_floats; i += stride) out[i] = src[i];
    } else {
        for (uint64_t i = t0; i < n_floats; i += stride) out[i] = src[i];
    }
}

// flags[channel][me] = seq in every peer's mailbox (lane = peer)
__global__ void signal_kernel(PeerPtrs mailboxes, int channel, int me, int world, uint64_t seq) {
    const int p = threadIdx.x;
    if (p >= world || p == me) return;
    Mailbox *mb = static_cast<Mailbox *>(mailboxes.p[p]);
    __hip_atomic_store(&mb->flags[channel][me], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// until every peer's flag of `channel` in the OWN mailbox has reached seq
__global__ void wait_kernel(Mailbox *mb, int channel, int me, int world, uint64_t seq, uint64_t budget_ticks) {
    const int p = threadIdx.x;
    if (p >= world || p == me) return;
    const uint64_t t0 = wall_clock64();
    while (__hip_atomic_load(&mb->flags[channel][p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > budget_ticks) {
            __hip_atomic_store(&mb->error, (uint64_t)(channel + 1) | ((uint64_t)p << 8) | (seq << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
    }
}

// out[i] = sum over ranks (in rank order) of their posted vectors; src.p[r] = rank r's scratch half (own one included)
template <class T>
__global__ __launch_bounds__(256) void reduce_kernel(PeerPtrs src, int world, uint64_t n, T *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        T s = 0;
        for (int r = 0; r < world; ++r) {
            // a load that does not stop at this device's non-coherent caches: the peer wrote the value in ITS memory
            const T *p = static_cast<const T *>(src.p[r]) + i;
            if constexpr (sizeof(T) == 8) {
                const uint64_t bits = __hip_atomic_load(reinterpret_cast<const uint64_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                s += __builtin_bit_cast(T, bits);
            } else {
                const uint32_t bits = __hip_atomic_load(reinterpret_cast<const uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                s += __builtin_bit_cast(T, bits);
            }
        }
        out[i] = s;
    }
}

__global__ __launch_bounds__(256) void fetch_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, uint64_t n_words) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * 256)
        dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// PULL form of the all-gather: blockIdx.y = peer slot; the peer's shard out of ITS replica (src.p[]) into the same offsets of this
// rank's (dst.p[], already offset), with loads that do not stop at this device's non-coherent caches
__global__ __launch_bounds__(256) void pull_kernel(PeerPtrs src, PeerPtrs dst, PeerPtrs count) {
    const uint64_t n_floats = reinterpret_cast<uint64_t>(count.p[blockIdx.y]);
    const float *in = static_cast<const float *>(src.p[blockIdx.y]);
    float *out = static_cast<float *>(dst.p[blockIdx.y]);
    const uint64_t stride = (uint64_t)gridDim.x * 256, t0 = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 7u) == 0) {
        const uint64_t n2 = n_floats >> 1;
        const uint64_t *i2 = reinterpret_cast<const uint64_t *>(in);
        uint64_t *o2 = reinterpret_cast<uint64_t *>(out);
        for (uint64_t i = t0; i < n2; i += stride) o2[i] = __hip_atomic_load(i2 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (uint64_t i = (n2 << 1) + t0; i < n_floats; i += stride)
            out[i] = __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const uint32_t *>(in) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    } else {
        for (uint64_t i = t0; i < n_floats; i += stride)
            out[i] = __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const uint32_t *>(in) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    }
}

// ---- self-test kernels (peer_selftest) ----
__device__ __forceinline__ float probe_pattern(uint32_t round, uint32_t rank, uint32_t i) {
    return __builtin_bit_cast(float, 0x3f800000u | ((round * 2654435761u + rank * 40503u + i * 7919u) & 0x007fffffu));    // in [1, 2): never NaN
}
// every CU reads the whole probe buffer with plain loads: whatever the caches hold of it afterwards is what a later plain load may see
__global__ __launch_bounds__(256) void probe_warm_kernel(const float *buf, uint64_t n, float *sink) {
    float s = 0.f;
    for (uint64_t i = threadIdx.x; i < n; i += 256) s += buf[i];
    if (s == -1.f) sink[0] = s;                // (never: keeps the loads)
}
__global__ __launch_bounds__(256) void probe_fill_kernel(float *slot, uint32_t round, uint32_t rank) {
    for (uint32_t i = threadIdx.x; i < kProbeFloats; i += 256) slot[i] = probe_pattern(round, rank, i);
}
// plain loads again — the load path of the SpMM's gathers — against what every rank must have written; skew != 0 expects another pattern
// (fault injection)
__global__ __launch_bounds__(256) void probe_check_kernel(const float *buf, int world, uint32_t round, uint32_t skew, uint32_t *bad) {
    const uint32_t p = blockIdx.x;
    if ((int)p >= world) return;
    uint32_t mism = 0;
    for (uint32_t i = threadIdx.x; i < kProbeFloats; i += 256)
        mism += __builtin_bit_cast(uint32_t, buf[(uint64_t)p * kProbeFloats + i]) != __builtin_bit_cast(uint32_t, probe_pattern(round + skew, p, i));
    if (mism) atomicAdd(bad, mism);
}

int signal_and_wait(cleora_comm *c, int channel, uint64_t seq, hipStream_t stream) {
    PeerLayer *pl = c->peer;
    if (pl->inproc) {
        // record, meet on the host (a wait on an event nobody has recorded yet is a no-op), wait for every peer's record.  The
        // parity's event is recorded again two operations later — after the meeting of the operation in between, which every
        // rank reaches only once it has enqueued its waits of this one.
        CL_HIP(hipEventRecord(pl->ev[channel][seq & 1], stream));
        const int b = barrier_host(c);
        if (b != CLEORA_OK) return b;
        for (int p = 0; p < c->world; ++p)
            if (p != c->rank) CL_HIP(hipStreamWaitEvent(stream, pl->peer_layer[p]->ev[channel][seq & 1], 0));
        return CLEORA_OK;
    }
    PeerPtrs mb{};
    for (int p = 0; p < c->world; ++p) mb.p[p] = p == c->rank ? (void *)pl->mailbox : (void *)pl->mailboxes.peer[p];
    hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(kMaxWorld), 0, stream, mb, channel, c->rank, c->world, seq);
    hipLaunchKernelGGL(wait_kernel, dim3(1), dim3(kMaxWorld), 0, stream, pl->mailbox, channel, c->rank, c->world, seq, kWaitTicks);
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
}

// staging for all-reduce / broadcast: two halves of at least `bytes`; growing it is collective and host-synchronous
int ensure_scratch(cleora_comm *c, uint64_t bytes, hipStream_t stream) {
    PeerLayer *pl = c->peer;
    if (pl->scratch && pl->scratch_half >= bytes) return CLEORA_OK;
    CL_HIP(hipStreamSynchronize(stream));                   // whatever still reads the old staging
    int rc;
    if (pl->scratch) {
        if ((rc = peer_unregister(c, pl->scratch)) != CLEORA_OK) return rc;
        (void)hipFree(pl->scratch);
        pl->scratch = nullptr;
    }
    uint64_t half = 1ull << 20;
    while (half < bytes) half <<= 1;
    void *p = nullptr;
    CL_HIP(hipMalloc(&p, 2 * half));
    pl->scratch = static_cast<char *>(p);
    pl->scratch_half = half;
    return peer_register(c, pl->scratch, 2 * half);
}

Registration *find_registration(PeerLayer *pl, const void *ptr, uint64_t bytes) {
    const char *b = static_cast<const char *>(ptr);
    for (Registration &r : pl->regs)
        if (b >= r.local && b + bytes <= r.local + r.bytes) return &r;
    return nullptr;
}

}  // namespace

// the communicator will never see all of its ranks again (a partly built multi-device handle is being torn down): its host
// barriers return at once instead of waiting out their budget for ranks that do not exist
void peer_abandon(cleora_comm *c) {
    if (c && c->peer) c->peer->poisoned = true;
}

int peer_host_barrier(cleora_comm *c) {
    if (!c->peer || c->world == 1) return CLEORA_OK;
    return barrier_host(c);
}

int peer_enable(cleora_comm *c) {
    if (c->peer) return CLEORA_OK;
    CL_REQUIRE(c->world <= kMaxWorld, "the peer transport serves at most 64 ranks (one node)");
    CL_HIP(hipSetDevice(c->device));
    PeerLayer *pl = new (std::nothrow) PeerLayer();
    if (!pl) { set_error("host allocation failed"); return CLEORA_E_OOM; }
    c->peer = pl;
    auto fail = [&](int rc) { peer_destroy(c); return rc; };
    // the mailbox: uncached device memory where the platform offers it (flags polled by a running kernel must not sit in a
    // non-coherent cache), else fine-grained, else plain (enough when the ranks share one device)
    void *mb = nullptr;
    for (int kind = 0; kind < 3 && !mb; ++kind) {
        void *p = nullptr;
        const hipError_t e = kind == 0 ? hipExtMallocWithFlags(&p, sizeof(Mailbox), hipDeviceMallocUncached)
                             : kind == 1 ? hipExtMallocWithFlags(&p, sizeof(Mailbox), hipDeviceMallocFinegrained) : hipMalloc(&p, sizeof(Mailbox));
        if (e != hipSuccess) { (void)hipGetLastError(); continue; }
        hipIpcMemHandle_t probe;                              // a kind that cannot be exported is of no use here
        if (c->world > 1 && hipIpcGetMemHandle(&probe, p) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); continue; }
        mb = p;
    }
    if (!mb) { set_error("peer transport: no exportable device memory for the mailbox (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)"); return fail(CLEORA_E_HIP); }
    pl->mailbox = static_cast<Mailbox *>(mb);
    if (hipMemset(mb, 0, sizeof(Mailbox)) != hipSuccess) { (void)hipGetLastError(); set_error("peer transport: hipMemset failed"); return fail(CLEORA_E_HIP); }
    if (c->world == 1) return CLEORA_OK;
    char name[64];
    std::snprintf(name, sizeof name, "/cleora.%016llx", (unsigned long long)fnv1a(c->id, sizeof c->id));
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { set_error(std::string("peer transport: shm_open failed: ") + std::strerror(errno)); return fail(CLEORA_E_RCCL); }
    pl->shm_bytes = sizeof(ShmSegment);
    if (ftruncate(fd, (off_t)pl->shm_bytes) != 0) { close(fd); set_error("peer transport: ftruncate of the shared segment failed"); return fail(CLEORA_E_RCCL); }
    void *m = mmap(nullptr, pl->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { set_error("peer transport: mmap of the shared segment failed"); return fail(CLEORA_E_RCCL); }
    pl->shm = static_cast<ShmSegment *>(m);                  // a fresh segment is zero-filled: barrier state starts at 0 everywhere
    int rc = barrier_host(c);
    if (c->rank == 0) (void)shm_unlink(name);               // the mappings live on; nothing is left behind if a rank dies later
    if (rc != CLEORA_OK) return fail(rc);
    if ((rc = exchange_and_map(c, pl->mailbox, sizeof(Mailbox), &pl->mailboxes)) != CLEORA_OK) return fail(rc);
    // are all ranks threads of this process?  (the records of the exchange above are still in place: the next write to them
    // comes after the barrier below)
    bool same = true;
    for (int p = 0; p < c->world; ++p) same = same && same_process(pl->shm->rec[p]);
    if ((rc = barrier_host(c)) != CLEORA_OK) return fail(rc);
    if (same) {
        for (int ch = 0; ch < kChannels; ++ch)
            for (int k = 0; k < 2; ++k)
                if (hipEventCreateWithFlags(&pl->ev[ch][k], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); set_error("peer transport: hipEventCreate failed"); same = false; }
        pl->shm->rec[c->rank].raw = (uint64_t)reinterpret_cast<uintptr_t>(pl);
        pl->shm->rec[c->rank].ok = same ? 1 : 0;
        if ((rc = barrier_host(c)) != CLEORA_OK) return fail(rc);
        bool all = true;
        for (int p = 0; p < c->world; ++p) {
            all = all && pl->shm->rec[p].ok == 1;
            pl->peer_layer[p] = reinterpret_cast<PeerLayer *>((uintptr_t)pl->shm->rec[p].raw);
        }
        if ((rc = barrier_host(c)) != CLEORA_OK) return fail(rc);
        if (!all) { set_error("peer transport: creating the in-process signal events failed on a rank"); return fail(CLEORA_E_HIP); }
        pl->inproc = true;
    }
    // the handshake: refuse (or downgrade) the transport NOW, in milliseconds, not at the first all-gather of a 10 GB iterate
    if ((rc = peer_selftest(c, 0)) != CLEORA_OK) return fail(rc);
    return CLEORA_OK;
}

void peer_destroy(cleora_comm *c) {
    PeerLayer *pl = c->peer;
    if (!pl) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (Registration &r : pl->regs) close_registration(pl, r, c->world);
    close_registration(pl, pl->mailboxes, c->world);
    if (pl->shm && c->world > 1) (void)barrier_host(c);      // peers have closed their mappings of OUR memory before we free it
    if (pl->scratch) (void)hipFree(pl->scratch);
    if (pl->probe) (void)hipFree(pl->probe);
    if (pl->probe_bad) (void)hipFree(pl->probe_bad);
    if (pl->mailbox) (void)hipFree(pl->mailbox);
    for (int ch = 0; ch < kChannels; ++ch)
        for (int k = 0; k < 2; ++k)
            if (pl->ev[ch][k]) (void)hipEventDestroy(pl->ev[ch][k]);      // (after the barrier above: no peer enqueues a wait on them any more)
    if (pl->shm) munmap(pl->shm, pl->shm_bytes);
    delete pl;
    c->peer = nullptr;
}

int peer_register(cleora_comm *c, void *buf, uint64_t bytes) {
    CL_REQUIRE(c->peer != nullptr, "the peer transport is not enabled on this communicator");
    CL_REQUIRE(buf != nullptr && bytes > 0, "buf is NULL / empty");
    PeerLayer *pl = c->peer;
    // Registering is collective, so "already registered" must be a decision every rank takes alike: only the SAME buffer (address and
    // size) counts — the same call repeated on every rank.  A buffer that merely lies inside, or overlaps, an existing registration
    // (a freed allocation whose address came back on some ranks, a sub-buffer) would make this rank skip the exchange the others
    // enter: refused instead.
    const char *b0 = static_cast<const char *>(buf);
    for (const Registration &r : pl->regs) {
        if (r.local == b0 && r.bytes == bytes) return CLEORA_OK;
        if (b0 < r.local + r.bytes && r.local < b0 + bytes) {
            set_error("peer transport: the buffer overlaps a registered one without being it; unregister that one first (cleora_comm_unregister)");
            return CLEORA_E_INVALID;
        }
    }
    CL_HIP(hipSetDevice(c->device));
    Registration reg;
    if (c->world > 1) {
        const int rc = exchange_and_map(c, buf, bytes, &reg);
        if (rc != CLEORA_OK) return rc;
    } else {
        reg.local = static_cast<char *>(buf);
        reg.bytes = bytes;
    }
    pl->regs.push_back(reg);
    return CLEORA_OK;
}

int peer_unregister(cleora_comm *c, void *buf) {
    if (!c->peer) return CLEORA_OK;
    PeerLayer *pl = c->peer;
    for (size_t k = 0; k < pl->regs.size(); ++k) {
        if (pl->regs[k].local != static_cast<char *>(buf)) continue;
        CL_HIP(hipSetDevice(c->device));
        CL_HIP(hipDeviceSynchronize());                     // our kernels that store into the peers' copies
        int rc = c->world > 1 ? barrier_host(c) : CLEORA_OK;
        close_registration(pl, pl->regs[k], c->world);
        if (rc == CLEORA_OK && c->world > 1) rc = barrier_host(c);   // every mapping of `buf` is closed: the owner may free it
        pl->regs.erase(pl->regs.begin() + (long)k);
        return rc;
    }
    return CLEORA_OK;
}

int peer_allgatherv_f32(cleora_comm *c, float *buf, const uint64_t *offsets, hipStream_t stream) {
    CL_REQUIRE(c->peer != nullptr, "the peer transport is not enabled on this communicator");
    PeerLayer *pl = c->peer;
    const int P = c->world, me = c->rank;
    if (P == 1) return CLEORA_OK;
    Registration *reg = find_registration(pl, buf, offsets[P] * sizeof(float));
    CL_REQUIRE(reg != nullptr, "peer-direct all-gather: the buffer is not registered (cleora_comm_register)");
    CL_HIP(hipSetDevice(c->device));
    const uint64_t mine = offsets[me + 1] - offsets[me];
    const uint64_t seq = ++pl->seq[0];
    if (pl->mode == 1) {
        // PULL: "my shard is final" (it was written by kernels ahead of this call on `stream`: a kernel boundary), wait for every peer's
        // word, then copy their shards out of their replicas.  A peer overwrites a shard only two iterations later, after an
        // all-gather that every rank enters behind this copy (sharded.hip joins the gathers of an iteration before the next one starts).
        const int rc = signal_and_wait(c, 0, seq, stream);
        if (rc != CLEORA_OK) return rc;
        PeerPtrs src{}, dst{}, cnt{};
        uint64_t most = 0;
        int slots = 0;
        for (int k = 1; k < P; ++k) {
            const int p = (me + k) % P;
            const uint64_t np = offsets[p + 1] - offsets[p];
            if (!np) continue;
            const uint64_t byte_off = (uint64_t)(reinterpret_cast<char *>(buf + offsets[p]) - reg->local);
            src.p[slots] = reg->peer[p] + byte_off;
            dst.p[slots] = buf + offsets[p];
            cnt.p[slots] = reinterpret_cast<void *>((uintptr_t)np);
            most = np > most ? np : most;
            ++slots;
        }
        if (slots) {
            uint64_t bx = (most / 2 + 255) / 256;
            bx = bx > 160 ? 160 : (bx ? bx : 1);
            hipLaunchKernelGGL(pull_kernel, dim3((unsigned)bx, (unsigned)slots), dim3(256), 0, stream, src, dst, cnt);
            CL_HIP(hipGetLastError());
        }
        return CLEORA_OK;
    }
    if (mine) {
        const uint64_t byte_off = (uint64_t)(reinterpret_cast<char *>(buf + offsets[me]) - reg->local);
        PeerPtrs dst{};
        for (int k = 1; k < P; ++k) dst.p[k - 1] = reg->peer[(me + k) % P] + byte_off;       // staggered: rank r starts with r + 1
        bool vec4 = (byte_off & 15u) == 0 && (reinterpret_cast<uintptr_t>(reg->local) & 15u) == 0;
        for (int p = 0; p < P; ++p)                                                          // the peers' mapped copies need not be aligned like ours
            if (p != me && (reinterpret_cast<uintptr_t>(reg->peer[p]) & 15u) != 0) vec4 = false;
        const uint64_t units = vec4 ? (mine + 3) / 4 : mine;
        uint64_t bx = (units + 255) / 256;
        if (bx > 160) bx = 160;                                                              // bandwidth-bound on the links: a few blocks per CU in total
        hipLaunchKernelGGL(push_kernel, dim3((unsigned)bx, (unsigned)(P - 1)), dim3(256), 0, stream, buf + offsets[me], dst, mine, vec4 ? 1 : 0);
    }
    return signal_and_wait(c, 0, seq, stream);
}

int peer_allreduce(cleora_comm *c, void *buf, uint64_t n, bool f64, hipStream_t stream) {
    CL_REQUIRE(c->peer != nullptr, "the peer transport is not enabled on this communicator");
    PeerLayer *pl = c->peer;
    if (c->world == 1 || n == 0) return CLEORA_OK;
    CL_HIP(hipSetDevice(c->device));
    const uint64_t bytes = n * (f64 ? 8 : 4);
    int rc = ensure_scratch(c, bytes, stream);
    if (rc != CLEORA_OK) return rc;
    const uint64_t seq = ++pl->seq[1];
    const uint64_t half = (seq & 1) * pl->scratch_half;
    CL_HIP(hipMemcpyAsync(pl->scratch + half, buf, bytes, hipMemcpyDeviceToDevice, stream));
    if ((rc = signal_and_wait(c, 1, seq, stream)) != CLEORA_OK) return rc;
    Registration *reg = find_registration(pl, pl->scratch, 2 * pl->scratch_half);
    PeerPtrs src{};
    for (int p = 0; p < c->world; ++p) src.p[p] = (p == c->rank ? pl->scratch : reg->peer[p]) + half;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    if (f64) hipLaunchKernelGGL(reduce_kernel<double>, dim3(blocks), dim3(256), 0, stream, src, c->world, n, static_cast<double *>(buf));
    else hipLaunchKernelGGL(reduce_kernel<float>, dim3(blocks), dim3(256), 0, stream, src, c->world, n, static_cast<float *>(buf));
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
}

int peer_broadcast(cleora_comm *c, void *buf, uint64_t bytes, int root, hipStream_t stream) {
    CL_REQUIRE(c->peer != nullptr, "the peer transport is not enabled on this communicator");
    PeerLayer *pl = c->peer;
    if (c->world == 1 || bytes == 0) return CLEORA_OK;
    CL_REQUIRE(bytes % 4 == 0, "broadcast of whole 4-byte words");
    CL_HIP(hipSetDevice(c->device));
    int rc = ensure_scratch(c, bytes, stream);
    if (rc != CLEORA_OK) return rc;
    const uint64_t seq = ++pl->seq[1];
    const uint64_t half = (seq & 1) * pl->scratch_half;
    if (c->rank == root) CL_HIP(hipMemcpyAsync(pl->scratch + half, buf, bytes, hipMemcpyDeviceToDevice, stream));
    if ((rc = signal_and_wait(c, 1, seq, stream)) != CLEORA_OK) return rc;
    if (c->rank != root) {
        Registration *reg = find_registration(pl, pl->scratch, 2 * pl->scratch_half);
        const uint64_t words = bytes / 4;
        const unsigned blocks = (unsigned)((words + 255) / 256 > 1024 ? 1024 : (words + 255) / 256);
        hipLaunchKernelGGL(fetch_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const uint32_t *>(reg->peer[root] + half),
                           static_cast<uint32_t *>(buf), words);
        CL_HIP(hipGetLastError());
    }
    return CLEORA_OK;
}

// Does data handed over by the all-gather arrive where the NEXT KERNEL'S PLAIN LOADS see it?  (The product loops trust exactly that:
// the SpMM gathers from a replica the peers stored into.)  Three rounds on a 4 KiB-per-rank probe buffer: every CU first reads the
// whole buffer (so stale copies sit in its caches), every rank then writes a fresh pattern into its slot, the all-gather under test
// runs, and a checker kernel compares every word with plain loads.  A failing PUSH form (plain stores into the peers' memory) is
// replaced by the PULL form (system-scope loads from the peers' memory) and tested again; if that fails too the transport is refused.
// The verdict is taken together (one flag in the shared segment): every rank ends in the same mode.  Collective, host-synchronous.
// flags: 1 = the PUSH checker expects a pattern nobody wrote (exercises the fallback), 2 = the PULL checker as well (the refusal).
int peer_selftest(cleora_comm *c, uint32_t flags) {
    CL_REQUIRE(c->peer != nullptr, "the peer transport is not enabled on this communicator");
    PeerLayer *pl = c->peer;
    if (c->world == 1) { pl->selftested = true; return CLEORA_OK; }
    CL_HIP(hipSetDevice(c->device));
    int rc;
    if (!pl->probe) {
        void *q = nullptr;
        CL_HIP(hipMalloc(&q, (uint64_t)c->world * kProbeFloats * sizeof(float)));
        pl->probe = static_cast<float *>(q);
        CL_HIP(hipMemset(q, 0, (uint64_t)c->world * kProbeFloats * sizeof(float)));
        CL_HIP(hipMalloc(reinterpret_cast<void **>(&pl->probe_bad), sizeof(uint32_t)));
        if ((rc = peer_register(c, pl->probe, (uint64_t)c->world * kProbeFloats * sizeof(float))) != CLEORA_OK) return rc;
    }
    std::vector<uint64_t> offsets((size_t)c->world + 1);
    for (int p = 0; p <= c->world; ++p) offsets[p] = (uint64_t)p * kProbeFloats;
    const int saved = pl->mode;
    // its own stream: the ranks may be threads of one process on one device (csrc/multi.hip), whose null stream they would share
    struct TestStream {
        hipStream_t s = nullptr;
        ~TestStream() { if (s) (void)hipStreamDestroy(s); }
    } ts;
    CL_HIP(hipStreamCreateWithFlags(&ts.s, hipStreamNonBlocking));
    hipStream_t st = ts.s;
    auto run_mode = [&](int mode, uint32_t skew, bool *ok) -> int {
        pl->mode = mode;
        uint32_t bad_total = 0;
        for (uint32_t round = 1; round <= 3; ++round) {
            CL_HIP(hipMemsetAsync(pl->probe_bad, 0, sizeof(uint32_t), st));
            hipLaunchKernelGGL(probe_warm_kernel, dim3(512), dim3(256), 0, st, pl->probe, (uint64_t)c->world * kProbeFloats, pl->probe);
            CL_HIP(hipStreamSynchronize(st));
            int b = barrier_host(c);                                  // everybody's caches are warm with the old content: peers may write now
            if (b != CLEORA_OK) return b;
            hipLaunchKernelGGL(probe_fill_kernel, dim3(1), dim3(256), 0, st, pl->probe + (uint64_t)c->rank * kProbeFloats, round, (uint32_t)c->rank);
            if ((b = peer_allgatherv_f32(c, pl->probe, offsets.data(), st)) != CLEORA_OK) return b;
            hipLaunchKernelGGL(probe_check_kernel, dim3((unsigned)c->world), dim3(256), 0, st, pl->probe, c->world, round, skew, pl->probe_bad);
            uint32_t bad = 0;
            CL_HIP(hipMemcpyAsync(&bad, pl->probe_bad, sizeof bad, hipMemcpyDeviceToHost, st));
            CL_HIP(hipStreamSynchronize(st));
            bad_total += bad;
            if ((b = peer_check(c)) != CLEORA_OK) return b;
            if ((b = barrier_host(c)) != CLEORA_OK) return b;        // nobody refills a slot a peer is still checking
        }
        // one verdict for all ranks
        if (bad_total) pl->shm->failed.store(1, std::memory_order_release);
        int b = barrier_host(c);
        if (b != CLEORA_OK) return b;
        *ok = pl->shm->failed.load(std::memory_order_acquire) == 0;
        if ((b = barrier_host(c)) != CLEORA_OK) return b;
        if (c->rank == 0) pl->shm->failed.store(0, std::memory_order_release);
        return barrier_host(c);
    };
    bool ok = false;
    if ((rc = run_mode(0, (flags & 1u) ? 1u : 0u, &ok)) != CLEORA_OK) { pl->mode = saved; return rc; }
    if (!ok) {
        if ((rc = run_mode(1, (flags & 2u) ? 1u : 0u, &ok)) != CLEORA_OK) { pl->mode = saved; return rc; }
        if (!ok) {
            pl->mode = saved;
            set_error("peer transport: data stored into a peer's buffer (and data loaded from it) did not arrive where the next kernel's loads see it: "
                      "the mappings are not coherent enough for the peer-direct all-gather on this node; use the RCCL all-gather (cleora_comm_set_allgather)");
            return CLEORA_E_RCCL;
        }
    }
    pl->selftested = true;
    return CLEORA_OK;
}

int peer_mode(const cleora_comm *c) { return c->peer ? c->peer->mode : -1; }

// the all-gather form, decided together by the callers (sharded.hip's first-use check switches every rank to PULL at once)
void peer_set_mode(cleora_comm *c, int mode) {
    if (c->peer) c->peer->mode = mode ? 1 : 0;
}

int peer_check(cleora_comm *c) {
    if (!c->peer || !c->peer->mailbox) return CLEORA_OK;
    uint64_t err = 0;
    CL_HIP(hipMemcpy(&err, &c->peer->mailbox->error, sizeof err, hipMemcpyDeviceToHost));
    if (err) {
        (void)hipMemset(&c->peer->mailbox->error, 0, sizeof err);       // reported once
        set_error("peer transport: rank " + std::to_string(c->rank) + " waited 60 s for rank " + std::to_string((int)((err >> 8) & 0xff)) +
                  " (channel " + std::to_string((int)(err & 0xff) - 1) + ", operation " + std::to_string((unsigned long long)(err >> 16)) + "): a peer died or the mappings are not coherent");
        return CLEORA_E_RCCL;
    }
    return CLEORA_OK;
}

}  // namespace cleora
