// stager.hip — host <-> device copies for the host-pointer entry points at PCIe speed.
//
// What the reference's unmodified embed() loop calls 40 times is `left_markov_propagate(numpy) -> numpy`
// (src/lib.rs:29-47): pageable host memory in, pageable host memory out.  A plain hipMemcpy on pageable memory moves
// ~19 GB/s (measured round 1: 54.7 ms for 0.5M x 256 in + out).  Here the copy is a two-stage pipeline over a ring of
// pinned chunks: worker threads memcpy user memory <-> pinned chunk while the DMA engine moves the previous chunk over
// PCIe (hipMemcpyAsync on a stream of its own), so both stages run at their own speed and the slower one (PCIe,
// ~55 GB/s pinned) sets the rate.  The ring and the workers are created on first use and live for the process.
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "common.h"

namespace cleora {
namespace {

class CopyPool {   // N persistent workers; run() = a parallel memcpy, the caller's thread takes a share too
public:
    explicit CopyPool(int n) {
        for (int i = 0; i < n; ++i) workers_.emplace_back([this, i] { loop(i); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lock(mu_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void run(void *dst, const void *src, size_t bytes) {
        const int parts = (int)workers_.size() + 1;
        if (bytes < (1u << 20) || parts == 1) {
            std::memcpy(dst, src, bytes);
            return;
        }
        {
            std::lock_guard<std::mutex> lock(mu_);
            dst_ = static_cast<char *>(dst);
            src_ = static_cast<const char *>(src);
            bytes_ = bytes;
            pending_ = (int)workers_.size();
            ++gen_;
        }
        cv_.notify_all();
        slice(parts - 1, parts);
        std::unique_lock<std::mutex> lock(mu_);
        done_.wait(lock, [this] { return pending_ == 0; });
    }

private:
    void slice(int part, int parts) {
        const size_t per = ((bytes_ + parts - 1) / parts + 4095) & ~size_t(4095);
        const size_t b = per * part;
        if (b >= bytes_) return;
        const size_t len = b + per < bytes_ ? per : bytes_ - b;
        std::memcpy(dst_ + b, src_ + b, len);
    }
    void loop(int id) {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lock(mu_);
                cv_.wait(lock, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
            }
            slice(id, (int)workers_.size() + 1);
            std::lock_guard<std::mutex> lock(mu_);
            if (--pending_ == 0) done_.notify_one();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    char *dst_ = nullptr;
    const char *src_ = nullptr;
    size_t bytes_ = 0;
    int pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

constexpr size_t kChunk = 32u << 20;
constexpr int kSlots = 4;

struct Stager {
    std::mutex mu;                  // one staged copy at a time per device (they would only share the link anyway)
    void *slot[kSlots] = {};
    hipEvent_t ev[kSlots] = {};
    hipStream_t stream = nullptr;
    CopyPool *pool = nullptr;
    int device = -1;
    bool ok = false;

    int init() {
        int dev = 0;
        CL_HIP(hipGetDevice(&dev));
        if (ok && dev == device) return CLEORA_OK;
        if (ok) {   // another device: the stream belongs to the old one
            (void)hipStreamDestroy(stream);
            for (int i = 0; i < kSlots; ++i) (void)hipEventDestroy(ev[i]);
            ok = false;
        }
        for (int i = 0; i < kSlots; ++i)
            if (!slot[i]) CL_HIP(hipHostMalloc(&slot[i], kChunk, hipHostMallocDefault));
        CL_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        for (int i = 0; i < kSlots; ++i) CL_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        if (!pool) {
            int n = 8;
            if (const char *e = std::getenv("CLEORA_COPY_THREADS")) n = std::atoi(e);
            const int hw = (int)std::thread::hardware_concurrency();
            if (hw > 0 && n > hw) n = hw;
            if (n < 1) n = 1;
            pool = new CopyPool(n - 1);
        }
        device = dev;
        ok = true;
        return CLEORA_OK;
    }
};

// one pipeline per device: the threads of a one-process multi-device job (multi.hip) move their shards over their own PCIe links
Stager &stager() {
    static std::mutex mu;
    static std::vector<Stager *> per_device;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    if ((size_t)dev >= per_device.size()) per_device.resize((size_t)dev + 1, nullptr);
    if (!per_device[(size_t)dev]) per_device[(size_t)dev] = new Stager();
    return *per_device[(size_t)dev];
}

}  // namespace

// Everything enqueued on `after` (may be the null stream) before the call is complete before the first byte moves.
int staged_h2d(void *dst_dev, const void *src_host, uint64_t bytes, hipStream_t after) {
    if (bytes == 0) return CLEORA_OK;
    if (bytes < (4u << 20)) {   // small: one plain copy is quicker than waking the pipeline
        CL_HIP(hipStreamSynchronize(after));
        CL_HIP(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
        return CLEORA_OK;
    }
    Stager &s = stager();
    std::lock_guard<std::mutex> lock(s.mu);
    int rc = s.init();
    if (rc != CLEORA_OK) return rc;
    CL_HIP(hipStreamSynchronize(after));
    const char *src = static_cast<const char *>(src_host);
    char *dst = static_cast<char *>(dst_dev);
    uint64_t off = 0;
    for (uint64_t i = 0; off < bytes; ++i, off += kChunk) {
        const int k = (int)(i % kSlots);
        const size_t len = bytes - off < kChunk ? (size_t)(bytes - off) : kChunk;
        if (i >= kSlots) CL_HIP(hipEventSynchronize(s.ev[k]));           // the DMA that last read this slot is done
        s.pool->run(s.slot[k], src + off, len);                           // user memory -> pinned, all workers
        CL_HIP(hipMemcpyAsync(dst + off, s.slot[k], len, hipMemcpyHostToDevice, s.stream));
        CL_HIP(hipEventRecord(s.ev[k], s.stream));
    }
    CL_HIP(hipStreamSynchronize(s.stream));
    return CLEORA_OK;
}

int staged_d2h(void *dst_host, const void *src_dev, uint64_t bytes, hipStream_t after) {
    if (bytes == 0) return CLEORA_OK;
    if (bytes < (4u << 20)) {
        CL_HIP(hipStreamSynchronize(after));
        CL_HIP(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
        return CLEORA_OK;
    }
    Stager &s = stager();
    std::lock_guard<std::mutex> lock(s.mu);
    int rc = s.init();
    if (rc != CLEORA_OK) return rc;
    CL_HIP(hipStreamSynchronize(after));
    char *dst = static_cast<char *>(dst_host);
    const char *src = static_cast<const char *>(src_dev);
    const uint64_t chunks = (bytes + kChunk - 1) / kChunk;
    auto issue = [&](uint64_t i) -> int {
        const uint64_t off = i * kChunk;
        const size_t len = bytes - off < kChunk ? (size_t)(bytes - off) : kChunk;
        const int k = (int)(i % kSlots);
        CL_HIP(hipMemcpyAsync(s.slot[k], src + off, len, hipMemcpyDeviceToHost, s.stream));
        CL_HIP(hipEventRecord(s.ev[k], s.stream));
        return CLEORA_OK;
    };
    for (uint64_t i = 0; i < chunks && i < (uint64_t)kSlots; ++i)
        if ((rc = issue(i)) != CLEORA_OK) return rc;
    for (uint64_t i = 0; i < chunks; ++i) {
        const uint64_t off = i * kChunk;
        const size_t len = bytes - off < kChunk ? (size_t)(bytes - off) : kChunk;
        const int k = (int)(i % kSlots);
        CL_HIP(hipEventSynchronize(s.ev[k]));
        s.pool->run(dst + off, s.slot[k], len);                           // pinned -> user memory while later chunks fly
        if (i + kSlots < chunks && (rc = issue(i + kSlots)) != CLEORA_OK) return rc;
    }
    return CLEORA_OK;
}

}  // namespace cleora
