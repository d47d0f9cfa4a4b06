// shm_barrier.h — the barrier for the HOST threads of the ranks of one node (csrc/peer.hip: the bootstrap of the peer-direct
// transport), on two words of a POSIX shared-memory segment.  Plain C++ without the HIP runtime, so that the CPU suite can run it
// with as many processes as a node has GPUs (tests/test_shm_barrier_cpu.py) — the GPU box has one.
//
// Sense-reversing: the last rank to arrive resets the count and flips the shared sense; the others spin on the sense (yielding
// after a short while), bounded by a wall-clock budget — a rank that never arrives is an error, not a hang.  A fresh segment is
// zero-filled, so every rank starts with local sense 0.  The reference has no counterpart (pycleora is single-process).
#pragma once
#include <sched.h>

#include <atomic>
#include <chrono>
#include <cstdint>

namespace cleora {

struct ShmBarrier {
    std::atomic<uint32_t> count, sense;
};
static_assert(std::atomic<uint32_t>::is_always_lock_free, "the barrier words are shared between processes");

// true: every one of the `world` ranks has arrived; false: the budget ran out first
inline bool shm_barrier_wait(ShmBarrier *b, uint32_t world, uint32_t *local_sense, double budget_seconds) {
    const uint32_t my = *local_sense ^= 1u;
    if (b->count.fetch_add(1, std::memory_order_acq_rel) + 1 == world) {
        b->count.store(0, std::memory_order_relaxed);
        b->sense.store(my, std::memory_order_release);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (b->sense.load(std::memory_order_acquire) != my) {
        if (++spins > 200) sched_yield();
        if ((spins & 1023u) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > budget_seconds) return false;
    }
    return true;
}

}  // namespace cleora
