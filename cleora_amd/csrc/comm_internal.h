// comm_internal.h — the communicator behind cleora_comm_* (comm.hip: RCCL; peer.hip: the peer-direct transport) and what the
// multi-GPU loops of sharded.hip need from both.
#pragma once
#include <rccl/rccl.h>

#include "common.h"

namespace cleora {
struct PeerLayer;
}

struct cleora_comm {
    ncclComm_t comm = nullptr;          // nullptr: a local communicator (peer-direct transport only)
    int rank = 0, world = 1, device = 0;
    int allgather_algo = 0;   // CLEORA_ALLGATHER_*: 0 ncclAllGather / grouped broadcasts, 1 send/recv mesh, 2 peer-direct stores
    unsigned char id[CLEORA_COMM_ID_BYTES] = {0};
    cleora::PeerLayer *peer = nullptr;
    std::mutex mu;
};

namespace cleora {

// peer.hip — all collective over the communicator's ranks, all enqueue-only on `stream` except enable / register / unregister
int peer_enable(cleora_comm *c);                                   // bootstrap: shared-memory segment, mailboxes (collective, host-synchronous)
void peer_destroy(cleora_comm *c);
int peer_register(cleora_comm *c, void *buf, uint64_t bytes);     // map `buf` of every rank into every rank (collective, host-synchronous)
int peer_unregister(cleora_comm *c, void *buf);
int peer_allgatherv_f32(cleora_comm *c, float *buf, const uint64_t *offsets, hipStream_t stream);
int peer_allreduce(cleora_comm *c, void *buf, uint64_t n, bool f64, hipStream_t stream);
int peer_broadcast(cleora_comm *c, void *buf, uint64_t bytes, int root, hipStream_t stream);
int peer_selftest(cleora_comm *c, uint32_t flags);                 // data-visibility self-test; may switch every rank to the PULL all-gather (collective, host-synchronous)
int peer_mode(const cleora_comm *c);                               // 0 PUSH, 1 PULL, -1 no peer transport
void peer_set_mode(cleora_comm *c, int mode);
int peer_check(cleora_comm *c);                                    // CLEORA_E_RCCL if a wait ever timed out on this rank (reads the mailbox)
int peer_host_barrier(cleora_comm *c);                             // the ranks' host threads meet (shared memory; bounded wait)
void peer_abandon(cleora_comm *c);                                 // its host barriers stop waiting (teardown of a group that never became complete)

}  // namespace cleora
