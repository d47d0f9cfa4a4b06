// CPU stress of cleora_amd/csrc/shm_barrier.h (the bootstrap barrier of the peer-direct transport) with as many PROCESSES as a
// node has GPUs: shm_barrier_stress WORLD ROUNDS [absent_rank].  Every rank writes its round number into its slot, passes the
// barrier, and checks that every slot holds that round (nobody may be a round behind or ahead), then passes a second barrier
// before the slots are overwritten.  With an absent rank the others must come back with "timeout" after the budget, not hang.
// Exit code 0 = all ranks agree; prints one line.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../cleora_amd/csrc/shm_barrier.h"

struct Segment {
    cleora::ShmBarrier barrier;
    std::atomic<uint32_t> slot[64];
    std::atomic<uint32_t> bad;
};

int main(int argc, char **argv) {
    const int world = argc > 1 ? atoi(argv[1]) : 8;
    const unsigned rounds = argc > 2 ? (unsigned)atoi(argv[2]) : 100000u;
    const int absent = argc > 3 ? atoi(argv[3]) : -1;
    const double budget = absent >= 0 ? 0.5 : 60.0;
    char name[64];
    snprintf(name, sizeof name, "/cleora.test.%d", (int)getpid());
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Segment)) != 0) { perror("shm"); return 2; }
    Segment *s = static_cast<Segment *>(mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
    shm_unlink(name);
    if (s == MAP_FAILED) { perror("mmap"); return 2; }
    int timeouts = 0;
    for (int r = 0; r < world; ++r) {
        if (r == absent) continue;
        const pid_t pid = fork();
        if (pid == 0) {
            uint32_t sense = 0;
            for (unsigned k = 1; k <= rounds; ++k) {
                s->slot[r].store(k, std::memory_order_relaxed);
                if (!cleora::shm_barrier_wait(&s->barrier, (uint32_t)world, &sense, budget)) _exit(3);
                for (int q = 0; q < world; ++q)
                    if (s->slot[q].load(std::memory_order_relaxed) != k) s->bad.fetch_add(1);
                if (!cleora::shm_barrier_wait(&s->barrier, (uint32_t)world, &sense, budget)) _exit(3);
            }
            _exit(0);
        }
    }
    int failed = 0;
    for (int r = 0; r < world - (absent >= 0 ? 1 : 0); ++r) {
        int st = 0;
        wait(&st);
        if (WIFEXITED(st) && WEXITSTATUS(st) == 3) ++timeouts;
        else if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) ++failed;
    }
    printf("{\"world\": %d, \"rounds\": %u, \"disagreements\": %u, \"timeouts\": %d, \"crashed\": %d}\n", world, rounds, s->bad.load(), timeouts, failed);
    if (absent >= 0) return (timeouts == world - 1 && failed == 0) ? 0 : 1;
    return (s->bad.load() == 0 && timeouts == 0 && failed == 0) ? 0 : 1;
}
