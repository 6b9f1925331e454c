// TEST INFRASTRUCTURE, not product: a stand-in for librccl.so that lets N processes SHARING ONE GPU rehearse the code path
// `bench.py --gpus N` takes (unast_comm_init with world > 1, the unique-id broadcast, unast_allreduce issued eagerly and by the
// stream-replay executor at marker nodes).  RCCL itself refuses two ranks on one device, so the collective is staged through the host:
// selected with UNAST_COMM_LIB=<this .so> (csrc/comm.cpp), never loaded otherwise.
//
// Exports the five symbols comm.cpp binds (ncclGetUniqueId, ncclCommInitRank, ncclAllReduce, ncclCommDestroy, ncclGetErrorString) with
// RCCL's stream semantics: ncclAllReduce returns at once and the reduction is ordered on `stream`.  Per piece of at most PIECE floats:
//   D2H copy of this rank's values into its slot of a POSIX shared-memory segment (named by the unique id; two slot sets, used
//   alternately) -> a host function (hipLaunchHostFunc) that announces arrival, waits until every rank has arrived at the same
//   collective (sequence numbers, so a rank that issues collectives in ANOTHER ORDER or of another size is caught: the counts are
//   compared and a mismatch is reported instead of reducing garbage), sums the slots in rank order into a private pinned buffer ->
//   H2D copy of the sums.  Every wait is bounded (30 s): on a time-out or a mismatch the error is latched, printed, and the buffer is
//   filled with NaN so that the test fails instead of hanging the box.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <atomic>
#include <limits>

namespace {

constexpr size_t PIECE = 1u << 21;          // floats per piece (8 MB)
constexpr int MAX_RANKS = 8;

struct Shared {
    std::atomic<long long> arrived[MAX_RANKS];     // sequence number of the newest collective piece rank r has copied into its slot
    std::atomic<long long> count[MAX_RANKS][2];    // its size, per slot set
    std::atomic<int> error;
    std::atomic<int> attached;
};

struct FakeComm {
    int rank, world;
    char name[64];
    Shared* sh;
    float* slots;               // [2 sets][world][PIECE], in the shared segment
    size_t seg_bytes;
    float* result;              // pinned, private
    long long seq;              // pieces issued so far (host side)
};

struct Job { FakeComm* c; long long seq; size_t n; };

double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

void host_reduce(void* arg) {
    Job* j = (Job*)arg;
    FakeComm* c = j->c;
    const int set = (int)(j->seq & 1);
    c->sh->count[c->rank][set].store((long long)j->n, std::memory_order_relaxed);
    c->sh->arrived[c->rank].store(j->seq, std::memory_order_release);
    const double t0 = now_s();
    bool ok = true;
    for (int r = 0; r < c->world && ok; ++r) {
        while (c->sh->arrived[r].load(std::memory_order_acquire) < j->seq) {
            if (c->sh->error.load() || now_s() - t0 > 30.0) { ok = false; break; }
            usleep(50);
        }
        if (ok && c->sh->count[r][set].load(std::memory_order_relaxed) != (long long)j->n) {
            fprintf(stderr, "fake_rccl: rank %d collective #%lld has %zu floats, rank %d has %lld there: the ranks issue different collectives\n",
                    c->rank, j->seq, j->n, r, c->sh->count[r][set].load());
            ok = false;
        }
    }
    if (!ok) {
        if (!c->sh->error.exchange(1)) fprintf(stderr, "fake_rccl: rank %d gave up at collective #%lld (time-out or mismatch)\n", c->rank, j->seq);
        for (size_t i = 0; i < j->n; ++i) c->result[i] = std::numeric_limits<float>::quiet_NaN();
    } else {
        const float* base = c->slots + (size_t)set * c->world * PIECE;
        for (size_t i = 0; i < j->n; ++i) {
            float s = 0.f;
            for (int r = 0; r < c->world; ++r) s += base[(size_t)r * PIECE + i];          // rank order: every rank forms the same sum
            c->result[i] = s;
        }
    }
    delete j;
}

}  // namespace

struct FakeId { char b[128]; };

extern "C" int ncclGetUniqueId(void* id128) {
    memset(id128, 0, 128);
    snprintf((char*)id128, 64, "/unast_fake_rccl_%d_%lld", (int)getpid(), (long long)(now_s() * 1e6));
    return 0;
}

extern "C" int ncclCommInitRank(void** comm, int world, FakeId id, int rank) {
    if (world < 1 || world > MAX_RANKS || rank < 0 || rank >= world) return 4;       // ncclInvalidArgument
    FakeComm* c = new FakeComm();
    c->rank = rank; c->world = world; c->seq = 0;
    memcpy(c->name, id.b, 63); c->name[63] = 0;
    c->seg_bytes = sizeof(Shared) + 4096 + (size_t)2 * world * PIECE * sizeof(float);
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->seg_bytes) != 0) { perror("fake_rccl: shm_open / ftruncate"); delete c; return 2; }
    void* p = mmap(nullptr, c->seg_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { perror("fake_rccl: mmap"); delete c; return 2; }
    c->sh = (Shared*)p;                                                              // (a fresh segment is zero-filled: all counters start at 0)
    c->slots = (float*)((char*)p + ((sizeof(Shared) + 4095) / 4096) * 4096);
    if (hipHostRegister(c->slots, (size_t)2 * world * PIECE * sizeof(float), hipHostRegisterDefault) != hipSuccess) { fprintf(stderr, "fake_rccl: hipHostRegister failed\n"); return 2; }
    if (hipHostMalloc((void**)&c->result, PIECE * sizeof(float), hipHostMallocDefault) != hipSuccess) return 2;
    if (c->sh->attached.fetch_add(1) + 1 == world) shm_unlink(c->name);              // the last rank to attach removes the name; the mappings live on
    *comm = c;
    return 0;
}

extern "C" int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) {
    FakeComm* c = (FakeComm*)comm;
    if (!c || dtype != 7 || op != 0 || send != recv) return 4;                       // float32 sum in place is all comm.cpp asks for
    float* buf = (float*)recv;
    for (size_t off = 0; off < count; off += PIECE) {
        const size_t n = count - off < PIECE ? count - off : PIECE;
        const long long seq = ++c->seq;
        float* slot = c->slots + ((size_t)(seq & 1) * c->world + c->rank) * PIECE;
        if (hipMemcpyAsync(slot, buf + off, n * sizeof(float), hipMemcpyDeviceToHost, stream) != hipSuccess) return 1;
        if (hipLaunchHostFunc(stream, host_reduce, new Job{c, seq, n}) != hipSuccess) return 1;
        if (hipMemcpyAsync(buf + off, c->result, n * sizeof(float), hipMemcpyHostToDevice, stream) != hipSuccess) return 1;
    }
    return 0;
}

extern "C" int ncclCommDestroy(void* comm) {
    FakeComm* c = (FakeComm*)comm;
    if (!c) return 0;
    (void)hipDeviceSynchronize();
    (void)hipHostUnregister(c->slots);
    (void)hipHostFree(c->result);
    munmap((void*)c->sh, c->seg_bytes);
    delete c;
    return 0;
}

extern "C" const char* ncclGetErrorString(int rc) {
    return rc == 0 ? "success" : rc == 4 ? "invalid argument (fake_rccl)" : "error (fake_rccl)";
}
