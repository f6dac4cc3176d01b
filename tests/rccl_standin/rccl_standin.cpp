// TEST-ONLY stand-in for the thirteen RCCL entry points libmocha_hip.so resolves at run time (struct Rccl in
// mocha_sigasia2023_amd/csrc/mocha_api.cpp), so that the multi-rank C-ABI path - mocha_comm_init, mocha_bank_broadcast's
// scatter + all-gather + tail broadcast, the receiving side's allocation branch - runs with 2..8 ranks on a box that has ONE
// GPU.  Real RCCL refuses two ranks on one device; here every rank is an ordinary process on the same GPU and the "wire" is
// a POSIX shared-memory segment: device -> host staging -> device with hipMemcpy.  Selected with mocha_set_rccl_library /
// MOCHA_RCCL_LIBRARY.  It is not a product component: nothing under mocha_sigasia2023_amd/ refers to it, and it makes no
// performance claim.
//
// Semantics kept from RCCL: the call signatures (rccl.h), rank / count queries, in-place all-gather, point-to-point calls
// taking effect at ncclGroupEnd, matching of sends and receives by (peer, byte count), error codes instead of hangs
// (every wait is bounded).  Simplifications: every call is host-synchronous (it synchronises `stream` first, so stream order
// is respected trivially); point-to-point calls are only supported inside a group that EVERY rank of the communicator
// enters (mocha_bank_broadcast's pattern); one communicator per process.
//
// Fault injection for the library's error paths (tests/test_multirank_standin.py):
//   MOCHA_STANDIN_FAIL_SEND=<rank>   ncclSend on that rank returns ncclInternalError (once)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr int MAXW = 8;
constexpr size_t SLOT = 4u << 20;                 // staging bytes per rank
constexpr double TIMEOUT_S = 120.0;

struct Shm {
    std::atomic<int> arrived;
    std::atomic<int> sense;
    std::atomic<int> joined;
    std::atomic<int> error;                       // sticky: a rank saw a usage error inside a collective section
    long long announce[MAXW][MAXW];               // bytes rank s sends to rank d in the current group
    alignas(4096) unsigned char stage[MAXW][SLOT];
};

struct Comm {
    Shm* shm = nullptr;
    int rank = 0, world = 1;
    int local_sense = 0;
    char name[64] = {0};
};

struct P2P { bool send; void* buf; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
thread_local int g_depth = 0;
thread_local std::vector<P2P> g_queue;
bool g_fail_send_armed = true;
Comm* g_comm = nullptr;                            // the process's communicator (one per process): an EMPTY group still runs the
                                                  // collective schedule on it, e.g. after an injected ncclSend failure

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

bool barrier(Comm* c) {
    Shm* s = c->shm;
    const int sense = (c->local_sense ^= 1);
    if (s->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
        s->arrived.store(0, std::memory_order_relaxed);
        s->sense.store(sense, std::memory_order_release);
        return true;
    }
    const double t0 = now();
    while (s->sense.load(std::memory_order_acquire) != sense) {
        sched_yield();
        if (now() - t0 > TIMEOUT_S) return false;
    }
    return true;
}

size_t dtype_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

#define HIPOK(e) do { if ((e) != hipSuccess) return ncclUnhandledCudaError; } while (0)
#define BAR(c) do { if (!barrier(c)) return ncclSystemError; } while (0)

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id->internal, 0, NCCL_UNIQUE_ID_BYTES);
    snprintf(id->internal, 64, "/mocha_rccl_standin_%d_%llx", (int)getpid(), (unsigned long long)(now() * 1e6));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > MAXW || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm* c = new Comm();
    c->rank = rank; c->world = nranks;
    memcpy(c->name, id.internal, 63);
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { delete c; return ncclSystemError; }
    if (ftruncate(fd, sizeof(Shm)) != 0) { close(fd); delete c; return ncclSystemError; }       // fresh pages are zero: barrier state starts clean
    void* p = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->shm = (Shm*)p;
    c->shm->joined.fetch_add(1);
    const double t0 = now();
    while (c->shm->joined.load() < nranks) {
        sched_yield();
        if (now() - t0 > TIMEOUT_S) { munmap(p, sizeof(Shm)); delete c; return ncclSystemError; }
    }
    if (!barrier(c)) { munmap(p, sizeof(Shm)); delete c; return ncclSystemError; }
    if (rank == 0) shm_unlink(c->name);                   // everyone has it mapped: the name can go (no leak if a rank dies later)
    *comm = (ncclComm_t)c;
    g_comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = (Comm*)comm;
    if (!c) return ncclInvalidArgument;
    if (g_comm == c) g_comm = nullptr;
    munmap(c->shm, sizeof(Shm));
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = ((Comm*)comm)->world;
    return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) {
    if (!comm || !rank) return ncclInvalidArgument;
    *rank = ((Comm*)comm)->rank;
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm,
                           hipStream_t stream) {
    Comm* c = (Comm*)comm;
    const size_t es = dtype_bytes(datatype);
    if (!c || !es || root < 0 || root >= c->world || g_depth) return ncclInvalidArgument;
    HIPOK(hipStreamSynchronize(stream));
    const size_t bytes = count * es, cap = SLOT * MAXW;
    unsigned char* st = &c->shm->stage[0][0];
    for (size_t off = 0; off < bytes; off += cap) {
        const size_t n = bytes - off < cap ? bytes - off : cap;
        if (c->rank == root) HIPOK(hipMemcpy(st, (const char*)sendbuff + off, n, hipMemcpyDeviceToHost));
        BAR(c);
        if (c->rank != root) HIPOK(hipMemcpy((char*)recvbuff + off, st, n, hipMemcpyHostToDevice));
        else if (sendbuff != recvbuff) HIPOK(hipMemcpy((char*)recvbuff + off, (const char*)sendbuff + off, n, hipMemcpyDeviceToDevice));
        BAR(c);
    }
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
    Comm* c = (Comm*)comm;
    const size_t es = dtype_bytes(datatype);
    if (!c || !es || g_depth) return ncclInvalidArgument;
    HIPOK(hipStreamSynchronize(stream));
    const size_t bytes = sendcount * es;
    const bool in_place = (const char*)sendbuff == (char*)recvbuff + (size_t)c->rank * bytes;
    for (size_t off = 0; off < bytes; off += SLOT) {
        const size_t n = bytes - off < SLOT ? bytes - off : SLOT;
        HIPOK(hipMemcpy(c->shm->stage[c->rank], (const char*)sendbuff + off, n, hipMemcpyDeviceToHost));
        BAR(c);
        for (int r = 0; r < c->world; ++r) {
            if (r == c->rank && in_place) continue;
            HIPOK(hipMemcpy((char*)recvbuff + (size_t)r * bytes + off, c->shm->stage[r], n, hipMemcpyHostToDevice));
        }
        BAR(c);
    }
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }

static ncclResult_t run_group(std::vector<P2P>& ops) {
    Comm* c = ops.empty() ? g_comm : ops[0].comm;
    if (!c) return ncclSuccess;                           // no communicator yet: nothing collective to do
    for (auto& o : ops) if (o.comm != c) return ncclInvalidUsage;
    Shm* s = c->shm;
    for (auto& o : ops) HIPOK(hipStreamSynchronize(o.stream));
    for (int d = 0; d < c->world; ++d) s->announce[c->rank][d] = 0;
    for (auto& o : ops) if (o.send) s->announce[c->rank][o.peer] = (long long)o.bytes;
    BAR(c);
    bool bad = false;
    for (int src = 0; src < c->world; ++src)
        for (int dst = 0; dst < c->world; ++dst) {
            const size_t bytes = (size_t)s->announce[src][dst];
            if (src == dst || bytes == 0) continue;
            const P2P* mine = nullptr;
            if (c->rank == src) for (auto& o : ops) if (o.send && o.peer == dst) mine = &o;
            if (c->rank == dst) {
                for (auto& o : ops) if (!o.send && o.peer == src) mine = &o;
                if (!mine || mine->bytes != bytes) { bad = true; s->error.store(1); mine = nullptr; }     // unmatched send: keep the barrier schedule
            }
            const size_t cap = SLOT * MAXW;
            unsigned char* st = &s->stage[0][0];
            for (size_t off = 0; off < bytes; off += cap) {
                const size_t n = bytes - off < cap ? bytes - off : cap;
                if (c->rank == src && mine) HIPOK(hipMemcpy(st, (const char*)mine->buf + off, n, hipMemcpyDeviceToHost));
                BAR(c);
                if (c->rank == dst && mine) HIPOK(hipMemcpy((char*)mine->buf + off, st, n, hipMemcpyHostToDevice));
                BAR(c);
            }
        }
    // a receive nobody sends to is a usage error too (the real library would hang)
    for (auto& o : ops) if (!o.send && s->announce[o.peer][c->rank] == 0) { bad = true; s->error.store(1); }
    BAR(c);
    const bool any = s->error.load() != 0;
    BAR(c);
    if (c->rank == 0) s->error.store(0);
    BAR(c);
    return (bad || any) ? ncclInvalidUsage : ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    std::vector<P2P> ops;
    ops.swap(g_queue);
    return run_group(ops);
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    const size_t es = dtype_bytes(datatype);
    if (!c || !es || peer < 0 || peer >= c->world || peer == c->rank) return ncclInvalidArgument;
    if (!g_depth) return ncclInvalidUsage;                // stand-in contract: point-to-point only inside a group
    const char* f = getenv("MOCHA_STANDIN_FAIL_SEND");
    if (f && g_fail_send_armed && atoi(f) == c->rank) { g_fail_send_armed = false; return ncclInternalError; }
    g_queue.push_back(P2P{true, const_cast<void*>(sendbuff), count * es, peer, c, stream});
    return ncclSuccess;
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    const size_t es = dtype_bytes(datatype);
    if (!c || !es || peer < 0 || peer >= c->world || peer == c->rank) return ncclInvalidArgument;
    if (!g_depth) return ncclInvalidUsage;
    g_queue.push_back(P2P{false, recvbuff, count * es, peer, c, stream});
    return ncclSuccess;
}

// a version code no real RCCL carries (0.0.1): a bench line that names it ran on the stand-in
ncclResult_t ncclGetVersion(int* version) {
    if (!version) return ncclInvalidArgument;
    *version = 1;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled HIP error (stand-in)";
        case ncclSystemError: return "system error or peer timeout (stand-in)";
        case ncclInternalError: return "internal error (stand-in, injected)";
        case ncclInvalidArgument: return "invalid argument (stand-in)";
        case ncclInvalidUsage: return "invalid usage (stand-in): unmatched point-to-point call";
        default: return "unknown result code (stand-in)";
    }
}

}  // extern "C"
