// shm_rccl.cpp -- TEST DOUBLE of the eight RCCL entry points carma_shard.hip binds (ncclGetUniqueId, ncclCommInitRank,
// ncclCommDestroy, ncclGetErrorString, ncclGroupStart, ncclGroupEnd, ncclSend, ncclRecv), with a shared-memory transport
// between processes of one box.  Test infrastructure only (tests/test_gpu_ladder_shard.py builds it and points the
// library at it through CARMA_RCCL_LIB): RCCL refuses two ranks on one device, so on a one-GPU box the code paths of
// carma_pt_iterate_sharded for nranks > 1 -- who sends to whom, in which order, the boundary self-check -- could
// otherwise only run with every block on rank 0.  Same semantics as the real thing where the library depends on them:
// stream ordered, sends and receives between a pair of ranks matched in issue order, operations of a group issued
// together (sends first, so that two ranks -- or a rank with itself -- that send to each other never wait in a circle).
//
//   hipcc -O1 -shared -fPIC -o libshm_rccl.so shm_rccl.cpp -lrt -lpthread
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

constexpr int NSLOT = 16;                 // messages a channel holds before its sender waits
constexpr size_t SLOT_BYTES = 64 * 1024;  // largest message (the boundary chains of a few hundred replicas)
constexpr double TIMEOUT_S = 120.0;

struct Channel {                          // one per ordered pair (src, dst)
    std::atomic<unsigned long long> head;  // messages written
    std::atomic<unsigned long long> tail;  // messages consumed
    size_t bytes[NSLOT];
    unsigned char data[NSLOT][SLOT_BYTES];
};

struct Header {
    std::atomic<int> ready;
};

struct ShmComm {
    int rank = 0, nranks = 1;
    void* base = nullptr;
    size_t map_bytes = 0;
    char name[64] = {0};
    std::vector<void*> garbage;           // pinned staging buffers of operations already enqueued
    int inflight = 0;
    Channel* chan(int src, int dst) const
    {
        return reinterpret_cast<Channel*>(static_cast<unsigned char*>(base) + 4096) + (size_t)src * nranks + dst;
    }
};

struct Op {
    bool send;
    void* buf;
    size_t bytes;
    int peer;
    ShmComm* cm;
    hipStream_t st;
};

thread_local int group_depth = 0;
thread_local std::vector<Op> pending;

struct CbArg {
    Channel* ch;
    void* stage;
    size_t bytes;
};

[[noreturn]] void die(const char* what)
{
    std::fprintf(stderr, "shm_rccl (test transport): %s\n", what);
    std::fflush(stderr);
    _exit(3);
}

template <class F>
void wait_until(F ok, const char* what)
{
    const auto t0 = std::chrono::steady_clock::now();
    while (!ok()) {
        sched_yield();
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > TIMEOUT_S) die(what);
    }
}

void send_cb(void* p)
{
    CbArg* a = static_cast<CbArg*>(p);
    Channel* ch = a->ch;
    wait_until([&] { return ch->head.load(std::memory_order_relaxed) - ch->tail.load(std::memory_order_acquire) < NSLOT; },
               "a send waited two minutes for room in its channel");
    const unsigned long long h = ch->head.load(std::memory_order_relaxed);
    ch->bytes[h % NSLOT] = a->bytes;
    std::memcpy(ch->data[h % NSLOT], a->stage, a->bytes);
    ch->head.store(h + 1, std::memory_order_release);
    delete a;
}

void recv_cb(void* p)
{
    CbArg* a = static_cast<CbArg*>(p);
    Channel* ch = a->ch;
    wait_until([&] { return ch->head.load(std::memory_order_acquire) > ch->tail.load(std::memory_order_relaxed); },
               "a receive waited two minutes for its message");
    const unsigned long long t = ch->tail.load(std::memory_order_relaxed);
    if (ch->bytes[t % NSLOT] != a->bytes) die("a receive met a message of another size: sends and receives are not matched");
    std::memcpy(a->stage, ch->data[t % NSLOT], a->bytes);
    ch->tail.store(t + 1, std::memory_order_release);
    delete a;
}

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
        case ncclInt8:
        case ncclUint8: return 1;
        case ncclFloat16: return 2;
        case ncclInt32:
        case ncclUint32:
        case ncclFloat32: return 4;
        case ncclInt64:
        case ncclUint64:
        case ncclFloat64: return 8;
        default: return 0;
    }
}

ncclResult_t run(std::vector<Op>& ops)
{
    // sends first: the messages are then in their channels (or will be, independently of what this rank receives)
    for (int pass = 0; pass < 2; pass++) {
        for (Op& o : ops) {
            if (o.send != (pass == 0)) continue;
            if (o.bytes > SLOT_BYTES || o.peer < 0 || o.peer >= o.cm->nranks) return ncclInvalidArgument;
            void* stage = nullptr;
            if (hipHostMalloc(&stage, o.bytes ? o.bytes : 1, hipHostMallocDefault) != hipSuccess) return ncclUnhandledCudaError;
            o.cm->garbage.push_back(stage);
            if (o.send) {
                if (hipMemcpyAsync(stage, o.buf, o.bytes, hipMemcpyDeviceToHost, o.st) != hipSuccess) return ncclUnhandledCudaError;
                if (hipLaunchHostFunc(o.st, send_cb, new CbArg{o.cm->chan(o.cm->rank, o.peer), stage, o.bytes}) != hipSuccess)
                    return ncclUnhandledCudaError;
            } else {
                if (hipLaunchHostFunc(o.st, recv_cb, new CbArg{o.cm->chan(o.peer, o.cm->rank), stage, o.bytes}) != hipSuccess)
                    return ncclUnhandledCudaError;
                if (hipMemcpyAsync(o.buf, stage, o.bytes, hipMemcpyHostToDevice, o.st) != hipSuccess) return ncclUnhandledCudaError;
            }
            o.cm->inflight++;
        }
    }
    // staging buffers are recycled once the stream has drained (a test transport: no need to be clever)
    for (Op& o : ops) {
        ShmComm* cm = o.cm;
        if (cm->inflight > 48) {
            if (hipStreamSynchronize(o.st) != hipSuccess) return ncclUnhandledCudaError;
            for (void* g : cm->garbage) (void)hipHostFree(g);
            cm->garbage.clear();
            cm->inflight = 0;
        }
    }
    return ncclSuccess;
}

ncclResult_t enqueue(bool send, const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t st)
{
    ShmComm* cm = reinterpret_cast<ShmComm*>(comm);
    if (!cm || !buf || type_bytes(type) == 0) return ncclInvalidArgument;
    pending.push_back(Op{send, const_cast<void*>(buf), count * type_bytes(type), peer, cm, st});
    if (group_depth == 0) {
        std::vector<Op> ops;
        ops.swap(pending);
        return run(ops);
    }
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    if (!id) return ncclInvalidArgument;
    std::memset(id->internal, 0, NCCL_UNIQUE_ID_BYTES);
    const unsigned long long a = (unsigned long long)getpid(),
                             b = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
    std::snprintf(id->internal, NCCL_UNIQUE_ID_BYTES, "/carma_shm_%llx_%llx", a, b);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    ShmComm* cm = new ShmComm();
    cm->rank = rank;
    cm->nranks = nranks;
    std::memcpy(cm->name, id.internal, sizeof cm->name - 1);
    cm->map_bytes = 4096 + sizeof(Channel) * (size_t)nranks * nranks;
    const int fd = shm_open(cm->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)cm->map_bytes) != 0) {
        delete cm;
        return ncclSystemError;
    }
    cm->base = mmap(nullptr, cm->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);   // (a fresh object reads as zeros)
    close(fd);
    if (cm->base == MAP_FAILED) {
        delete cm;
        return ncclSystemError;
    }
    Header* h = static_cast<Header*>(cm->base);
    h->ready.fetch_add(1, std::memory_order_acq_rel);
    wait_until([&] { return h->ready.load(std::memory_order_acquire) >= nranks; }, "not every rank reached ncclCommInitRank");
    // every rank has the object mapped: the name can go (nothing is left behind if a process dies later)
    if (rank == 0) {
        usleep(20000);
        shm_unlink(cm->name);
    }
    *comm = reinterpret_cast<ncclComm_t>(cm);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    ShmComm* cm = reinterpret_cast<ShmComm*>(comm);
    if (!cm) return ncclSuccess;
    (void)hipDeviceSynchronize();
    for (void* g : cm->garbage) (void)hipHostFree(g);
    munmap(cm->base, cm->map_bytes);
    delete cm;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled HIP error (shared-memory test transport)";
        case ncclSystemError: return "system error (shared-memory test transport)";
        case ncclInvalidArgument: return "invalid argument (shared-memory test transport)";
        default: return "error (shared-memory test transport)";
    }
}

ncclResult_t ncclGroupStart()
{
    group_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (group_depth <= 0) return ncclInvalidUsage;
    if (--group_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(pending);
    return run(ops);
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    return enqueue(true, sendbuff, count, datatype, peer, comm, stream);
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    return enqueue(false, recvbuff, count, datatype, peer, comm, stream);
}

}  // extern "C"
