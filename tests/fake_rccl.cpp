// fake_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl.so that lets N PROCESSES SHARING ONE GPU (or none) run the rank != 0
// paths of csrc/smr_comm.cpp (fill_neutral, the local_ops offset rule, gather -> ncclAllReduce -> scatter of strided destinations)
// on the single MI355X a builder has.  It exports exactly the entry points libstrided_hip dlopens -- ncclGetUniqueId, ncclCommInitRank,
// ncclCommDestroy, ncclAllReduce, ncclGetErrorString, ncclCommCount, ncclCommUserRank -- and all-reduces through a POSIX
// shared-memory segment: every rank copies its send buffer device -> shm slot, a sense-reversing barrier, every rank folds the slots
// IN RANK ORDER (so all ranks compute bit-identical results, like a ring all-reduce's fixed order), copies the result back to its
// device buffer, second barrier.  Selected with SMR_RCCL_LIB=<path to libfake_rccl.so>; never loaded by the product otherwise.
//
// Build: hipcc -O2 -std=c++17 -shared -fPIC tests/fake_rccl.cpp -o tests/libfake_rccl.so -lrt   (done by __graft_entry__.build())
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {
constexpr size_t SLOT = 4u << 20;  // bytes per rank per round; larger messages go in rounds
constexpr int MAXR = 16;
struct Shared {
    std::atomic<int> arrived;
    std::atomic<int> sense;
    std::atomic<int> attached;
    int nranks;
    alignas(64) unsigned char slots[1];  // nranks * SLOT
};
struct Comm {
    int nranks, rank, sense;
    Shared* sh;
    size_t bytes;
    char name[64];
    unsigned char* bounce;  // pinned host staging
};

const char* g_err = "fake rccl: ok";

bool barrier(Comm* c) {
    c->sense ^= 1;
    const int me = c->sh->arrived.fetch_add(1, std::memory_order_acq_rel) + 1;
    if (me == c->nranks) {
        c->sh->arrived.store(0, std::memory_order_relaxed);
        c->sh->sense.store(c->sense, std::memory_order_release);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (c->sh->sense.load(std::memory_order_acquire) != c->sense) {
        std::this_thread::yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
            g_err = "fake rccl: a rank did not reach the all-reduce within 120 s";
            return false;
        }
    }
    return true;
}

template <class T> void fold(T* acc, const T* x, size_t n, ncclRedOp_t op) {
    switch (op) {
        case ncclSum: for (size_t i = 0; i < n; ++i) acc[i] = (T)(acc[i] + x[i]); break;
        case ncclProd: for (size_t i = 0; i < n; ++i) acc[i] = (T)(acc[i] * x[i]); break;
        case ncclMin: for (size_t i = 0; i < n; ++i) acc[i] = x[i] < acc[i] ? x[i] : acc[i]; break;
        case ncclMax: for (size_t i = 0; i < n; ++i) acc[i] = acc[i] < x[i] ? x[i] : acc[i]; break;
        default: break;
    }
}
// integer sums / products wrap (unsigned arithmetic on the bit patterns), like the device's
template <class S, class U> void fold_int(S* acc, const S* x, size_t n, ncclRedOp_t op) {
    if (op == ncclSum) {
        for (size_t i = 0; i < n; ++i) acc[i] = (S)((U)acc[i] + (U)x[i]);
    } else if (op == ncclProd) {
        for (size_t i = 0; i < n; ++i) acc[i] = (S)((U)acc[i] * (U)x[i]);
    } else {
        fold<S>(acc, x, n, op);
    }
}
size_t size_of(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}
void fold_any(void* acc, const void* x, size_t n, ncclDataType_t t, ncclRedOp_t op) {
    switch (t) {
        case ncclInt8: fold_int<int8_t, uint8_t>((int8_t*)acc, (const int8_t*)x, n, op); break;
        case ncclUint8: fold_int<uint8_t, uint8_t>((uint8_t*)acc, (const uint8_t*)x, n, op); break;
        case ncclInt32: fold_int<int32_t, uint32_t>((int32_t*)acc, (const int32_t*)x, n, op); break;
        case ncclUint32: fold_int<uint32_t, uint32_t>((uint32_t*)acc, (const uint32_t*)x, n, op); break;
        case ncclInt64: fold_int<int64_t, uint64_t>((int64_t*)acc, (const int64_t*)x, n, op); break;
        case ncclUint64: fold_int<uint64_t, uint64_t>((uint64_t*)acc, (const uint64_t*)x, n, op); break;
        case ncclFloat32: fold<float>((float*)acc, (const float*)x, n, op); break;
        case ncclFloat64: fold<double>((double*)acc, (const double*)x, n, op); break;
        default: break;
    }
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::memset(id->internal, 0, NCCL_UNIQUE_ID_BYTES);
    const auto t = std::chrono::steady_clock::now().time_since_epoch().count();
    std::snprintf(id->internal, NCCL_UNIQUE_ID_BYTES, "/smr_fake_rccl_%d_%llx", (int)getpid(), (unsigned long long)t);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks || id.internal[0] != '/') {
        g_err = "fake rccl: bad arguments to ncclCommInitRank";
        return ncclInvalidArgument;
    }
    Comm* c = new Comm();
    c->nranks = nranks;
    c->rank = rank;
    c->sense = 0;
    std::snprintf(c->name, sizeof c->name, "%s", id.internal);
    c->bytes = sizeof(Shared) + (size_t)nranks * SLOT;
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) {
        g_err = "fake rccl: shm_open / ftruncate failed";
        return ncclSystemError;
    }
    void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        g_err = "fake rccl: mmap failed";
        return ncclSystemError;
    }
    c->sh = (Shared*)p;  // a fresh segment is zero-filled: arrived = sense = attached = 0
    c->sh->nranks = nranks;
    c->sh->attached.fetch_add(1, std::memory_order_acq_rel);
    c->bounce = (unsigned char*)std::malloc(SLOT);
    // like the real bootstrap: return when every rank has joined
    const auto t0 = std::chrono::steady_clock::now();
    while (c->sh->attached.load(std::memory_order_acquire) < nranks) {
        std::this_thread::yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
            g_err = "fake rccl: not every rank called ncclCommInitRank within 120 s";
            return ncclSystemError;
        }
    }
    *out = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = (Comm*)comm;
    if (!c) return ncclSuccess;
    const bool last = c->sh->attached.fetch_sub(1, std::memory_order_acq_rel) == 1;
    munmap(c->sh, c->bytes);
    if (last) shm_unlink(c->name);
    std::free(c->bounce);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
    *count = ((Comm*)comm)->nranks;
    return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) {
    *rank = ((Comm*)comm)->rank;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    const size_t es = size_of(t);
    if (!c || !es) {
        g_err = "fake rccl: unsupported data type";
        return ncclInvalidArgument;
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;  // the gather kernel queued before the collective
    const size_t per = SLOT / es;
    for (size_t done = 0; done < count; done += per) {
        const size_t n = count - done < per ? count - done : per;
        unsigned char* mine = c->sh->slots + (size_t)c->rank * SLOT;
        if (hipMemcpy(mine, (const char*)send + done * es, n * es, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        if (!barrier(c)) return ncclSystemError;
        std::memcpy(c->bounce, c->sh->slots, n * es);  // rank 0's contribution first, then 1, 2, ... on every rank alike
        for (int r = 1; r < c->nranks; ++r) fold_any(c->bounce, c->sh->slots + (size_t)r * SLOT, n, t, op);
        if (hipMemcpy((char*)recv + done * es, c->bounce, n * es, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        if (!barrier(c)) return ncclSystemError;  // nobody overwrites a slot before everyone has folded it
    }
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t) { return g_err; }

}  // extern "C"
