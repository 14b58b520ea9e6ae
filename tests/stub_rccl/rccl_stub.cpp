// tests/stub_rccl/rccl_stub.cpp -- TEST INFRASTRUCTURE ONLY.  A stand-in for librccl.so with the five entry points csrc/dist.hip binds
// (ncclGetUniqueId, ncclCommInitRank, ncclAllGather, ncclCommDestroy, ncclGetErrorString), so that the multi-rank branch of
// fd_dist_gather_records and ffp_detect_app --gpus N can run with several processes on ONE GPU (the test boxes have one).  The ranks
// meet in a POSIX shared-memory segment named after the unique id; an all-gather is: device -> my slot, barrier, all slots -> device,
// barrier.  Selected with FD_RCCL_LIB=<this library>; nothing in the product links or loads it otherwise.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>

namespace {
constexpr size_t SLOT = 32u << 20;   // bytes a rank may contribute per call
struct Shared {
    std::atomic<int> arrived;
    std::atomic<int> generation;
    std::atomic<int> attached;
};
struct Comm {
    int rank, world;
    Shared* sh;
    unsigned char* slots;
    size_t bytes;
    char name[64];
};
bool barrier(Comm* c) {
    const int gen = c->sh->generation.load();
    if (c->sh->arrived.fetch_add(1) + 1 == c->world) {
        c->sh->arrived.store(0);
        c->sh->generation.fetch_add(1);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (c->sh->generation.load() == gen) {
        std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return false;   // a rank died: do not hang the test
    }
    return true;
}
size_t elem(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::memset(id, 0, sizeof(*id));
    const unsigned long long v = (unsigned long long)getpid() * 1000003ull ^ (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
    std::snprintf(id->internal, sizeof(id->internal), "fdstub_%llx", v);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    Comm* c = new Comm();
    c->rank = rank; c->world = nranks;
    // the segment's name: hex of the id's first bytes (whoever made the id -- this stub or the real ncclGetUniqueId)
    std::snprintf(c->name, sizeof(c->name), "/fdstub_");
    for (int i = 0; i < 16; ++i) std::snprintf(c->name + 8 + 2 * i, 3, "%02x", (unsigned char)id.internal[i]);
    c->bytes = 4096 + SLOT * (size_t)nranks;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { delete c; return ncclSystemError; }
    void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->sh = reinterpret_cast<Shared*>(p);   // a fresh segment is zero-filled: counters start at 0
    c->slots = reinterpret_cast<unsigned char*>(p) + 4096;
    c->sh->attached.fetch_add(1);
    *comm = reinterpret_cast<ncclComm_t>(c);
    return barrier(c) ? ncclSuccess : ncclSystemError;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t bytes = sendcount * elem(datatype);
    if (bytes > SLOT) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->slots + SLOT * (size_t)c->rank, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    for (int r = 0; r < c->world; ++r)
        if (hipMemcpy(reinterpret_cast<unsigned char*>(recvbuff) + bytes * (size_t)r, c->slots + SLOT * (size_t)r, bytes, hipMemcpyHostToDevice) != hipSuccess)
            return ncclUnhandledCudaError;
    return barrier(c) ? ncclSuccess : ncclSystemError;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (c->sh->attached.fetch_sub(1) == 1) shm_unlink(c->name);
    munmap(c->sh, c->bytes);
    delete c;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : "rccl stub error"; }

}  // extern "C"
