// featuredetection_amd/csrc/dist.hip -- image-shard data parallelism in the product (north_star: "images shard
// embarrassingly across the 8 GPUs of one node with a single RCCL gather of detections over xGMI").
//
// One process per GPU.  Image i belongs to rank i mod world (fd_dist_owner); models are replicated; there is NO data-path
// collective.  The only exchange is fd_dist_gather_records (or its two halves fd_dist_gather_begin / _end): every rank contributes its
// detection records of the images it owned since the last gather, and every rank receives all records ordered by (image, detector,
// original order) -- the order a single process would have produced them in.  A gather is a 64-byte header exchange (every rank's
// count and stride) followed by ONE ncclAllGather of max-count + 1 rows per rank: only the rows the fullest rank used travel (round 5
// moved the full fixed-stride buffers: 8.4 MB per rank and gather in config 5 for ~0.3 MB of records).  Both run on the handle's OWN
// stream: the gather never waits for, or stalls, the detection kernels queued on the context's stream, and with _begin / _end the
// records of one interval travel while the next interval's images are processed.  Payloads are KBs to MBs, i.e. latency-bound.
//
// librccl.so is loaded on first use (dlopen): a single-GPU process never pays for it, and libfd_hip.so has no link-time dependency.
#include "fd_internal.hpp"
#include <dlfcn.h>
#include <algorithm>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>
#include <mutex>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl& rccl() {
    static Rccl r;
    static std::string loadError;   // dlerror() text of the one attempt (dlerror returns NULL once it has been read)
    static std::once_flag once;
    std::call_once(once, [] {
        // FD_RCCL_LIB: another library with the same five entry points (tests/stub_rccl: ranks of one GPU meeting in shared memory)
        const char* env = getenv("FD_RCCL_LIB");
        const char* names[] = {env && *env ? env : "librccl.so", "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
            if (const char* e = dlerror()) { if (loadError.empty()) loadError = e; }
            if (env && *env) break;   // an explicit library that does not load is an error, not a reason to take another one
        }
        if (!r.lib) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
        r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    });
    if (!r.lib || !r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy)
        FD_THROW(FD_ERR_RUNTIME, "librccl.so is not available: %s", r.lib ? "missing symbols" : (loadError.empty() ? "not found" : loadError.c_str()));
    return r;
}

#define RCCL_CHECK(expr)                                                                                          \
    do {                                                                                                          \
        ncclResult_t _r = (expr);                                                                                 \
        if (_r != ncclSuccess) FD_THROW(FD_ERR_RUNTIME, "%s failed: %s", #expr, rccl().GetErrorString ? rccl().GetErrorString(_r) : "?"); \
    } while (0)

}  // namespace

struct fd_dist {
    fd_ctx* ctx = nullptr;
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t gstream = nullptr;   // the collectives' own stream
    hipEvent_t gdone = nullptr;
    DevBuf dsend, drecv, dhdr;       // dhdr: [1 + world] header rows
    HostBuf hsend, hrecv, hhdr;
    // a gather between _begin and _end
    bool inflight = false;
    int capInflight = 0;
    size_t rowsInflight = 0;         // rows per rank of the payload collective (0: nobody had a record)
    std::vector<fd_record> hdrs;     // every rank's header row of the gather in flight
    // result of the last collective, kept until it has been delivered: a caller whose `all` was too small (FD_ERR_CAPACITY) calls again
    // with a larger buffer and gets THESE records -- no second ncclAllGather that only some ranks would enter
    std::vector<fd_record> pending;
    bool havePending = false, pendingTrunc = false;
};

static_assert(sizeof(fd_record) == 64, "fd_record is eight doubles");
static_assert(sizeof(ncclUniqueId) == FD_DIST_ID_BYTES, "fd_dist id size");

// first half of a gather: the header exchange (a short wait on the handle's own stream) and the payload collective, queued
static void dist_gather_begin(fd_dist* d, const fd_record* local, int n_local, int cap_per_rank) {
    if (d->inflight) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_dist_gather_begin: the previous gather has not been collected (fd_dist_gather_end)");
    if (d->havePending) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_dist_gather_begin: a gathered set is waiting in the handle (fetch or discard it)");
    const int W = d->world;
    const int n = std::min(n_local, cap_per_rank);
    d->hsend.reserve(sizeof(fd_record) * ((size_t)n + 1));
    fd_record* hs = d->hsend.as<fd_record>();
    std::memset(hs, 0, sizeof(fd_record));
    hs[0].image = (double)n_local;           // row 0: how many records this rank had (more than cap: the receivers flag the truncation)
    hs[0].detector = (double)cap_per_rank;   // ... and the stride limit it was called with (checked by the receivers)
    if (n) std::memcpy(hs + 1, local, sizeof(fd_record) * (size_t)n);
    d->hdrs.assign((size_t)W, hs[0]);
    d->capInflight = cap_per_rank;
    d->rowsInflight = (size_t)n + 1;
    if (d->comm) {
        HIP_CHECK(hipSetDevice(d->ctx->device));
        hipStream_t st = d->gstream;
        // 1. every rank's header row (64 bytes each)
        d->dhdr.reserve(sizeof(fd_record) * ((size_t)W + 1));
        d->hhdr.reserve(sizeof(fd_record) * (size_t)W);
        HIP_CHECK(hipMemcpyAsync(d->dhdr.p, hs, sizeof(fd_record), hipMemcpyHostToDevice, st));
        RCCL_CHECK(rccl().AllGather(d->dhdr.p, d->dhdr.as<fd_record>() + 1, sizeof(fd_record), ncclUint8, d->comm, st));
        HIP_CHECK(hipMemcpyAsync(d->hhdr.p, d->dhdr.as<fd_record>() + 1, sizeof(fd_record) * (size_t)W, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));   // this stream carries nothing but gathers
        int64_t most = 0;
        for (int r = 0; r < W; ++r) {
            d->hdrs[(size_t)r] = d->hhdr.as<fd_record>()[r];
            // (every rank sees the same headers, so every rank throws here or none does: nobody is left alone in the payload collective)
            if ((int64_t)d->hdrs[(size_t)r].detector != (int64_t)cap_per_rank)
                FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_dist_gather_records: rank %d was called with cap_per_rank %lld, this rank with %d", r,
                         (long long)d->hdrs[(size_t)r].detector, cap_per_rank);
            most = std::max<int64_t>(most, std::min<int64_t>((int64_t)d->hdrs[(size_t)r].image, cap_per_rank));
        }
        // 2. the payload: max-count + 1 rows per rank (nothing at all when no rank has a record)
        d->rowsInflight = most > 0 ? (size_t)most + 1 : 0;
        if (d->rowsInflight) {
            const size_t bytes = d->rowsInflight * sizeof(fd_record);
            d->dsend.reserve(bytes);
            d->drecv.reserve(bytes * (size_t)W);
            d->hrecv.reserve(bytes * (size_t)W);
            HIP_CHECK(hipMemcpyAsync(d->dsend.p, hs, sizeof(fd_record) * ((size_t)n + 1), hipMemcpyHostToDevice, st));   // the used prefix only
            RCCL_CHECK(rccl().AllGather(d->dsend.p, d->drecv.p, bytes, ncclUint8, d->comm, st));
            HIP_CHECK(hipMemcpyAsync(d->hrecv.p, d->drecv.p, bytes * (size_t)W, hipMemcpyDeviceToHost, st));
        }
        HIP_CHECK(hipEventRecord(d->gdone, st));
    }
    d->inflight = true;
}

// second half: wait for the payload, merge into d->pending
static void dist_gather_finish(fd_dist* d) {
    if (!d->inflight) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_dist_gather_end: no gather in flight");
    d->inflight = false;
    const int W = d->world;
    const fd_record* hr = d->hsend.as<fd_record>();
    size_t rows = d->rowsInflight;
    if (d->comm) {
        HIP_CHECK(hipEventSynchronize(d->gdone));
        hr = d->hrecv.as<fd_record>();
    }
    // records of all ranks, ordered by (image, detector, original order): stable sort of (key, position)
    std::vector<const fd_record*> ptrs;
    bool trunc = false;
    for (int r = 0; r < W; ++r) {
        const int64_t cnt = (int64_t)d->hdrs[(size_t)r].image;
        trunc = trunc || cnt > d->capInflight;
        const int64_t take = std::min<int64_t>(cnt, d->capInflight);
        const fd_record* part = hr + (size_t)r * rows;
        for (int64_t i = 0; i < take; ++i) ptrs.push_back(part + 1 + i);
    }
    std::stable_sort(ptrs.begin(), ptrs.end(), [](const fd_record* a, const fd_record* b) {
        return a->image != b->image ? a->image < b->image : a->detector < b->detector;
    });
    d->pending.resize(ptrs.size());
    for (size_t i = 0; i < ptrs.size(); ++i) d->pending[i] = *ptrs[i];
    d->pendingTrunc = trunc;
    d->havePending = true;
}

static void dist_deliver(fd_dist* d, fd_record* all, int64_t all_cap, int64_t* n_all, int* truncated) {
    *n_all = (int64_t)d->pending.size();
    if (truncated) *truncated = d->pendingTrunc ? 1 : 0;
    if (!all) return;   // count only: the records stay for the call that brings a buffer
    if ((int64_t)d->pending.size() > all_cap)
        FD_THROW(FD_ERR_CAPACITY, "fd_dist_gather_records: %zu records, capacity %lld (call again with a larger buffer: no new collective)", d->pending.size(), (long long)all_cap);
    if (!d->pending.empty()) std::memcpy(all, d->pending.data(), sizeof(fd_record) * d->pending.size());
    d->pending.clear();
    d->havePending = false;
}

extern "C" {

int fd_dist_owner(int64_t image_index, int world) { return world > 0 ? (int)(image_index % world) : 0; }

int fd_dist_unique_id(uint8_t* id) {
    return fd_guard(nullptr, [&] {
        if (!id) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_dist_unique_id: NULL argument");
        ncclUniqueId u;
        RCCL_CHECK(rccl().GetUniqueId(&u));
        std::memcpy(id, &u, sizeof(u));
    });
}

int fd_dist_init(fd_ctx* ctx, int rank, int world, const uint8_t* id, fd_dist** out) {
    return fd_guard(ctx, [&] {
        if (!ctx || !out || world < 1 || rank < 0 || rank >= world) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_dist_init: bad argument");
        if (world > 1 && !id) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_dist_init: the communicator id is needed for more than one rank");
        std::unique_ptr<fd_dist> d(new fd_dist());
        d->ctx = ctx;
        d->rank = rank;
        d->world = world;
        // A single rank gathers from itself: no communicator, no librccl, whatever `id` holds (a caller that always passes
        // its id buffer must not need librccl on a one-GPU host).  FD_DIST_FORCE_COMM=1 asks for a real one-rank communicator
        // (ncclCommInitRank(world = 1), the gather then goes through ncclAllGather like with N ranks): the one-GPU test of the binding.
        const char* force = getenv("FD_DIST_FORCE_COMM");
        if (world > 1 || (id && force && atoi(force) != 0)) {
            HIP_CHECK(hipSetDevice(ctx->device));
            ncclUniqueId u;
            std::memcpy(&u, id, sizeof(u));
            RCCL_CHECK(rccl().CommInitRank(&d->comm, world, u, rank));
            HIP_CHECK(hipStreamCreateWithFlags(&d->gstream, hipStreamNonBlocking));
            HIP_CHECK(hipEventCreateWithFlags(&d->gdone, hipEventDisableTiming));
        }
        *out = d.release();
    });
}

void fd_dist_destroy(fd_dist* d) {
    if (!d) return;
    if (d->gstream) (void)hipStreamSynchronize(d->gstream);
    if (d->comm) (void)rccl().CommDestroy(d->comm);
    if (d->gdone) (void)hipEventDestroy(d->gdone);
    if (d->gstream) (void)hipStreamDestroy(d->gstream);
    delete d;
}

// Drops the records a finished collective left in the handle (a count-only call, or FD_ERR_CAPACITY, that the caller does not follow up):
// the next fd_dist_gather_records is a new collective again.  EVERY rank must drop (or take) a set -- a rank that still holds one
// would answer the next call from its handle while the others enter ncclAllGather.
void fd_dist_gather_discard(fd_dist* d) {
    if (!d) return;
    d->pending.clear();
    d->havePending = false;
    d->pendingTrunc = false;
}
int fd_dist_gather_pending(const fd_dist* d) { return d && d->havePending ? 1 : 0; }

int fd_dist_rank(const fd_dist* d) { return d ? d->rank : 0; }
int fd_dist_world(const fd_dist* d) { return d ? d->world : 1; }

// {image, detector, cx, cy, w, h, score, probability} of every detection, as doubles (ints and the fp32 score are exact in fp64)
int fd_pack_records(int64_t image_id, int32_t detector_id, const fd_detection* dets, int n, fd_record* out) {
    if (n < 0 || (n > 0 && (!dets || !out))) return FD_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < n; ++i) {
        fd_record& r = out[i];
        r.image = (double)image_id; r.detector = (double)detector_id;
        r.cx = dets[i].cx; r.cy = dets[i].cy; r.w = dets[i].w; r.h = dets[i].h;
        r.score = (double)dets[i].score; r.probability = dets[i].probability;
    }
    return FD_OK;
}

int fd_dist_gather_records(fd_dist* d, const fd_record* local, int n_local, int cap_per_rank, fd_record* all, int64_t all_cap, int64_t* n_all,
                           int* truncated) {
    return fd_guard(d ? d->ctx : nullptr, [&] {
        if (!d || n_local < 0 || cap_per_rank < 1 || (n_local > 0 && !local) || !n_all) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_dist_gather_records: bad argument");
        if (!d->havePending) {
            // The collective.  EVERY rank must make this call, with the SAME cap_per_rank (fd_hip.h).  It runs once per set of records:
            // what it delivered stays in the handle until a call takes it.
            dist_gather_begin(d, local, n_local, cap_per_rank);
            dist_gather_finish(d);
        }
        dist_deliver(d, all, all_cap, n_all, truncated);
    });
}

int fd_dist_gather_begin(fd_dist* d, const fd_record* local, int n_local, int cap_per_rank) {
    return fd_guard(d ? d->ctx : nullptr, [&] {
        if (!d || n_local < 0 || cap_per_rank < 1 || (n_local > 0 && !local)) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_dist_gather_begin: bad argument");
        dist_gather_begin(d, local, n_local, cap_per_rank);
    });
}

int fd_dist_gather_end(fd_dist* d, fd_record* all, int64_t all_cap, int64_t* n_all, int* truncated) {
    return fd_guard(d ? d->ctx : nullptr, [&] {
        if (!d || !n_all) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_dist_gather_end: bad argument");
        if (!d->havePending) dist_gather_finish(d);   // (a retry after FD_ERR_CAPACITY finds the set in the handle)
        dist_deliver(d, all, all_cap, n_all, truncated);
    });
}

}  // extern "C"
