// featuredetection_amd/csrc/ctx.hip -- context management of libfd_hip.so
#include "fd_internal.hpp"
#include <cstdlib>
#include <cstring>

// Deployment note: the batch entry points spread independent frames / detectors over a pool of HIP streams; with the runtime's
// default of four hardware queues the pool, the context stream and the read-back stream share queues and the short dependent
// kernels of different frames serialise (640x480 five-stage batch: 115 -> 145 Mpatches/s with GPU_MAX_HW_QUEUES=8).  The
// variable is read when the HIP runtime initialises, so it is the host process's to set (bench.py does); the library does
// not touch the process environment.

extern "C" {

const char* fd_version(void) { return "fd_hip 0.1 (gfx950)"; }

int fd_device_count(int* n) {
    if (!n) return FD_ERR_INVALID_ARGUMENT;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) { *n = 0; return FD_ERR_HIP; }
    *n = c;
    return FD_OK;
}

int fd_ctx_create(int device_id, void* hip_stream, fd_ctx** out) {
    if (!out) return FD_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    fd_ctx* ctx = new fd_ctx();
    int rc = fd_guard(ctx, [&] {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) FD_THROW(FD_ERR_HIP, "no HIP device available (%s); this library has no CPU fallback", hipGetErrorString(e));
        if (device_id < 0 || device_id >= n) FD_THROW(FD_ERR_INVALID_ARGUMENT, "device %d out of range (0..%d)", device_id, n - 1);
        HIP_CHECK(hipSetDevice(device_id));
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            FD_THROW(FD_ERR_HIP, "device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
        ctx->device = device_id;
        ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (hip_stream) {
            ctx->stream = (hipStream_t)hip_stream;
            ctx->own_stream = false;
        } else {
            HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
            ctx->own_stream = true;
        }
        HIP_CHECK(hipEventCreate(&ctx->ev0));
        HIP_CHECK(hipEventCreate(&ctx->ev1));
    });
    if (rc != FD_OK) {
        // keep the message reachable for the caller through a static buffer
        static thread_local std::string last;
        last = ctx->error;
        fprintf(stderr, "fd_ctx_create: %s\n", last.c_str());
        delete ctx;
        return rc;
    }
    *out = ctx;
    return FD_OK;
}

void fd_ctx_destroy(fd_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    ctx->scratch.clear();   // device buffers of the per-context scratch objects
    if (ctx->pinned.p) (void)hipHostFree(ctx->pinned.p);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (hipEvent_t e : ctx->evx) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->evg) if (e) (void)hipEventDestroy(e);
    if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
    if (ctx->tail) (void)hipStreamDestroy(ctx->tail);
    for (hipStream_t ps : ctx->pool) if (ps) (void)hipStreamDestroy(ps);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// Creates the streams a context otherwise creates on first use (batch pool, high-priority tail, auxiliary).  The HIP runtime deals
// streams to its hardware queues in creation order: a process that runs other work first (and so creates other streams first) gets a
// different mapping for these -- bench.py measured the 15-detector batch at 8.4 G patches/s behind two other workloads and 9.4 G alone.
int fd_ctx_warm_streams(fd_ctx* ctx) {
    return fd_guard(ctx, [&] {
        if (!ctx) FD_THROW(FD_ERR_INVALID_ARGUMENT, "NULL context");
        HIP_CHECK(hipSetDevice(ctx->device));
        for (int i = 0; i < 8; ++i) (void)fd_pool_stream(ctx, i);
        (void)fd_tail_stream(ctx);
        (void)fd_aux_stream(ctx);
    });
}

const char* fd_last_error(const fd_ctx* ctx) { return ctx ? ctx->error.c_str() : "NULL context"; }

int fd_ctx_synchronize(fd_ctx* ctx) {
    return fd_guard(ctx, [&] {
        if (!ctx) FD_THROW(FD_ERR_INVALID_ARGUMENT, "NULL context");
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int fd_ctx_set_kernel_timing(fd_ctx* ctx, int enable) {
    if (!ctx) return FD_ERR_INVALID_ARGUMENT;
    ctx->kernel_timing = enable != 0;
    ctx->kernel_timing_mode = enable == 2 ? 2 : (enable == 3 ? 3 : 1);
    return FD_OK;
}

int fd_last_kernel_ms(fd_ctx* ctx, const char** kernel_name, float* ms) {
    if (!ctx) return FD_ERR_INVALID_ARGUMENT;
    if (kernel_name) *kernel_name = ctx->last_kernel;
    if (ms) *ms = ctx->last_kernel_ms;
    return FD_OK;
}

}  // extern "C"

FdAsyncQueue& fd_async_queue() {
    static FdAsyncQueue q([] {
        const char* e = getenv("FD_ASYNC_THREADS");
        const int n = e ? atoi(e) : 2;
        return n < 1 ? 1 : (n > 16 ? 16 : n);
    }());
    return q;
}

FdAsyncQueue& fd_batch_queue() {
    static FdAsyncQueue q([] {
        const char* e = getenv("FD_BATCH_THREADS");
        const int n = e ? atoi(e) : 8;
        return n < 1 ? 1 : (n > 16 ? 16 : n);
    }());
    return q;
}
