// featuredetection_amd/csrc/fd_internal.hpp -- internal declarations of libfd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <map>
#include <memory>
#include <string>
#include <typeindex>
#include <typeinfo>
#include <vector>
#include <stdexcept>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <deque>
#include <exception>
#include "../../include/fd_hip.h"
#include "../../include/fd_hip_bench.h"

struct FdPinned {   // pinned host scratch (allocated lazily): pageable async copies cost ~1 ms each on this stack
    void* p = nullptr;
    size_t cap = 0;
};

// A few persistent host threads for the per-detector host stages of the batch entry points (ordering the positives, overlap
// elimination, the SVM launch, NMS): run(f) executes f on every worker and on the caller and returns when all are done.
struct FdWorkerPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, cvDone;
    std::function<void()> job;
    uint64_t gen = 0;
    int pending = 0;
    bool stop = false;
    explicit FdWorkerPool(int n) {
        for (int i = 0; i < n; ++i) th.emplace_back([this] { loop(); });
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            std::function<void()> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
                j = job;
            }
            try { j(); } catch (...) {}   // jobs report their errors themselves; an escaping exception would terminate the process
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--pending == 0) cvDone.notify_one();
            }
        }
    }
    void run(const std::function<void()>& f) {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = f;
            pending = (int)th.size();
            ++gen;
        }
        cv.notify_all();
        // the workers reference the caller's stack through f: wait for them even when the caller's own share throws
        std::exception_ptr err;
        try { f(); } catch (...) { err = std::current_exception(); }
        {
            std::unique_lock<std::mutex> lk(mu);
            cvDone.wait(lk, [&] { return pending == 0; });
        }
        if (err) std::rethrow_exception(err);
    }
    ~FdWorkerPool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
};

// Process-wide queue of host continuations (a few persistent threads, FD_ASYNC_THREADS, default 2): the ticket entry points
// (fd_detect_five_stage_frames_begin) hand the host stages that follow their kernels -- wait, order the positives, overlap
// elimination, SVM launch, wait, NMS -- to it, so the calling thread is free to queue the next call's kernels and the GPU never waits
// for a host stage of one call while another call has work ready.  submit() returns a handle; wait() rethrows the task's exception.
struct FdAsyncTask {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    std::exception_ptr error;
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        if (error) std::rethrow_exception(error);
    }
};
struct FdAsyncQueue {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::pair<std::function<void()>, std::shared_ptr<FdAsyncTask>>> q;
    bool stop = false;
    explicit FdAsyncQueue(int n) {
        for (int i = 0; i < n; ++i) th.emplace_back([this] { loop(); });
    }
    void loop() {
        for (;;) {
            std::pair<std::function<void()>, std::shared_ptr<FdAsyncTask>> item;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;   // stop requested and nothing left
                item = std::move(q.front());
                q.pop_front();
            }
            std::exception_ptr err;
            try { item.first(); } catch (...) { err = std::current_exception(); }
            {
                std::lock_guard<std::mutex> lk(item.second->mu);
                item.second->error = err;
                item.second->done = true;
            }
            item.second->cv.notify_all();
        }
    }
    std::shared_ptr<FdAsyncTask> submit(std::function<void()> f) {
        auto t = std::make_shared<FdAsyncTask>();
        {
            std::lock_guard<std::mutex> lk(mu);
            q.emplace_back(std::move(f), t);
        }
        cv.notify_one();
        return t;
    }
    ~FdAsyncQueue() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
};
FdAsyncQueue& fd_async_queue();   // ctx.hip
FdAsyncQueue& fd_batch_queue();   // ctx.hip: the per-detector host stages of batches in flight (FD_BATCH_THREADS threads)

struct fd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string error;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // bracket the dominant kernel of a detect call when kernel_timing is on (fd_hip_bench.h)
    bool kernel_timing = false;
    int kernel_timing_mode = 1;      // WVM cascades: 1 = all cascade kernels of the call, 2 = the dense pre-filter only, 3 = stage B's chain kernels (fd_hip_bench.h)
    hipEvent_t evx[8] = {};          // mode 3: one event pair per stage-B phase around its k_wvb_chain2 launch (created on first use)
    int evxN = 0;                    // pairs recorded by the last timed launch
    hipEvent_t evg[2] = {};          // mode 2: around the last k_wvm_prefilter_group launch (fd_last_group_prefilter_ms)
    int evgMembers = 0;              // its member count (0: none recorded)
    const char* last_kernel = "";
    float last_kernel_ms = 0.f;
    int num_cus = 256;
    FdPinned pinned;
    hipStream_t aux = nullptr;   // second stream (created on first use): small follow-up work that must not queue behind ctx->stream
    hipStream_t pool[8] = {};   // batch jobs are spread over these (created on first use)
    hipStream_t tail = nullptr; // high-priority stream of the batch entry points' follow-up kernels (created on first use)
    std::unique_ptr<FdWorkerPool> workers;   // created on first use by the batch entry points (FD_BATCH_THREADS)
    // per-context device scratch of the translation units (fd_scratch<T>): owned by the context, freed with it
    std::map<std::type_index, std::shared_ptr<void>> scratch;
};

// the context's scratch object of type T (one per context and type, created on first use); contexts are not shared between
// host threads, so no locking
template <class T>
static inline T& fd_scratch(fd_ctx* ctx) {
    std::shared_ptr<void>& slot = ctx->scratch[std::type_index(typeid(T))];
    if (!slot) slot = std::shared_ptr<void>(new T(), [](void* p) { delete static_cast<T*>(p); });
    return *static_cast<T*>(slot.get());
}

struct FdError {
    int code;
    std::string msg;
};

#define FD_THROW(code, ...)                               \
    do {                                                  \
        char _b[512];                                     \
        snprintf(_b, sizeof(_b), __VA_ARGS__);            \
        throw FdError{code, std::string(_b)};             \
    } while (0)

#define HIP_CHECK(expr)                                                                       \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) FD_THROW(FD_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// Runs body, converts exceptions into status codes + ctx->error (no exceptions cross the C ABI).
template <class F>
static inline int fd_guard(fd_ctx* ctx, F&& body) {
    try {
        body();
        return FD_OK;
    } catch (const FdError& e) {
        if (ctx) ctx->error = e.msg;
        return e.code;
    } catch (const std::exception& e) {
        if (ctx) ctx->error = e.what();
        return FD_ERR_RUNTIME;
    } catch (...) {
        if (ctx) ctx->error = "unknown error";
        return FD_ERR_RUNTIME;
    }
}

// Simple owning device buffer that only grows (reused across calls: no hipMalloc in steady state).
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (p) HIP_CHECK(hipFree(p));
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        HIP_CHECK(hipMalloc(&p, want));
        cap = want;
    }
    template <class T> T* as() const { return (T*)p; }
};

// Pinned host staging buffer
struct HostBuf {
    void* p = nullptr;
    size_t cap = 0;
    ~HostBuf() { if (p) (void)hipHostFree(p); }
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (p) HIP_CHECK(hipHostFree(p));
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        HIP_CHECK(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
    }
    template <class T> T* as() const { return (T*)p; }
};

// pinned scratch of the context, at least `bytes` large
static inline void* fd_pinned(fd_ctx* ctx, size_t bytes) {
    if (bytes > ctx->pinned.cap) {
        if (ctx->pinned.p) HIP_CHECK(hipHostFree(ctx->pinned.p));
        ctx->pinned.p = nullptr;
        ctx->pinned.cap = 0;
        size_t want = bytes + bytes / 2 + 4096;
        HIP_CHECK(hipHostMalloc(&ctx->pinned.p, want, hipHostMallocDefault));
        ctx->pinned.cap = want;
    }
    return ctx->pinned.p;
}

static inline hipStream_t fd_aux_stream(fd_ctx* ctx) {
    if (!ctx->aux) HIP_CHECK(hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
    return ctx->aux;
}

// Stream of the small follow-up kernels of the batch entry points (the SVM stage and read-backs of a detector whose cascade has
// finished): highest priority and never behind the cascades still queued on the pool streams, so that the host can finish the
// detectors one by one while the GPU works through the remaining cascades.
static inline hipStream_t fd_tail_stream(fd_ctx* ctx) {
    if (!ctx->tail) {
        int least = 0, greatest = 0;
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_CHECK(hipStreamCreateWithPriority(&ctx->tail, hipStreamNonBlocking, greatest));
    }
    return ctx->tail;
}

static inline hipStream_t fd_pool_stream(fd_ctx* ctx, int i) {
    static const int nstreams = [] { const char* e = getenv("FD_BATCH_STREAMS"); int v = e ? atoi(e) : 8; return v < 1 ? 1 : (v > 8 ? 8 : v); }();   // (8 since round 6: config 3 with the batch queue 9280 against 8910 Mpatches/s with 4)
    hipStream_t& s = ctx->pool[i % nstreams];
    if (!s) HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return s;
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): `done` is the caller's static per-kernel bit mask
static inline void fd_allow_lds(fd_ctx* ctx, const void* kernel, int bytes, uint64_t& done) {
    const uint64_t bit = 1ull << (ctx->device & 63);
    if (done & bit) return;
    HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done |= bit;
}

// what the fused HOG + SVM kernel (hog_svm_fused.hpp, launched from hog.hip) needs of an f32 RBF model (svm.hip)
struct FdSvmFusedView {
    const float* svFrag;   // fragment-major support vectors [nsv_pad][KP]
    const float* ss;       // |s|^2, [nsv_pad]
    const float* coeff;    // [nsv_pad], zero padded
    int nsv_pad, KP;
    float bias;
    double gamma;
};

struct FdStreamSwap {   // temporarily redirects everything that launches on ctx->stream
    fd_ctx* c;
    hipStream_t keep;
    FdStreamSwap(fd_ctx* c_, hipStream_t s_) : c(c_), keep(c_->stream) { c->stream = s_; }
    ~FdStreamSwap() { c->stream = keep; }
};

static inline int fd_cvRound(double v) { return (int)std::lrint(v); }  // cvRound: half-to-even

// ---------------- pyramid ----------------
struct LayerDesc {          // device-visible layer descriptor
    int32_t w, h;           // layer size (pixels)
    int32_t ch;             // channels of the filtered layer
    int32_t pad;
    uint32_t gray_off;      // byte offset of the gray layer in the arena
    uint32_t filt_off;      // byte offset of the filtered layer (== gray_off when no layer filter)
};

struct HostLayer {
    int index;
    double scale;
    int w, h, ch;
    uint32_t gray_off, filt_off;
    bool kept;
    int chain, depth;       // first-octave slot i and pyrDown depth j
    uint32_t blur_off = 0;  // kept layers of a gradient pyramid with blurring: the blurred gray layer (GradientFilter.cpp:43-46)
};

struct WindowLayer {        // per kept layer, for the window enumeration of one (patch, step, roi)
    int32_t layer;          // position in the kept-layer list
    int32_t bx, by;         // first window origin
    int32_t nx, ny;         // number of window positions
    int32_t ow, oh;         // original (image) patch size
    int64_t first;          // index of the first window of this layer
};

struct fd_pyramid {
    fd_ctx* ctx;
    size_t octl;
    double inc, minS, maxS;
    int filter_kind = FD_LAYER_NONE, bins = 9, signed_gradients = 0, interpolate = 0, grad_kernel = 1, lbp_type = 0;
    int grad_blur = 0;               // GradientFilter's blurKernelSize (0: none); blurred copies of the kept layers live at blur_off
    int img_w = 0, img_h = 0;
    std::vector<HostLayer> all;      // every computed layer (kept or only a pyrDown source)
    std::vector<int> kept;           // indices into all, sorted by layer index
    DevBuf arena;                    // gray full-res + all layers + filtered layers
    DevBuf input;                    // staging for host images
    DevBuf lut;                      // gradient binning LUT (65536 * 2|4 bytes)
    DevBuf rtab;                     // cv::resize coordinate / weight tables of the first-octave layers (k_resize_down, pyramid.hip)
    std::vector<uint32_t> rtab_x, rtab_y;   // per entry of `all` (depth-0 layers with a pyrDown successor): offsets into rtab, ~0u = none
    std::vector<uint32_t> rtile_off;        // per k_resize_down launch (MAXJ chains): its tile list in rtab (offset in int2 entries) ...
    std::vector<int> rtile_cnt;             // ... and the number of tiles per frame
    DevBuf layer_table;              // LayerDesc per kept layer
    std::vector<LayerDesc> h_layer_table;
    uint32_t gray_full_off = 0;
    size_t arena_bytes = 0;
    uint64_t version = 0;            // bumped by every update
    // several frames of identical size in one pyramid (fd_pyramid_set_frames): frame f's layers live at arena + f * image_stride
    // with the layout of frame 0.  One launch per pyramid stage and ONE cascade / SVM launch serve all frames of a call
    // (fd_detect_five_stage_frames); the per-frame launch chain is what bounds small frames.
    int nimg = 1;
    size_t image_stride = 0;
    // fd_pyramid_select: layer sub-range (pyramid layer indices, -1 = open end), layer step (over the kept layers, starting at the
    // first) and default region of interest of every window enumeration that follows (DirectPyramidFeatureExtractor.cpp:75-123)
    int sel_first = -1, sel_last = -1, sel_step = 1;
    // fd_pyramid_select_view: index range of the layers a pyramid built on this one exposes (ImagePyramid(pyramid, min, max)): the
    // layers outside do not exist for the window enumeration, and the layer step counts from the first layer inside
    int view_first = -1, view_last = -1;
    bool sel_has_roi = false;
    int sel_roi[4] = {0, 0, 0, 0};
    // recorded on the updating stream after the last kernel of an update: consumers on OTHER streams (the stream pool of the
    // batch entry points) wait for it; consumers on the same stream are ordered anyway
    hipEvent_t ready = nullptr;
    hipStream_t readyStream = nullptr;
    ~fd_pyramid() { if (ready) (void)hipEventDestroy(ready); }
};

// entry points that only know single-frame pyramids
static inline void fd_pyramid_require_single(const fd_pyramid* p, const char* who) {
    if (p->nimg > 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "%s: the pyramid holds %d frames (fd_pyramid_set_frames); use fd_detect_five_stage_frames", who, p->nimg);
}

// makes `consumer` wait for the pyramid's last update when that ran on another stream
static inline void fd_pyramid_wait(const fd_pyramid* p, hipStream_t consumer) {
    if (p->ready && consumer != p->readyStream) HIP_CHECK(hipStreamWaitEvent(consumer, p->ready, 0));
}

void fd_pyramid_update_on(fd_pyramid* p, const uint8_t* image, int w, int h, int ch, int is_device, hipStream_t st);
constexpr int FD_MAX_FRAMES = 64;
void fd_enumerate_layers(const fd_pyramid* p, int pw, int ph, int sx, int sy, const int* roi,
                         std::vector<WindowLayer>& out, int64_t& total);

// ---------------- host-side detection logic (hostalgo.cpp) ----------------
void fd_host_overlap_elimination(const fd_detection* in, int n, float dist, float ratio, std::vector<int>& keep);
void fd_host_block_nms_sparse(const std::vector<fd_detection>& pos, int imgW, int imgH, int sz, bool masked,
                              std::vector<int>& maxima_xy /* pairs x,y in row-major order */);
