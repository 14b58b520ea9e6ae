// featuredetection_amd/csrc/rvm.hip -- classification::RvmClassifier / ProbabilisticRvmClassifier
// (RvmClassifier.cpp:75-126, ProbabilisticRvmClassifier.cpp:52-64; the "prvm" classifier of
// ffpDetectApp.cpp:484), SURVEY.md 8(f) row 1: the dense cascaded reduced-vector machine.
//
// Level k of the cascade costs one dense kernel evaluation K(x, rsv_k) over the whole patch and (through
// the reference's cached evaluation path, see oracle/orc_classify.cpp Rvm::eval) adds c[k][k] * K_k to the
// running fp64 distance; a window leaves at the first level whose threshold it misses.  Unlike the WVM there
// is no per-window preparation and every level costs the same, so the mapping is lane == window: a lane walks
// its own patch in the reference's element order (the fp32 sum of squared differences is bit-identical),
// the reduced set vector is wave-uniform (scalar loads), and the transcendental runs 64-wide.  Early exits
// thin the lanes out, so the cascade runs as a few passes over level ranges with stream compaction in
// between (survivor queues in HBM, counts read on the device: no host round trip between passes).
//
// Feature vectors: f32 vectors (per-Mat API) or the u8 patch feature spaces of ffpDetectApp.cpp:446-461
// (gray, hq64 = HistEq64Filter, histeq = cv::equalizeHist) followed by ConversionFilter(CV_32F, scale, shift),
// which is applied on the fly (x = float(u8) * scale + shift, like cv::Mat::convertTo).
#include "fd_internal.hpp"
#include "fd_device.hpp"
#include <algorithm>
#include <cstring>
#include <memory>

void fd_window_to_detection(const fd_pyramid* p, const std::vector<WindowLayer>& wls, int sx, int sy, int64_t wid, fd_detection& d);

constexpr int RVM_MAX_LAYERS = 64;
constexpr int RVM_DEEP_FROM = 16;    // levels from here on are evaluated 64 at a time per surviving window
constexpr int RVM_MAX_DIM = 32;       // patch width/height limit of the window feature kernel

struct RvmWinLayer {
    int32_t bx, by, nx, ny;
    int32_t lw;
    uint32_t off;
    int64_t first;
};
struct RvmWinTable {
    int32_t n, sx, sy, pad;
    int64_t total;
    RvmWinLayer l[RVM_MAX_LAYERS];
};

struct RvmDev {
    int32_t kernel, dim, numFilters, numUse;
    double p0, p1;
    int32_t degree;
    float bias;
    const float* sv;      // [numFilters][dim]
    const float* svT;     // [dim][numFilters]: element i of every reduced set vector (lane == level reads are coalesced)
    const float* diag;    // [numFilters]: coefficients[k][k]
    const float* thr;     // [numFilters]
};

struct RvmRec {   // survivor between passes / positive record
    uint32_t wid_lo, wid_hi;
    double d;
};

struct fd_rvm {
    fd_ctx* ctx;
    RvmDev dev;
    int filter_w, filter_h;
    double logisticA, logisticB;
    std::vector<float> h_thr;
    DevBuf sv, svT, diag, thr;
    DevBuf feats, q0, q1, counters, level, dist, pos;   // scratch reused across calls
};

namespace {

using namespace fd_dev;

__device__ __forceinline__ double powi_d(double base, int exponent) {  // PolynomialKernel.hpp:73-81
    double tmp = base, ret = 1.0;
    for (int t = exponent; t > 0; t /= 2) {
        if (t % 2 == 1) ret *= tmp;
        tmp = tmp * tmp;
    }
    return ret;
}

// u8 patch features of every window: gray (copy), HistEq64Filter, cv::equalizeHist.  One wave per window.
__global__ __launch_bounds__(64) void k_window_patches(const uint8_t* __restrict__ arena, RvmWinTable wt, int pw, int ph, int mode,
                                                       uint8_t* __restrict__ out) {
    __shared__ unsigned char px[RVM_MAX_DIM * RVM_MAX_DIM];
    __shared__ unsigned char eq[RVM_MAX_DIM * RVM_MAX_DIM];
    __shared__ int hist[256];
    __shared__ int lut[256];
    const int lane = threadIdx.x;
    const int n = pw * ph;
    for (int64_t wid = blockIdx.x; wid < wt.total; wid += gridDim.x) {
        int li = 0;
        for (int l = 1; l < wt.n; ++l) li = wt.l[l].first <= wid ? l : li;
        const RvmWinLayer& wl = wt.l[li];
        const int local = (int)(wid - wl.first);
        const int iy = local / wl.nx, ix = local - iy * wl.nx;
        const uint8_t* src = arena + wl.off + (size_t)(wl.by + iy * wt.sy) * wl.lw + (wl.bx + ix * wt.sx);
        for (int i = lane; i < n; i += 64) {
            const int y = i / pw, x = i - y * pw;
            px[i] = src[(size_t)y * wl.lw + x];
        }
        wave_sync();
        const unsigned char* res = px;
        if (mode == FD_FEATURE_HQ64) { histeq64_wave(px, eq, n, hist, lane); res = eq; }
        else if (mode == FD_FEATURE_HISTEQ) { equalize_hist_wave(px, eq, n, hist, lut, lane); res = eq; }
        for (int i = lane; i < n; i += 64) out[(size_t)wid * n + i] = res[i];
        wave_sync();
    }
}

// kernel value of the reference for one lane's vector x against the (wave-uniform) reduced set vector s.
// f32 paths of RbfKernel.hpp:97-108 (fp32 sum, element order), HistogramIntersectionKernel.hpp:81-93,
// LinearKernel.hpp / PolynomialKernel.hpp (cv::Mat::dot: fp64 accumulation).
template <bool U8IN>
__device__ __forceinline__ double rvm_kernel_value(const RvmDev& m, const void* xv, const float* __restrict__ s, float scale, float shift) {
    const int dim = m.dim;
    auto X = [&](int i) -> float {
        if (U8IN) return (float)((const unsigned char*)xv)[i] * scale + shift;   // cv::Mat::convertTo(CV_32F, scale, shift)
        return ((const float*)xv)[i];
    };
    if (m.kernel == FD_KERNEL_RBF) {
        float sum = 0.f;
        int i = 0;
        if (U8IN && (dim & 15) == 0) {
            // 16 pixels per 128-bit load, the next chunk in flight while the current one is consumed; the fp32
            // chain itself stays in element order
            const uint4* xw = (const uint4*)xv;   // rows are 16-byte aligned when dim % 16 == 0
            const float4* sw = (const float4*)s;
            uint4 cur = xw[0];
            for (; i + 16 <= dim; i += 16) {
                const uint4 nxt = xw[(i + 16 < dim ? i + 16 : i) >> 4];
                const unsigned int w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 sv4 = sw[(i >> 2) + c];
                    const float sa[4] = {sv4.x, sv4.y, sv4.z, sv4.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float diff = ((float)((w[c] >> (8 * q)) & 255u) * scale + shift) - sa[q];
                        sum = sum + diff * diff;
                    }
                }
                cur = nxt;
            }
        } else if (U8IN && (dim & 3) == 0) {
            const unsigned int* xw = (const unsigned int*)xv;   // rows are 4-byte aligned when dim % 4 == 0
            for (; i + 4 <= dim; i += 4) {
                const unsigned int w4 = xw[i >> 2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float diff = ((float)((w4 >> (8 * q)) & 255u) * scale + shift) - s[i + q];
                    sum = sum + diff * diff;
                }
            }
        }
        for (; i < dim; ++i) {
            const float diff = X(i) - s[i];
            sum = sum + diff * diff;
        }
        return exp(-m.p0 * (double)sum);
    }
    if (m.kernel == FD_KERNEL_HIK) {
        float sum = 0.f;
        for (int i = 0; i < dim; ++i) sum = sum + fminf(X(i), s[i]);
        return (double)sum;
    }
    double dot = 0.0;
    for (int i = 0; i < dim; ++i) dot = dot + (double)X(i) * (double)s[i];
    if (m.kernel == FD_KERNEL_LINEAR) return dot;
    return powi_d(m.p0 * dot + m.p1, m.degree);
}

// One pass over the levels [k0, k1).  inq == NULL: the items are the windows 0..n0-1 (first pass).
template <bool U8IN>
__global__ __launch_bounds__(256) void k_rvm_pass(RvmDev m, const void* __restrict__ feats, int64_t rowBytes, float scale, float shift,
                                                  int64_t n0, const RvmRec* __restrict__ inq, const unsigned int* __restrict__ in_count,
                                                  int k0, int k1, RvmRec* __restrict__ outq, unsigned int* __restrict__ out_count,
                                                  int32_t* __restrict__ all_level, double* __restrict__ all_dist,
                                                  RvmRec* __restrict__ pos, unsigned int* __restrict__ pos_count, unsigned int pos_cap) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t n = inq ? (int64_t)*in_count : n0;
    for (int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * 64; i0 < n; i0 += (int64_t)gridDim.x * 256) {
        const int64_t item = i0 + lane;
        bool alive = item < n;
        int64_t wid = item;
        double d = 0.0;
        if (alive && inq) {
            wid = (int64_t)(((uint64_t)inq[item].wid_hi << 32) | inq[item].wid_lo);
            d = inq[item].d;
        }
        const void* x = (const char*)feats + (size_t)(alive ? wid : 0) * rowBytes;
        for (int k = k0; k < k1; ++k) {
            if (!__any(alive)) break;
            if (alive) {
                const double K = rvm_kernel_value<U8IN>(m, x, m.sv + (size_t)k * m.dim, scale, shift);
                const double term = (double)m.diag[k] * K;
                d = (k == 0) ? (-(double)m.bias) + term : d + term;      // RvmClassifier.cpp:94-112 (cached path)
                const float thr = m.thr[k];
                if (!(d >= (double)thr && k + 1 < m.numUse)) {           // :84 leaves the cascade
                    if (all_level) all_level[wid] = k;
                    if (all_dist) all_dist[wid] = d;
                    if (k + 1 == m.numUse && d >= (double)thr) {         // classify, :68-73
                        const unsigned int slot = atomicAdd(pos_count, 1u);
                        if (slot < pos_cap) pos[slot] = RvmRec{(uint32_t)wid, (uint32_t)(wid >> 32), d};
                    }
                    alive = false;
                }
            }
        }
        // survivors of this pass: wave-aggregated append
        const unsigned long long mask = __ballot(alive);
        if (mask) {
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(out_count, (unsigned int)__popcll(mask));
            base = __builtin_amdgcn_readfirstlane(base);
            if (alive) {
                const unsigned int idx = base + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull));
                outq[idx] = RvmRec{(uint32_t)wid, (uint32_t)(wid >> 32), d};
            }
        }
    }
}

// Deep part of the cascade: the few windows that survive the first levels run dozens more, and a lane == window
// mapping would leave them on a handful of wavefronts, each bound by its 400-step fp32 chains.  The kernel values of
// different levels are independent, so here a wavefront takes ONE window and evaluates 64 levels at once (lane ==
// level; the patch is wave-uniform, the reduced set vectors are read transposed), then forms the running distance
// in level order and stops at the first missed threshold.
template <bool U8IN>
__global__ __launch_bounds__(256) void k_rvm_deep(RvmDev m, const void* __restrict__ feats, int64_t rowBytes, float scale, float shift,
                                                  const RvmRec* __restrict__ inq, const unsigned int* __restrict__ in_count, int k0,
                                                  int32_t* __restrict__ all_level, double* __restrict__ all_dist,
                                                  RvmRec* __restrict__ pos, unsigned int* __restrict__ pos_count, unsigned int pos_cap) {
    __shared__ float xs[4][RVM_MAX_DIM * RVM_MAX_DIM];   // the window's vector, converted once
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t n = (int64_t)*in_count;
    const int dim = m.dim, F = m.numFilters;
    float* x = xs[wave];
    for (int64_t item = (int64_t)blockIdx.x * 4 + wave; item < n; item += (int64_t)gridDim.x * 4) {
        const int64_t wid = (int64_t)(((uint64_t)inq[item].wid_hi << 32) | inq[item].wid_lo);
        double d = inq[item].d;
        const char* row = (const char*)feats + (size_t)wid * rowBytes;
        for (int i = lane; i < dim; i += 64)
            x[i] = U8IN ? (float)((const unsigned char*)row)[i] * scale + shift : ((const float*)row)[i];
        wave_sync();
        int level = -1;
        for (int kb = k0; kb < m.numUse && level < 0; kb += 64) {
            const int k = kb + lane;
            const bool valid = k < m.numUse;
            const float* st = m.svT + (valid ? k : kb);
            double K;
            if (m.kernel == FD_KERNEL_RBF) {
                float sum = 0.f;
                for (int i = 0; i < dim; ++i) {
                    const float diff = x[i] - st[(size_t)i * F];
                    sum = sum + diff * diff;
                }
                K = exp(-m.p0 * (double)sum);
            } else if (m.kernel == FD_KERNEL_HIK) {
                float sum = 0.f;
                for (int i = 0; i < dim; ++i) sum = sum + fminf(x[i], st[(size_t)i * F]);
                K = (double)sum;
            } else {
                double dot = 0.0;
                for (int i = 0; i < dim; ++i) dot = dot + (double)x[i] * (double)st[(size_t)i * F];
                K = m.kernel == FD_KERNEL_LINEAR ? dot : powi_d(m.p0 * dot + m.p1, m.degree);
            }
            const double term = valid ? (double)m.diag[k] * K : 0.0;
            // running distance in level order (RvmClassifier.cpp:94-112); lane j keeps d_{kb+j}
            double mine = 0.0;
            const int cnt = min(64, m.numUse - kb);
            for (int j = 0; j < cnt; ++j) {
                const double t = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(term), j),
                                                  __builtin_amdgcn_readlane(__double2loint(term), j));
                d = (kb + j == 0) ? (-(double)m.bias) + t : d + t;
                mine = (lane == j) ? d : mine;
            }
            const float thr = valid ? m.thr[k] : 0.f;
            const bool leaves = valid && !(mine >= (double)thr && k + 1 < m.numUse);
            const unsigned long long lm = __ballot(leaves);
            if (lm) {
                const int e = __builtin_ctzll(lm);
                level = kb + e;
                if (lane == e) {
                    if (all_level) all_level[wid] = level;
                    if (all_dist) all_dist[wid] = mine;
                    if (level + 1 == m.numUse && mine >= (double)thr) {
                        const unsigned int slot = atomicAdd(pos_count, 1u);
                        if (slot < pos_cap) pos[slot] = RvmRec{(uint32_t)wid, (uint32_t)(wid >> 32), mine};
                    }
                }
            }
            // (d now holds the distance after the last level of the block: the entry value of the next one)
        }
        wave_sync();
    }
}

// level ranges of the passes: short at the start (most windows leave early), longer later
int pass_bounds(int numUse, int* b) {
    static const int cuts[] = {2, 6, RVM_DEEP_FROM};   // the rest runs lane == level (k_rvm_deep)
    int np = 0, k = 0;
    b[0] = 0;
    for (int c : cuts) {
        if (k >= numUse) break;
        k = std::min(c, numUse);
        b[++np] = k;
    }
    return np;
}

// runs the cascade over n vectors resident on the device (rows of rowBytes bytes); leaves counters[0] = positives
template <bool U8IN>
void run_cascade(fd_ctx* ctx, fd_rvm* m, const void* dfeats, int64_t rowBytes, float scale, float shift, int64_t n, bool want_all,
                 unsigned int pos_cap) {
    hipStream_t st = ctx->stream;
    int b[16];
    const int np = pass_bounds(m->dev.numUse, b);
    m->q0.reserve(sizeof(RvmRec) * (size_t)n);
    m->q1.reserve(sizeof(RvmRec) * (size_t)n);
    m->counters.reserve(256);
    m->pos.reserve(sizeof(RvmRec) * (size_t)std::max<unsigned int>(pos_cap, 1));
    if (want_all) {
        m->level.reserve(sizeof(int32_t) * (size_t)n);
        m->dist.reserve(sizeof(double) * (size_t)n);
    }
    HIP_CHECK(hipMemsetAsync(m->counters.p, 0, 256, st));
    unsigned int* cnt = m->counters.as<unsigned int>();   // [0] positives, [1 + p] survivors of pass p
    for (int p = 0; p < np; ++p) {
        const RvmRec* inq = p == 0 ? nullptr : (p % 2 ? m->q0.as<RvmRec>() : m->q1.as<RvmRec>());
        RvmRec* outq = p % 2 ? m->q1.as<RvmRec>() : m->q0.as<RvmRec>();
        const int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->num_cus * 8);
        hipLaunchKernelGGL(k_rvm_pass<U8IN>, dim3(grid), dim3(256), 0, st, m->dev, dfeats, rowBytes, scale, shift, n, inq,
                           p == 0 ? nullptr : cnt + p, b[p], b[p + 1], outq, cnt + 1 + p, want_all ? m->level.as<int32_t>() : nullptr,
                           want_all ? m->dist.as<double>() : nullptr, m->pos.as<RvmRec>(), cnt, pos_cap);
        HIP_CHECK(hipGetLastError());
    }
    if (m->dev.numUse > b[np]) {
        if (m->dev.dim > RVM_MAX_DIM * RVM_MAX_DIM) FD_THROW(FD_ERR_INVALID_ARGUMENT, "RvmClassifier: vectors longer than %d are not supported", RVM_MAX_DIM * RVM_MAX_DIM);
        const RvmRec* inq = np % 2 ? m->q0.as<RvmRec>() : m->q1.as<RvmRec>();
        const int grid = (int)std::min<int64_t>((n + 3) / 4, (int64_t)ctx->num_cus * 8);
        hipLaunchKernelGGL(k_rvm_deep<U8IN>, dim3(grid), dim3(256), 0, st, m->dev, dfeats, rowBytes, scale, shift, inq, cnt + np, b[np],
                           want_all ? m->level.as<int32_t>() : nullptr, want_all ? m->dist.as<double>() : nullptr, m->pos.as<RvmRec>(), cnt,
                           pos_cap);
        HIP_CHECK(hipGetLastError());
    }
}

}  // namespace

extern "C" {

int fd_rvm_create(fd_ctx* ctx, const fd_rvm_model* md, fd_rvm** out) {
    return fd_guard(ctx, [&] {
        if (!ctx || !md || !out) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_rvm_create: NULL argument");
        if (md->kernel < 0 || md->kernel > 3) FD_THROW(FD_ERR_RUNTIME, "RvmClassifier: Unsupported kernel type: %d", md->kernel);
        if (md->num_filters < 1 || md->filter_w < 1 || md->filter_h < 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "RvmClassifier: empty model");
        if (!md->support_vectors || !md->coefficients || !md->thresholds) FD_THROW(FD_ERR_INVALID_ARGUMENT, "RvmClassifier: NULL model array");
        HIP_CHECK(hipSetDevice(ctx->device));
        std::unique_ptr<fd_rvm> m(new fd_rvm());
        m->ctx = ctx;
        const int F = md->num_filters, dim = md->filter_w * md->filter_h;
        RvmDev& d = m->dev;
        std::memset(&d, 0, sizeof(d));
        d.kernel = md->kernel; d.dim = dim; d.numFilters = F;
        d.numUse = (md->num_used <= 0 || md->num_used > F) ? F : md->num_used;   // setNumFiltersToUse, RvmClassifier.cpp:119-126
        d.p0 = md->p0; d.p1 = md->p1; d.degree = (int)md->p2; d.bias = md->bias;
        std::vector<float> diag(F);
        for (int k = 0; k < F; ++k) diag[k] = md->coefficients[(size_t)k * (k + 1) / 2 + k];
        auto up = [&](DevBuf& b, const void* src, size_t bytes) {
            b.reserve(std::max<size_t>(bytes, 16));
            HIP_CHECK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
        };
        up(m->sv, md->support_vectors, sizeof(float) * (size_t)F * dim);
        {
            std::vector<float> svT((size_t)F * dim);
            for (int k = 0; k < F; ++k)
                for (int i = 0; i < dim; ++i) svT[(size_t)i * F + k] = md->support_vectors[(size_t)k * dim + i];
            up(m->svT, svT.data(), sizeof(float) * svT.size());
        }
        up(m->diag, diag.data(), sizeof(float) * F);
        up(m->thr, md->thresholds, sizeof(float) * F);
        d.sv = m->sv.as<float>(); d.svT = m->svT.as<float>(); d.diag = m->diag.as<float>(); d.thr = m->thr.as<float>();
        m->filter_w = md->filter_w; m->filter_h = md->filter_h;
        m->logisticA = md->logistic_a; m->logisticB = md->logistic_b;
        m->h_thr.assign(md->thresholds, md->thresholds + F);
        *out = m.release();
    });
}

void fd_rvm_destroy(fd_rvm* m) { delete m; }

int fd_rvm_eval_batch(fd_ctx* ctx, const fd_rvm* rvm_, const float* features, int64_t n, int32_t* out_level, double* out_distance) {
    return fd_guard(ctx, [&] {
        if (!ctx || !rvm_ || n < 0 || (n > 0 && !features)) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_rvm_eval_batch: bad argument");
        if (n == 0) return;
        fd_rvm* m = const_cast<fd_rvm*>(rvm_);
        HIP_CHECK(hipSetDevice(ctx->device));
        const size_t bytes = sizeof(float) * (size_t)n * m->dev.dim;
        m->feats.reserve(bytes);
        HIP_CHECK(hipMemcpyAsync(m->feats.p, features, bytes, hipMemcpyHostToDevice, ctx->stream));
        run_cascade<false>(ctx, m, m->feats.p, (int64_t)m->dev.dim * 4, 1.f, 0.f, n, true, 0u);
        if (out_level) HIP_CHECK(hipMemcpyAsync(out_level, m->level.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        if (out_distance) HIP_CHECK(hipMemcpyAsync(out_distance, m->dist.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

// SlidingWindowDetector::detect (SlidingWindowDetector.cpp:87-98) with a ProbabilisticRvmClassifier on the
// u8 patch feature spaces of ffpDetectApp.cpp:446-461 + ConversionFilter(CV_32F, scale, shift)
int fd_detect_rvm(fd_ctx* ctx, fd_pyramid* p, const fd_rvm* rvm_, const fd_rvm_detect_params* dp, const int* roi, fd_detection* out,
                  int64_t cap, int64_t* count, int32_t* all_level, double* all_distance) {
    return fd_guard(ctx, [&] {
        if (!ctx || !p || !rvm_ || !dp || !count) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_detect_rvm: NULL argument");
        fd_rvm* m = const_cast<fd_rvm*>(rvm_);
        if (p->ctx != ctx || m->ctx != ctx) FD_THROW(FD_ERR_INVALID_ARGUMENT, "objects belong to different contexts");
        if (p->filter_kind != FD_LAYER_NONE) FD_THROW(FD_ERR_INVALID_ARGUMENT, "RVM detection needs a gray pyramid (no layer filter)");
        fd_pyramid_require_single(p, "fd_detect_rvm");
        if (p->all.empty()) FD_THROW(FD_ERR_RUNTIME, "pyramid has not been updated with an image");
        if (dp->feature_space < FD_FEATURE_GRAY || dp->feature_space > FD_FEATURE_HISTEQ)
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "unknown feature space %d", dp->feature_space);
        const int pw = m->filter_w, ph = m->filter_h, dim = pw * ph;
        if (pw > RVM_MAX_DIM || ph > RVM_MAX_DIM) FD_THROW(FD_ERR_INVALID_ARGUMENT, "patch size must be within %d x %d", RVM_MAX_DIM, RVM_MAX_DIM);
        HIP_CHECK(hipSetDevice(ctx->device));
        std::vector<WindowLayer> wls;
        int64_t total;
        fd_enumerate_layers(p, pw, ph, dp->step_x, dp->step_y, roi, wls, total);
        *count = 0;
        if (total == 0) return;
        if (wls.size() > (size_t)RVM_MAX_LAYERS) FD_THROW(FD_ERR_INVALID_ARGUMENT, "too many pyramid layers (%zu)", wls.size());
        RvmWinTable wt;
        std::memset(&wt, 0, sizeof(wt));
        wt.sx = dp->step_x; wt.sy = dp->step_y; wt.total = total;
        for (const WindowLayer& w : wls) {
            if (w.nx == 0 || w.ny == 0) continue;
            const HostLayer& L = p->all[p->kept[w.layer]];
            RvmWinLayer& dl = wt.l[wt.n++];
            dl.bx = w.bx; dl.by = w.by; dl.nx = w.nx; dl.ny = w.ny; dl.lw = L.w; dl.off = L.gray_off; dl.first = w.first;
        }
        hipStream_t st = ctx->stream;
        m->feats.reserve((size_t)total * dim + 16);
        hipLaunchKernelGGL(k_window_patches, dim3((unsigned)std::min<int64_t>(total, (int64_t)ctx->num_cus * 32)), dim3(64), 0, st,
                           p->arena.as<uint8_t>(), wt, pw, ph, dp->feature_space, m->feats.as<uint8_t>());
        HIP_CHECK(hipGetLastError());
        const bool want_all = all_level || all_distance;
        const unsigned int pos_cap = (unsigned int)std::min<int64_t>(total, 1 << 22);
        run_cascade<true>(ctx, m, m->feats.p, (int64_t)dim, dp->conv_scale, dp->conv_shift, total, want_all, pos_cap);
        unsigned int* hcnt = (unsigned int*)fd_pinned(ctx, 64);
        HIP_CHECK(hipMemcpyAsync(hcnt, m->counters.p, 4, hipMemcpyDeviceToHost, st));
        if (all_level) HIP_CHECK(hipMemcpyAsync(all_level, m->level.p, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, st));
        if (all_distance) HIP_CHECK(hipMemcpyAsync(all_distance, m->dist.p, sizeof(double) * (size_t)total, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        const unsigned int cnt = *hcnt;
        if (cnt > pos_cap) FD_THROW(FD_ERR_DEVICE_CAPACITY, "fd_detect_rvm: %u positives exceed the device buffer", cnt);
        std::vector<RvmRec> raw(cnt);
        if (cnt) HIP_CHECK(hipMemcpy(raw.data(), m->pos.p, sizeof(RvmRec) * cnt, hipMemcpyDeviceToHost));
        auto widof = [](const RvmRec& r) { return ((uint64_t)r.wid_hi << 32) | r.wid_lo; };
        std::sort(raw.begin(), raw.end(), [&](const RvmRec& a, const RvmRec& b) { return widof(a) < widof(b); });
        *count = cnt;
        for (unsigned int i = 0; i < cnt && out && (int64_t)i < cap; ++i) {
            fd_detection d;
            std::memset(&d, 0, sizeof(d));
            fd_window_to_detection(p, wls, dp->step_x, dp->step_y, (int64_t)widof(raw[i]), d);
            d.level = m->dev.numUse - 1;
            d.positive = 1;
            d.score = (float)raw[i].d;
            d.probability = 1.0f / (1.0f + std::exp(m->logisticA + m->logisticB * raw[i].d));   // ProbabilisticRvmClassifier.cpp:62
            out[i] = d;
        }
        if (out && (int64_t)cnt > cap) FD_THROW(FD_ERR_CAPACITY, "fd_detect_rvm: %u positives, capacity %lld", cnt, (long long)cap);
    });
}

}  // extern "C"
