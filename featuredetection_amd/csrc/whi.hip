// featuredetection_amd/csrc/whi.hip -- the "whi" patch filter chain of ffpDetectApp.cpp:449-454:
//   WhiteningFilter (WhiteningFilter.cpp:20-81) -> HistogramEqualizationFilter (cv::equalizeHist) ->
//   ConversionFilter(CV_32F, 1/127.5, -1) -> UnitNormFilter(NORM_L2) (UnitNormFilter.cpp, eps 1e-4)
// and the stand-alone HistogramEqualizationFilter ("histeq" feature space, :446-448).
//
// One wavefront per patch.  The two DFTs of the whitening filter are evaluated as separable plain
// DFTs in fp64 (row pass, column pass; patch sizes are 16..32, so a pass is a 20-term sum per
// output), every lane owning complete output sums in the k order of oracle/orc_filters.cpp whi(): the
// whitened u8 image and the equalised image are bit-identical to the CPU restatement; only the final
// L2 norm is reduced across lanes.  All intermediates live in LDS (32 bytes per pixel); HBM traffic
// is the patch read (L2 hits for overlapping windows) and the 4*w*h-byte feature write.
#include "fd_internal.hpp"
#include "fd_device.hpp"
#include <algorithm>
#include <complex>
#include <cstring>
#include <memory>

struct fd_svm;
float fd_svm_threshold(const fd_svm* m);
int fd_svm_dim(const fd_svm* m);
bool fd_svm_is_u8(const fd_svm* m);
void fd_svm_generic_launch(fd_ctx* ctx, const fd_svm* m, const void* dfeat, const uint32_t* didx, int64_t stride_bytes, int64_t n, double* dout);
void fd_svm_positives_to_detections(fd_ctx* ctx, const fd_pyramid* p, const fd_svm* svm, const std::vector<WindowLayer>& wls, int sx, int sy,
                                    const double* ddist, int64_t N, fd_detection* out, int64_t cap, int64_t* count, double* all_distance);

constexpr int WHI_MAX_LAYERS = 64;
constexpr int WHI_MAX_DIM = 32;

struct WhiWinLayer {
    int32_t bx, by, nx, ny;
    int32_t lw;
    uint32_t off;
    int64_t first;
};
struct WhiWinTable {
    int32_t n, sx, sy, raw;   // raw != 0: `total` contiguous w x h patches
    int64_t total;
    WhiWinLayer l[WHI_MAX_LAYERS];
};
struct WhiDev {
    int32_t w, h;
    const double* twRow;   // [w][w] (re, im) = polar(1, -2 pi x k / w)
    const double* twCol;   // [h][h]
    const float* filt;     // [h][w]  WhiteningFilter.cpp:62-81
};

namespace {

using namespace fd_dev;

__device__ __forceinline__ const uint8_t* locate(const uint8_t* arena, const WhiWinTable& wt, int64_t wid, int w, int h, int& stride) {
    if (wt.raw) {
        stride = w;
        return arena + (size_t)wid * w * h;
    }
    int li = 0;
    for (int l = 1; l < wt.n; ++l) li = wt.l[l].first <= wid ? l : li;
    const WhiWinLayer& wl = wt.l[li];
    const int local = (int)(wid - wl.first);
    const int iy = local / wl.nx, ix = local - iy * wl.nx;
    stride = wl.lw;
    return arena + wl.off + (size_t)(wl.by + iy * wt.sy) * wl.lw + (wl.bx + ix * wt.sx);
}

// EQ_ONLY: HistogramEqualizationFilter alone, u8 output
// MODE 0: the whole whi chain -> unit-norm floats; 1: cv::equalizeHist only -> u8; 2: WhiteningFilter only -> u8
template <int MODE>
__global__ __launch_bounds__(64) void k_whi(const uint8_t* __restrict__ arena, WhiWinTable wt, WhiDev d, float* __restrict__ feat,
                                            uint8_t* __restrict__ eqOut) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int w = d.w, h = d.h, n = w * h;
    double* Are = (double*)smem;          // T, later U
    double* Aim = Are + n;
    double* Bre = Aim + n;                // filtered spectrum
    double* Bim = Bre + n;
    int* hist = (int*)(Bim + n);
    int* lut = hist + 256;
    unsigned char* px = (unsigned char*)(lut + 256);   // [n] input patch, later the whitened u8 image
    unsigned char* eq = px + ((n + 15) & ~15);         // [n]
    for (int64_t wid = blockIdx.x; wid < wt.total; wid += gridDim.x) {
        int stride;
        const uint8_t* src = locate(arena, wt, wid, w, h, stride);
        for (int i = lane; i < n; i += 64) {
            const int y = i / w, x = i - y * w;
            px[i] = src[(size_t)y * stride + x];
        }
        wave_sync();
        if (MODE != 1) {
            // forward DFT (DFT_SCALE | DFT_COMPLEX_OUTPUT): rows
            for (int o = lane; o < n; o += 64) {
                const int v = o / w, x = o - v * w;
                const double* tw = d.twRow + (size_t)x * w * 2;
                double sr = 0, si = 0;
                for (int k = 0; k < w; ++k) {
                    const double p = (double)px[v * w + k];
                    sr = sr + p * tw[2 * k];
                    si = si + p * tw[2 * k + 1];
                }
                Are[o] = sr;
                Aim[o] = si;
            }
            wave_sync();
            // columns, 1/n, float spectrum, whitening filter in float (WhiteningFilter.cpp:38-45)
            for (int o = lane; o < n; o += 64) {
                const int y = o / w, u = o - y * w;
                const double* tw = d.twCol + (size_t)y * h * 2;
                double sr = 0, si = 0;
                for (int k = 0; k < h; ++k) {
                    const double ar = Are[k * w + u], ai = Aim[k * w + u];
                    const double br = tw[2 * k], bi = tw[2 * k + 1];
                    sr = sr + (ar * br - ai * bi);
                    si = si + (ar * bi + ai * br);
                }
                const float f = d.filt[o];
                Bre[o] = (double)((float)(sr / n) * f);
                Bim[o] = (double)((float)(si / n) * f);
            }
            wave_sync();
            // inverse DFT (DFT_INVERSE | DFT_REAL_OUTPUT): rows over the conjugate-symmetric completion of the half spectrum
            for (int o = lane; o < n; o += 64) {
                const int v = o / w, x = o - v * w;
                const double* tw = d.twRow + (size_t)x * w * 2;
                const int rr = (h - v) % h;
                double sr = 0, si = 0;
                for (int u = 0; u < w; ++u) {
                    const int cc = (w - u) % w;
                    const bool own = u < cc || (u == cc && v <= rr);
                    const double gr = own ? Bre[v * w + u] : Bre[rr * w + cc];
                    const double gi = own ? Bim[v * w + u] : -Bim[rr * w + cc];
                    const double br = tw[2 * u], bi = -tw[2 * u + 1];
                    sr = sr + (gr * br - gi * bi);
                    si = si + (gr * bi + gi * br);
                }
                Are[o] = sr;
                Aim[o] = si;
            }
            wave_sync();
            // columns (real part), convertTo(CV_8U, 1, 127)
            for (int o = lane; o < n; o += 64) {
                const int y = o / w, x = o - y * w;
                const double* tw = d.twCol + (size_t)y * h * 2;
                double sr = 0;
                for (int v = 0; v < h; ++v) {
                    const double br = tw[2 * v], bi = -tw[2 * v + 1];
                    sr = sr + (Are[v * w + x] * br - Aim[v * w + x] * bi);
                }
                const float val = (float)sr;
                px[o] = sat_u8_d((double)(val * 1.0f + 127.0f));
            }
            wave_sync();
        }
        if (MODE == 2) {   // WhiteningFilter::applyTo alone (WhiteningFilter.cpp:20-58): the whitened u8 image
            for (int i = lane; i < n; i += 64) eqOut[(size_t)wid * n + i] = px[i];
            wave_sync();
            continue;
        }
        equalize_hist_wave(px, eq, n, hist, lut, lane);
        if (MODE == 1) {
            for (int i = lane; i < n; i += 64) eqOut[(size_t)wid * n + i] = eq[i];
        } else {
            const float a = (float)(1.0 / 127.5), b = -1.0f;
            double part = 0;
            for (int i = lane; i < n; i += 64) {
                const float v = (float)eq[i] * a + b;
                part += (double)v * v;
            }
            for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
            const double norm = sqrt(part);
            const float eps = 1e-4f;
            const float inv = (float)(1.0 / (norm + eps));
            for (int i = lane; i < n; i += 64) {
                const float v = (float)eq[i] * a + b;
                feat[(size_t)wid * n + i] = v * inv;
            }
        }
        wave_sync();
    }
}

struct WhiScratch {
    DevBuf tables, in, feat, dist;
    int w = 0, h = 0;
    float alpha = 0, cutoff = 0;
    bool valid = false;
    WhiDev dev;
};
WhiScratch& scratch(fd_ctx* ctx) { return fd_scratch<WhiScratch>(ctx); }

// twiddles + whitening filter (same expressions as oracle/orc_filters.cpp whi_tables: both sides evaluate
// std::polar / powf / expf with the host libm)
const WhiDev& tables(fd_ctx* ctx, WhiScratch& S, int w, int h, float alpha, float cutoff) {
    if (w < 2 || h < 2 || w > WHI_MAX_DIM || h > WHI_MAX_DIM) FD_THROW(FD_ERR_INVALID_ARGUMENT, "patch size must be within 2..%d", WHI_MAX_DIM);
    if (S.valid && S.w == w && S.h == h && S.alpha == alpha && S.cutoff == cutoff) return S.dev;
    const double PI2 = 6.283185307179586476925286766559;
    std::vector<double> twRow((size_t)w * w * 2), twCol((size_t)h * h * 2);
    std::vector<float> filt((size_t)w * h);
    for (int x = 0; x < w; ++x)
        for (int k = 0; k < w; ++k) {
            std::complex<double> t = std::polar(1.0, -PI2 * x * k / w);
            twRow[((size_t)x * w + k) * 2] = t.real();
            twRow[((size_t)x * w + k) * 2 + 1] = t.imag();
        }
    for (int y = 0; y < h; ++y)
        for (int k = 0; k < h; ++k) {
            std::complex<double> t = std::polar(1.0, -PI2 * y * k / h);
            twCol[((size_t)y * h + k) * 2] = t.real();
            twCol[((size_t)y * h + k) * 2 + 1] = t.imag();
        }
    for (int row = 0; row < h; ++row)
        for (int col = 0; col < w; ++col) {
            int shiftedRow = (row + h / 2) % h, shiftedCol = (col + w / 2) % w;
            float fx = -0.5f + shiftedCol * (2 * 0.5f) / (w - 1);
            float fy = -0.5f + shiftedRow * (2 * 0.5f) / (h - 1);
            float rho = std::sqrt(fx * fx + fy * fy);
            float f = std::pow(rho, alpha);
            if (cutoff > 0) f *= std::exp(-std::pow(rho / cutoff, 4));
            filt[(size_t)row * w + col] = f;
        }
    const size_t b0 = sizeof(double) * twRow.size(), b1 = sizeof(double) * twCol.size(), b2 = sizeof(float) * filt.size();
    S.tables.reserve(b0 + b1 + b2);
    HIP_CHECK(hipMemcpy(S.tables.p, twRow.data(), b0, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy((char*)S.tables.p + b0, twCol.data(), b1, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy((char*)S.tables.p + b0 + b1, filt.data(), b2, hipMemcpyHostToDevice));
    S.dev.w = w; S.dev.h = h;
    S.dev.twRow = (const double*)S.tables.p;
    S.dev.twCol = (const double*)((char*)S.tables.p + b0);
    S.dev.filt = (const float*)((char*)S.tables.p + b0 + b1);
    S.w = w; S.h = h; S.alpha = alpha; S.cutoff = cutoff; S.valid = true;
    return S.dev;
}

size_t lds_bytes(int w, int h) {
    const size_t n = (size_t)w * h;
    return 4 * n * sizeof(double) + 512 * sizeof(int) + 2 * ((n + 15) & ~(size_t)15);
}

template <int MODE>
void launch(fd_ctx* ctx, const uint8_t* arena, const WhiWinTable& wt, const WhiDev& d, float* feat, uint8_t* eqOut) {
    const size_t lds = lds_bytes(d.w, d.h);
    static uint64_t lds_allowed = 0;   // per instantiation
    fd_allow_lds(ctx, (const void*)k_whi<MODE>, 64 * 1024, lds_allowed);
    const int grid = (int)std::min<int64_t>(wt.total, (int64_t)ctx->num_cus * 16);
    hipLaunchKernelGGL(k_whi<MODE>, dim3(grid), dim3(64), lds, ctx->stream, arena, wt, d, feat, eqOut);
    HIP_CHECK(hipGetLastError());
}

// ConversionFilter::applyTo (cv::Mat::convertTo, OpenCV 2.4 cvtScale_<T, DT, float>): value * (float)alpha + (float)beta in float,
// then saturate_cast<uchar>(cvRound(.)) for a CV_8U destination
__global__ void k_convert(const void* __restrict__ src, int srcF32, void* __restrict__ dst, int dstF32, int64_t n, double alpha, double beta) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (dstF32) {
            const float v = srcF32 ? ((const float*)src)[i] : (float)((const uint8_t*)src)[i];
            ((float*)dst)[i] = v * (float)alpha + (float)beta;
        } else {
            const float v = srcF32 ? ((const float*)src)[i] : (float)((const uint8_t*)src)[i];
            ((uint8_t*)dst)[i] = sat_u8_d((double)(v * (float)alpha + (float)beta));
        }
    }
}

// UnitNormFilter::applyTo (UnitNormFilter.cpp:24-43): image / (cv::norm(image, normType) + 1e-4f) per image, one wavefront each;
// the norm is accumulated in double over the float values like cv::norm
__global__ __launch_bounds__(64) void k_unit_norm(const float* __restrict__ src, float* __restrict__ dst, int64_t nimg, int len, int normType) {
    const int lane = threadIdx.x;
    for (int64_t im = blockIdx.x; im < nimg; im += gridDim.x) {
        const float* s = src + (size_t)im * len;
        double part = 0;
        for (int i = lane; i < len; i += 64) {
            const double v = (double)s[i];
            if (normType == 4) part += v * v;             // cv::NORM_L2
            else if (normType == 2) part += fabs(v);      // cv::NORM_L1
            else part = fmax(part, fabs(v));              // cv::NORM_INF
        }
        for (int o = 32; o > 0; o >>= 1) {
            const double other = __shfl_xor(part, o, 64);
            part = (normType == 4 || normType == 2) ? part + other : fmax(part, other);
        }
        const double norm = normType == 4 ? sqrt(part) : part;
        const float inv = (float)(1.0 / (norm + (double)1e-4f));   // image / (norm + eps): MatExpr scale 1/d, applied in float
        for (int i = lane; i < len; i += 64) dst[(size_t)im * len + i] = s[i] * inv;
    }
}

void build_table(const fd_pyramid* p, const fd_whi_params* wp, WhiWinTable& wt, std::vector<WindowLayer>& wls) {
    if (p->filter_kind != FD_LAYER_NONE) FD_THROW(FD_ERR_INVALID_ARGUMENT, "the whi chain needs a gray pyramid (no layer filter)");
    fd_pyramid_require_single(p, "the whi chain");
    if (p->all.empty()) FD_THROW(FD_ERR_RUNTIME, "pyramid has not been updated with an image");
    int64_t total;
    fd_enumerate_layers(p, wp->patch_w, wp->patch_h, wp->step_x, wp->step_y, nullptr, wls, total);
    if (wls.size() > (size_t)WHI_MAX_LAYERS) FD_THROW(FD_ERR_INVALID_ARGUMENT, "too many pyramid layers (%zu)", wls.size());
    std::memset(&wt, 0, sizeof(wt));
    wt.sx = wp->step_x; wt.sy = wp->step_y; wt.total = total;
    for (const WindowLayer& w : wls) {
        if (w.nx == 0 || w.ny == 0) continue;
        const HostLayer& L = p->all[p->kept[w.layer]];
        WhiWinLayer& dl = wt.l[wt.n++];
        dl.bx = w.bx; dl.by = w.by; dl.nx = w.nx; dl.ny = w.ny; dl.lw = L.w; dl.off = L.gray_off; dl.first = w.first;
    }
}

// whi features of every window into S.feat ([N][w*h] floats); returns N
int64_t run_whi(fd_ctx* ctx, fd_pyramid* p, const fd_whi_params* wp, std::vector<WindowLayer>& wls, WhiScratch& S) {
    if (p->ctx != ctx) FD_THROW(FD_ERR_INVALID_ARGUMENT, "objects belong to different contexts");
    HIP_CHECK(hipSetDevice(ctx->device));
    const WhiDev& d = tables(ctx, S, wp->patch_w, wp->patch_h, wp->alpha, wp->cutoff);
    WhiWinTable wt;
    build_table(p, wp, wt, wls);
    if (wt.total == 0) return 0;
    S.feat.reserve(sizeof(float) * (size_t)wt.total * d.w * d.h);
    launch<0>(ctx, p->arena.as<uint8_t>(), wt, d, S.feat.as<float>(), nullptr);
    return wt.total;
}

}  // namespace

extern "C" {

int fd_whi_batch(fd_ctx* ctx, const uint8_t* patches, int64_t n, int w, int h, float alpha, float cutoff, float* dst) {
    return fd_guard(ctx, [&] {
        if (!ctx || n < 0 || (n > 0 && (!patches || !dst))) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_whi_batch: bad argument");
        if (n == 0) return;
        HIP_CHECK(hipSetDevice(ctx->device));
        WhiScratch& S = scratch(ctx);
        const WhiDev& d = tables(ctx, S, w, h, alpha, cutoff);
        const size_t bytes = (size_t)n * w * h;
        S.in.reserve(bytes);
        S.feat.reserve(bytes * sizeof(float));
        HIP_CHECK(hipMemcpyAsync(S.in.p, patches, bytes, hipMemcpyHostToDevice, ctx->stream));
        WhiWinTable wt;
        std::memset(&wt, 0, sizeof(wt));
        wt.raw = 1;
        wt.total = n;
        launch<0>(ctx, S.in.as<uint8_t>(), wt, d, S.feat.as<float>(), nullptr);
        HIP_CHECK(hipMemcpyAsync(dst, S.feat.p, bytes * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int fd_equalize_hist_batch(fd_ctx* ctx, const uint8_t* patches, int64_t n, int w, int h, uint8_t* dst) {
    return fd_guard(ctx, [&] {
        if (!ctx || n < 0 || (n > 0 && (!patches || !dst))) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_equalize_hist_batch: bad argument");
        if (n == 0) return;
        if (w < 1 || h < 1 || w > WHI_MAX_DIM || h > WHI_MAX_DIM) FD_THROW(FD_ERR_INVALID_ARGUMENT, "patch size must be within 1..%d", WHI_MAX_DIM);
        HIP_CHECK(hipSetDevice(ctx->device));
        WhiScratch& S = scratch(ctx);
        const size_t bytes = (size_t)n * w * h;
        S.in.reserve(bytes);
        S.feat.reserve(bytes);
        HIP_CHECK(hipMemcpyAsync(S.in.p, patches, bytes, hipMemcpyHostToDevice, ctx->stream));
        WhiWinTable wt;
        std::memset(&wt, 0, sizeof(wt));
        wt.raw = 1;
        wt.total = n;
        WhiDev d;
        std::memset(&d, 0, sizeof(d));
        d.w = w; d.h = h;
        launch<1>(ctx, S.in.as<uint8_t>(), wt, d, nullptr, S.feat.as<uint8_t>());
        HIP_CHECK(hipMemcpyAsync(dst, S.feat.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

// WhiteningFilter::applyTo alone (WhiteningFilter.cpp:20-58) on n contiguous w x h u8 patches -> u8 (convertTo(CV_8U, 1, 127))
int fd_whitening_batch(fd_ctx* ctx, const uint8_t* patches, int64_t n, int w, int h, float alpha, float cutoff, uint8_t* dst) {
    return fd_guard(ctx, [&] {
        if (!ctx || n < 0 || (n > 0 && (!patches || !dst))) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_whitening_batch: bad argument");
        if (n == 0) return;
        HIP_CHECK(hipSetDevice(ctx->device));
        WhiScratch& S = scratch(ctx);
        const WhiDev& d = tables(ctx, S, w, h, alpha, cutoff);
        const size_t bytes = (size_t)n * w * h;
        S.in.reserve(bytes);
        S.feat.reserve(bytes);
        HIP_CHECK(hipMemcpyAsync(S.in.p, patches, bytes, hipMemcpyHostToDevice, ctx->stream));
        WhiWinTable wt;
        std::memset(&wt, 0, sizeof(wt));
        wt.raw = 1;
        wt.total = n;
        launch<2>(ctx, S.in.as<uint8_t>(), wt, d, nullptr, S.feat.as<uint8_t>());
        HIP_CHECK(hipMemcpyAsync(dst, S.feat.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int fd_convert_batch(fd_ctx* ctx, const void* src, int src_dtype, int64_t count, double alpha, double beta, void* dst, int dst_dtype) {
    return fd_guard(ctx, [&] {
        if (!ctx || count < 0 || (count > 0 && (!src || !dst))) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_convert_batch: bad argument");
        if ((src_dtype != FD_DTYPE_U8 && src_dtype != FD_DTYPE_F32) || (dst_dtype != FD_DTYPE_U8 && dst_dtype != FD_DTYPE_F32))
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "ConversionFilter: CV_8U and CV_32F are supported on this backend");
        if (count == 0) return;
        HIP_CHECK(hipSetDevice(ctx->device));
        WhiScratch& S = scratch(ctx);
        const size_t sb = (size_t)count * (src_dtype == FD_DTYPE_F32 ? 4 : 1), db = (size_t)count * (dst_dtype == FD_DTYPE_F32 ? 4 : 1);
        S.in.reserve(sb);
        S.feat.reserve(db);
        HIP_CHECK(hipMemcpyAsync(S.in.p, src, sb, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_convert, dim3((unsigned)std::min<int64_t>((count + 255) / 256, 4096)), dim3(256), 0, ctx->stream, S.in.p,
                           src_dtype == FD_DTYPE_F32, S.feat.p, dst_dtype == FD_DTYPE_F32, count, alpha, beta);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(dst, S.feat.p, db, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int fd_unit_norm_batch(fd_ctx* ctx, const float* src, int64_t n, int len, int norm_type, float* dst) {
    return fd_guard(ctx, [&] {
        if (!ctx || n < 0 || len < 1 || (n > 0 && (!src || !dst))) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_unit_norm_batch: bad argument");
        if (norm_type != 1 && norm_type != 2 && norm_type != 4) FD_THROW(FD_ERR_INVALID_ARGUMENT, "UnitNormFilter: norm type must be NORM_INF (1), NORM_L1 (2) or NORM_L2 (4)");
        if (n == 0) return;
        HIP_CHECK(hipSetDevice(ctx->device));
        WhiScratch& S = scratch(ctx);
        const size_t bytes = sizeof(float) * (size_t)n * len;
        S.in.reserve(bytes);
        S.feat.reserve(bytes);
        HIP_CHECK(hipMemcpyAsync(S.in.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_unit_norm, dim3((unsigned)std::min<int64_t>(n, 65535)), dim3(64), 0, ctx->stream, S.in.as<float>(), S.feat.as<float>(), n, len, norm_type);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(dst, S.feat.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int fd_extract_whi(fd_ctx* ctx, fd_pyramid* p, const fd_whi_params* wp, float* features, int64_t cap_windows, int64_t* count) {
    return fd_guard(ctx, [&] {
        if (!ctx || !p || !wp || !count) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_extract_whi: NULL argument");
        std::vector<WindowLayer> wls;
        if (!features) {
            int64_t total;
            fd_enumerate_layers(p, wp->patch_w, wp->patch_h, wp->step_x, wp->step_y, nullptr, wls, total);
            *count = total;
            return;
        }
        WhiScratch& S = scratch(ctx);
        const int64_t N = run_whi(ctx, p, wp, wls, S);
        *count = N;
        if (N == 0) return;
        if (N > cap_windows) FD_THROW(FD_ERR_CAPACITY, "fd_extract_whi: %lld windows, capacity %lld", (long long)N, (long long)cap_windows);
        HIP_CHECK(hipMemcpyAsync(features, S.feat.p, sizeof(float) * (size_t)N * wp->patch_w * wp->patch_h, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

// SlidingWindowDetector::detect with the whi feature space and a ProbabilisticSvmClassifier on the f32
// vectors (ffpDetectApp.cpp:446-500: featurespace "whi", classifier "psvm")
int fd_detect_whi_svm(fd_ctx* ctx, fd_pyramid* p, const fd_svm* svm, const fd_whi_params* wp, fd_detection* out, int64_t cap,
                      int64_t* count, double* all_distance) {
    return fd_guard(ctx, [&] {
        if (!ctx || !p || !svm || !wp || !count) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_detect_whi_svm: NULL argument");
        if (fd_svm_dim(svm) != wp->patch_w * wp->patch_h || fd_svm_is_u8(svm))
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "SVM must work on f32 vectors of length %d", wp->patch_w * wp->patch_h);
        WhiScratch& S = scratch(ctx);
        std::vector<WindowLayer> wls;
        const int64_t N = run_whi(ctx, p, wp, wls, S);
        *count = 0;
        if (N == 0) return;
        const int dlen = wp->patch_w * wp->patch_h;
        S.dist.reserve(sizeof(double) * (size_t)N);
        fd_svm_generic_launch(ctx, svm, S.feat.p, nullptr, (int64_t)dlen * 4, N, S.dist.as<double>());
        fd_svm_positives_to_detections(ctx, p, svm, wls, wp->step_x, wp->step_y, S.dist.as<double>(), N, out, cap, count, all_distance);
    });
}

}  // extern "C"
