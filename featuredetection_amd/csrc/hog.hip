// featuredetection_amd/csrc/hog.hip -- HOG patch features (HogFilter.cpp:58-122 on top of
// HistogramFilter.cpp:23-197, non-interpolating path) for every sliding window, and the
// single-stage HOG + RBF-SVM detector of BASELINE config 2 (BenchmarkRunner.cpp:118-126,235-242).
//
// One wavefront per window.  The (bin, weight) layer image produced by the pyramid's
// GradientFilter + GradientBinningFilter layer filter is read through LDS; lane e owns one
// (cell, bin) accumulator and walks its cell's pixels in the reference's row-major order, so the
// fp32 cell histograms, cell energies, block normalisers and the final block vectors are
// bit-identical to the CPU path.  Features are written once, in the fragment-major layout the
// MFMA SVM kernel consumes (svm.hip), together with |x|^2.
// Algorithmic HBM bytes per window: 2*pw*ph (layer read, mostly L2 hits) + 4*F (feature write).
#include "fd_internal.hpp"
#include <algorithm>
#include <cstring>
#include <memory>

constexpr int HOG_MAX_LAYERS = 64;

struct HogWinLayer {
    int32_t bx, by, nx, ny;
    int32_t lw;          // layer width in pixels
    uint32_t off;        // byte offset of the filtered (bin, weight) layer
    int64_t first;
};
struct HogWinTable {
    int32_t n, sx, sy, pad;
    int64_t total;
    HogWinLayer l[HOG_MAX_LAYERS];
};
struct HogDev {
    int32_t pw, ph, bins, cell, block, sau;
    int32_t rows, cols;      // cell grid
    int32_t brows, bcols;    // block grid
    int32_t perBlock;        // floats per block
    int32_t F;               // feature length
    int32_t KP;              // padded length for the fragment layout
};

// from svm.hip
struct fd_svm;
bool fd_svm_has_mfma_path(const fd_svm* m);
int fd_svm_KP(const fd_svm* m);
float fd_svm_threshold(const fd_svm* m);
double fd_svm_probability(const fd_svm* m, double d);
void fd_svm_rbf_mfma_launch(fd_ctx* ctx, const fd_svm* m, const float* xFrag, const float* xx, int64_t npatches, double* out);
void fd_svm_generic_launch(fd_ctx* ctx, const fd_svm* m, const void* dfeat, const uint32_t* didx, int64_t stride_bytes, int64_t n, double* dout);

namespace {

constexpr int HOG_MAX_PIX = 32 * 32;
constexpr int HOG_MAX_HIST = 1024;   // cells * bins
constexpr int HOG_MAX_CELLS = 64;
constexpr int HOG_MAX_BLOCKS = 64;

struct __attribute__((aligned(16))) HogLds {
    unsigned short px[HOG_MAX_PIX];   // bin | weight << 8
    float hist[HOG_MAX_HIST];
    float energy[HOG_MAX_CELLS];
    float norm[HOG_MAX_BLOCKS];
};

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ size_t frag_index_dev(int64_t row, int k, int KP) {
    const int64_t tile = row >> 5;
    const int r = (int)(row & 31), q = k >> 3, h = (k >> 2) & 1, t = k & 3;
    return (size_t)tile * 32 * KP + (size_t)q * 256 + (size_t)(h * 32 + r) * 4 + t;
}

template <bool FRAG>
__global__ __launch_bounds__(128) void k_hog_features(const uint8_t* __restrict__ arena, HogWinTable wt, HogDev hp,
                                                      float* __restrict__ feat, float* __restrict__ xx) {
    __shared__ HogLds lds[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    HogLds& L = lds[wave];
    const int pw = hp.pw, ph = hp.ph, B = hp.bins;
    const int ncells = hp.rows * hp.cols, nhist = ncells * B, nblocks = hp.brows * hp.bcols;
    const float factor = 1.f / 255.f;
    const float eps = 1e-4f;
    const int64_t nwaves = (int64_t)gridDim.x * 2;
    for (int64_t wid = (int64_t)blockIdx.x * 2 + wave; wid < wt.total; wid += nwaves) {
        int li;
        {
            bool le = lane < wt.n && wt.l[lane < wt.n ? lane : 0].first <= wid;
            li = __popcll(__ballot(le)) - 1;
        }
        li = __builtin_amdgcn_readfirstlane(li);
        const HogWinLayer wl = wt.l[li];
        const int local = (int)(wid - wl.first);
        const int iy = local / wl.nx, ix = local - iy * wl.nx;
        const int lx = wl.bx + ix * wt.sx, ly = wl.by + iy * wt.sy;
        const unsigned short* src = (const unsigned short*)(arena + wl.off) + (size_t)ly * wl.lw + lx;
        // stage the window's (bin, weight) pixels: two rows per wave instruction
        {
            const int half = lane >> 5, col = lane & 31;
            for (int r = half; r < ph; r += 2)
                if (col < pw) L.px[r * pw + col] = src[(size_t)r * wl.lw + col];
        }
        wave_sync();
        // cell histograms: lane e = (cell, bin); sequential fp32 adds in row-major pixel order
        for (int e = lane; e < nhist; e += 64) {
            const int cell = e / B, bin = e - cell * B;
            const int cr = cell / hp.cols, cc = cell - cr * hp.cols;
            const int startRow = (cr * ph) / hp.rows, endRow = ((cr + 1) * ph) / hp.rows;
            const int startCol = (cc * pw) / hp.cols, endCol = ((cc + 1) * pw) / hp.cols;
            float h = 0.f;
            for (int y = startRow; y < endRow; ++y)
                for (int x = startCol; x < endCol; ++x) {
                    const unsigned int v = L.px[y * pw + x];
                    const float wgt = factor * (float)(v >> 8);
                    if ((int)(v & 255u) == bin) h = h + wgt;
                }
            L.hist[e] = h;
        }
        wave_sync();
        // cell energies (HogFilter.cpp:102-122)
        for (int c = lane; c < ncells; c += 64) {
            const float* hv = L.hist + c * B;
            float energy = 0.f;
            if (hp.sau) {
                const int hb = B / 2;
                for (int b = 0; b < hb; ++b) { const float uw = hv[b] + hv[hb + b]; energy = energy + uw * uw; }
            } else {
                for (int b = 0; b < B; ++b) energy = energy + hv[b] * hv[b];
            }
            L.energy[c] = energy;
        }
        wave_sync();
        // block normalisers (HogFilter.cpp:78-84)
        for (int bl = lane; bl < nblocks; bl += 64) {
            const int br = bl / hp.bcols, bc = bl - br * hp.bcols;
            float energy = 0.f;
            for (int cr = br; cr < br + hp.block; ++cr)
                for (int cc = bc; cc < bc + hp.block; ++cc) energy = energy + L.energy[cr * hp.cols + cc];
            L.norm[bl] = 1.f / sqrtf(energy + eps);
        }
        wave_sync();
        // block vectors (HogFilter.cpp:85-97): one output element per lane
        const int hb = B / 2;
        const int perCell = hp.sau ? B + hb : B;
        float sq = 0.f;
        for (int o = lane; o < hp.KP; o += 64) {
            float val = 0.f;
            if (o < hp.F) {
                const int bl = o / hp.perBlock, w = o - bl * hp.perBlock;
                const int ci = w / perCell, b = w - ci * perCell;
                const int br = bl / hp.bcols, bc = bl - br * hp.bcols;
                const int cr = br + ci / hp.block, cc = bc + (ci - (ci / hp.block) * hp.block);
                const float* hv = L.hist + (cr * hp.cols + cc) * B;
                const float nrm = L.norm[bl];
                val = b < B ? nrm * hv[b] : nrm * (hv[b - B] + hv[hb + b - B]);
            }
            sq += val * val;
            if (FRAG) feat[frag_index_dev(wid, o, hp.KP)] = val;
            else if (o < hp.F) feat[(size_t)wid * hp.F + o] = val;
        }
        if (FRAG) {
            for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o, 64);
            if (lane == 0) xx[wid] = sq;
        }
        wave_sync();
    }
}

struct HogPos {
    uint32_t wid_lo, wid_hi;
    double dist;
};

__global__ void k_select_positives(const double* __restrict__ dist, int64_t n, float threshold, HogPos* __restrict__ pos,
                                   unsigned int* __restrict__ count, unsigned int cap) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double dv = dist[i];
        if (dv >= (double)threshold) {  // SvmClassifier.cpp:44-46
            unsigned int s = atomicAdd(count, 1u);
            if (s < cap) pos[s] = HogPos{(uint32_t)i, (uint32_t)(i >> 32), dv};
        }
    }
}

struct HogScratch {
    DevBuf feat, xx, dist, pos, counter;
};
HogScratch& scratch(fd_ctx* ctx) {
    static thread_local std::vector<std::pair<fd_ctx*, std::unique_ptr<HogScratch>>> tab;
    for (auto& kv : tab)
        if (kv.first == ctx) return *kv.second;
    tab.emplace_back(ctx, std::unique_ptr<HogScratch>(new HogScratch()));
    return *tab.back().second;
}

HogDev make_hogdev(const fd_hog_params* hp, int KPwant) {
    if (!hp) FD_THROW(FD_ERR_INVALID_ARGUMENT, "NULL hog parameters");
    if (hp->bins <= 0) FD_THROW(FD_ERR_INVALID_ARGUMENT, "HogFilter: binCount must be greater than zero");
    if (hp->cell_size <= 0) FD_THROW(FD_ERR_INVALID_ARGUMENT, "HogFilter: cellSize must be greater than zero");
    if (hp->block_size <= 0) FD_THROW(FD_ERR_INVALID_ARGUMENT, "HogFilter: blockSize must be greater than zero");
    if (hp->signed_and_unsigned && hp->bins % 2 != 0)
        FD_THROW(FD_ERR_INVALID_ARGUMENT, "HogFilter: the bin size must be even for signed and unsigned gradients to be combined");
    if (hp->patch_w < 1 || hp->patch_h < 1 || hp->patch_w > 32 || hp->patch_h > 32)
        FD_THROW(FD_ERR_INVALID_ARGUMENT, "patch size must be within 1..32 on this backend");
    HogDev d;
    d.pw = hp->patch_w; d.ph = hp->patch_h; d.bins = hp->bins; d.cell = hp->cell_size; d.block = hp->block_size;
    d.sau = hp->signed_and_unsigned ? 1 : 0;
    d.rows = fd_cvRound((double)hp->patch_h / (double)hp->cell_size);
    d.cols = fd_cvRound((double)hp->patch_w / (double)hp->cell_size);
    d.brows = d.rows - d.block + 1;
    d.bcols = d.cols - d.block + 1;
    if (d.rows < 1 || d.cols < 1 || d.brows < 1 || d.bcols < 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "HogFilter: patch smaller than one block");
    d.perBlock = d.block * d.block * (d.sau ? d.bins + d.bins / 2 : d.bins);
    d.F = d.brows * d.bcols * d.perBlock;
    if (d.rows * d.cols > HOG_MAX_CELLS || d.rows * d.cols * d.bins > HOG_MAX_HIST || d.brows * d.bcols > HOG_MAX_BLOCKS)
        FD_THROW(FD_ERR_INVALID_ARGUMENT, "HogFilter: too many cells/bins for this backend");
    d.KP = KPwant > 0 ? KPwant : ((d.F + 7) & ~7);
    return d;
}

void build_table(const fd_pyramid* p, const fd_hog_params* hp, HogWinTable& wt, std::vector<WindowLayer>& wls) {
    if (p->filter_kind != FD_LAYER_GRADBIN || p->interpolate)
        FD_THROW(FD_ERR_INVALID_ARGUMENT, "HOG extraction needs a pyramid with the FD_LAYER_GRADBIN layer filter (no bin interpolation)");
    if (p->bins != hp->bins) FD_THROW(FD_ERR_INVALID_ARGUMENT, "HogFilter bins (%d) differ from the layer filter bins (%d)", hp->bins, p->bins);
    if (p->all.empty()) FD_THROW(FD_ERR_RUNTIME, "pyramid has not been updated with an image");
    int64_t total;
    fd_enumerate_layers(p, hp->patch_w, hp->patch_h, hp->step_x, hp->step_y, nullptr, wls, total);
    if (wls.size() > (size_t)HOG_MAX_LAYERS) FD_THROW(FD_ERR_INVALID_ARGUMENT, "too many pyramid layers (%zu)", wls.size());
    std::memset(&wt, 0, sizeof(wt));
    wt.sx = hp->step_x; wt.sy = hp->step_y; wt.total = total;
    for (const WindowLayer& w : wls) {
        if (w.nx == 0 || w.ny == 0) continue;
        const HostLayer& L = p->all[p->kept[w.layer]];
        HogWinLayer& dl = wt.l[wt.n++];
        dl.bx = w.bx; dl.by = w.by; dl.nx = w.nx; dl.ny = w.ny; dl.lw = L.w; dl.off = L.filt_off; dl.first = w.first;
    }
}

void window_geometry(const fd_pyramid* p, const std::vector<WindowLayer>& wls, int sx, int sy, int64_t wid, fd_detection& d) {
    size_t i = 0;
    while (i + 1 < wls.size() && (wls[i + 1].first <= wid)) ++i;
    while (wls[i].nx == 0 || wls[i].ny == 0) --i;
    const WindowLayer& w = wls[i];
    const HostLayer& L = p->all[p->kept[w.layer]];
    int64_t local = wid - w.first;
    int iy = (int)(local / w.nx), ix = (int)(local % w.nx);
    d.layer = w.layer;
    d.lx = w.bx + ix * sx;
    d.ly = w.by + iy * sy;
    d.w = w.ow;
    d.h = w.oh;
    d.cx = fd_cvRound(d.lx / L.scale) + w.ow / 2;
    d.cy = fd_cvRound(d.ly / L.scale) + w.oh / 2;
}

// features for all windows (fragment-major + |x|^2) then the MFMA SVM; returns window count
int64_t run_hog_svm(fd_ctx* ctx, fd_pyramid* p, const fd_svm* svm, const fd_hog_params* hp, std::vector<WindowLayer>& wls,
                    HogScratch& S, bool time_kernel) {
    if (!fd_svm_has_mfma_path(svm)) FD_THROW(FD_ERR_INVALID_ARGUMENT, "the HOG detector needs an RBF SVM on f32 feature vectors");
    HogDev hd = make_hogdev(hp, fd_svm_KP(svm));
    if (((hd.F + 7) & ~7) != hd.KP) FD_THROW(FD_ERR_INVALID_ARGUMENT, "SVM dimension does not match the HOG feature length %d", hd.F);
    HogWinTable wt;
    build_table(p, hp, wt, wls);
    const int64_t N = wt.total;
    if (N == 0) return 0;
    HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t Npad = (N + 63) & ~(int64_t)63;
    S.feat.reserve(sizeof(float) * (size_t)Npad * hd.KP);
    S.xx.reserve(sizeof(float) * (size_t)Npad);
    S.dist.reserve(sizeof(double) * (size_t)Npad);
    if (Npad > N) {  // zero the partially used last tile pair (rows N..Npad-1 live in the last 2 tiles)
        const int64_t firstTile = N >> 5;
        HIP_CHECK(hipMemsetAsync(S.feat.as<float>() + (size_t)firstTile * 32 * hd.KP, 0, sizeof(float) * (size_t)(Npad / 32 - firstTile) * 32 * hd.KP, st));
        HIP_CHECK(hipMemsetAsync(S.xx.as<float>() + N, 0, sizeof(float) * (size_t)(Npad - N), st));
    }
    const int grid = (int)std::min<int64_t>((N + 1) / 2, (int64_t)ctx->num_cus * 16);
    hipLaunchKernelGGL(k_hog_features<true>, dim3(grid), dim3(128), 0, st, p->arena.as<uint8_t>(), wt, hd, S.feat.as<float>(), S.xx.as<float>());
    HIP_CHECK(hipGetLastError());
    if (time_kernel) HIP_CHECK(hipEventRecord(ctx->ev0, st));
    fd_svm_rbf_mfma_launch(ctx, svm, S.feat.as<float>(), S.xx.as<float>(), N, S.dist.as<double>());
    if (time_kernel) HIP_CHECK(hipEventRecord(ctx->ev1, st));
    return N;
}

}  // namespace

extern "C" {

int fd_hog_feature_length(const fd_hog_params* hp) {
    try { return make_hogdev(hp, 0).F; } catch (...) { return -1; }
}

int fd_extract_hog(fd_ctx* ctx, fd_pyramid* p, const fd_hog_params* hp, float* features, int64_t cap_windows, int64_t* count) {
    return fd_guard(ctx, [&] {
        if (!ctx || !p || !hp || !count) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_extract_hog: NULL argument");
        HogDev hd = make_hogdev(hp, 0);
        HogWinTable wt;
        std::vector<WindowLayer> wls;
        build_table(p, hp, wt, wls);
        *count = wt.total;
        if (!features || wt.total == 0) return;
        if (wt.total > cap_windows) FD_THROW(FD_ERR_CAPACITY, "fd_extract_hog: %lld windows, capacity %lld", (long long)wt.total, (long long)cap_windows);
        HIP_CHECK(hipSetDevice(ctx->device));
        HogScratch& S = scratch(ctx);
        const size_t bytes = sizeof(float) * (size_t)wt.total * hd.F;
        S.feat.reserve(bytes);
        const int grid = (int)std::min<int64_t>((wt.total + 1) / 2, (int64_t)ctx->num_cus * 16);
        hipLaunchKernelGGL(k_hog_features<false>, dim3(grid), dim3(128), 0, ctx->stream, p->arena.as<uint8_t>(), wt, hd, S.feat.as<float>(), (float*)nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(features, S.feat.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int fd_detect_hog_svm(fd_ctx* ctx, fd_pyramid* p, const fd_svm* svm, const fd_hog_params* hp, fd_detection* out, int64_t cap,
                      int64_t* count, double* all_distance) {
    return fd_guard(ctx, [&] {
        if (!ctx || !p || !svm || !hp || !count) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_detect_hog_svm: NULL argument");
        HogScratch& S = scratch(ctx);
        std::vector<WindowLayer> wls;
        const int64_t N = run_hog_svm(ctx, p, svm, hp, wls, S, false);
        *count = 0;
        if (N == 0) return;
        hipStream_t st = ctx->stream;
        const unsigned int pcap = (unsigned int)std::min<int64_t>(N, 1 << 22);
        S.pos.reserve(sizeof(HogPos) * (size_t)pcap);
        S.counter.reserve(256);
        HIP_CHECK(hipMemsetAsync(S.counter.p, 0, 4, st));
        hipLaunchKernelGGL(k_select_positives, dim3((unsigned)std::min<int64_t>((N + 255) / 256, 2048)), dim3(256), 0, st,
                           S.dist.as<double>(), N, fd_svm_threshold(svm), S.pos.as<HogPos>(), S.counter.as<unsigned int>(), pcap);
        HIP_CHECK(hipGetLastError());
        unsigned int cnt = 0;
        HIP_CHECK(hipMemcpyAsync(&cnt, S.counter.p, 4, hipMemcpyDeviceToHost, st));
        if (all_distance) HIP_CHECK(hipMemcpyAsync(all_distance, S.dist.p, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (cnt > pcap) FD_THROW(FD_ERR_CAPACITY, "fd_detect_hog_svm: %u positives exceed the device buffer", cnt);
        std::vector<HogPos> raw(cnt);
        if (cnt) HIP_CHECK(hipMemcpy(raw.data(), S.pos.p, sizeof(HogPos) * cnt, hipMemcpyDeviceToHost));
        auto widof = [](const HogPos& r) { return ((uint64_t)r.wid_hi << 32) | r.wid_lo; };
        std::sort(raw.begin(), raw.end(), [&](const HogPos& a, const HogPos& b) { return widof(a) < widof(b); });
        *count = cnt;
        for (unsigned int i = 0; i < cnt && out && (int64_t)i < cap; ++i) {
            fd_detection d;
            std::memset(&d, 0, sizeof(d));
            window_geometry(p, wls, hp->step_x, hp->step_y, (int64_t)widof(raw[i]), d);
            d.level = -1;
            d.positive = 1;
            d.score = (float)raw[i].dist;
            d.probability = fd_svm_probability(svm, raw[i].dist);
            out[i] = d;
        }
        if (out && (int64_t)cnt > cap) FD_THROW(FD_ERR_CAPACITY, "fd_detect_hog_svm: %u positives, capacity %lld", cnt, (long long)cap);
    });
}

int fd_bench_hog_svm(fd_ctx* ctx, fd_pyramid* p, const fd_svm* svm, const fd_hog_params* hp, int64_t* count, int64_t* positives) {
    return fd_guard(ctx, [&] {
        if (!ctx || !p || !svm || !hp) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_bench_hog_svm: NULL argument");
        HogScratch& S = scratch(ctx);
        std::vector<WindowLayer> wls;
        const int64_t N = run_hog_svm(ctx, p, svm, hp, wls, S, true);
        if (count) *count = N;
        unsigned int cnt = 0;
        if (N) {
            hipStream_t st = ctx->stream;
            const unsigned int pcap = (unsigned int)std::min<int64_t>(N, 1 << 22);
            S.pos.reserve(sizeof(HogPos) * (size_t)pcap);
            S.counter.reserve(256);
            HIP_CHECK(hipMemsetAsync(S.counter.p, 0, 4, st));
            hipLaunchKernelGGL(k_select_positives, dim3((unsigned)std::min<int64_t>((N + 255) / 256, 2048)), dim3(256), 0, st,
                               S.dist.as<double>(), N, fd_svm_threshold(svm), S.pos.as<HogPos>(), S.counter.as<unsigned int>(), pcap);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipMemcpyAsync(&cnt, S.counter.p, 4, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            HIP_CHECK(hipEventElapsedTime(&ctx->last_kernel_ms, ctx->ev0, ctx->ev1));
            ctx->last_kernel = "k_svm_rbf_mfma";
        }
        if (positives) *positives = cnt;
    });
}

}  // extern "C"
