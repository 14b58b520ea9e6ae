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

constexpr int HOG_MAX_HIST = 1024;   // cells * bins
constexpr int HOG_MAX_CELLS = 64;
constexpr int HOG_MAX_BLOCKS = 64;

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// fp32 add executed by the LDS unit (ds_add_f32, no return value): fire-and-forget, in program order per wave
__device__ __forceinline__ void lds_fadd(float* p, float v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

__device__ __forceinline__ size_t frag_index_dev(int64_t row, int k, int KP) {
    const int64_t tile = row >> 5;
    const int r = (int)(row & 31), q = k >> 3, h = (k >> 2) & 1, t = k & 3;
    return (size_t)tile * 32 * KP + (size_t)q * 256 + (size_t)(h * 32 + r) * 4 + t;
}

// Tables built once on the host per HOG parameter set (no per-element integer divisions in the kernel):
//  cTab[c]  cell rectangle startRow | endRow<<8 | startCol<<16 | endCol<<24 (HistogramFilter.cpp:134-137)
//  oTab[o]  (o = output element): cell | bin<<8 | block<<16 | (1<<30 if the element is the "unsigned"
//           sum hv[bin] + hv[bin + bins/2]) | (1<<31 if o is padding and must be zero)
struct HogTables {
    const uint32_t* cTab;
    const uint32_t* oTab;
};

// A wavefront works on WPW = 64 / cells windows at once: lane = (window, cell).  Each lane walks its
// cell's pixels (straight from the L1/L2-resident layer) in the reference's row-major order and
// accumulates into its private histogram column in LDS (hist[bin][lane]); per-address order is the
// program order, so the fp32 sums are bit-identical to the CPU loop.  Block vectors are written as
// float4 (one fragment slot = 4 consecutive feature elements) directly into the fragment-major
// tile of the MFMA SVM kernel; the WPW windows of a pass fill WPW*16 contiguous bytes per slot.
// Only ~3 KB of LDS per wave, so occupancy (and with it latency hiding) is register-bound.
template <bool FRAG>
__global__ __launch_bounds__(256) void k_hog_tile(const uint8_t* __restrict__ arena, HogWinTable wt, HogDev hp, HogTables tab,
                                                  int64_t npad, float* __restrict__ feat, float* __restrict__ xx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int B = hp.bins;
    const int C = hp.rows * hp.cols, nblocks = hp.brows * hp.bcols;
    const int WPW = 64 / C;                       // windows per wavefront pass
    const int LPW = 64 / WPW;                     // lanes per window in the output stage
    const int KP = hp.KP, F = hp.F;
    uint32_t* cTab = (uint32_t*)smem;                             // 64
    uint32_t* oTab = cTab + 64;                                   // KP
    unsigned char* wbase = (unsigned char*)(oTab + ((KP + 3) & ~3));
    const int perWave = (B * 64 + 64 + 64 + 64) * 4;
    float* hist = (float*)(wbase + (size_t)wave * perWave);       // [B][64]
    float* energy = hist + B * 64;                                // [64]  (window, cell)
    float* norm = energy + 64;                                    // [64]  (window, block)
    float* part = norm + 64;                                      // [64]  partial |x|^2 per lane
    for (int i = threadIdx.x; i < C; i += 256) cTab[i] = tab.cTab[i];
    for (int i = threadIdx.x; i < KP; i += 256) oTab[i] = tab.oTab[i];
    __syncthreads();
    const float factor = 1.f / 255.f;
    const float eps = 1e-4f;
    const int lw_ = lane / C, lc = lane - lw_ * C;          // this lane's window slot and cell
    const bool laneUsed = lw_ < WPW;
    const int nw_ = lane / nblocks, nb_ = lane - nw_ * nblocks;   // (window, block) role for the normalisers
    const int ow = lane / LPW, ol = lane - ow * LPW;        // (window, sub-lane) role for the output stage
    const uint32_t crect = cTab[laneUsed ? lc : 0];
    const int cr0 = crect & 255, cr1 = (crect >> 8) & 255, cc0 = (crect >> 16) & 255, cc1 = crect >> 24;
    const int npass = (32 + WPW - 1) / WPW;
    const int64_t ntiles = (npad + 31) >> 5;
    // work item = (tile, pass); consecutive waves take consecutive passes of the same tile
    const int64_t nitems = ntiles * npass;
    for (int64_t item = (int64_t)blockIdx.x * 4 + wave; item < nitems; item += (int64_t)gridDim.x * 4) {
        const int64_t tileId = item / npass;
        const int pass = (int)(item - tileId * npass);
        const int r0 = pass * WPW;
        const int r = r0 + lw_;                       // tile row of this lane's window
        const int64_t wid = tileId * 32 + r;
        const bool valid = laneUsed && r < 32 && wid < wt.total;
        for (int b = 0; b < B; ++b) hist[b * 64 + lane] = 0.f;
        if (valid) {
            // window geometry per lane (windows of one pass may straddle rows / layers)
            int li = 0;
            for (int l = 1; l < wt.n; ++l) li = wt.l[l].first <= wid ? l : li;
            const HogWinLayer wl = wt.l[li];
            const int local = (int)(wid - wl.first);
            const int iy = local / wl.nx, ix = local - iy * wl.nx;
            const unsigned short* src = (const unsigned short*)(arena + wl.off) + (size_t)(wl.by + iy * wt.sy) * wl.lw + (wl.bx + ix * wt.sx);
            // cell histograms (HistogramFilter.cpp:150-163)
            for (int y = cr0; y < cr1; ++y) {
                const unsigned short* rowp = src + (size_t)y * wl.lw;
                for (int x = cc0; x < cc1; ++x) {
                    const unsigned int v = rowp[x];
                    float* hp_ = &hist[(v & 255u) * 64 + lane];
                    *hp_ = *hp_ + factor * (float)(v >> 8);
                }
            }
        }
        wave_sync();
        // ---- cell energies (HogFilter.cpp:102-122)
        {
            float en = 0.f;
            if (hp.sau) {
                const int hb = B / 2;
                for (int b = 0; b < hb; ++b) { const float uw = hist[b * 64 + lane] + hist[(hb + b) * 64 + lane]; en = en + uw * uw; }
            } else {
                for (int b = 0; b < B; ++b) { const float hv = hist[b * 64 + lane]; en = en + hv * hv; }
            }
            energy[lane] = en;
        }
        wave_sync();
        // ---- block normalisers (HogFilter.cpp:78-84): lane = (window, block)
        if (nw_ < WPW) {
            const int br = nb_ / hp.bcols, bc = nb_ - br * hp.bcols;
            float en = 0.f;
            for (int cr = br; cr < br + hp.block; ++cr)
                for (int cc = bc; cc < bc + hp.block; ++cc) en = en + energy[nw_ * C + cr * hp.cols + cc];
            norm[lane] = 1.f / sqrtf(en + eps);
        }
        wave_sync();
        // ---- block vectors (HogFilter.cpp:85-97): lane = (window, sub-lane); 4 consecutive elements per store
        float sq = 0.f;
        const int rr = r0 + ow;
        if (ow < WPW && rr < 32) {
            const bool wvalid = tileId * 32 + rr < wt.total;
            for (int s4 = ol; s4 < KP / 4; s4 += LPW) {
                float vals[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t t = oTab[4 * s4 + k];
                    float val = 0.f;
                    if (wvalid && !(t >> 31)) {
                        const int cell = t & 255, bin = (t >> 8) & 255, bl = (t >> 16) & 255;
                        const float nrm = norm[ow * nblocks + bl];
                        const float h0 = hist[bin * 64 + ow * C + cell];
                        val = (t >> 30) & 1 ? nrm * (h0 + hist[(bin + B / 2) * 64 + ow * C + cell]) : nrm * h0;
                    }
                    vals[k] = val;
                    sq += val * val;
                }
                if (FRAG) {
                    float4 o4 = make_float4(vals[0], vals[1], vals[2], vals[3]);
                    *(float4*)(feat + (size_t)tileId * 32 * KP + (size_t)(s4 >> 1) * 256 + ((s4 & 1) * 32 + rr) * 4) = o4;
                } else if (wvalid) {
                    float* d = feat + (size_t)(tileId * 32 + rr) * F + 4 * s4;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (4 * s4 + k < F) d[k] = vals[k];
                }
            }
        }
        if (FRAG) {
            part[lane] = sq;
            wave_sync();
            if (ol == 0 && ow < WPW && rr < 32 && tileId * 32 + rr < npad) {
                float tot = 0.f;
                for (int k = 0; k < LPW; ++k) tot += part[ow * LPW + k];
                xx[tileId * 32 + rr] = tot;
            }
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
    DevBuf feat, xx, dist, pos, counter, tables;
    HostBuf hcount;       // pinned read-back slot
    HogDev tabFor;        // parameters the tables were built for
    bool tabValid = false;
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
    if (d.rows * d.cols > HOG_MAX_CELLS || d.rows * d.cols * d.bins > HOG_MAX_HIST || d.brows * d.bcols > HOG_MAX_BLOCKS ||
        d.bins > 64 || d.pw > 255 || d.ph > 255)
        FD_THROW(FD_ERR_INVALID_ARGUMENT, "HogFilter: too many cells/bins for this backend");
    d.KP = KPwant > 0 ? KPwant : ((d.F + 7) & ~7);
    return d;
}

// host-built index tables of k_hog_tile (HistogramFilter.cpp:134-137 cell bounds, HogFilter.cpp:85-97 output order)
HogTables hog_tables(fd_ctx* ctx, HogScratch& S, const HogDev& d) {
    const int C = d.rows * d.cols;
    if (!S.tabValid || std::memcmp(&S.tabFor, &d, sizeof(HogDev)) != 0) {
        std::vector<uint32_t> t((size_t)64 + d.KP, 0u);
        for (int c = 0; c < C; ++c) {
            const int cr = c / d.cols, cc = c % d.cols;
            const uint32_t sr = (cr * d.ph) / d.rows, er = ((cr + 1) * d.ph) / d.rows;
            const uint32_t sc = (cc * d.pw) / d.cols, ec = ((cc + 1) * d.pw) / d.cols;
            t[c] = sr | (er << 8) | (sc << 16) | (ec << 24);
        }
        const int hb = d.bins / 2, perCell = d.sau ? d.bins + hb : d.bins;
        for (int o = 0; o < d.KP; ++o) {
            uint32_t v = 1u << 31;
            if (o < d.F) {
                const int bl = o / d.perBlock, w = o % d.perBlock, ci = w / perCell, b = w % perCell;
                const int br = bl / d.bcols, bc = bl % d.bcols;
                const int cell = (br + ci / d.block) * d.cols + (bc + ci % d.block);
                v = b < d.bins ? (uint32_t)cell | ((uint32_t)b << 8) | ((uint32_t)bl << 16)
                               : (uint32_t)cell | ((uint32_t)(b - d.bins) << 8) | ((uint32_t)bl << 16) | (1u << 30);
            }
            t[64 + o] = v;
        }
        S.tables.reserve(sizeof(uint32_t) * t.size());
        HIP_CHECK(hipMemcpyAsync(S.tables.p, t.data(), sizeof(uint32_t) * t.size(), hipMemcpyHostToDevice, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));  // t is a stack-lifetime host buffer
        S.tabFor = d;
        S.tabValid = true;
    }
    HogTables tb;
    tb.cTab = S.tables.as<uint32_t>();
    tb.oTab = tb.cTab + 64;
    return tb;
}

size_t hog_lds_bytes(const HogDev& d, bool /*frag*/) {
    const size_t perWave = ((size_t)d.bins * 64 + 64 + 64 + 64) * 4;
    return (64 + (((size_t)d.KP + 3) & ~(size_t)3)) * 4 + 4 * perWave;
}

template <bool FRAG>
void launch_hog(fd_ctx* ctx, const fd_pyramid* p, const HogWinTable& wt, const HogDev& hd, HogScratch& S, int64_t npad, float* feat, float* xx) {
    const HogTables tb = hog_tables(ctx, S, hd);
    const size_t lds = hog_lds_bytes(hd, FRAG);
    if (lds > 160 * 1024) FD_THROW(FD_ERR_INVALID_ARGUMENT, "HOG feature vector too long for the LDS tile (%d floats)", hd.F);
    static bool attr_set[2] = {false, false};
    if (!attr_set[FRAG]) {
        HIP_CHECK(hipFuncSetAttribute((const void*)k_hog_tile<FRAG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[FRAG] = true;
    }
    const int C = hd.rows * hd.cols, WPW = 64 / C, npass = (32 + WPW - 1) / WPW;
    const int64_t nitems = ((npad + 31) / 32) * npass;
    const int grid = (int)std::min<int64_t>((nitems + 3) / 4, (int64_t)ctx->num_cus * 16);
    hipLaunchKernelGGL(k_hog_tile<FRAG>, dim3(grid), dim3(256), lds, ctx->stream, p->arena.as<uint8_t>(), wt, hd, tb, npad, feat, xx);
    HIP_CHECK(hipGetLastError());
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
    launch_hog<true>(ctx, p, wt, hd, S, Npad, S.feat.as<float>(), S.xx.as<float>());
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
        launch_hog<false>(ctx, p, wt, hd, S, wt.total, S.feat.as<float>(), nullptr);
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
        unsigned int* hcnt = (unsigned int*)fd_pinned(ctx, 64);
        HIP_CHECK(hipMemcpyAsync(hcnt, S.counter.p, 4, hipMemcpyDeviceToHost, st));
        if (all_distance) HIP_CHECK(hipMemcpyAsync(all_distance, S.dist.p, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        const unsigned int cnt = *hcnt;
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
        // positives == NULL: fully asynchronous (nothing is read back, no host synchronisation; the
        // caller brackets many calls with fd_ctx_synchronize).  Otherwise the call is synchronous and
        // also records the hipEvent-timed duration of the dominant kernel.
        const bool sync = positives != nullptr;
        const int64_t N = run_hog_svm(ctx, p, svm, hp, wls, S, sync);
        if (count) *count = N;
        if (!N) { if (positives) *positives = 0; return; }
        hipStream_t st = ctx->stream;
        const unsigned int pcap = (unsigned int)std::min<int64_t>(N, 1 << 22);
        S.pos.reserve(sizeof(HogPos) * (size_t)pcap);
        S.counter.reserve(256);
        S.hcount.reserve(64);
        HIP_CHECK(hipMemsetAsync(S.counter.p, 0, 4, st));
        hipLaunchKernelGGL(k_select_positives, dim3((unsigned)std::min<int64_t>((N + 255) / 256, 2048)), dim3(256), 0, st,
                           S.dist.as<double>(), N, fd_svm_threshold(svm), S.pos.as<HogPos>(), S.counter.as<unsigned int>(), pcap);
        HIP_CHECK(hipGetLastError());
        if (sync) {
            HIP_CHECK(hipMemcpyAsync(S.hcount.p, S.counter.p, 4, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            HIP_CHECK(hipEventElapsedTime(&ctx->last_kernel_ms, ctx->ev0, ctx->ev1));
            ctx->last_kernel = "k_svm_rbf_mfma";
            *positives = *S.hcount.as<unsigned int>();
        }
    });
}

}  // extern "C"
