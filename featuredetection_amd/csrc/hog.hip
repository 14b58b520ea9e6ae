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
int fd_svm_dim(const fd_svm* m);
bool fd_svm_is_u8(const fd_svm* m);
double fd_svm_probability(const fd_svm* m, double d);
void fd_svm_rbf_mfma_launch(fd_ctx* ctx, const fd_svm* m, const float* xFrag, const float* xx, int64_t npatches, double* out);
void fd_svm_generic_launch(fd_ctx* ctx, const fd_svm* m, const void* dfeat, const uint32_t* didx, int64_t stride_bytes, int64_t n, double* dout);
bool fd_svm_fused_view(const fd_svm* m, FdSvmFusedView* v);

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
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
                    *hp_ = *hp_ + factor * (float)(v >> 8);   // (ds_add_f32 instead of the read-add-write: bit-exact, but 131 against 168 Mpatches/s)
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

#include "hog_svm_fused.hpp"

struct HogScratch {
    DevBuf feat, xx, dist, pos, counter, tables, histTables, patchIn, part;
    DevBuf header;        // fused path: {positives, retired workgroups}, left zero by every run
    bool headerClean = false;
    HostBuf hcount;       // pinned read-back slot
    HogDev tabFor;        // parameters the tables were built for
    bool tabValid = false;
};
HogScratch& scratch(fd_ctx* ctx) { return fd_scratch<HogScratch>(ctx); }

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
    static uint64_t lds_allowed = 0;   // per instantiation
    fd_allow_lds(ctx, (const void*)k_hog_tile<FRAG>, 160 * 1024, lds_allowed);
    const int C = hd.rows * hd.cols, WPW = 64 / C, npass = (32 + WPW - 1) / WPW;
    const int64_t nitems = ((npad + 31) / 32) * npass;
    const int grid = (int)std::min<int64_t>((nitems + 3) / 4, (int64_t)ctx->num_cus * 16);
    hipLaunchKernelGGL(k_hog_tile<FRAG>, dim3(grid), dim3(256), lds, ctx->stream, p->arena.as<uint8_t>(), wt, hd, tb, npad, feat, xx);
    HIP_CHECK(hipGetLastError());
}

void build_table(const fd_pyramid* p, const fd_hog_params* hp, HogWinTable& wt, std::vector<WindowLayer>& wls, bool any_bin_image = false) {
    if (!any_bin_image) {
        if (p->filter_kind != FD_LAYER_GRADBIN || p->interpolate)
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "HOG extraction needs a pyramid with the FD_LAYER_GRADBIN layer filter (no bin interpolation)");
        if (p->bins != hp->bins) FD_THROW(FD_ERR_INVALID_ARGUMENT, "HogFilter bins (%d) differ from the layer filter bins (%d)", hp->bins, p->bins);
    }
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
// *selected: the positives are already in S.hcount (pinned: record 0 = their number) when the stream reaches this point -- the fused
// path; otherwise the caller queues k_select_positives and the copy
int64_t run_hog_svm(fd_ctx* ctx, fd_pyramid* p, const fd_svm* svm, const fd_hog_params* hp, std::vector<WindowLayer>& wls,
                    HogScratch& S, bool time_kernel, bool* selected = nullptr, unsigned int* pcapOut = nullptr) {
    if (selected) *selected = false;
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
    // config 2's shape (20 x 20 patches, 5-pixel cells, 2 x 2 blocks, 9 bins -> 324 elements) through ONE kernel: HOG vectors are
    // produced in the registers of the MFMA operand, nothing but the distances is written (hog_svm_fused.hpp).  FD_HOG_FUSED=0:
    // the two-kernel path below (any other shape takes it anyway; the tests compare the two).
    FdSvmFusedView fv;
    const char* fusedEnv = getenv("FD_HOG_FUSED");   // read per call: the tests switch between the two paths inside one process
    const bool fusedOn = !(fusedEnv && atoi(fusedEnv) == 0);
    if (fusedOn && hd.pw == 20 && hd.ph == 20 && hd.cell == 5 && hd.block == 2 && hd.bins == 9 && !hd.sau && hd.KP == 8 * HSF_Q &&
        fd_svm_fused_view(svm, &fv) && fv.KP == hd.KP && fv.nsv_pad / 32 <= 64) {
        HsfSvm hm;
        hm.svFrag = fv.svFrag; hm.ss = fv.ss; hm.coeff = fv.coeff; hm.nsvt = fv.nsv_pad / 32; hm.bias = fv.bias; hm.negGamma = (float)(-fv.gamma);
        const HsfPlan pl = hsf_plan(N, ctx->num_cus, hm.nsvt);
        S.dist.reserve(sizeof(double) * (size_t)pl.npadRows);
        if (pl.rem > 0) S.part.reserve(sizeof(double) * (size_t)pl.S * pl.npadRows);
        const char* me = getenv("FD_HSF_SUBS");
        const int subs = me ? atoi(me) : 1;
        const size_t lds = hsf_lds_bytes(subs, hm.nsvt);
        static uint64_t a0 = 0, a1 = 0, a2 = 0;
        fd_allow_lds(ctx, (const void*)k_hog_svm_fused<0>, 160 * 1024, a0);
        fd_allow_lds(ctx, (const void*)k_hog_svm_fused<1>, 160 * 1024, a1);
        fd_allow_lds(ctx, (const void*)k_hog_svm_fused<2>, 160 * 1024, a2);
        HsfOut o;
        o.dist = S.dist.as<double>(); o.part = S.part.as<double>();
        o.cap = (unsigned int)std::min<int64_t>(N, 1 << 22);
        o.threshold = fd_svm_threshold(svm);
        S.hcount.reserve(sizeof(HogPos) * ((size_t)o.cap + 1));
        o.pos = S.hcount.as<HogPos>();          // pinned host memory is device-accessible under the same address
        if (!S.header.p) { S.header.reserve(64); S.headerClean = false; }
        if (!S.headerClean) HIP_CHECK(hipMemsetAsync(S.header.p, 0, 64, st));
        S.headerClean = true;                   // k_hsf_finish leaves it zero
        o.header = S.header.as<unsigned int>();
        o.pos[0].wid_lo = 0xffffffffu;          // overwritten by the last workgroup of k_hsf_finish
        const int grid = pl.R > 0 ? pl.G : pl.rem * pl.S;
        if (time_kernel) HIP_CHECK(hipEventRecord(ctx->ev0, st));
        if (subs == 0)
            hipLaunchKernelGGL(k_hog_svm_fused<0>, dim3(grid), dim3(512), lds, st, p->arena.as<uint8_t>(), wt, hm, pl, o);
        else if (subs == 2)
            hipLaunchKernelGGL(k_hog_svm_fused<2>, dim3(grid), dim3(512), lds, st, p->arena.as<uint8_t>(), wt, hm, pl, o);
        else
            hipLaunchKernelGGL(k_hog_svm_fused<1>, dim3(grid), dim3(512), lds, st, p->arena.as<uint8_t>(), wt, hm, pl, o);
        if (time_kernel) HIP_CHECK(hipEventRecord(ctx->ev1, st));
        const int64_t first = pl.rem > 0 ? (int64_t)pl.R * pl.G * 256 : N;   // first window of the remainder round
        hipLaunchKernelGGL(k_hsf_finish, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((N - first + 255) / 256, 1024))), dim3(256), 0, st,
                           o, pl.S, pl.npadRows, first, N, hm.bias);
        if (selected) *selected = true;
        if (pcapOut) *pcapOut = o.cap;
        HIP_CHECK(hipGetLastError());
        return N;
    }
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
        fd_pyramid_require_single(p, "fd_extract_hog");
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

// Asynchronous half of SlidingWindowDetector::detect for the HOG chain: pyramid layers -> HOG tiles -> MFMA SVM -> positive
// selection are queued on the context's stream together with the read-back of the counter and the first positives into pinned
// memory; nothing blocks.  One ticket per context at a time (the scratch buffers belong to one run).
struct fd_hog_svm_ticket {
    fd_pyramid* p = nullptr;
    const fd_svm* svm = nullptr;
    fd_hog_params hp;
    std::vector<WindowLayer> wls;
    int64_t N = 0;
    unsigned int pcap = 0;
    hipEvent_t done = nullptr;
    bool timed = false;
    bool zerocopy = false;   // the fused path: the kernels wrote the positives into the pinned buffer themselves
    ~fd_hog_svm_ticket() { if (done) (void)hipEventDestroy(done); }
};
constexpr unsigned int HOG_FIRST_CHUNK = 4096;   // positives fetched together with the counter

static void hog_svm_begin(fd_ctx* ctx, fd_pyramid* p, const fd_svm* svm, const fd_hog_params* hp, fd_hog_svm_ticket& t) {
    HogScratch& S = scratch(ctx);
    t.p = p; t.svm = svm; t.hp = *hp;
    t.timed = ctx->kernel_timing;
    t.N = run_hog_svm(ctx, p, svm, hp, t.wls, S, t.timed, &t.zerocopy, &t.pcap);
    if (t.N == 0) return;
    hipStream_t st = ctx->stream;
    if (!t.zerocopy) {
        t.pcap = (unsigned int)std::min<int64_t>(t.N, 1 << 22);
        // record 0 of the positive buffer is the header (counter), so that counter + first positives come back in one copy
        S.pos.reserve(sizeof(HogPos) * ((size_t)t.pcap + 1));
        S.hcount.reserve(sizeof(HogPos) * ((size_t)t.pcap + 1));
        HIP_CHECK(hipMemsetAsync(S.pos.p, 0, sizeof(HogPos), st));
        hipLaunchKernelGGL(k_select_positives, dim3((unsigned)std::min<int64_t>((t.N + 255) / 256, 2048)), dim3(256), 0, st,
                           S.dist.as<double>(), t.N, fd_svm_threshold(svm), S.pos.as<HogPos>() + 1, S.pos.as<unsigned int>(), t.pcap);
        HIP_CHECK(hipGetLastError());
        const size_t first = std::min<size_t>(t.pcap, HOG_FIRST_CHUNK);
        HIP_CHECK(hipMemcpyAsync(S.hcount.p, S.pos.p, sizeof(HogPos) * (first + 1), hipMemcpyDeviceToHost, st));
    }
    if (!t.done) HIP_CHECK(hipEventCreateWithFlags(&t.done, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(t.done, st));
}

static void hog_svm_end(fd_ctx* ctx, fd_hog_svm_ticket& t, fd_detection* out, int64_t cap, int64_t* count, double* all_distance) {
    *count = 0;
    if (t.N == 0) return;
    HogScratch& S = scratch(ctx);
    hipStream_t st = ctx->stream;
    HIP_CHECK(hipEventSynchronize(t.done));
    if (t.timed) {
        HIP_CHECK(hipEventElapsedTime(&ctx->last_kernel_ms, ctx->ev0, ctx->ev1));
        ctx->last_kernel = t.zerocopy ? "k_hog_svm_fused" : "k_svm_rbf_mfma";
    }
    HogPos* h = S.hcount.as<HogPos>();
    const unsigned int cnt = h[0].wid_lo;
    if (cnt > t.pcap) FD_THROW(FD_ERR_DEVICE_CAPACITY, "fd_detect_hog_svm: %u positives exceed the device buffer", cnt);
    const size_t first = t.zerocopy ? (size_t)t.pcap : std::min<size_t>(t.pcap, HOG_FIRST_CHUNK);
    if (cnt > first) HIP_CHECK(hipMemcpyAsync(h + 1 + first, S.pos.as<HogPos>() + 1 + first, sizeof(HogPos) * (cnt - first), hipMemcpyDeviceToHost, st));
    if (all_distance) HIP_CHECK(hipMemcpyAsync(all_distance, S.dist.p, sizeof(double) * (size_t)t.N, hipMemcpyDeviceToHost, st));
    if (cnt > first || all_distance) HIP_CHECK(hipStreamSynchronize(st));
    HogPos* raw = h + 1;
    auto widof = [](const HogPos& r) { return ((uint64_t)r.wid_hi << 32) | r.wid_lo; };
    std::sort(raw, raw + cnt, [&](const HogPos& a, const HogPos& b) { return widof(a) < widof(b); });   // extraction order
    *count = cnt;
    for (unsigned int i = 0; i < cnt && out && (int64_t)i < cap; ++i) {
        fd_detection d;
        std::memset(&d, 0, sizeof(d));
        window_geometry(t.p, t.wls, t.hp.step_x, t.hp.step_y, (int64_t)widof(raw[i]), d);
        d.level = -1;
        d.positive = 1;
        d.score = (float)raw[i].dist;
        d.probability = fd_svm_probability(t.svm, raw[i].dist);
        out[i] = d;
    }
    if (out && (int64_t)cnt > cap) FD_THROW(FD_ERR_CAPACITY, "fd_detect_hog_svm: %u positives, capacity %lld", cnt, (long long)cap);
}

int fd_detect_hog_svm(fd_ctx* ctx, fd_pyramid* p, const fd_svm* svm, const fd_hog_params* hp, fd_detection* out, int64_t cap,
                      int64_t* count, double* all_distance) {
    return fd_guard(ctx, [&] {
        if (!ctx || !p || !svm || !hp || !count) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_detect_hog_svm: NULL argument");
        fd_pyramid_require_single(p, "fd_detect_hog_svm");
        fd_hog_svm_ticket t;
        hog_svm_begin(ctx, p, svm, hp, t);
        hog_svm_end(ctx, t, out, cap, count, all_distance);
    });
}

int fd_detect_hog_svm_begin(fd_ctx* ctx, fd_pyramid* p, const fd_svm* svm, const fd_hog_params* hp, fd_hog_svm_ticket** ticket) {
    if (ticket) *ticket = nullptr;
    return fd_guard(ctx, [&] {
        if (!ctx || !p || !svm || !hp || !ticket) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_detect_hog_svm_begin: NULL argument");
        fd_pyramid_require_single(p, "fd_detect_hog_svm_begin");
        std::unique_ptr<fd_hog_svm_ticket> t(new fd_hog_svm_ticket());
        hog_svm_begin(ctx, p, svm, hp, *t);
        *ticket = t.release();
    });
}

int fd_detect_hog_svm_end(fd_ctx* ctx, fd_hog_svm_ticket* ticket, fd_detection* out, int64_t cap, int64_t* count) {
    std::unique_ptr<fd_hog_svm_ticket> t(ticket);   // released whatever happens
    return fd_guard(ctx, [&] {
        if (!ctx || !t || !count) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_detect_hog_svm_end: NULL argument");
        hog_svm_end(ctx, *t, out, cap, count, nullptr);
    });
}

}  // extern "C"

// =====================================================================================================
// Generic histogram features: every HistogramFilter subclass of the reference on top of
// HistogramFilter::createCellHistograms (HistogramFilter.cpp:23-197, both the interpolating and the
// non-interpolating path, 1/2/4-channel bin images):
//   FD_HIST_HOG              HogFilter.cpp:58-122
//   FD_HIST_SPATIAL          SpatialHistogramFilter.cpp:56-94 (block 1x1 and general blocks)
//   FD_HIST_PYRAMID_HOG      PyramidHogFilter.cpp:33-113
//   FD_HIST_SPATIAL_PYRAMID  SpatialPyramidHistogramFilter.cpp:37-81
// One wavefront per window.  The patch of the filtered layer is staged in LDS; lane == cell walks the
// pixels that contribute to its cell in the reference's scan order and accumulates into its private
// histogram row, so every fp32 accumulator sees its addends in the reference order.  The block /
// pyramid / normalisation stage runs with lane == block (or histogram) in the reference's loop order.
// The k_hog_tile kernel above is the tuned special case (HOG, no interpolation, <= 64 cells) of this.
struct HistDev {
    int32_t kind, pw, ph, ch, bins, interpolate, sau, concatenate, normalization;
    int32_t rows, cols;        // finest cell grid
    int32_t blockW, blockH, brows, bcols;
    int32_t maxLevel, histCount;
    int32_t realBins;          // bins (+ bins/2 with signedAndUnsigned)
    int32_t F;                 // feature length
    int32_t cellsOff;          // float offset of the cell histograms inside the LDS vector area
    int32_t ldsFloats;         // floats in the vector area
    int32_t patchBytes;        // pw*ph*ch rounded up to 16
};
struct HistCache { int32_t i1, i2; float w1, w2; };   // HistogramFilter.cpp:199-220
struct HistTables {
    const HistCache* rowCache;   // [ph]
    const HistCache* colCache;   // [pw]
    const int32_t* rowRange;     // [rows] lo | hi << 16 (pixel rows contributing to a cell row)
    const int32_t* colRange;     // [cols]
};

namespace {

__device__ __forceinline__ double wave_sum_dd(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// HistogramFilter::normalize (HistogramFilter.cpp:222-251) of v[0..n) by ONE lane (sequential, like the reference)
__device__ __forceinline__ void hist_normalize_seq(float* v, int n, int normalization) {
    const float eps = 1e-4f;
    auto l2 = [&]() {
        double s = 0;
        for (int i = 0; i < n; ++i) s += (double)v[i] * v[i];
        const float norm = (float)sqrt(s);
        const float inv = (float)(1.0 / (double)(norm + eps));
        for (int i = 0; i < n; ++i) v[i] = v[i] * inv;
    };
    auto l1 = [&]() {
        double s = 0;
        for (int i = 0; i < n; ++i) s += fabs((double)v[i]);
        const float norm = (float)s;
        const float inv = (float)(1.0 / (double)(norm + eps));
        for (int i = 0; i < n; ++i) v[i] = v[i] * inv;
    };
    if (normalization == 1) l2();
    else if (normalization == 2) { l2(); for (int i = 0; i < n; ++i) v[i] = fminf(v[i], 0.2f); l2(); }
    else if (normalization == 3) l1();
    else if (normalization == 4) { l1(); for (int i = 0; i < n; ++i) v[i] = sqrtf(v[i]); }
}

// the same by the whole wave on a long vector (block 1x1 of SpatialHistogramFilter: one normalisation of
// the concatenated cell histograms); the fp64 partial sums are combined in a different order than the
// sequential loop, which the fp32 result tolerates (compared at 1e-6 relative in the tests)
__device__ __forceinline__ void hist_normalize_wave(float* v, int n, int normalization, int lane) {
    const float eps = 1e-4f;
    auto l2 = [&]() {
        double s = 0;
        for (int i = lane; i < n; i += 64) s += (double)v[i] * v[i];
        s = wave_sum_dd(s);
        const float norm = (float)sqrt(s);
        const float inv = (float)(1.0 / (double)(norm + eps));
        for (int i = lane; i < n; i += 64) v[i] = v[i] * inv;
        wave_sync();
    };
    auto l1 = [&]() {
        double s = 0;
        for (int i = lane; i < n; i += 64) s += fabs((double)v[i]);
        s = wave_sum_dd(s);
        const float norm = (float)s;
        const float inv = (float)(1.0 / (double)(norm + eps));
        for (int i = lane; i < n; i += 64) v[i] = v[i] * inv;
        wave_sync();
    };
    if (normalization == 1) l2();
    else if (normalization == 2) {
        l2();
        for (int i = lane; i < n; i += 64) v[i] = fminf(v[i], 0.2f);
        wave_sync();
        l2();
    } else if (normalization == 3) l1();
    else if (normalization == 4) {
        l1();
        for (int i = lane; i < n; i += 64) v[i] = sqrtf(v[i]);
        wave_sync();
    }
}

__global__ __launch_bounds__(64) void k_hist_features(const uint8_t* __restrict__ arena, HogWinTable wt, HistDev hd, HistTables tab,
                                                      float* __restrict__ feat) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int pw = hd.pw, ph = hd.ph, ch = hd.ch, B = hd.bins;
    const int R = hd.rows, Cc = hd.cols, ncells = R * Cc;
    unsigned char* patch = smem;                                           // [ph][pw][ch]
    HistCache* rowCache = (HistCache*)(smem + hd.patchBytes);               // [ph]
    HistCache* colCache = rowCache + ph;                                    // [pw]
    int32_t* rowRange = (int32_t*)(colCache + pw);                          // [R]
    int32_t* colRange = rowRange + R;                                       // [Cc]
    float* vec = (float*)(((size_t)(colRange + Cc) + 15) & ~(size_t)15);    // [ldsFloats]: output vector (+ raw cells, scratch)
    float* cells = vec + hd.cellsOff;                                       // [ncells][B]
    if (hd.interpolate) {
        for (int i = lane; i < ph; i += 64) rowCache[i] = tab.rowCache[i];
        for (int i = lane; i < pw; i += 64) colCache[i] = tab.colCache[i];
        for (int i = lane; i < R; i += 64) rowRange[i] = tab.rowRange[i];
        for (int i = lane; i < Cc; i += 64) colRange[i] = tab.colRange[i];
    }
    const float factor = 1.f / 255.f;
    const float eps = 1e-4f;
    const int rowBytes = pw * ch;

    for (int64_t wid = blockIdx.x; wid < wt.total; wid += gridDim.x) {
        // ---- window -> layer position (DirectPyramidFeatureExtractor.cpp:75-123 order)
        int li = 0;
        for (int l = 1; l < wt.n; ++l) li = wt.l[l].first <= wid ? l : li;
        const HogWinLayer& wl = wt.l[li];
        const int local = (int)(wid - wl.first);
        const int iy = local / wl.nx, ix = local - iy * wl.nx;
        const uint8_t* src = arena + wl.off + ((size_t)(wl.by + iy * wt.sy) * wl.lw + (wl.bx + ix * wt.sx)) * ch;
        const size_t srcStride = (size_t)wl.lw * ch;
        for (int i = lane; i < ph * rowBytes; i += 64) {
            const int y = i / rowBytes, xb = i - y * rowBytes;
            patch[i] = src[(size_t)y * srcStride + xb];
        }
        for (int i = lane; i < hd.ldsFloats; i += 64) vec[i] = 0.f;
        wave_sync();

        // ---- cell histograms: lane == cell
        for (int c = lane; c < ncells; c += 64) {
            const int cr = c / Cc, cc = c - cr * Cc;
            float* hv = cells + (size_t)c * B;
            if (!hd.interpolate) {   // HistogramFilter.cpp:130-163
                const int sr = (cr * ph) / R, er = ((cr + 1) * ph) / R;
                const int sc = (cc * pw) / Cc, ec = ((cc + 1) * pw) / Cc;
                for (int y = sr; y < er; ++y)
                    for (int x = sc; x < ec; ++x) {
                        const unsigned char* px = patch + (y * pw + x) * ch;
                        if (ch == 1) {
                            if (px[0] < B) hv[px[0]] = hv[px[0]] + 1.f;
                        } else {
                            if (px[0] < B) hv[px[0]] = hv[px[0]] + factor * (float)px[1];
                            if (ch == 4 && px[2] < B) hv[px[2]] = hv[px[2]] + factor * (float)px[3];
                        }
                    }
            } else {                 // HistogramFilter.cpp:36-128: bilinear cell interpolation
                const int ylo = rowRange[cr] & 0xffff, yhi = rowRange[cr] >> 16;
                const int xlo = colRange[cc] & 0xffff, xhi = colRange[cc] >> 16;
                for (int y = ylo; y <= yhi; ++y) {
                    const HistCache rc = rowCache[y];
                    for (int x = xlo; x <= xhi; ++x) {
                        const HistCache kc = colCache[x];
                        const unsigned char* px = patch + (y * pw + x) * ch;
                        const float wt0 = ch == 1 ? 1.f : factor * (float)px[1];
                        const float wt1 = ch == 4 ? factor * (float)px[3] : 0.f;
                        // the four (row, column) neighbours in the reference's order; only mine are added
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int rr = (q & 2) ? rc.i2 : rc.i1, qc = (q & 1) ? kc.i2 : kc.i1;
                            if (rr != cr || qc != cc) continue;
                            const float wr = (q & 2) ? rc.w2 : rc.w1, wc = (q & 1) ? kc.w2 : kc.w1;
                            if (ch == 1) {
                                if (px[0] < B) hv[px[0]] = hv[px[0]] + wr * wc;
                            } else {
                                if (px[0] < B) hv[px[0]] = hv[px[0]] + wt0 * wr * wc;
                                if (ch == 4 && px[2] < B) hv[px[2]] = hv[px[2]] + wt1 * wr * wc;
                            }
                        }
                    }
                }
            }
        }
        wave_sync();

        // ---- feature vector
        float* out = vec;
        if (hd.kind == FD_HIST_HOG) {
            // scratch behind the cells: energies [ncells], then normalisers [nblocks]
            float* energy = cells + (size_t)ncells * B;
            float* nrm = energy + ncells;
            const int hb = B / 2, nblocks = hd.brows * hd.bcols;
            for (int c = lane; c < ncells; c += 64) {   // HogFilter.cpp:102-122
                const float* hv = cells + (size_t)c * B;
                float en = 0.f;
                if (hd.sau) for (int b = 0; b < hb; ++b) { const float u = hv[b] + hv[hb + b]; en = en + u * u; }
                else for (int b = 0; b < B; ++b) en = en + hv[b] * hv[b];
                energy[c] = en;
            }
            wave_sync();
            const int perCell = hd.realBins, perBlock = hd.blockW * hd.blockH * perCell;
            for (int bl = lane; bl < nblocks; bl += 64) {   // HogFilter.cpp:69-100
                const int br = bl / hd.bcols, bc = bl - br * hd.bcols;
                float en = 0.f;
                for (int r2 = br; r2 < br + hd.blockH; ++r2)
                    for (int c2 = bc; c2 < bc + hd.blockW; ++c2) en = en + energy[r2 * Cc + c2];
                nrm[bl] = 1.f / sqrtf(en + eps);
            }
            wave_sync();
            for (int o = lane; o < hd.F; o += 64) {
                const int bl = o / perBlock, w = o - bl * perBlock, ci = w / perCell, b = w - ci * perCell;
                const int br = bl / hd.bcols, bc = bl - br * hd.bcols;
                const float* hv = cells + (size_t)((br + ci / hd.blockW) * Cc + (bc + ci % hd.blockW)) * B;
                out[o] = b < B ? nrm[bl] * hv[b] : nrm[bl] * (hv[b - B] + hv[b - B + hb]);
            }
            wave_sync();
        } else if (hd.kind == FD_HIST_SPATIAL) {
            if (hd.blockW == 1 && hd.blockH == 1) {   // SpatialHistogramFilter.cpp:60-63: cells are the vector
                hist_normalize_wave(out, hd.F, hd.normalization, lane);
            } else {               // :64-92
                const int nblocks = hd.brows * hd.bcols;
                const int bhs = hd.concatenate ? hd.blockW * hd.blockH * B : B;
                for (int bl = lane; bl < nblocks; bl += 64) {
                    const int br = bl / hd.bcols, bc = bl - br * hd.bcols;
                    float* o = out + (size_t)bl * bhs;
                    for (int r2 = br; r2 < br + hd.blockH; ++r2)
                        for (int c2 = bc; c2 < bc + hd.blockW; ++c2) {
                            const float* hv = cells + (size_t)(r2 * Cc + c2) * B;
                            for (int b = 0; b < B; ++b) o[b] = o[b] + hv[b];
                            if (hd.concatenate) o += B;
                        }
                    hist_normalize_seq(out + (size_t)bl * bhs, bhs, hd.normalization);
                }
                wave_sync();
            }
        } else {   // pyramid kinds
            const int RB = hd.realBins, hb = B / 2;
            const int finest = 1 << (2 * hd.maxLevel);
            float* level = out + (size_t)(hd.histCount - finest) * RB;
            if (hd.kind == FD_HIST_PYRAMID_HOG) {   // copyCellHistograms, PyramidHogFilter.cpp:57-74
                for (int i = lane; i < finest * RB; i += 64) {
                    const int c = i / RB, b = i - c * RB;
                    const float* hv = cells + (size_t)c * B;
                    level[i] = b < B ? hv[b] : hv[b - B] + hv[b - B + hb];
                }
                wave_sync();
            }   // FD_HIST_SPATIAL_PYRAMID: the cells were accumulated in place (cellsOff)
            for (int lv = hd.maxLevel - 1; lv >= 0; --lv) {   // combineHistograms: children in row-major order
                const int bcnt = 1 << lv, ccnt = bcnt << 1;
                float* parent = level - (size_t)(bcnt * bcnt) * RB;
                for (int i = lane; i < bcnt * bcnt * RB; i += 64) {
                    const int bl = i / RB, b = i - bl * RB;
                    const int br = bl / bcnt, bc = bl - br * bcnt;
                    float s = 0.f;
                    for (int r2 = 2 * br; r2 < 2 * br + 2; ++r2)
                        for (int c2 = 2 * bc; c2 < 2 * bc + 2; ++c2) s = s + level[(size_t)(r2 * ccnt + c2) * RB + b];
                    parent[i] = s;
                }
                wave_sync();
                level = parent;
            }
            for (int hgi = lane; hgi < hd.histCount; hgi += 64) {
                float* hv = out + (size_t)hgi * RB;
                if (hd.kind == FD_HIST_PYRAMID_HOG) {   // normalizeHistograms, PyramidHogFilter.cpp:94-113
                    float en = 0.f;
                    if (hd.sau) for (int b = B; b < RB; ++b) en = en + hv[b] * hv[b];
                    else for (int b = 0; b < B; ++b) en = en + hv[b] * hv[b];
                    const float normalizer = 1.f / sqrtf(en + eps);
                    for (int b = 0; b < RB; ++b) hv[b] = normalizer * hv[b];
                } else {
                    hist_normalize_seq(hv, RB, hd.normalization);
                }
            }
            wave_sync();
        }
        for (int i = lane; i < hd.F; i += 64) feat[(size_t)wid * hd.F + i] = out[i];
        wave_sync();
    }
}

HistDev make_histdev(const fd_hist_params* hp, int ch) {
    if (!hp) FD_THROW(FD_ERR_INVALID_ARGUMENT, "NULL histogram parameters");
    if (hp->bins <= 0 || hp->bins > 256) FD_THROW(FD_ERR_INVALID_ARGUMENT, "HistogramFilter: binCount must be within 1..256");
    if (hp->patch_w < 1 || hp->patch_h < 1 || hp->patch_w > 64 || hp->patch_h > 64)
        FD_THROW(FD_ERR_INVALID_ARGUMENT, "patch size must be within 1..64 on this backend");
    HistDev d;
    std::memset(&d, 0, sizeof(d));
    d.kind = hp->kind; d.pw = hp->patch_w; d.ph = hp->patch_h; d.ch = ch; d.bins = hp->bins;
    d.interpolate = hp->interpolate ? 1 : 0;
    d.sau = hp->signed_and_unsigned ? 1 : 0;
    d.concatenate = hp->concatenate ? 1 : 0;
    d.normalization = hp->normalization;
    d.realBins = d.bins;
    if (d.kind == FD_HIST_HOG || d.kind == FD_HIST_SPATIAL) {
        const char* who = d.kind == FD_HIST_HOG ? "HogFilter" : "SpatialHistogramFilter";
        if (hp->cell_size <= 0) FD_THROW(FD_ERR_INVALID_ARGUMENT, "%s: cellSize must be greater than zero", who);
        if (hp->block_size <= 0) FD_THROW(FD_ERR_INVALID_ARGUMENT, "%s: blockSize must be greater than zero", who);
        if (hp->cell_h < 0 || hp->block_h < 0) FD_THROW(FD_ERR_INVALID_ARGUMENT, "%s: negative cell/block height", who);
        const int cellH = hp->cell_h > 0 ? hp->cell_h : hp->cell_size;
        d.rows = fd_cvRound((double)d.ph / (double)cellH);
        d.cols = fd_cvRound((double)d.pw / (double)hp->cell_size);
        d.blockW = hp->block_size;
        d.blockH = hp->block_h > 0 ? hp->block_h : hp->block_size;
        d.brows = d.rows - d.blockH + 1;
        d.bcols = d.cols - d.blockW + 1;
        if (d.rows < 1 || d.cols < 1 || d.brows < 1 || d.bcols < 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "%s: patch smaller than one block", who);
        if (d.kind == FD_HIST_HOG) {
            if (d.sau && d.bins % 2 != 0)
                FD_THROW(FD_ERR_INVALID_ARGUMENT, "HogFilter: the bin size must be even for signed and unsigned gradients to be combined");
            d.realBins = d.sau ? d.bins + d.bins / 2 : d.bins;
            d.F = d.brows * d.bcols * d.blockW * d.blockH * d.realBins;
            d.cellsOff = (d.F + 3) & ~3;   // raw cells (+ energies, normalisers) behind the output vector
            d.ldsFloats = d.cellsOff + d.rows * d.cols * d.bins + d.rows * d.cols + d.brows * d.bcols;
        } else {
            d.sau = 0;
            if (d.normalization < 0 || d.normalization > 4) FD_THROW(FD_ERR_INVALID_ARGUMENT, "SpatialHistogramFilter: invalid normalization");
            if (d.blockW == 1 && d.blockH == 1) {
                d.F = d.rows * d.cols * d.bins;
                d.cellsOff = 0;
                d.ldsFloats = d.F;
            } else {
                d.F = d.brows * d.bcols * (d.concatenate ? d.blockW * d.blockH * d.bins : d.bins);
                d.cellsOff = (d.F + 3) & ~3;
                d.ldsFloats = d.cellsOff + d.rows * d.cols * d.bins;
            }
        }
    } else if (d.kind == FD_HIST_PYRAMID_HOG || d.kind == FD_HIST_SPATIAL_PYRAMID) {
        const char* who = d.kind == FD_HIST_PYRAMID_HOG ? "PyramidHogFilter" : "SpatialPyramidHistogramFilter";
        if (hp->levels <= 0 || hp->levels > 5) FD_THROW(FD_ERR_INVALID_ARGUMENT, "%s: levelCount must be within 1..5", who);
        d.maxLevel = hp->levels - 1;
        for (int l = 0; l < hp->levels; ++l) d.histCount += 1 << (2 * l);
        d.rows = d.cols = 1 << d.maxLevel;
        d.blockW = d.blockH = 1; d.brows = d.rows; d.bcols = d.cols;
        if (d.kind == FD_HIST_PYRAMID_HOG) {
            if (d.sau && d.bins % 2 != 0)
                FD_THROW(FD_ERR_INVALID_ARGUMENT, "PyramidHogFilter: the bin size must be even for signed and unsigned gradients to be combined");
            d.realBins = d.sau ? d.bins + d.bins / 2 : d.bins;
            d.F = d.histCount * d.realBins;
            d.cellsOff = (d.F + 3) & ~3;
            d.ldsFloats = d.cellsOff + d.rows * d.cols * d.bins;
        } else {
            d.sau = 0;
            if (d.normalization < 0 || d.normalization > 4) FD_THROW(FD_ERR_INVALID_ARGUMENT, "SpatialPyramidHistogramFilter: invalid normalization");
            d.F = d.histCount * d.bins;
            d.cellsOff = (d.histCount - d.rows * d.cols) * d.bins;   // finest level accumulates in place
            d.ldsFloats = d.F;
        }
    } else {
        FD_THROW(FD_ERR_INVALID_ARGUMENT, "invalid histogram feature kind %d", d.kind);
    }
    d.patchBytes = (d.pw * d.ph * d.ch + 15) & ~15;
    return d;
}

size_t hist_lds_bytes(const HistDev& d) {
    return (size_t)d.patchBytes + sizeof(HistCache) * (size_t)(d.pw + d.ph) + sizeof(int32_t) * (size_t)(d.rows + d.cols) + 16 +
           sizeof(float) * (size_t)d.ldsFloats;
}

// HistogramFilter::createCache (HistogramFilter.cpp:199-220) + the pixel range each cell row/column receives
void hist_tables(fd_ctx* ctx, HogScratch& S, const HistDev& d, HistTables& tb) {
    std::vector<unsigned char> blob;
    auto cache = [](int size, int count, std::vector<HistCache>& c, std::vector<int32_t>& range) {
        c.resize(size);
        std::vector<int> lo(count, 1 << 20), hi(count, -1);
        for (int m = 0; m < size; ++m) {
            HistCache e;
            const double realIndex = (double)count * ((double)m + 0.5) / (double)size - 0.5;
            e.i1 = (int)std::floor(realIndex);
            e.i2 = e.i1 + 1;
            e.w2 = (float)(realIndex - e.i1);
            e.w1 = 1.f - e.w2;
            if (e.i1 < 0) { e.i1 = e.i2; e.w1 = 0; }
            else if (e.i2 >= count) { e.i2 = e.i1; e.w2 = 0; }
            c[m] = e;
            for (int idx : {e.i1, e.i2}) { lo[idx] = std::min(lo[idx], m); hi[idx] = std::max(hi[idx], m); }
        }
        range.resize(count);
        for (int i = 0; i < count; ++i) range[i] = hi[i] < 0 ? (1 | (0 << 16)) : (lo[i] | (hi[i] << 16));   // empty: lo > hi
    };
    std::vector<HistCache> rc, cc;
    std::vector<int32_t> rr, cr;
    cache(d.ph, d.rows, rc, rr);
    cache(d.pw, d.cols, cc, cr);
    const size_t o1 = sizeof(HistCache) * rc.size(), o2 = o1 + sizeof(HistCache) * cc.size(), o3 = o2 + 4 * rr.size();
    blob.resize(o3 + 4 * cr.size());
    std::memcpy(blob.data(), rc.data(), o1);
    std::memcpy(blob.data() + o1, cc.data(), o2 - o1);
    std::memcpy(blob.data() + o2, rr.data(), o3 - o2);
    std::memcpy(blob.data() + o3, cr.data(), 4 * cr.size());
    S.histTables.reserve(blob.size());
    HIP_CHECK(hipMemcpyAsync(S.histTables.p, blob.data(), blob.size(), hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    tb.rowCache = (const HistCache*)S.histTables.p;
    tb.colCache = (const HistCache*)((const char*)S.histTables.p + o1);
    tb.rowRange = (const int32_t*)((const char*)S.histTables.p + o2);
    tb.colRange = (const int32_t*)((const char*)S.histTables.p + o3);
}

// k_hist_features over the windows of `wt` inside `arena` -> S.feat ([N][F] floats)
void launch_hist_features(fd_ctx* ctx, const uint8_t* arena, const HogWinTable& wt, const HistDev& hd, HogScratch& S) {
    HIP_CHECK(hipSetDevice(ctx->device));
    const size_t lds = hist_lds_bytes(hd);
    if (lds > 160 * 1024) FD_THROW(FD_ERR_INVALID_ARGUMENT, "histogram feature vector too long for the LDS (%d floats)", hd.F);
    static uint64_t lds_allowed = 0;
    fd_allow_lds(ctx, (const void*)k_hist_features, 160 * 1024, lds_allowed);
    HistTables tb;
    hist_tables(ctx, S, hd, tb);
    S.feat.reserve(sizeof(float) * (size_t)wt.total * hd.F);
    const int grid = (int)std::min<int64_t>(wt.total, (int64_t)ctx->num_cus * 32);
    hipLaunchKernelGGL(k_hist_features, dim3(grid), dim3(64), lds, ctx->stream, arena, wt, hd, tb, S.feat.as<float>());
    HIP_CHECK(hipGetLastError());
}

// features of every window, plain [N][F] layout, into S.feat; returns N
int64_t run_hist_features(fd_ctx* ctx, fd_pyramid* p, const fd_hist_params* hp, std::vector<WindowLayer>& wls, HogScratch& S, HistDev& hd) {
    if (!hp) FD_THROW(FD_ERR_INVALID_ARGUMENT, "NULL histogram parameters");
    if (p->filter_kind != FD_LAYER_GRADBIN && p->filter_kind != FD_LAYER_LBP)
        FD_THROW(FD_ERR_INVALID_ARGUMENT, "histogram features need a pyramid whose layers are bin images (FD_LAYER_GRADBIN or FD_LAYER_LBP)");
    if (p->all.empty()) FD_THROW(FD_ERR_RUNTIME, "pyramid has not been updated with an image");
    const int ch = p->all[p->kept[0]].ch;
    if ((hp->kind == FD_HIST_HOG || hp->kind == FD_HIST_PYRAMID_HOG) && ch == 1)
        FD_THROW(FD_ERR_INVALID_ARGUMENT, "HOG features need a gradient bin image (FD_LAYER_GRADBIN)");
    if (p->filter_kind == FD_LAYER_GRADBIN && p->bins != hp->bins)
        FD_THROW(FD_ERR_INVALID_ARGUMENT, "histogram bins (%d) differ from the layer filter bins (%d)", hp->bins, p->bins);
    hd = make_histdev(hp, ch);
    fd_hog_params g;
    std::memset(&g, 0, sizeof(g));
    g.patch_w = hp->patch_w; g.patch_h = hp->patch_h; g.step_x = hp->step_x; g.step_y = hp->step_y; g.bins = hp->bins;
    HogWinTable wt;
    build_table(p, &g, wt, wls, /*any_bin_image=*/true);
    const int64_t N = wt.total;
    if (N == 0) return 0;
    launch_hist_features(ctx, p->arena.as<uint8_t>(), wt, hd, S);
    return N;
}

}  // namespace

// classifier positives (distance >= threshold, SvmClassifier.cpp:44-46) of N scored windows -> detection
// records in extraction order, with the logistic probability (ProbabilisticSvmClassifier.cpp:50-58)
void fd_svm_positives_to_detections(fd_ctx* ctx, const fd_pyramid* p, const fd_svm* svm, const std::vector<WindowLayer>& wls, int sx, int sy,
                                    const double* ddist, int64_t N, fd_detection* out, int64_t cap, int64_t* count, double* all_distance) {
    HogScratch& S = scratch(ctx);
    hipStream_t st = ctx->stream;
    const unsigned int pcap = (unsigned int)std::min<int64_t>(N, 1 << 22);
    S.pos.reserve(sizeof(HogPos) * (size_t)pcap);
    S.counter.reserve(256);
    HIP_CHECK(hipMemsetAsync(S.counter.p, 0, 4, st));
    hipLaunchKernelGGL(k_select_positives, dim3((unsigned)std::min<int64_t>((N + 255) / 256, 2048)), dim3(256), 0, st, ddist, N,
                       fd_svm_threshold(svm), S.pos.as<HogPos>(), S.counter.as<unsigned int>(), pcap);
    HIP_CHECK(hipGetLastError());
    unsigned int* hcnt = (unsigned int*)fd_pinned(ctx, 64);
    HIP_CHECK(hipMemcpyAsync(hcnt, S.counter.p, 4, hipMemcpyDeviceToHost, st));
    if (all_distance) HIP_CHECK(hipMemcpyAsync(all_distance, ddist, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const unsigned int cnt = *hcnt;
    if (cnt > pcap) FD_THROW(FD_ERR_DEVICE_CAPACITY, "%u positives exceed the device buffer", cnt);
    std::vector<HogPos> raw(cnt);
    if (cnt) HIP_CHECK(hipMemcpy(raw.data(), S.pos.p, sizeof(HogPos) * cnt, hipMemcpyDeviceToHost));
    auto widof = [](const HogPos& r) { return ((uint64_t)r.wid_hi << 32) | r.wid_lo; };
    std::sort(raw.begin(), raw.end(), [&](const HogPos& a, const HogPos& b) { return widof(a) < widof(b); });
    *count = cnt;
    for (unsigned int i = 0; i < cnt && out && (int64_t)i < cap; ++i) {
        fd_detection d;
        std::memset(&d, 0, sizeof(d));
        window_geometry(p, wls, sx, sy, (int64_t)widof(raw[i]), d);
        d.level = -1;
        d.positive = 1;
        d.score = (float)raw[i].dist;
        d.probability = fd_svm_probability(svm, raw[i].dist);
        out[i] = d;
    }
    if (out && (int64_t)cnt > cap) FD_THROW(FD_ERR_CAPACITY, "%u positives, capacity %lld", cnt, (long long)cap);
}

extern "C" {

int fd_hist_feature_length(const fd_hist_params* hp, int channels) {
    try { return make_histdev(hp, channels).F; } catch (...) { return -1; }
}

int fd_extract_hist(fd_ctx* ctx, fd_pyramid* p, const fd_hist_params* hp, float* features, int64_t cap_windows, int64_t* count) {
    return fd_guard(ctx, [&] {
        if (!ctx || !p || !hp || !count) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_extract_hist: NULL argument");
        fd_pyramid_require_single(p, "fd_extract_hist");
        HogScratch& S = scratch(ctx);
        std::vector<WindowLayer> wls;
        HistDev hd;
        if (!features) {   // count only
            int64_t total;
            fd_enumerate_layers(p, hp->patch_w, hp->patch_h, hp->step_x, hp->step_y, nullptr, wls, total);
            *count = total;
            return;
        }
        const int64_t N = run_hist_features(ctx, p, hp, wls, S, hd);
        *count = N;
        if (N == 0) return;
        if (N > cap_windows) FD_THROW(FD_ERR_CAPACITY, "fd_extract_hist: %lld windows, capacity %lld", (long long)N, (long long)cap_windows);
        HIP_CHECK(hipMemcpyAsync(features, S.feat.p, sizeof(float) * (size_t)N * hd.F, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

// SlidingWindowDetector::detect (SlidingWindowDetector.cpp:87-98) with a histogram patch filter and a
// ProbabilisticSvmClassifier on f32 feature vectors (any kernel; BenchmarkRunner.cpp:185-263 wiring)
// HistogramFilter::applyTo(Mat) of the four histogram patch filters (HogFilter.cpp:58-122, SpatialHistogramFilter.cpp:56-94,
// PyramidHogFilter.cpp:33-113, SpatialPyramidHistogramFilter.cpp:37-81) on n contiguous bin-image patches of hp->patch_w x
// hp->patch_h pixels with `channels` bytes per pixel (1: bin, 2: bin + weight, 4: two bins + weights): the batch is addressed as
// one layer whose windows are hp->patch_h rows apart.  out: n x fd_hist_feature_length(hp, channels) floats.
int fd_hist_patch_batch(fd_ctx* ctx, const uint8_t* bin_patches, int64_t n, int channels, const fd_hist_params* hp, float* out) {
    return fd_guard(ctx, [&] {
        if (!ctx || !hp || n < 0 || (n > 0 && (!bin_patches || !out))) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_hist_patch_batch: bad argument");
        if (channels != 1 && channels != 2 && channels != 4) FD_THROW(FD_ERR_INVALID_ARGUMENT, "HistogramFilter: the image must have one, two or four channels");
        if ((hp->kind == FD_HIST_HOG || hp->kind == FD_HIST_PYRAMID_HOG) && channels == 1)
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "HOG features need a gradient bin image (two or four channels)");
        if (n == 0) return;
        if (n > (int64_t)1 << 24) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_hist_patch_batch: too many patches in one call");
        HogScratch& S = scratch(ctx);
        const HistDev hd = make_histdev(hp, channels);
        HIP_CHECK(hipSetDevice(ctx->device));
        const size_t bytes = (size_t)n * hp->patch_w * hp->patch_h * channels;
        S.patchIn.reserve(bytes);
        HIP_CHECK(hipMemcpyAsync(S.patchIn.p, bin_patches, bytes, hipMemcpyHostToDevice, ctx->stream));
        HogWinTable wt;
        std::memset(&wt, 0, sizeof(wt));
        wt.n = 1; wt.sx = 1; wt.sy = hp->patch_h; wt.total = n;
        wt.l[0].bx = 0; wt.l[0].by = 0; wt.l[0].nx = 1; wt.l[0].ny = (int32_t)n; wt.l[0].lw = hp->patch_w; wt.l[0].off = 0; wt.l[0].first = 0;
        launch_hist_features(ctx, S.patchIn.as<uint8_t>(), wt, hd, S);
        HIP_CHECK(hipMemcpyAsync(out, S.feat.p, sizeof(float) * (size_t)n * hd.F, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int fd_detect_hist_svm(fd_ctx* ctx, fd_pyramid* p, const fd_svm* svm, const fd_hist_params* hp, fd_detection* out, int64_t cap,
                       int64_t* count, double* all_distance) {
    return fd_guard(ctx, [&] {
        if (!ctx || !p || !svm || !hp || !count) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_detect_hist_svm: NULL argument");
        fd_pyramid_require_single(p, "fd_detect_hist_svm");
        HogScratch& S = scratch(ctx);
        std::vector<WindowLayer> wls;
        HistDev hd;
        const int64_t N = run_hist_features(ctx, p, hp, wls, S, hd);
        *count = 0;
        if (N == 0) return;
        if (fd_svm_dim(svm) != hd.F || fd_svm_is_u8(svm))
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "SVM must work on f32 vectors of length %d", hd.F);
        S.dist.reserve(sizeof(double) * (size_t)N);
        fd_svm_generic_launch(ctx, svm, S.feat.p, nullptr, (int64_t)hd.F * 4, N, S.dist.as<double>());
        fd_svm_positives_to_detections(ctx, p, svm, wls, hp->step_x, hp->step_y, S.dist.as<double>(), N, out, cap, count, all_distance);
    });
}

}  // extern "C"
