// featuredetection_amd/csrc/wvm.hip -- fused patch-extract + HistEq64 + integral image + WVM cascade.
//
// One wavefront per sliding window (DirectPyramidFeatureExtractor.cpp:113-118 hot loop 1 and
// SlidingWindowDetector.cpp:92-96 hot loop 2 fused; nothing per-window is materialised in HBM):
//   1. 64 lanes read the pw x ph window of the pyramid layer (two rows per wave instruction),
//   2. HistEq64Filter.cpp:32-125: 64-bin histogram in LDS (lane == bin), the fp32 prefix sum is done
//      in the reference's sequential order (bit-exact; exact .5 ties occur),
//   3. IImg.cpp:26-65: integral image of the equalised patch in LDS (exact integers) and the fp32
//      row-sequential sum of squares,
//   4. WvmClassifier.cpp:100-149,191-346: level loop with early exit.  Rect sums -> per-grey-level sums
//      (LDS, exact ints) -> fp64 chain in reference order -> exp -> the fp32 filter-output sums
//      res_k = -bias + sum_{p<=k} w[k][p] K[p] are kept as 64-lane-distributed running sums P_m (one
//      per future level m), each extended by exactly one term per level in the reference order, so
//      the sequential fp32 rounding is reproduced without a serial loop.
// Compiled with -ffp-contract=off: the float/double operation order is part of the spec.
//
// Roofline: nominally HBM (compulsory bytes/window = layer bytes / windows + 12 B record), in
// practice VALU/LDS bound; both figures are reported by bench.py --workload wvm.
#include "fd_internal.hpp"
#include <chrono>
#include <algorithm>
#include <cstring>
#include <memory>


// from svm.hip
struct fd_svm;
int fd_svm_dim(const fd_svm* m);
bool fd_svm_is_u8(const fd_svm* m);
float fd_svm_threshold(const fd_svm* m);
void fd_svm_generic_launch(fd_ctx* ctx, const fd_svm* m, const void* dfeat, const uint32_t* didx, int64_t stride_bytes, int64_t n, double* dout);

constexpr int WVM_MAX_LAYERS = 64;
constexpr int WVM_MAX_DIM = 32;      // patch width/height limit of this kernel
constexpr int WVM_MAX_VALS = 16;     // grey values per filter
constexpr int WVM_PJ = 5;            // up to 320 filters (largest cfg-implied WVM: 280)

struct WinLayerDev {
    int32_t bx, by, nx, ny;
    int32_t lw;
    uint32_t off;
    uint32_t magic;   // min(floor(2^32 / nx), 2^32 - 1): mulhi(local, magic) is local / nx or one less
    uint32_t pad;
    int64_t first;
};
struct WinTable {
    int32_t n;
    int32_t sx, sy;
    int32_t raw;   // != 0: arena holds `total` contiguous, already equalised patches (fd_wvm_eval_batch)
    int64_t total;
    WinLayerDev l[WVM_MAX_LAYERS];
};

// Per-level header, fetched with one scalar load
struct WvmLevelHdr {
    int32_t nrects, cntval;
    float thr;
    int32_t pad;
    double pp;
    double pad2;
};

struct WvmDev {
    int32_t fw, fh, d;
    int32_t numFilters, numUsed, numPer;
    float negBasis;   // -basisParam
    float negBias;    // -lin_thresholds[level]
    float stretch;    // 255.0f / (float)(fw*fh)
    const float* thresholds;
    const float* wT;          // wT[p * numFilters + k] = hkWeights[k][p]
    const double* pp;
    const int32_t* valOff;
    const double* val;
    const int32_t* rectBegin;  // [numFilters + 1]
    const uint32_t* rects;     // x1 | y1 << 8 | x2 << 16 | y2 << 24
    const uint8_t* rectV;      // grey-value index v (>= 1) of each rect
    const uint4* lvlRec;       // [numFilters][64]: lane l -> {rect l, v tag of rect l, val[l] (double bits)}
    const struct WvmLevelHdr* lvlHdr;   // [numFilters]
};

struct PosRec {
    uint32_t wid_lo, wid_hi;
    int32_t level;
    float fout;
};

struct fd_wvm {
    fd_ctx* ctx;
    WvmDev dev;
    DevBuf thresholds, wT, pp, valOff, val, rectBegin, rects, rectV, lvlRec, lvlHdr;
    double logisticA, logisticB;
    std::vector<float> h_thresholds;
    // scratch reused across calls
    DevBuf all_level, all_fout, pos, pos_patches, counter;
    HostBuf h_pos;
    int64_t pos_cap = 0;
};

namespace {

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142;

// Inclusive prefix sum inside each 32-lane half of the wave (integer, so the order is free):
// Hillis-Steele inside the 16-lane DPP rows, then lane 15 of rows 0/2 is added to rows 1/3.
__device__ __forceinline__ int scan_half(int s) {
    s += __builtin_amdgcn_update_dpp(0, s, DPP_ROW_SHR1, 0xf, 0xf, true);
    s += __builtin_amdgcn_update_dpp(0, s, DPP_ROW_SHR2, 0xf, 0xf, true);
    s += __builtin_amdgcn_update_dpp(0, s, DPP_ROW_SHR4, 0xf, 0xf, true);
    s += __builtin_amdgcn_update_dpp(0, s, DPP_ROW_SHR8, 0xf, 0xf, true);
    s += __builtin_amdgcn_update_dpp(0, s, DPP_ROW_BCAST15, 0xa, 0xf, false);
    return s;
}
__device__ __forceinline__ double readlane_d(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// LDS per wave: hist 256 + sv 64 + kernel-value history 1280 + integral image (1600 for 20x20, else 4096)
template <int PW_, int PH_>
struct __attribute__((aligned(16))) WaveLds {
    unsigned int hist[64];
    int sv[WVM_MAX_VALS];
    float kh[64 * WVM_PJ];
    unsigned int ii[PW_ ? PW_ * PH_ : WVM_MAX_DIM * WVM_MAX_DIM];
};

// One wave per window.  PW_/PH_ != 0: patch size known at compile time (the 20x20 detectors of the
// reference configs); 0: sizes from the model, up to 32x32.  RAW: the input is `total` contiguous,
// already equalised patches (fd_wvm_eval_batch).
//
// Lane layout of the patch: lanes 0-31 = columns of the top rows [0, rh), lanes 32-63 = columns of the
// bottom rows [rh, ph); register j of a lane = row r0 + j.  HistEq64, the integral image and the sum of
// squares run on registers + DPP; only the histogram, the finished integral image and the per-level
// grey-value sums go through LDS.
template <int PW_, int PH_, bool RAW>
__global__ __launch_bounds__(256) void k_wvm_cascade(const uint8_t* __restrict__ arena, WinTable wt, WvmDev m,
                                                      int32_t* __restrict__ all_level, float* __restrict__ all_fout,
                                                      PosRec* __restrict__ pos, uint8_t* __restrict__ pos_patches,
                                                      unsigned int* __restrict__ pos_count, unsigned int pos_cap) {
    __shared__ WaveLds<PW_, PH_> lds[4];
    constexpr int RHMAX = PW_ ? (PH_ + 1) / 2 : WVM_MAX_DIM / 2;
    constexpr bool ALLROWS = PW_ && (PH_ % 2 == 0);   // every (half, j) is a real row
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveLds<PW_, PH_>& L = lds[wave];
    const int pw = PW_ ? PW_ : m.fw, ph = PW_ ? PH_ : m.fh, d = pw * ph;
    const int rh = PW_ ? RHMAX : (ph + 1) / 2;
    const int half = lane >> 5, col = lane & 31;
    const bool colok = col < pw;
    const int r0 = half * rh;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int F = m.numFilters;
    auto rowok = [&](int j) { return ALLROWS ? true : (j < rh && r0 + j < ph); };

    // layer cursor: window ids only grow, so the layer index only moves forward (all scalar)
    int li = 0;
    int64_t nextFirst = (!RAW && wt.n > 1) ? wt.l[1].first : INT64_MAX;
    if (lane < WVM_MAX_VALS) L.sv[lane] = 0;
    wave_sync();

    for (int64_t wid = (int64_t)blockIdx.x * 4 + wave; wid < wt.total; wid += nwaves) {
        const uint8_t* src;
        int srcStride;
        if (RAW) {
            src = arena + (size_t)wid * d;
            srcStride = pw;
        } else {
            while (wid >= nextFirst) {
                ++li;
                nextFirst = (li + 1 < wt.n) ? wt.l[li + 1].first : INT64_MAX;
            }
            const WinLayerDev& wl = wt.l[li];
            const unsigned int local = (unsigned int)(wid - wl.first);
            unsigned int iy = __umulhi(local, wl.magic);   // floor(local / nx) or one less
            unsigned int ix = local - iy * (unsigned int)wl.nx;
            if (ix >= (unsigned int)wl.nx) { ix -= wl.nx; ++iy; }
            const int lx = wl.bx + (int)ix * wt.sx, ly = wl.by + (int)iy * wt.sy;
            src = arena + wl.off + (size_t)ly * wl.lw + lx;
            srcStride = wl.lw;
        }

        // ---- 0. level-0 model data: requested now, consumed after the fixed part
        uint4 lv = m.lvlRec[lane];
        WvmLevelHdr hd = m.lvlHdr[0];
        float w = m.wT[lane];

        // ---- 1. load the window (columns beyond the patch read column 0 and are masked later)
        unsigned int px[RHMAX];
        {
            const uint8_t* sp = src + (size_t)r0 * srcStride + (colok ? col : 0);
#pragma unroll
            for (int j = 0; j < RHMAX; ++j) px[j] = rowok(j) ? sp[(size_t)j * srcStride] : 0u;
        }
        if (!RAW) {
            // ---- 2. HistEq64 (HistEq64Filter.cpp:32-125): histogram with lane == bin
            L.hist[lane] = 0;
            wave_sync();
            if (colok) {
#pragma unroll
                for (int j = 0; j < RHMAX; ++j)
                    if (rowok(j)) atomicAdd(&L.hist[px[j] >> 2], 1u);
            }
            wave_sync();
            const float pdf = (float)L.hist[lane] * m.stretch;
            // sequential fp32 cdf: x_t[l] = x_{t-1}[l-1] + pdf[l]; lane l holds cdf[l] from step l on
            float x = pdf;
#pragma unroll
            for (int t = 1; t < 64; ++t) {
                const float sh = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), DPP_WAVE_SHR1, 0xf, 0xf, true));
                x = sh + pdf;
            }
            const int lutv = (int)(unsigned int)(unsigned char)floor((double)x + 0.5);
            // ---- 3. equalise through the crossbar (lane b holds lut[b])
#pragma unroll
            for (int j = 0; j < RHMAX; ++j) {
                const unsigned int e = (unsigned int)__builtin_amdgcn_ds_bpermute((int)(px[j] & 0xfcu), lutv);
                px[j] = (colok && rowok(j)) ? e : 0u;
            }
        } else {
#pragma unroll
            for (int j = 0; j < RHMAX; ++j) px[j] = (colok && rowok(j)) ? px[j] : 0u;
        }
        // ---- integral image (IImg.cpp:22-47): row prefix sums by DPP, column sums in registers
        float sxx;
        {
            int s[RHMAX];
            int qTop[RHMAX], qBot[RHMAX];
#pragma unroll
            for (int j = 0; j < RHMAX; ++j) {
                s[j] = scan_half((int)px[j]);
                const int q = scan_half((int)(px[j] * px[j]));
                qTop[j] = __builtin_amdgcn_readlane(q, 31);
                qBot[j] = __builtin_amdgcn_readlane(q, 63);
            }
#pragma unroll
            for (int j = 1; j < RHMAX; ++j) s[j] += s[j - 1];
            // column totals of the top half go to the bottom half
            int topTotal = s[0];
#pragma unroll
            for (int j = 1; j < RHMAX; ++j)
                if (j < rh) topTotal = s[j];
            topTotal = __builtin_amdgcn_ds_bpermute(col << 2, topTotal);
            if (half == 0) topTotal = 0;
            if (colok) {
#pragma unroll
                for (int j = 0; j < RHMAX; ++j)
                    if (rowok(j)) L.ii[(r0 + j) * pw + col] = (unsigned int)(s[j] + topTotal);
            }
            // sum of squares: last column, fp32, row by row (IImg.cpp:33-47)
            sxx = (float)qTop[0];
#pragma unroll
            for (int j = 1; j < RHMAX; ++j)
                if (j < rh) sxx = sxx + (float)qTop[j];
#pragma unroll
            for (int j = 0; j < RHMAX; ++j)
                if (j < rh && rh + j < ph) sxx = sxx + (float)qBot[j];
        }
        wave_sync();
        const int sx_total = (int)L.ii[(ph - 1) * pw + (pw - 1)];

        // ---- 4. cascade (WvmClassifier.cpp:100-149, linEvalWvmHisteq64 :190-350)
        // Pb: lane l holds the running fp32 sum of level 64*b + l (b = block of the current level), in
        // the reference's summation order; blocks are caught up from the kernel-value history when entered.
        float Pb = m.negBias;
        float u = 0.f;  // lane n holds u_kernel_eval[n]
        int level = 0, n = 0;
        float fout = 0.f;
        float thr = 0.f;
        for (int k = 0;; ++k) {
            // software pipeline: the (packed) model data of level k+1 is requested before level k is
            // evaluated; an early exit simply drops it.
            const int kn = min(k + 1, m.numUsed - 1);
            const uint4 lvN = m.lvlRec[(size_t)kn * 64 + lane];
            const WvmLevelHdr hdN = m.lvlHdr[kn];
            const float wN = m.wT[(size_t)kn * F + (kn & ~63) + lane];

            if (k > 0 && (k & 63) == 0) {   // entering block b: replay levels 0..k-1 for its lanes
                Pb = m.negBias;
                for (int kk = 0; kk < k; ++kk) {
                    const float t = m.wT[(size_t)kk * F + k + lane] * L.kh[kk];
                    Pb = Pb + t;
                }
            }
            if (lane < hd.nrects) {   // first 64 rects of the level come from the prefetched record
                const unsigned int rc = lv.x;
                const int x1 = rc & 255, y1 = (rc >> 8) & 255, x2 = (rc >> 16) & 255, y2 = rc >> 24;
                int s = (int)L.ii[y2 * pw + x2];
                if (x1 > 0) s -= (int)L.ii[y2 * pw + x1 - 1];
                if (y1 > 0) s -= (int)L.ii[(y1 - 1) * pw + x2];
                if (x1 > 0 && y1 > 0) s += (int)L.ii[(y1 - 1) * pw + x1 - 1];
                atomicAdd(&L.sv[lv.y], s);
            }
            if (hd.nrects > 64) {     // rare: remaining rects straight from the flat arrays
                const int rb = m.rectBegin[k];
                for (int r = rb + 64 + lane; r < rb + hd.nrects; r += 64) {
                    const unsigned int rc = m.rects[r];
                    const int x1 = rc & 255, y1 = (rc >> 8) & 255, x2 = (rc >> 16) & 255, y2 = rc >> 24;
                    int s = (int)L.ii[y2 * pw + x2];
                    if (x1 > 0) s -= (int)L.ii[y2 * pw + x1 - 1];
                    if (y1 > 0) s -= (int)L.ii[(y1 - 1) * pw + x2];
                    if (x1 > 0 && y1 > 0) s += (int)L.ii[(y1 - 1) * pw + x1 - 1];
                    atomicAdd(&L.sv[m.rectV[r]], s);
                }
            }
            wave_sync();
            int svr = 0;
            if (lane < WVM_MAX_VALS) {   // lane v takes the sum of grey value v and clears it for the next level
                svr = L.sv[lane];
                L.sv[lane] = 0;
            }
            const double valL = __hiloint2double((int)lv.w, (int)lv.z);
            const double prod = (double)svr * valL;
            const int cntval = hd.cntval;
            double sum_xp = 0.0;
            int sumv0 = sx_total;
            for (int v = 1; v < cntval; ++v) {
                sumv0 -= __builtin_amdgcn_readlane(svr, v);
                sum_xp = sum_xp + readlane_d(prod, v);
            }
            const double t0 = (double)sumv0 * readlane_d(valL, 0);
            sum_xp = sum_xp + t0;
            sum_xp = sum_xp + (double)readlane_f(u, n);
            const float unew = (float)sum_xp;
            u = (lane == n) ? unew : u;
            double norm = (double)sxx;
            norm = norm - 2 * sum_xp;
            norm = norm + hd.pp;
            const float Kk = (float)exp((double)m.negBasis * norm);
            if (m.numUsed > 64 && lane == 0) L.kh[k] = Kk;
            {
                const float t = w * Kk;   // weights above the diagonal are stored as 0
                Pb = Pb + t;
            }
            fout = readlane_f(Pb, k & 63);
            level = k;
            thr = hd.thr;
            if (!(fout >= thr && k + 1 < m.numUsed)) break;
            lv = lvN;
            hd = hdN;
            w = wN;
            if (++n == m.numPer) n = 0;
        }
        // ---- 5. results
        const bool positive = (level + 1 == m.numFilters) && (fout >= thr);
        if (lane == 0) {
            if (all_level) all_level[wid] = level;
            if (all_fout) all_fout[wid] = fout;
        }
        if (positive) {
            unsigned int slot = 0;
            if (lane == 0) slot = atomicAdd(pos_count, 1u);
            slot = __builtin_amdgcn_readfirstlane(slot);
            if (slot < pos_cap) {
                if (lane == 0) pos[slot] = PosRec{(uint32_t)wid, (uint32_t)(wid >> 32), level, fout};
                uint8_t* dst = pos_patches + (size_t)slot * d;
                if (colok) {
#pragma unroll
                    for (int j = 0; j < RHMAX; ++j)
                        if (rowok(j)) dst[(r0 + j) * pw + col] = (uint8_t)px[j];
                }
            }
        }
        wave_sync();
    }
}

// HistEq64 only (fd_histeq64_batch): same steps 1-3 on contiguous patches
__global__ __launch_bounds__(256) void k_histeq64(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int64_t n, int pw,
                                                  int ph, float stretch) {
    __shared__ unsigned int hist[4][64];
    __shared__ unsigned int lut[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d = pw * ph;
    for (int64_t id = (int64_t)blockIdx.x * 4 + wave; id < n; id += (int64_t)gridDim.x * 4) {
        const uint8_t* src = in + id * d;
        hist[wave][lane] = 0;
        wave_sync();
        for (int i = lane; i < d; i += 64) atomicAdd(&hist[wave][src[i] >> 2], 1u);
        wave_sync();
        const float pdf = (float)hist[wave][lane] * stretch;
        float c = readlane_f(pdf, 0);
        float mycdf = c;
#pragma unroll
        for (int b = 1; b < 64; ++b) {
            c = c + readlane_f(pdf, b);
            mycdf = (lane == b) ? c : mycdf;
        }
        lut[wave][lane] = (unsigned int)(unsigned char)floor((double)mycdf + 0.5);
        wave_sync();
        for (int i = lane; i < d; i += 64) out[id * d + i] = (uint8_t)lut[wave][src[i] >> 2];
        wave_sync();
    }
}

template <bool RAW>
void launch_cascade(hipStream_t st, int grid, const WvmDev& dev, const uint8_t* arena, const WinTable& wt, int32_t* all_level,
                    float* all_fout, PosRec* pos, uint8_t* pos_patches, unsigned int* counter, unsigned int pos_cap) {
    if (dev.fw == 20 && dev.fh == 20)
        hipLaunchKernelGGL((k_wvm_cascade<20, 20, RAW>), dim3(grid), dim3(256), 0, st, arena, wt, dev, all_level, all_fout, pos,
                           pos_patches, counter, pos_cap);
    else
        hipLaunchKernelGGL((k_wvm_cascade<0, 0, RAW>), dim3(grid), dim3(256), 0, st, arena, wt, dev, all_level, all_fout, pos,
                           pos_patches, counter, pos_cap);
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------

void fd_wvm_build_table(const fd_pyramid* p, int pw, int ph, int sx, int sy, const int* roi, WinTable& wt,
                        std::vector<WindowLayer>& wls) {
    int64_t total;
    fd_enumerate_layers(p, pw, ph, sx, sy, roi, wls, total);
    if (wls.size() > (size_t)WVM_MAX_LAYERS)
        FD_THROW(FD_ERR_INVALID_ARGUMENT, "pyramid has %zu layers, this backend supports %d", wls.size(), WVM_MAX_LAYERS);
    std::memset(&wt, 0, sizeof(wt));
    wt.n = 0;
    wt.sx = sx;
    wt.sy = sy;
    wt.total = total;
    for (const WindowLayer& w : wls) {
        if (w.nx == 0 || w.ny == 0) continue;  // layers without windows never match (first == next first)
        const HostLayer& L = p->all[p->kept[w.layer]];
        WinLayerDev& dl = wt.l[wt.n++];
        dl.bx = w.bx; dl.by = w.by; dl.nx = w.nx; dl.ny = w.ny; dl.lw = L.w; dl.off = L.gray_off; dl.first = w.first;
        dl.magic = (uint32_t)std::min<uint64_t>((1ull << 32) / (uint64_t)w.nx, 0xffffffffull);
    }
}

// window id -> detection record geometry (host, mirrors DirectPyramidFeatureExtractor.cpp:113-118)
void fd_window_to_detection(const fd_pyramid* p, const std::vector<WindowLayer>& wls, int sx, int sy, int64_t wid,
                            fd_detection& d) {
    size_t i = 0;
    while (i + 1 < wls.size() && (wls[i + 1].first <= wid)) ++i;
    while (wls[i].nx == 0 || wls[i].ny == 0) --i;  // skip empty layers that share the same first index
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

struct WvmRun {
    std::vector<WindowLayer> wls;
    int64_t total = 0;
    std::vector<PosRec> pos;       // sorted by window id (= extraction order)
    std::vector<uint32_t> slots;   // device slot of each sorted positive (index into pos_patches)
};

void fd_wvm_run(fd_ctx* ctx, fd_pyramid* p, fd_wvm* m, int sx, int sy, const int* roi, bool want_all, WvmRun& run,
                bool time_kernel) {
    if (p->ctx != ctx || m->ctx != ctx) FD_THROW(FD_ERR_INVALID_ARGUMENT, "objects belong to different contexts");
    if (p->filter_kind != FD_LAYER_NONE)
        FD_THROW(FD_ERR_INVALID_ARGUMENT, "WVM detection needs a gray pyramid (no layer filter)");
    if (p->all.empty()) FD_THROW(FD_ERR_RUNTIME, "pyramid has not been updated with an image");
    HIP_CHECK(hipSetDevice(ctx->device));
    WinTable wt;
    fd_wvm_build_table(p, m->dev.fw, m->dev.fh, sx, sy, roi, wt, run.wls);
    run.total = wt.total;
    run.pos.clear();
    run.slots.clear();
    if (wt.total == 0) return;
    hipStream_t st = ctx->stream;
    if (want_all) {
        m->all_level.reserve(sizeof(int32_t) * (size_t)wt.total);
        m->all_fout.reserve(sizeof(float) * (size_t)wt.total);
    }
    if (m->pos_cap == 0) {
        m->pos_cap = 1 << 18;
        const char* e = getenv("FD_WVM_POS_CAP");
        if (e && atoll(e) > 0) m->pos_cap = atoll(e);
    }
    // pos buffer: record 0 is the header (positive counter), records 1.. are the positives, so that the
    // counter and the first records come back in a single read
    m->pos.reserve(sizeof(PosRec) * ((size_t)m->pos_cap + 1));
    m->pos_patches.reserve((size_t)m->dev.d * (size_t)m->pos_cap);
    HIP_CHECK(hipMemsetAsync(m->pos.p, 0, sizeof(PosRec), st));
    const int64_t blocks_needed = (wt.total + 3) / 4;
    const int grid = (int)std::min<int64_t>(blocks_needed, (int64_t)ctx->num_cus * 8);
    if (time_kernel) HIP_CHECK(hipEventRecord(ctx->ev0, st));
    launch_cascade<false>(st, grid, m->dev, p->arena.as<uint8_t>(), wt, want_all ? m->all_level.as<int32_t>() : nullptr,
                          want_all ? m->all_fout.as<float>() : nullptr, m->pos.as<PosRec>() + 1, m->pos_patches.as<uint8_t>(),
                          m->pos.as<unsigned int>(), (unsigned int)m->pos_cap);
    if (time_kernel) HIP_CHECK(hipEventRecord(ctx->ev1, st));
    HIP_CHECK(hipGetLastError());
    const size_t firstChunk = (size_t)std::min<int64_t>(m->pos_cap, 2048);
    PosRec* hraw = (PosRec*)fd_pinned(ctx, sizeof(PosRec) * ((size_t)m->pos_cap + 1));
    HIP_CHECK(hipMemcpyAsync(hraw, m->pos.p, sizeof(PosRec) * (firstChunk + 1), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const unsigned int cnt = hraw[0].wid_lo;
    if (time_kernel) {
        HIP_CHECK(hipEventElapsedTime(&ctx->last_kernel_ms, ctx->ev0, ctx->ev1));
        ctx->last_kernel = "k_wvm_cascade";
    }
    if ((int64_t)cnt > m->pos_cap)
        FD_THROW(FD_ERR_CAPACITY, "WVM produced %u positives, device buffer holds %lld (set FD_WVM_POS_CAP)", cnt, (long long)m->pos_cap);
    if (cnt) {
        if (cnt > firstChunk) {
            HIP_CHECK(hipMemcpyAsync(hraw + 1 + firstChunk, m->pos.as<PosRec>() + 1 + firstChunk, sizeof(PosRec) * (cnt - firstChunk),
                                     hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
        }
        const PosRec* raw = hraw + 1;
        std::vector<uint32_t> order(cnt);
        for (uint32_t i = 0; i < cnt; ++i) order[i] = i;
        auto widof = [&](uint32_t i) { return ((uint64_t)raw[i].wid_hi << 32) | raw[i].wid_lo; };
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return widof(a) < widof(b); });
        run.pos.resize(cnt);
        run.slots.resize(cnt);
        for (uint32_t i = 0; i < cnt; ++i) { run.pos[i] = raw[order[i]]; run.slots[i] = order[i]; }
    }
}

// ProbabilisticWvmClassifier.cpp:52 -- evaluated on the host with libm, like the reference
static inline double wvm_probability(const fd_wvm* m, double fout) {
    return 1.0f / (1.0f + std::exp(m->logisticA + m->logisticB * fout));
}

void fd_wvm_positives_to_detections(const fd_pyramid* p, const fd_wvm* m, const WvmRun& run, int sx, int sy,
                                    std::vector<fd_detection>& out) {
    out.resize(run.pos.size());
    for (size_t i = 0; i < run.pos.size(); ++i) {
        fd_detection d;
        std::memset(&d, 0, sizeof(d));
        int64_t wid = (int64_t)(((uint64_t)run.pos[i].wid_hi << 32) | run.pos[i].wid_lo);
        fd_window_to_detection(p, run.wls, sx, sy, wid, d);
        d.level = run.pos[i].level;
        d.positive = 1;
        d.score = run.pos[i].fout;
        d.probability = wvm_probability(m, (double)run.pos[i].fout);
        out[i] = d;
    }
}

const uint8_t* fd_wvm_patch_buffer(const fd_wvm* m) { return m->pos_patches.as<uint8_t>(); }
int fd_wvm_dim(const fd_wvm* m) { return m->dev.d; }

extern "C" {

int fd_wvm_create(fd_ctx* ctx, const fd_wvm_model* md, fd_wvm** out) {
    return fd_guard(ctx, [&] {
        if (!ctx || !md || !out) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_wvm_create: NULL argument");
        const int F = md->num_filters;
        if (F < 1 || F > 64 * WVM_PJ) FD_THROW(FD_ERR_INVALID_ARGUMENT, "WvmClassifier: num_filters must be in 1..%d", 64 * WVM_PJ);
        if (md->filter_w < 1 || md->filter_h < 1 || md->filter_w > WVM_MAX_DIM || md->filter_h > WVM_MAX_DIM)
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "WvmClassifier: filter size must be within 1..%d", WVM_MAX_DIM);
        if (md->num_per_level < 1 || md->num_per_level > 64)
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "WvmClassifier: num_per_level must be in 1..64");
        if (!md->thresholds || !md->hk_weights || !md->pp || !md->val_off || !md->val || !md->rec_off || !md->rects)
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "WvmClassifier: NULL model array");
        HIP_CHECK(hipSetDevice(ctx->device));
        fd_wvm* m = new fd_wvm();
        std::unique_ptr<fd_wvm> guard(m);
        m->ctx = ctx;
        const int nval = md->val_off[F];
        std::vector<float> wT((size_t)F * F + 64, 0.f);   // +64: a block of lanes may read past the last row
        for (int k = 0; k < F; ++k)
            for (int pidx = 0; pidx <= k; ++pidx) wT[(size_t)pidx * F + k] = md->hk_weights[(size_t)k * F + pidx];
        std::vector<int32_t> rectBegin(F + 1);
        std::vector<uint32_t> rects;
        std::vector<uint8_t> rectV;
        for (int k = 0; k < F; ++k) {
            rectBegin[k] = (int32_t)rects.size();
            const int v0 = md->val_off[k], cntval = md->val_off[k + 1] - v0;
            if (cntval < 1 || cntval > WVM_MAX_VALS)
                FD_THROW(FD_ERR_INVALID_ARGUMENT, "WvmClassifier: filter %d has %d grey values (1..%d supported)", k, cntval, WVM_MAX_VALS);
            for (int v = 1; v < cntval; ++v) {
                long area255 = 0;
                for (int r = md->rec_off[v0 + v]; r < md->rec_off[v0 + v + 1]; ++r) {
                    const uint8_t* rc = md->rects + 4 * (size_t)r;
                    if (rc[0] > rc[2] || rc[1] > rc[3] || rc[2] >= md->filter_w || rc[3] >= md->filter_h)
                        FD_THROW(FD_ERR_INVALID_ARGUMENT, "WvmClassifier: rectangle %d of filter %d lies outside the patch", r, k);
                    rects.push_back((uint32_t)rc[0] | ((uint32_t)rc[1] << 8) | ((uint32_t)rc[2] << 16) | ((uint32_t)rc[3] << 24));
                    rectV.push_back((uint8_t)v);
                    area255 += 255L * (rc[2] - rc[0] + 1) * (rc[3] - rc[1] + 1);
                }
                // the reference accumulates these sums in fp32; they are exact (and equal to our int
                // sums) as long as they stay below 2^24
                if (area255 >= (1L << 24))
                    FD_THROW(FD_ERR_INVALID_ARGUMENT, "WvmClassifier: rect sums of filter %d may exceed 2^24 (fp32-exact range)", k);
            }
        }
        rectBegin[F] = (int32_t)rects.size();
        if (rects.empty()) { rects.push_back(0); rectV.push_back(1); }
        auto up = [&](DevBuf& b, const void* src, size_t bytes) {
            b.reserve(std::max<size_t>(bytes, 16));
            HIP_CHECK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
        };
        up(m->thresholds, md->thresholds, sizeof(float) * F);
        up(m->wT, wT.data(), sizeof(float) * wT.size());
        up(m->pp, md->pp, sizeof(double) * F);
        up(m->valOff, md->val_off, sizeof(int32_t) * (F + 1));
        up(m->val, md->val, sizeof(double) * std::max(nval, 1));
        up(m->rectBegin, rectBegin.data(), sizeof(int32_t) * (F + 1));
        up(m->rects, rects.data(), sizeof(uint32_t) * rects.size());
        up(m->rectV, rectV.data(), rectV.size());
        {   // packed per-level records for the software-pipelined cascade
            std::vector<uint32_t> rec((size_t)F * 64 * 4, 0u);
            std::vector<WvmLevelHdr> hdr(F);
            for (int k = 0; k < F; ++k) {
                const int v0 = md->val_off[k], cntval = md->val_off[k + 1] - v0;
                const int nrects = rectBegin[k + 1] - rectBegin[k];
                std::memset(&hdr[k], 0, sizeof(WvmLevelHdr));
                hdr[k].nrects = nrects; hdr[k].cntval = cntval; hdr[k].thr = md->thresholds[k]; hdr[k].pp = md->pp[k];
                for (int l = 0; l < 64; ++l) {
                    uint32_t* r4 = &rec[((size_t)k * 64 + l) * 4];
                    if (l < nrects) { r4[0] = rects[rectBegin[k] + l]; r4[1] = rectV[rectBegin[k] + l]; }
                    if (l < cntval) { uint64_t bits; std::memcpy(&bits, &md->val[v0 + l], 8); r4[2] = (uint32_t)bits; r4[3] = (uint32_t)(bits >> 32); }
                }
            }
            up(m->lvlRec, rec.data(), sizeof(uint32_t) * rec.size());
            up(m->lvlHdr, hdr.data(), sizeof(WvmLevelHdr) * hdr.size());
        }
        WvmDev& d = m->dev;
        d.fw = md->filter_w; d.fh = md->filter_h; d.d = md->filter_w * md->filter_h;
        d.numFilters = F;
        d.numUsed = (md->num_used > F || md->num_used <= 0) ? F : md->num_used;  // WvmClassifier.cpp:151-158
        d.numPer = md->num_per_level;
        d.negBasis = -md->basis_param;
        d.negBias = -md->bias;
        d.stretch = 255.0f / (float)(md->filter_w * md->filter_h);
        d.thresholds = m->thresholds.as<float>(); d.wT = m->wT.as<float>(); d.pp = m->pp.as<double>();
        d.valOff = m->valOff.as<int32_t>(); d.val = m->val.as<double>(); d.rectBegin = m->rectBegin.as<int32_t>();
        d.rects = m->rects.as<uint32_t>(); d.rectV = m->rectV.as<uint8_t>();
        d.lvlRec = m->lvlRec.as<uint4>(); d.lvlHdr = m->lvlHdr.as<WvmLevelHdr>();
        m->logisticA = md->logistic_a;
        m->logisticB = md->logistic_b;
        m->h_thresholds.assign(md->thresholds, md->thresholds + F);
        *out = guard.release();
    });
}

void fd_wvm_destroy(fd_wvm* m) { delete m; }

int fd_detect_wvm(fd_ctx* ctx, fd_pyramid* p, const fd_wvm* wvm_, int sx, int sy, const int* roi, fd_detection* out,
                  int64_t cap, int64_t* count, int32_t* all_level, float* all_score) {
    return fd_guard(ctx, [&] {
        if (!ctx || !p || !wvm_ || !count) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_detect_wvm: NULL argument");
        fd_wvm* m = const_cast<fd_wvm*>(wvm_);
        WvmRun run;
        const bool want_all = all_level || all_score;
        fd_wvm_run(ctx, p, m, sx, sy, roi, want_all, run, false);
        if (want_all && run.total) {
            if (all_level) HIP_CHECK(hipMemcpy(all_level, m->all_level.p, sizeof(int32_t) * (size_t)run.total, hipMemcpyDeviceToHost));
            if (all_score) HIP_CHECK(hipMemcpy(all_score, m->all_fout.p, sizeof(float) * (size_t)run.total, hipMemcpyDeviceToHost));
        }
        std::vector<fd_detection> dets;
        fd_wvm_positives_to_detections(p, m, run, sx, sy, dets);
        *count = (int64_t)dets.size();
        for (size_t i = 0; i < dets.size() && (int64_t)i < cap && out; ++i) out[i] = dets[i];
        if ((int64_t)dets.size() > cap && out)
            FD_THROW(FD_ERR_CAPACITY, "fd_detect_wvm: %zu positives, capacity %lld", dets.size(), (long long)cap);
    });
}

int fd_wvm_eval_batch(fd_ctx* ctx, const fd_wvm* wvm_, const uint8_t* patches, int64_t n, int32_t* out_level, float* out_score) {
    return fd_guard(ctx, [&] {
        if (!ctx || !wvm_ || (n > 0 && !patches) || n < 0) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_wvm_eval_batch: bad argument");
        if (n == 0) return;
        fd_wvm* m = const_cast<fd_wvm*>(wvm_);
        HIP_CHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        const size_t bytes = (size_t)n * m->dev.d;
        DevBuf in;
        in.reserve(bytes);
        m->all_level.reserve(sizeof(int32_t) * (size_t)n);
        m->all_fout.reserve(sizeof(float) * (size_t)n);
        m->pos.reserve(sizeof(PosRec) * 16);
        m->pos_patches.reserve((size_t)m->dev.d * 16);
        m->counter.reserve(256);
        HIP_CHECK(hipMemcpyAsync(in.p, patches, bytes, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemsetAsync(m->counter.p, 0, 4, st));
        WinTable wt;
        std::memset(&wt, 0, sizeof(wt));
        wt.raw = 1;
        wt.total = n;
        const int grid = (int)std::min<int64_t>((n + 3) / 4, (int64_t)ctx->num_cus * 8);
        launch_cascade<true>(st, grid, m->dev, in.as<uint8_t>(), wt, m->all_level.as<int32_t>(), m->all_fout.as<float>(),
                             m->pos.as<PosRec>(), m->pos_patches.as<uint8_t>(), m->counter.as<unsigned int>(), 0u);
        HIP_CHECK(hipGetLastError());
        if (out_level) HIP_CHECK(hipMemcpyAsync(out_level, m->all_level.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
        if (out_score) HIP_CHECK(hipMemcpyAsync(out_score, m->all_fout.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
    });
}

int fd_bench_wvm(fd_ctx* ctx, fd_pyramid* p, const fd_wvm* wvm_, int sx, int sy, int64_t* count, int64_t* positives) {
    return fd_guard(ctx, [&] {
        if (!ctx || !p || !wvm_) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_bench_wvm: NULL argument");
        WvmRun run;
        fd_wvm_run(ctx, p, const_cast<fd_wvm*>(wvm_), sx, sy, nullptr, false, run, true);
        if (count) *count = run.total;
        if (positives) *positives = (int64_t)run.pos.size();
    });
}

// detection::FiveStageSlidingWindowDetector::detect, FiveStageSlidingWindowDetector.cpp:187-320 / :331-380
int fd_detect_five_stage(fd_ctx* ctx, fd_pyramid* p, const fd_wvm* wvm_, const fd_svm* svm, float oe_dist, float oe_ratio,
                         int sx, int sy, const int* roi, fd_detection* out, int cap, int* count, int32_t* stage_counts) {
    return fd_guard(ctx, [&] {
        if (!ctx || !p || !wvm_ || !svm || !count) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_detect_five_stage: NULL argument");
        fd_wvm* m = const_cast<fd_wvm*>(wvm_);
        if (fd_svm_dim(svm) != m->dev.d || !fd_svm_is_u8(svm))
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "second classifier must work on the %d-byte HistEq64 patch", m->dev.d);
        // stage 1: WVM over all windows (SlidingWindowDetector::detect), positives in extraction order
        static const bool trace = getenv("FD_TRACE") != nullptr;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto t0 = now();
        auto lap = [&](const char* what) {
            if (!trace) return;
            auto t1 = now();
            fprintf(stderr, "[fd five-stage] %-12s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t1 - t0).count());
            t0 = t1;
        };
        WvmRun run;
        fd_wvm_run(ctx, p, m, sx, sy, roi, false, run, true);
        lap("wvm");
        std::vector<fd_detection> wvmPos;
        fd_wvm_positives_to_detections(p, m, run, sx, sy, wvmPos);
        if (stage_counts) stage_counts[0] = (int)wvmPos.size();
        lap("to_dets");
        // stage 2: overlap elimination
        std::vector<int> keep;
        fd_host_overlap_elimination(wvmPos.data(), (int)wvmPos.size(), oe_dist, oe_ratio, keep);
        if (stage_counts) stage_counts[1] = (int)keep.size();
        lap("oe");
        // stage 3: SVM on the survivors' HistEq64 patches (still resident in HBM, gathered by slot)
        std::vector<fd_detection> svmPos;
        if (!keep.empty()) {
            std::vector<uint32_t> slots(keep.size());
            for (size_t i = 0; i < keep.size(); ++i) slots[i] = run.slots[keep[i]];
            DevBuf& idx = m->all_level;  // reuse scratch (not used by this call)
            idx.reserve(sizeof(uint32_t) * slots.size());
            m->all_fout.reserve(sizeof(double) * slots.size());
            // pinned staging: [slots (u32) | distances (f64)]
            const size_t distOff = (sizeof(uint32_t) * slots.size() + 15) & ~(size_t)15;
            char* pin = (char*)fd_pinned(ctx, distOff + sizeof(double) * slots.size());
            std::memcpy(pin, slots.data(), sizeof(uint32_t) * slots.size());
            HIP_CHECK(hipMemcpyAsync(idx.p, pin, sizeof(uint32_t) * slots.size(), hipMemcpyHostToDevice, ctx->stream));
            fd_svm_generic_launch(ctx, svm, m->pos_patches.p, idx.as<uint32_t>(), (int64_t)m->dev.d, (int64_t)slots.size(), m->all_fout.as<double>());
            HIP_CHECK(hipMemcpyAsync(pin + distOff, m->all_fout.p, sizeof(double) * slots.size(), hipMemcpyDeviceToHost, ctx->stream));
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
            const double* dist = (const double*)(pin + distOff);
            for (size_t i = 0; i < keep.size(); ++i) {
                if (dist[i] >= (double)fd_svm_threshold(svm)) {  // strongClassifier->classify(): bool only
                    fd_detection d = wvmPos[keep[i]];
                    d.score = (float)dist[i];
                    d.positive = 1;
                    d.probability = 0.5;  // ClassifiedPatch(patch, bool) default probability (ClassifiedPatch.hpp:29-30)
                    svmPos.push_back(d);
                }
            }
        }
        if (stage_counts) stage_counts[2] = (int)svmPos.size();
        lap("svm");
        auto byProb = [](const fd_detection& a, const fd_detection& b) { return a.probability > b.probability; };
        bool sortAtEnd = true;
        if (!roi) {
            std::vector<int> maxima;
            fd_host_block_nms_sparse(svmPos, p->img_w, p->img_h, 35, true, maxima);
            if (maxima.empty()) fd_host_block_nms_sparse(svmPos, p->img_w, p->img_h, 35, false, maxima);
            if (maxima.empty()) {
                sortAtEnd = false;  // "return svmPatchesPositive; // Should be empty." (:292-294), unsorted
            } else {
                std::sort(svmPos.begin(), svmPos.end(), byProb);
                std::vector<fd_detection> res;
                for (size_t i = 0; i + 1 < maxima.size(); i += 2) {
                    const int x = maxima[i], y = maxima[i + 1];
                    auto it = std::find_if(svmPos.begin(), svmPos.end(), [&](const fd_detection& a) { return a.cx == x && a.cy == y; });
                    if (it != svmPos.end()) res.push_back(*it);
                }
                svmPos.swap(res);
            }
        }
        if (sortAtEnd) std::sort(svmPos.begin(), svmPos.end(), byProb);
        if (stage_counts) stage_counts[3] = (int)svmPos.size();
        lap("nms");
        *count = (int)svmPos.size();
        for (size_t i = 0; i < svmPos.size() && (int)i < cap && out; ++i) out[i] = svmPos[i];
        if (out && (int)svmPos.size() > cap) FD_THROW(FD_ERR_CAPACITY, "fd_detect_five_stage: %zu detections, capacity %d", svmPos.size(), cap);
    });
}

int fd_histeq64_batch(fd_ctx* ctx, const uint8_t* patches, int64_t n, int w, int h, uint8_t* dst) {
    return fd_guard(ctx, [&] {
        if (!ctx || !patches || !dst || n < 0 || w < 1 || h < 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_histeq64_batch: bad argument");
        if (n == 0) return;
        HIP_CHECK(hipSetDevice(ctx->device));
        DevBuf in, out;
        const size_t bytes = (size_t)n * w * h;
        in.reserve(bytes);
        out.reserve(bytes);
        HIP_CHECK(hipMemcpyAsync(in.p, patches, bytes, hipMemcpyHostToDevice, ctx->stream));
        const int grid = (int)std::min<int64_t>((n + 3) / 4, 2048);
        hipLaunchKernelGGL(k_histeq64, dim3(grid), dim3(256), 0, ctx->stream, in.as<uint8_t>(), out.as<uint8_t>(), n, w, h,
                           255.0f / (float)(w * h));
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(dst, out.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

}  // extern "C"
