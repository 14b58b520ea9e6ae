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
#include <thread>
#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <atomic>


// from svm.hip
struct fd_svm;
int fd_svm_dim(const fd_svm* m);
bool fd_svm_is_u8(const fd_svm* m);
float fd_svm_threshold(const fd_svm* m);
double fd_svm_probability(const fd_svm* m, double d);
void fd_svm_generic_launch(fd_ctx* ctx, const fd_svm* m, const void* dfeat, const uint32_t* didx, int64_t stride_bytes, int64_t n, double* dout);
void fd_svm_generic_launch_on(hipStream_t st, const fd_svm* m, const void* dfeat, const uint32_t* didx, int64_t stride_bytes, int64_t n, double* dout);
bool fd_svm_u8_mfma_available(const fd_svm* m);
void fd_svm_u8_mfma_launch_counted(hipStream_t st, const fd_svm* m, const void* dfeat, const uint32_t* didx, int64_t stride_bytes, int64_t nmax,
                                   const unsigned int* dcount, double* dout);

constexpr int WVM_MAX_LAYERS = 64;
constexpr int WVM_MAX_DIM = 32;      // patch width/height limit of this kernel
constexpr int WVM_MAX_VALS = 16;     // grey values per filter
constexpr int WVM_FIRST_CHUNK = 4096;  // positives fetched together with the counter
constexpr int WVM_LCAP = 16;         // filters evaluated by the one-wave-per-window stage; survivors go to k_wvm_deep
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
    const int32_t* list;   // != NULL: `total` explicit windows {layer position, lx, ly}; l[i] then describes kept layer i
    // multi-frame pyramid (fd_pyramid_set_frames): window id = frame * per_image + id inside the frame; frame f's layers are at
    // arena + f * image_stride.  total = nimg * per_image.
    int32_t nimg, pad_;
    int64_t per_image;
    uint64_t image_stride;
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

// ---- stage B as dense contractions (wvm_stageb.hpp) ------------------------------------------------------------------
// Model tables: the rect sums of level k, grey value v >= 1 are x . M_{k,v} with M_{k,v}[pixel] = number of rects of (k, v) that
// cover the pixel (small integers), so all of them are one exact int8 contraction on v_mfma_i32_32x32x32_i8.  The rows
// (k, v) are grouped class by class (class n = k mod numPer: the levels that depend on each other through u_kernel_eval[n],
// WvmClassifier.cpp:308-333) into tiles of 32 rows; a level never straddles a tile.
constexpr int WVB_MAXPHASE = 4;
typedef int wvb_v4i __attribute__((ext_vector_type(4)));
typedef int wvb_v16i __attribute__((ext_vector_type(16)));
struct WvbDev {
    const wvb_v4i* A;         // [tile][KS][64 lanes]: lane l = row l & 31 of the tile, pixels ks * 32 + (l >> 5) * 16 .. + 15 (zero padded)
    const int4* lvl;          // [numFilters] {tile, first row inside the tile, grey-value count, val offset}
    const int32_t* c128;      // [tile * 32 + row] 128 * sum of the row (the pixels are stored as x - 128)
    const double* pp;         // [numFilters]
    const double* val;        // the model's val[]
    const float* thr;         // [numFilters]
    const float* wR;          // wR[p * Fr + k] = hkWeights[k][p] (p <= k), rows padded so that eight weights from any k can be read
    const int32_t* rec;       // [numFilters][64 dwords] the chain's per-level record (layout: wvm_stageb.hpp, WVB_REC_DW)
    int32_t KS, dstride, Fr;  // k-steps (32 pixels each), bytes per equalised patch row of the state (KS * 32 + 16: an odd number of 16-byte slots)
    int32_t KSP;              // k-steps per tile in A (KS rounded up to 8: the fragment ring of k_wvb_chain wraps from tile to tile)
    int32_t numPer, numUsed, numFilters, d;
    int32_t maxCnt;           // largest grey-value count of a used filter
    int32_t nphase;
    int32_t phaseGen[WVB_MAXPHASE + 1];   // phase i = generations [phaseGen[i], phaseGen[i + 1])
    float negBasis, negBias;
};
// Per-run state of the queued windows, two sets (a phase reads set phase & 1; its survivors are packed densely into the other).
// Position pos = place in the current phase's dense list.
struct WvbState {
    int8_t* X[2];             // [pos][dstride] equalised patch as x - 128, row-major, zero padded
    int64_t* wid[2];          // [pos] window id
    int2* aux[2];             // [pos] {sum of the equalised patch, fp32 bits of the reference's sum of squares (IImg.cpp:33-47)}
    float* U[2];              // [pos / 64][class][64] u_kernel_eval[class]
    float* K[2];              // [pos / 64][level][64] kernel values (filter_output[level]): the history of a tile of 64 windows is contiguous
    unsigned long long* exitKey;   // [pos] first failed level of the phase << 32 | fp32 bits of its filter-output sum; ~0: none
    unsigned int* cnt;        // [1 + WVB_MAXPHASE]: cnt[i] = windows alive at the start of phase i (i >= 1; phase 0 reads the queue length)
    int64_t cap;
};

struct fd_wvm {
    fd_ctx* ctx;
    WvmDev dev;
    DevBuf thresholds, wT, pp, valOff, val, rectBegin, rects, rectV, lvlRec, lvlHdr;
    double logisticA, logisticB;
    std::vector<float> h_thresholds;
    // scratch reused across calls
    DevBuf all_level, all_fout, pos, pos_patches, counter, deep_q, list;
    // dense pre-filter (wvm_dense.hpp): digit matrix, constants, queue of the windows it lets through
    DevBuf denseB, denseC;
    bool hdrClean = false;    // device header words are zero (left so by a zero-copy run): no memset needed
    bool zcRun = false;       // the run in flight reads back zero-copy
    int denseL = 0;   // 0: the model has no dense stage
    double denseScale = 0;   // 2^-s of the quantised residual images
    HostBuf h_pos;
    int64_t pos_cap = 0;
    hipEvent_t done = nullptr;   // recorded after the cascade kernels + first read-back of a run
    HostBuf h_tail;              // pinned staging of the SVM stage of a five-stage run: [slots | distances]
    hipEvent_t tailDone = nullptr;   // recorded after the SVM stage + its read-back
    hipEvent_t prep = nullptr;       // group launches (fd_wvm_launch_group): this member's header is cleared
    hipEvent_t grp = nullptr;        // group launches: the leader's shared pre-filter has been queued up to here
    // stage B as dense contractions (wvm_stageb.hpp): model tables, and the per-run state of the queued windows
    WvbDev wvb;
    bool wvbOk = false;
    DevBuf wvbA, wvbLvl, wvbC128, wvbWR, wvbRec;
    WvbState sb;
    DevBuf sbX[2], sbWid[2], sbAux[2], sbU[2], sbK[2], sbKey, sbCnt;
    int64_t deepCap = 0;             // windows stage B can hold in this run (the queue itself holds every window)
    bool sbRun = false;              // the run in flight uses the dense stage B
    // what the handle's previous runs saw (-1: unknown): windows queued for stage B, windows alive behind each phase cut of the model.
    // Sizes the grids and decides which cuts the next run keeps (launch_stageb).
    int64_t sbDeep = -1, sbCutAlive[WVB_MAXPHASE] = {-1, -1, -1, -1};
    uint32_t sbCutMask = ~0u, sbRuns = 0;
    int sbPlanN = 0, sbPlanCut[WVB_MAXPHASE] = {};   // cuts of the run in flight
    int sbLastN = 0, sbLastGen[WVB_MAXPHASE + 1] = {};   // phases of the last launch and their generation boundaries (fd_wvm_last_stage_b_plan)
    int64_t sbGrown = 0;             // capacity a queue overflow made the handle grow to
    int64_t prevPos = 0;             // positives of the handle's previous run (the batch entry points start the heaviest host tails first)
    std::shared_ptr<void> relaunch;   // WvbRelaunch: what fd_wvm_finish needs to run stage B again with a larger state (queue overflow)
    // five-stage tail on the device (fs_tail.hpp): overlap elimination + SVM queued behind the cascade
    bool tailWanted = false;         // set by the five-stage entry points before the launch
    bool tailRun = false;            // the run in flight keeps its positives on the device and is followed by k_fs_oe
    DevBuf fstHdr, fstSlots;         // FstHdr + the positive count stage B leaves; the SVM's slot list
    DevBuf fstFrameCount, fstFrameList;   // positives per frame and their slots (filled by k_wvb_exit, consumed and cleared by k_fs_oe)
    DevBuf fsbKeys, fsbGeo, fsbAcc, fsbMap;   // k_fs_oe_big (one frame, any number of positives): sort buffers, centres, accepted list, painted map
    HostBuf h_fst;                   // pinned: [hostHdr 16 B | FstFrame x frames | FstKeep x pos_cap | double x pos_cap]
    int64_t fstLaunched = 0;         // vectors the SVM launch of the run in flight covers
    int64_t fstPrevKeep = -1;        // survivors of the previous run (sizes the next SVM launch)
    int fstFrames = 0;
    bool fstDirty = false;           // a cascade with the device tail was queued and k_fs_oe (which clears the tail's counters) not yet
    // single-frame five-stage calls: the SVM scores of ALL WVM positives queued straight behind the cascade (five_stage.hpp, "spec")
    bool specWanted = false;         // set by fd_detect_five_stage before the launch
    bool specRun = false;            // the run in flight leaves its positive count on the device for the SVM launch behind it
    DevBuf specCnt;                  // that count (written by the last stage-B workgroup: CascadeOut::tail_count)
    HostBuf h_spec;                  // pinned: the distances, by positive slot
    int64_t specLaunched = 0;        // slots the SVM launch of the run in flight covers
    int64_t specPrev = -1;           // positives of the previous run (sizes the next launch)
    int specLastState = -1;          // test hook (fd_wvm_last_spec_state)
    int fstLastState = -1;           // measurement / test hook: -1 no device tail in the last run, else the flags it ended with (0: its results were used)
    ~fd_wvm() { if (done) (void)hipEventDestroy(done); if (tailDone) (void)hipEventDestroy(tailDone); if (prep) (void)hipEventDestroy(prep); if (grp) (void)hipEventDestroy(grp); }
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

// number of windows a cascade kernel has to evaluate and the id of the i-th one (all windows, or the pre-filter's queue)
__device__ __forceinline__ int64_t wt_total(const WinTable& wt) { return wt.total; }
__device__ __forceinline__ int64_t wt_wid(const WinTable&, int64_t i) { return i; }

// PW_/PH_ != 0: patch size known at compile time (the 20x20 detectors of the reference configs);
// 0: sizes from the model, up to 32x32.
//
// Lane layout of a patch inside one wave: lanes 0-31 = columns of the top rows [0, rh), lanes 32-63 =
// columns of the bottom rows [rh, ph); register j of a lane = row r0 + j.
template <int PW_, int PH_>
struct Geo {
    static constexpr int RHMAX = PW_ ? (PH_ + 1) / 2 : WVM_MAX_DIM / 2;
    static constexpr bool ALLROWS = PW_ && (PH_ % 2 == 0);   // every (half, j) is a real row
    static constexpr int IISZ = PW_ ? PW_ * PH_ : WVM_MAX_DIM * WVM_MAX_DIM;
    int pw, ph, d, rh, half, col, r0;
    bool colok;
    __device__ __forceinline__ Geo(const WvmDev& m, int lane) {
        pw = PW_ ? PW_ : m.fw;
        ph = PW_ ? PH_ : m.fh;
        d = pw * ph;
        rh = PW_ ? RHMAX : (ph + 1) / 2;
        half = lane >> 5;
        col = lane & 31;
        r0 = half * rh;
        colok = col < pw;
    }
    __device__ __forceinline__ bool rowok(int j) const { return ALLROWS ? true : (j < rh && r0 + j < ph); }
};

// window id -> first pixel of the window in the arena (DirectPyramidFeatureExtractor.cpp:75-123 order:
// layers, then rows, then columns).  RAW: `total` contiguous, already equalised patches.
template <bool RAW>
__device__ __forceinline__ const uint8_t* wvm_locate(const uint8_t* arena, const WinTable& wt, const int64_t* sFirst, int64_t wid,
                                                     int lane, int pw, int d, int& stride) {
    if (RAW) {
        stride = pw;
        return arena + (size_t)wid * d;
    }
    if (wt.nimg > 1) {   // multi-frame pyramid: frame = wid / per_image (wave-uniform)
        const int64_t f = wid / wt.per_image;
        wid -= f * wt.per_image;
        arena += (size_t)f * wt.image_stride;
    }
    if (wt.list) {   // explicit window list (single-patch extraction of sampled positions)
        const int32_t* e = wt.list + 3 * wid;
        const WinLayerDev& wl = wt.l[e[0]];
        stride = wl.lw;
        return arena + wl.off + (size_t)e[2] * wl.lw + e[1];
    }
    const int li = __builtin_amdgcn_readfirstlane(__popcll(__ballot(sFirst[lane] <= wid)) - 1);
    const WinLayerDev& wl = wt.l[li];
    const unsigned int local = (unsigned int)(wid - wl.first);
    unsigned int iy = __umulhi(local, wl.magic);   // floor(local / nx) or one less
    unsigned int ix = local - iy * (unsigned int)wl.nx;
    if (ix >= (unsigned int)wl.nx) { ix -= wl.nx; ++iy; }
    const int lx = wl.bx + (int)ix * wt.sx, ly = wl.by + (int)iy * wt.sy;
    stride = wl.lw;
    return arena + wl.off + (size_t)ly * wl.lw + lx;
}

// Fixed part of one window, executed by one wave: HistEq64 (HistEq64Filter.cpp:32-125), integral image
// and sum of squares (IImg.cpp:22-47).  Everything runs on registers + DPP; only the histogram and the
// finished integral image go through LDS.  On return px[] holds the equalised patch, ii the integral
// image (visible to this wave).
// PAD: the integral image is stored with a zero row and column in front ((pw + 1) x (ph + 1) entries, row 0 / column 0 are left
// untouched: the caller zeroes them once), so a rect sum is four unconditional reads.
template <int PW_, int PH_, bool RAW, bool PAD = false>
__device__ __forceinline__ void wvm_prepare(const Geo<PW_, PH_>& g, const uint8_t* src, int srcStride, float stretch, int lane,
                                            unsigned int* hist, unsigned int* ii,
                                            unsigned int (&px)[Geo<PW_, PH_>::RHMAX], float& sxx, int& sx_total) {
    constexpr int RHMAX = Geo<PW_, PH_>::RHMAX;
    {   // columns beyond the patch read column 0 and are masked later
        const uint8_t* sp = src + (size_t)g.r0 * srcStride + (g.colok ? g.col : 0);
#pragma unroll
        for (int j = 0; j < RHMAX; ++j) px[j] = g.rowok(j) ? sp[(size_t)j * srcStride] : 0u;
    }
    if (!RAW) {
        hist[lane] = 0;   // lane == bin
        wave_sync();
        if (g.colok) {
#pragma unroll
            for (int j = 0; j < RHMAX; ++j)
                if (g.rowok(j)) atomicAdd(&hist[px[j] >> 2], 1u);
        }
        wave_sync();
        const float pdf = (float)hist[lane] * stretch;
        // sequential fp32 cdf: x_t[l] = x_{t-1}[l-1] + pdf[l]; lane l holds cdf[l] from step l on
        float x = pdf;
#pragma unroll
        for (int t = 1; t < 64; ++t) {
            const float sh = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), DPP_WAVE_SHR1, 0xf, 0xf, true));
            x = sh + pdf;
        }
        const int lutv = (int)(unsigned int)(unsigned char)floor((double)x + 0.5);
        // equalise through the crossbar (lane b holds lut[b])
#pragma unroll
        for (int j = 0; j < RHMAX; ++j) {
            const unsigned int e = (unsigned int)__builtin_amdgcn_ds_bpermute((int)(px[j] & 0xfcu), lutv);
            px[j] = (g.colok && g.rowok(j)) ? e : 0u;
        }
    } else {
#pragma unroll
        for (int j = 0; j < RHMAX; ++j) px[j] = (g.colok && g.rowok(j)) ? px[j] : 0u;
    }
    // integral image: row prefix sums by DPP, column sums in registers
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
    int topTotal = s[0];   // column totals of the top half go to the bottom half
#pragma unroll
    for (int j = 1; j < RHMAX; ++j)
        if (j < g.rh) topTotal = s[j];
    topTotal = __builtin_amdgcn_ds_bpermute(g.col << 2, topTotal);
    if (g.half == 0) topTotal = 0;
    if (g.colok) {
#pragma unroll
        for (int j = 0; j < RHMAX; ++j)
            if (g.rowok(j)) ii[PAD ? (g.r0 + j + 1) * (g.pw + 1) + g.col + 1 : (g.r0 + j) * g.pw + g.col] = (unsigned int)(s[j] + topTotal);
    }
    // sum of squares: last column, fp32, row by row (IImg.cpp:33-47)
    sxx = (float)qTop[0];
#pragma unroll
    for (int j = 1; j < RHMAX; ++j)
        if (j < g.rh) sxx = sxx + (float)qTop[j];
#pragma unroll
    for (int j = 0; j < RHMAX; ++j)
        if (j < g.rh && g.rh + j < g.ph) sxx = sxx + (float)qBot[j];
    wave_sync();
    sx_total = (int)ii[PAD ? g.ph * (g.pw + 1) + g.pw : (g.ph - 1) * g.pw + (g.pw - 1)];
}

// Kernel value of one filter (WvmClassifier.cpp:190-350, linEvalWvmHisteq64): rect sums of the level
// from the integral image (lane == rect), grey-value sums through sv (must be all zero on entry, is
// all zero on return), then the scalar fp64 chain of the reference.  lv = packed level record of this
// lane, hd = level header.  u_n = u_kernel_eval[n] of the level; unew = its new value.
__device__ __forceinline__ float wvm_level_K(const WvmDev& m, const unsigned int* ii, int* sv, int pw, int lane, int k,
                                             const uint4& lv, const WvmLevelHdr& hd, int sx_total, float sxx, float u_n,
                                             float& unew) {
    if (lane < hd.nrects) {   // first 64 rects of the level come from the record
        const unsigned int rc = lv.x;
        const int x1 = rc & 255, y1 = (rc >> 8) & 255, x2 = (rc >> 16) & 255, y2 = rc >> 24;
        int s = (int)ii[y2 * pw + x2];
        if (x1 > 0) s -= (int)ii[y2 * pw + x1 - 1];
        if (y1 > 0) s -= (int)ii[(y1 - 1) * pw + x2];
        if (x1 > 0 && y1 > 0) s += (int)ii[(y1 - 1) * pw + x1 - 1];
        atomicAdd(&sv[lv.y], s);
    }
    if (hd.nrects > 64) {     // rare: remaining rects straight from the flat arrays
        const int rb = m.rectBegin[k];
        for (int r = rb + 64 + lane; r < rb + hd.nrects; r += 64) {
            const unsigned int rc = m.rects[r];
            const int x1 = rc & 255, y1 = (rc >> 8) & 255, x2 = (rc >> 16) & 255, y2 = rc >> 24;
            int s = (int)ii[y2 * pw + x2];
            if (x1 > 0) s -= (int)ii[y2 * pw + x1 - 1];
            if (y1 > 0) s -= (int)ii[(y1 - 1) * pw + x2];
            if (x1 > 0 && y1 > 0) s += (int)ii[(y1 - 1) * pw + x1 - 1];
            atomicAdd(&sv[m.rectV[r]], s);
        }
    }
    wave_sync();
    int svr = 0;
    if (lane < WVM_MAX_VALS) {   // lane v takes the sum of grey value v and clears it for the next level
        svr = sv[lane];
        sv[lane] = 0;
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
    sum_xp = sum_xp + (double)u_n;
    unew = (float)sum_xp;
    double norm = (double)sxx;
    norm = norm - 2 * sum_xp;
    norm = norm + hd.pp;
    return (float)exp((double)m.negBasis * norm);
}

struct CascadeOut {
    int32_t* all_level = nullptr;
    float* all_fout = nullptr;
    PosRec* pos = nullptr;
    uint8_t* pos_patches = nullptr;
    unsigned int* pos_count = nullptr;
    unsigned int pos_cap = 0;
    int64_t* deep_q = nullptr;           // windows that survive the first WVM_LCAP levels (finished by k_wvm_deep)
    unsigned int* deep_count = nullptr;
    // zero-copy read-back (host_count != NULL): `pos` is host-mapped pinned memory, and the last workgroup of stage B to retire
    // stores the positive count there and clears the device header {pos_count, deep_count, pre-queue count, done_blocks} for
    // the next run -- no copy, no memset on the stream, the event can be recorded straight after stage B.  The host reads the buffer
    // only after that event (fd_wvm_finish): its release is what makes the kernel's stores visible, the kernels themselves do not fence
    unsigned int* host_count = nullptr;
    unsigned int* done_blocks = nullptr;
    // five-stage tail on the device (fs_tail.hpp): `pos` stays in device memory, and stage B's last workgroup leaves the positive
    // count here for the overlap-elimination kernel queued behind it (the header's own counter is cleared for the next run)
    unsigned int* tail_count = nullptr;
    // ... and files every positive under its frame for that kernel (one workgroup per frame; with every workgroup scanning all ~10 K
    // records of a 64-frame call the scan was 13 us of hot-spotted L2 reads): frame_count[f] positives, their slots in
    // frame_list[f * frame_cap ..]
    unsigned int* frame_count = nullptr;
    uint32_t* frame_list = nullptr;
    unsigned int frame_cap = 0, frame_per_image = 0, frame_magic = 0, frame_n = 1;
};

struct WvbRelaunch {   // see fd_wvm::relaunch
    const uint8_t* arena;
    WinTable wt;
    CascadeOut o;
    hipStream_t st;
};

// end of a stage-B kernel: see CascadeOut::host_count
__device__ __forceinline__ void wvm_finalize(const CascadeOut& o) {
    if (!o.host_count) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        // no fences (see wvb_finalize): the count is an agent-scope atomic, and the host reads the pinned buffer only after the
        // stream's completion event
        const unsigned int done = atomicAdd(o.done_blocks, 1u);
        if (done == gridDim.x - 1) {
            const unsigned int cnt = __hip_atomic_load(o.pos_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o.host_count, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            // header words 0..3 of the device buffer: positives, stage-B queue, pre-filter queue, retired workgroups
            __hip_atomic_store(o.pos_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o.deep_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o.deep_count + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o.done_blocks, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int PW_, int PH_>
__device__ __forceinline__ void wvm_emit(const Geo<PW_, PH_>& g, const WvmDev& m, const CascadeOut& o, int64_t wid, int lane,
                                         int level, float fout, float thr, const unsigned int (&px)[Geo<PW_, PH_>::RHMAX]) {
    const bool positive = (level + 1 == m.numFilters) && (fout >= thr);
    if (lane == 0) {
        if (o.all_level) o.all_level[wid] = level;
        if (o.all_fout) o.all_fout[wid] = fout;
    }
    if (positive) {
        unsigned int slot = 0;
        if (lane == 0) slot = atomicAdd(o.pos_count, 1u);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot < o.pos_cap) {
            if (lane == 0) o.pos[slot] = PosRec{(uint32_t)wid, (uint32_t)(wid >> 32), level, fout};
            uint8_t* dst = o.pos_patches + (size_t)slot * g.d;
            if (g.colok) {
#pragma unroll
                for (int j = 0; j < Geo<PW_, PH_>::RHMAX; ++j)
                    if (g.rowok(j)) dst[(g.r0 + j) * g.pw + g.col] = (uint8_t)px[j];
            }
        }
    }
}

// ---- stage A: one wave per window, first WVM_LCAP levels -----------------------------------------
// Most windows leave the cascade here; the cost of the few that do not is two orders
// of magnitude higher, so they are queued for k_wvm_deep instead of stalling their wave.
template <int PW_, int PH_>
struct __attribute__((aligned(16))) WaveLds {
    unsigned int hist[64];
    int sv[WVM_MAX_VALS];
    unsigned int ii[Geo<PW_, PH_>::IISZ];
};

template <int PW_, int PH_, bool RAW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_wvm_cascade(const uint8_t* __restrict__ arena, WinTable wt, WvmDev m, CascadeOut o) {
    __shared__ WaveLds<PW_, PH_> lds[4];
    __shared__ int64_t sFirst[WVM_MAX_LAYERS];
    constexpr int RHMAX = Geo<PW_, PH_>::RHMAX;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveLds<PW_, PH_>& L = lds[wave];
    const Geo<PW_, PH_> g(m, lane);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int F = m.numFilters;
    const int nA = min(m.numUsed, WVM_LCAP);

    if (!RAW) {
        if (threadIdx.x < WVM_MAX_LAYERS) sFirst[threadIdx.x] = (int)threadIdx.x < wt.n ? wt.l[threadIdx.x].first : INT64_MAX;
        __syncthreads();
    }
    if (lane < WVM_MAX_VALS) L.sv[lane] = 0;
    wave_sync();

    const int64_t totalA = wt_total(wt);
    for (int64_t widx = (int64_t)blockIdx.x * 4 + wave; widx < totalA; widx += nwaves) {
        const int64_t wid = wt_wid(wt, widx);
        int srcStride;
        const uint8_t* src = wvm_locate<RAW>(arena, wt, sFirst, wid, lane, g.pw, g.d, srcStride);
        // level-0 model data: requested now, consumed after the fixed part
        uint4 lv = m.lvlRec[lane];
        WvmLevelHdr hd = m.lvlHdr[0];
        float w = m.wT[lane];
        unsigned int px[RHMAX];
        float sxx;
        int sx_total;
        wvm_prepare<PW_, PH_, RAW>(g, src, srcStride, m.stretch, lane, L.hist, L.ii, px, sxx, sx_total);

        // cascade (WvmClassifier.cpp:100-149).  Pb: lane l holds the running fp32 sum of level l in the
        // reference's summation order (weights above the diagonal are stored as 0).
        float Pb = m.negBias;
        float u = 0.f;  // lane n holds u_kernel_eval[n]
        int n = 0;
        bool deep = false;
        int level;
        float fout, thr;
        for (int k = 0;; ++k) {
            // software pipeline: the (packed) model data of level k+1 is requested before level k is
            // evaluated; an early exit simply drops it.
            const int kn = min(k + 1, nA - 1);
            const uint4 lvN = m.lvlRec[(size_t)kn * 64 + lane];
            const WvmLevelHdr hdN = m.lvlHdr[kn];
            const float wN = m.wT[(size_t)kn * F + lane];
            float unew;
            const float Kk = wvm_level_K(m, L.ii, L.sv, g.pw, lane, k, lv, hd, sx_total, sxx, readlane_f(u, n), unew);
            u = (lane == n) ? unew : u;
            {
                const float t = w * Kk;
                Pb = Pb + t;
            }
            fout = readlane_f(Pb, k);
            level = k;
            thr = hd.thr;
            if (!(fout >= thr && k + 1 < m.numUsed)) break;
            if (k + 1 == nA) { deep = true; break; }
            lv = lvN;
            hd = hdN;
            w = wN;
            if (++n == m.numPer) n = 0;
        }
        if (deep) {
            if (lane == 0) o.deep_q[atomicAdd(o.deep_count, 1u)] = wid;
        } else {
            wvm_emit<PW_, PH_>(g, m, o, wid, lane, level, fout, thr, px);
        }
        wave_sync();
    }
}

// DPP broadcast of lane V of every row of 16 lanes (a quarter of the wavefront) to the lanes of that row
// (v_mov_b32_dpp row_newbcast).  The empty asm keeps the compiler from folding the DPP control into the consuming VALU
// instruction: a folded `v_subrev_u32_dpp ... row_newbcast` returned wrong values on gfx950 (every window of a test frame
// left the cascade two levels early), the plain move is fine.
template <int V>
__device__ __forceinline__ int row_bcast_i(int x) {
    int y = __builtin_amdgcn_update_dpp(0, x, 0x150 + V, 0xf, 0xf, false);   // row_newbcast:V
    asm volatile("" : "+v"(y));
    return y;
}
#define FD_ROW_BCAST_I(x, V) row_bcast_i<V>(x)
template <int V>
__device__ __forceinline__ double row_bcast_d(double v) {
    const int lo = row_bcast_i<V>(__double2loint(v)), hi = row_bcast_i<V>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// ---- stage A, four windows per wavefront -----------------------------------------------------------
// Phase timing of the pair kernel: HistEq64 26 %, integral images 13 %, the filter loop 60 %, and the filter loop is bound by
// instruction issue: the wave-uniform part of a filter (the fp64 grey-value chain, exp, thresholds, prefetch, control:
// ~190 of ~250 instructions) runs once per instruction for only two windows.  Here a wavefront owns four consecutive
// windows: the fixed part runs twice in the pair layout above (lanes 0-31 / 32-63 = the two windows of a pair, lane ==
// column) and leaves four integral images in LDS; in the filter loop the rect sums still run in the pair layout (lane ==
// rect, 32 rects per pass, once per pair, the level record loaded once), but everything wave-uniform runs in a quarter
// layout: lanes 16q..16q+15 belong to window q, lane r of a quarter holds the running sum of cascade level r
// (WVM_LCAP == 16).  Only used for cascades that continue in k_wvm_deep (numUsed > WVM_LCAP), so no window can turn
// positive here and the equalised patches need not be kept.
template <int PW_, int PH_>
struct __attribute__((aligned(16))) QuadLds {
    unsigned int hist[2][64];   // histograms of the current pair, then its two LUTs
    int sv[4][WVM_MAX_VALS];
    float u[4][32];             // u_kernel_eval of each window
    unsigned int ii[4][PW_ * PH_];
};

constexpr int WVM_QUAD_WAVES = 2;   // wavefronts per workgroup (LDS granularity: 24x24 -> 7 workgroups per CU)

template <int PW_, int PH_, bool RAW>
__global__ __launch_bounds__(64 * WVM_QUAD_WAVES) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_wvm_cascade4(const uint8_t* __restrict__ arena,
                                                                                                                   WinTable wt, WvmDev m, CascadeOut o) {
    static_assert(PW_ > 0 && PW_ <= 32, "pair layout needs a compile-time width <= 32");
    static_assert(WVM_LCAP == 16, "quarter layout: one lane per cascade level of stage A");
    __shared__ QuadLds<PW_, PH_> lds[WVM_QUAD_WAVES];
    __shared__ int64_t sFirst[WVM_MAX_LAYERS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    QuadLds<PW_, PH_>& L = lds[wave];
    const int half = lane >> 5, c = lane & 31;   // pair layout
    const int q = lane >> 4, r = lane & 15;      // quarter layout
    const bool colok = c < PW_;
    constexpr int d = PW_ * PH_;
    const int64_t nwaves = (int64_t)gridDim.x * WVM_QUAD_WAVES;
    const int F = m.numFilters;
    const int nA = min(m.numUsed, WVM_LCAP);
    const int halfBase = lane & 32, quarterBase = lane & 48;

    if (!RAW) {
        if (threadIdx.x < WVM_MAX_LAYERS) sFirst[threadIdx.x] = (int)threadIdx.x < wt.n ? wt.l[threadIdx.x].first : INT64_MAX;
        __syncthreads();
    }
    L.sv[q][r] = 0;
    wave_sync();

    const int64_t totalA = wt_total(wt);
    const int64_t nquads = (totalA + 3) >> 2;
    for (int64_t quad = (int64_t)blockIdx.x * WVM_QUAD_WAVES + wave; quad < nquads; quad += nwaves) {
        // level-0 model data: requested now, consumed after the fixed part
        uint4 lv = m.lvlRec[c];
        uint4 lq = m.lvlRec[r];          // quarter layout: lane r holds val[r] of the level (z, w)
        WvmLevelHdr hd = m.lvlHdr[0];
        float w = m.wT[r];

        float sxxP0 = 0.f, sxxP1 = 0.f;
        int sxtP0 = 0, sxtP1 = 0;
#pragma unroll 1
        for (int p = 0; p < 2; ++p) {
            // ---- fixed part of the pair (windows 4*quad + 2p, + 2p + 1), as in k_wvm_cascade2
            const int64_t widx0 = 4 * quad + 2 * p, widx1 = widx0 + 1;
            const bool has0 = widx0 < totalA, has1 = widx1 < totalA;
            const int64_t wid0 = wt_wid(wt, has0 ? widx0 : 4 * quad), wid1 = has1 ? wt_wid(wt, widx1) : wid0;
            int stride0, stride1;
            const uint8_t* src0 = wvm_locate<RAW>(arena, wt, sFirst, wid0, lane, PW_, d, stride0);
            const uint8_t* src1 = src0;
            stride1 = stride0;
            if (has1) src1 = wvm_locate<RAW>(arena, wt, sFirst, wid1, lane, PW_, d, stride1);
            const uint8_t* src = half ? src1 : src0;
            const int stride = half ? stride1 : stride0;
            unsigned int px[PH_];
            {
                const uint8_t* sp = src + (colok ? c : 0);
#pragma unroll
                for (int rr = 0; rr < PH_; ++rr) px[rr] = sp[(size_t)rr * stride];
            }
            if (!RAW) {
                L.hist[0][lane] = 0;
                L.hist[1][lane] = 0;
                wave_sync();
                if (colok) {
#pragma unroll
                    for (int rr = 0; rr < PH_; ++rr) atomicAdd(&L.hist[half][px[rr] >> 2], 1u);
                }
                wave_sync();
                const float pdfA = (float)L.hist[0][lane] * m.stretch, pdfB = (float)L.hist[1][lane] * m.stretch;
                float xA = pdfA, xB = pdfB;
#pragma unroll
                for (int t = 1; t < 64; ++t) {
                    const float sa = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(xA), DPP_WAVE_SHR1, 0xf, 0xf, true));
                    const float sb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(xB), DPP_WAVE_SHR1, 0xf, 0xf, true));
                    xA = sa + pdfA;
                    xB = sb + pdfB;
                }
                wave_sync();
                L.hist[0][lane] = (unsigned int)(unsigned char)floor((double)xA + 0.5);
                L.hist[1][lane] = (unsigned int)(unsigned char)floor((double)xB + 0.5);
                wave_sync();
#pragma unroll
                for (int rr = 0; rr < PH_; ++rr) {
                    const unsigned int e = L.hist[half][px[rr] >> 2];
                    px[rr] = colok ? e : 0u;
                }
            } else {
#pragma unroll
                for (int rr = 0; rr < PH_; ++rr) px[rr] = colok ? px[rr] : 0u;
            }
            int colsum = 0;
            float sxxc = 0.f;
            unsigned int* iiw = L.ii[2 * p + half];
#pragma unroll
            for (int rr = 0; rr < PH_; ++rr) {
                colsum += scan_half((int)px[rr]);
                if (colok) iiw[rr * PW_ + c] = (unsigned int)colsum;
                const float qf = (float)scan_half((int)(px[rr] * px[rr]));
                sxxc = rr == 0 ? qf : sxxc + qf;
            }
            const float sxxH = __int_as_float(__builtin_amdgcn_ds_bpermute((lane | 31) << 2, __float_as_int(sxxc)));
            const int sxtH = __builtin_amdgcn_ds_bpermute((halfBase + PW_ - 1) << 2, colsum);
            if (p == 0) { sxxP0 = sxxH; sxtP0 = sxtH; }
            else { sxxP1 = sxxH; sxtP1 = sxtH; }
            wave_sync();
        }
        // ---- quarter layout: window q of the quad comes from half q & 1 of pair q >> 1
        const bool valid = 4 * quad + q < totalA;
        const int64_t myWid = wt_wid(wt, valid ? 4 * quad + q : 4 * quad);
        float sxx;
        int sx_total;
        {
            const float a0 = readlane_f(sxxP0, 0), a1 = readlane_f(sxxP0, 32), a2 = readlane_f(sxxP1, 0), a3 = readlane_f(sxxP1, 32);
            const int b0 = __builtin_amdgcn_readlane(sxtP0, 0), b1 = __builtin_amdgcn_readlane(sxtP0, 32);
            const int b2 = __builtin_amdgcn_readlane(sxtP1, 0), b3 = __builtin_amdgcn_readlane(sxtP1, 32);
            sxx = q == 0 ? a0 : (q == 1 ? a1 : (q == 2 ? a2 : a3));
            sx_total = q == 0 ? b0 : (q == 1 ? b1 : (q == 2 ? b2 : b3));
        }
        L.u[q][r] = 0.f;
        L.u[q][r + 16] = 0.f;
        wave_sync();

        // ---- first levels of the cascade, the four windows in lockstep
        float Pb = m.negBias;   // lane r of each quarter: running sum of level r
        bool alive = valid, deep = false;
        int level = 0, n = 0;
        float fout = 0.f, thr = 0.f;
        for (int k = 0;; ++k) {
            const int kn = min(k + 1, nA - 1);
            const uint4 lvN = m.lvlRec[(size_t)kn * 64 + c];
            const uint4 lqN = m.lvlRec[(size_t)kn * 64 + r];
            const WvmLevelHdr hdN = m.lvlHdr[kn];
            const float wN = m.wT[(size_t)kn * F + r];
            // rect sums in the pair layout: 32 rects per pass, the rect record loaded once for both pairs
            const unsigned long long aliveBits = __ballot(alive);
            const bool aliveP0 = (aliveBits >> (16 * half)) & 1, aliveP1 = (aliveBits >> (32 + 16 * half)) & 1;
            for (int rb = 0; rb < hd.nrects; rb += 32) {
                unsigned int rc, vt;
                if (rb == 0) { rc = lv.x; vt = lv.y; }
                else if (rb < 64) { const uint4 t = m.lvlRec[(size_t)k * 64 + rb + c]; rc = t.x; vt = t.y; }
                else { const int ri = m.rectBegin[k] + min(rb + c, hd.nrects - 1); rc = m.rects[ri]; vt = m.rectV[ri]; }
                if (rb + c < hd.nrects) {
                    const int x1 = rc & 255, y1 = (rc >> 8) & 255, x2 = (rc >> 16) & 255, y2 = rc >> 24;
                    const int i22 = y2 * PW_ + x2, i21 = y2 * PW_ + x1 - 1, i12 = (y1 - 1) * PW_ + x2, i11 = (y1 - 1) * PW_ + x1 - 1;
                    if (aliveP0) {
                        const unsigned int* ii = L.ii[half];
                        int s = (int)ii[i22];
                        if (x1 > 0) s -= (int)ii[i21];
                        if (y1 > 0) s -= (int)ii[i12];
                        if (x1 > 0 && y1 > 0) s += (int)ii[i11];
                        atomicAdd(&L.sv[half][vt], s);
                    }
                    if (aliveP1) {
                        const unsigned int* ii = L.ii[2 + half];
                        int s = (int)ii[i22];
                        if (x1 > 0) s -= (int)ii[i21];
                        if (y1 > 0) s -= (int)ii[i12];
                        if (x1 > 0 && y1 > 0) s += (int)ii[i11];
                        atomicAdd(&L.sv[2 + half][vt], s);
                    }
                }
            }
            wave_sync();
            // the reference's scalar chain (WvmClassifier.cpp:308-346), once per quarter: lane v of the quarter takes the sum of
            // grey value v (and clears it) and its product with val[v]; the chain runs in value order on row broadcasts
            const int svr = L.sv[q][r];
            L.sv[q][r] = 0;
            const double valL = __hiloint2double((int)lq.w, (int)lq.z);
            const double prod = (double)svr * valL;
            const int cntval = hd.cntval;
            double sum_xp = 0.0;
            int sumv0 = sx_total;
#define FD_CHAIN_STEP(V)                                                      \
    if ((V) < cntval) {                                                       \
        sumv0 -= FD_ROW_BCAST_I(svr, V);                                      \
        sum_xp = sum_xp + row_bcast_d<V>(prod);                               \
    }
            FD_CHAIN_STEP(1) FD_CHAIN_STEP(2) FD_CHAIN_STEP(3) FD_CHAIN_STEP(4) FD_CHAIN_STEP(5)
            FD_CHAIN_STEP(6) FD_CHAIN_STEP(7) FD_CHAIN_STEP(8) FD_CHAIN_STEP(9) FD_CHAIN_STEP(10)
            FD_CHAIN_STEP(11) FD_CHAIN_STEP(12) FD_CHAIN_STEP(13) FD_CHAIN_STEP(14) FD_CHAIN_STEP(15)
#undef FD_CHAIN_STEP
            const double t0 = (double)sumv0 * row_bcast_d<0>(valL);
            sum_xp = sum_xp + t0;
            sum_xp = sum_xp + (double)L.u[q][n];
            const float unew = (float)sum_xp;
            wave_sync();
            if (r == 0) L.u[q][n] = unew;
            double norm = (double)sxx;
            norm = norm - 2 * sum_xp;
            norm = norm + hd.pp;
            const float Kk = (float)exp((double)m.negBasis * norm);
            {
                const float t = w * Kk;   // weights above the diagonal are stored as 0
                Pb = Pb + t;
            }
            const float fk = __int_as_float(__builtin_amdgcn_ds_bpermute((quarterBase + k) << 2, __float_as_int(Pb)));
            if (alive) {
                if (!(fk >= hd.thr && k + 1 < m.numUsed)) {   // leaves the cascade here
                    level = k; fout = fk; thr = hd.thr;
                    alive = false;
                } else if (k + 1 == nA) {                     // survives stage A: finished by k_wvm_deep
                    deep = true;
                    alive = false;
                }
            }
            if (!__any(alive)) break;
            lv = lvN;
            lq = lqN;
            hd = hdN;
            w = wN;
            if (++n == m.numPer) n = 0;
        }
        // ---- results, per quarter
        if (valid && r == 0) {
            if (deep) {
                o.deep_q[atomicAdd(o.deep_count, 1u)] = myWid;
            } else {
                if (o.all_level) o.all_level[myWid] = level;
                if (o.all_fout) o.all_fout[myWid] = fout;
            }
        }
        (void)thr;
        wave_sync();
    }
}

// ---- stage B: one workgroup per surviving window ---------------------------------------------------
// The kernel values K_k of different filters are independent of each other except through
// u_kernel_eval[k % numPer] (written numPer filters earlier), so the four waves evaluate disjoint
// residue classes n = k % numPer concurrently (wave j: n = j, j+4, ...), keeping their u values in
// registers.  After every chunk of whole "generations" (numPer filters each, at most 64 filters) one
// wave forms the hierarchical sums of the chunk's filters lane-parallel -- lane == filter, each lane
// adding its terms in the reference order -- and the first failed threshold ends the window.
template <int PW_, int PH_, bool RAW, int NW>
__global__ __launch_bounds__(64 * NW) void k_wvm_deep(const uint8_t* __restrict__ arena, WinTable wt, WvmDev m, CascadeOut o) {
    constexpr int RHMAX = Geo<PW_, PH_>::RHMAX;
    constexpr int MAXOWN = (WVM_PJ + NW - 1) / NW;   // 64-level blocks owned by one wave (block j belongs to wave j % NW)
    __shared__ unsigned int ii[Geo<PW_, PH_>::IISZ];
    __shared__ unsigned int hist[NW][64];
    __shared__ int sv[NW][WVM_MAX_VALS];
    __shared__ float kh[64 * WVM_PJ];
    __shared__ int64_t sFirst[WVM_MAX_LAYERS];
    __shared__ unsigned long long sExit;   // (first failed level << 32 | fp32 bits of its sum), minimum over the candidates
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const Geo<PW_, PH_> g(m, lane);
    const int F = m.numFilters, NU = m.numUsed, NP = m.numPer;
    const int gensPerChunk = max(1, 64 / NP);
    const int chunk = gensPerChunk * NP;

    if (!RAW && threadIdx.x < WVM_MAX_LAYERS) sFirst[threadIdx.x] = (int)threadIdx.x < wt.n ? wt.l[threadIdx.x].first : INT64_MAX;
    if (lane < WVM_MAX_VALS) sv[wave][lane] = 0;
    __syncthreads();
    const unsigned int ndeep = *o.deep_count;

    for (unsigned int q = blockIdx.x; q < ndeep; q += gridDim.x) {
        const int64_t wid = o.deep_q[q];
        int srcStride;
        const uint8_t* src = wvm_locate<RAW>(arena, wt, sFirst, wid, lane, g.pw, g.d, srcStride);
        unsigned int px[RHMAX];
        float sxx;
        int sx_total;
        // every wave prepares the window (identical values; they all write the one integral image)
        wvm_prepare<PW_, PH_, RAW>(g, src, srcStride, m.stretch, lane, hist[wave], ii, px, sxx, sx_total);
        if (threadIdx.x == 0) sExit = ~0ull;
        __syncthreads();

        float u = 0.f;   // lane n holds u_kernel_eval[n] (only this wave's classes are used)
        // running hierarchical sums of the levels this wave owns (lane == level inside the block), in reference order:
        // every chunk appends its kernel values to ALL later levels, so no level ever has to catch up
        float Pacc[MAXOWN];
#pragma unroll
        for (int ow = 0; ow < MAXOWN; ++ow) Pacc[ow] = m.negBias;
        int level = NU - 1;
        float fout = 0.f;
        for (int c0 = 0; c0 < NU; c0 += chunk) {
            const int c1 = min(c0 + chunk, NU);
            // ---- kernel values of this wave's filters in [c0, c1): generation-major, classes wave, wave+NW, ...
            {
                int k = c0 + wave;          // c0 is a multiple of NP
                int n = wave, gbase = c0;
                if (n >= NP) k = c1;
                uint4 lv;
                WvmLevelHdr hd;
                if (k < c1) { lv = m.lvlRec[(size_t)k * 64 + lane]; hd = m.lvlHdr[k]; }
                while (k < c1) {
                    int n2 = n + NW, gb2 = gbase;
                    if (n2 >= NP) { n2 = wave; gb2 += NP; }
                    const int k2 = gb2 + n2;
                    const int kp = k2 < c1 ? k2 : k;
                    const uint4 lvN = m.lvlRec[(size_t)kp * 64 + lane];
                    const WvmLevelHdr hdN = m.lvlHdr[kp];
                    float unew;
                    const float Kk = wvm_level_K(m, ii, sv[wave], g.pw, lane, k, lv, hd, sx_total, sxx, readlane_f(u, n), unew);
                    u = (lane == n) ? unew : u;
                    if (lane == 0) kh[k] = Kk;
                    k = k2; n = n2; gbase = gb2;
                    lv = lvN;
                    hd = hdN;
                }
            }
            __syncthreads();
            // ---- hierarchical sums: add the chunk's terms to every owned level >= c0; check the levels inside the chunk
#pragma unroll
            for (int ow = 0; ow < MAXOWN; ++ow) {
                const int blk = wave + ow * NW;
                if (blk * 64 >= NU || blk * 64 + 63 < c0) continue;   // nothing owned here / already decided
                const int mm = blk * 64 + lane;
                const float* wp = m.wT + mm;
                float P = Pacc[ow];
                const float* wq = wp + (size_t)c0 * F;
#pragma unroll 8
                for (int i = c0; i < c1; ++i, wq += F) {
                    const float t = *wq * kh[i];
                    P = P + t;
                }
                Pacc[ow] = P;
                const bool mine = mm >= c0 && mm < c1;
                const float thrm = mine ? m.thresholds[mm] : 0.f;
                const bool fail = mine && !(P >= thrm && mm + 1 < NU);
                const unsigned long long fm = __ballot(fail);
                if (fm) {
                    const int e = __builtin_ctzll(fm);
                    if (lane == e) atomicMin(&sExit, ((unsigned long long)(unsigned int)mm << 32) | (unsigned int)__float_as_int(P));
                }
            }
            __syncthreads();
            const unsigned long long ex = sExit;
            if (ex != ~0ull) {
                level = (int)(ex >> 32);
                fout = __int_as_float((int)(unsigned int)ex);
                break;
            }
        }
        if (wave == 0) wvm_emit<PW_, PH_>(g, m, o, wid, lane, level, fout, m.thresholds[level], px);
        __syncthreads();
    }
    wvm_finalize(o);
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

// Launches stage A over all windows and stage B over its survivors (same stream, no host round trip:
// stage B is a persistent grid that reads the survivor count from device memory).
#include "wvm_stageb.hpp"

template <int PW_, int PH_, bool RAW>
void launch_sized(fd_ctx* ctx, hipStream_t st, int64_t total, fd_wvm* mh, const uint8_t* arena, const WinTable& wt,
                  const CascadeOut& o, bool skipA = false) {
    const WvmDev& dev = mh->dev;
    // Exact stage A (every window, when per-window outputs are requested or the model has no dense pre-filter): persistent grids,
    // exactly as many workgroups as fit on the device at once.  skipA: the dense pre-filter has filled the stage-B queue itself.
    bool launched = skipA;
    if constexpr (PW_ > 0 && PW_ <= 24 && PW_ * PH_ <= 576) {
        // four windows per wavefront where four integral images leave enough LDS for the occupancy of the fixed part; only for
        // cascades that continue in stage B (no window turns positive in the first WVM_LCAP filters)
        if (!launched && dev.numPer <= 32 && dev.numUsed > WVM_LCAP) {
            static int perCu4q = 0;
            if (perCu4q == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu4q, k_wvm_cascade4<PW_, PH_, RAW>, 64 * WVM_QUAD_WAVES, 0) != hipSuccess || perCu4q < 1))
                perCu4q = 4;
            const int grid4 = (int)std::min<int64_t>((total + 4 * WVM_QUAD_WAVES - 1) / (4 * WVM_QUAD_WAVES), (int64_t)ctx->num_cus * perCu4q * 2);
            hipLaunchKernelGGL((k_wvm_cascade4<PW_, PH_, RAW>), dim3(grid4), dim3(64 * WVM_QUAD_WAVES), 0, st, arena, wt, dev, o);
            launched = true;
        }
    }
    if (!launched) {   // one window per wavefront: any patch size, any filters per level
        static int perCuA = 0;
        if (perCuA == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCuA, k_wvm_cascade<PW_, PH_, RAW>, 256, 0) != hipSuccess || perCuA < 1)) perCuA = 4;
        const int gridA = (int)std::min<int64_t>((total + 3) / 4, (int64_t)ctx->num_cus * perCuA * 2);   // two full rounds
        hipLaunchKernelGGL((k_wvm_cascade<PW_, PH_, RAW>), dim3(gridA), dim3(256), 0, st, arena, wt, dev, o);
    }
    if (dev.numUsed <= WVM_LCAP) return;
    // stage B as dense contractions (wvm_stageb.hpp) -- every model whose rect counts fit the int8 operand
    if (mh->sbRun) {
        launch_stageb<PW_, PH_, RAW>(ctx, st, total, mh, arena, wt, o);
        return;
    }
    // the rect-lookup stage B (k_wvm_deep: one workgroup of eight wavefronts per queued window): models the int8 operand cannot
    // express (more than 127 rects of one grey value on a pixel), FD_WVM_STAGEB=old (the cross-check of the tests)
    static int perCu8 = 0;
    if (perCu8 == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu8, k_wvm_deep<PW_, PH_, RAW, 8>, 512, 0) != hipSuccess || perCu8 < 1)) perCu8 = 1;
    const int gridB = (int)std::min<int64_t>(total, (int64_t)ctx->num_cus * perCu8);
    hipLaunchKernelGGL((k_wvm_deep<PW_, PH_, RAW, 8>), dim3(gridB), dim3(512), 0, st, arena, wt, dev, o);
}

// patch sizes with compile-time geometry: the detectors of ffpDetectApp/*.cfg (20x20 faces, 24x24 lip / nose / eye
// corners, 16x24 ears, 32x16 eyes, 32x24 nose tip); anything else up to 32x32 takes the run-time-sized instance
#define FD_WVM_SIZES(X) X(20, 20) X(24, 24) X(16, 24) X(32, 16) X(32, 24)

template <bool RAW>
void launch_cascade(fd_ctx* ctx, hipStream_t st, int64_t total, fd_wvm* mh, const uint8_t* arena, const WinTable& wt,
                    const CascadeOut& o, bool skipA = false) {
#define FD_WVM_CASE(W, H) \
    if (mh->dev.fw == W && mh->dev.fh == H) return launch_sized<W, H, RAW>(ctx, st, total, mh, arena, wt, o, skipA);
    FD_WVM_SIZES(FD_WVM_CASE)
#undef FD_WVM_CASE
    launch_sized<0, 0, RAW>(ctx, st, total, mh, arena, wt, o, skipA);
}

}  // namespace

#include "wvm_dense.hpp"
#include "wvm_dense_group.hpp"
#include "fs_tail.hpp"

// ---- host side ---------------------------------------------------------------------------------

// Dense stage of a model (wvm_dense.hpp): residual images of the first L filters, quantised to 32-bit integers and split into
// four balanced base-256 digits in the B-operand layout of v_mfma_i32_32x32x32_i8.  L = 0 (no dense stage) when the model does
// not qualify: fewer than 17 used filters (the exact stage-A kernels would emit positives themselves), fewer than 4 filters per
// level, a patch size without a compile-time kernel, or FD_WVM_DENSE=0.
static void wvd_dev_from(const fd_wvm* m, int64_t* q, unsigned int* qcount, WvdDev& dv);
static void wvm_build_dense(fd_wvm* m, const fd_wvm_model* md) {
    m->denseL = 0;
    static const bool off = [] { const char* e = getenv("FD_WVM_DENSE"); return e && atoi(e) == 0; }();
    if (off) return;
    const int pw = md->filter_w, ph = md->filter_h, d = pw * ph;
    bool sized = false;
#define FD_WVM_CASE(W, H) if (pw == W && ph == H) sized = true;
    FD_WVM_CASE(20, 20) FD_WVM_CASE(24, 24) FD_WVM_CASE(16, 24) FD_WVM_CASE(32, 16) FD_WVM_CASE(32, 24)
#undef FD_WVM_CASE
    const int numUsed = m->dev.numUsed, numPer = m->dev.numPer;
    if (!sized || numUsed <= WVM_LCAP || numPer > 32) return;
    const int L = std::min(std::min(WVD_L, numPer), numUsed - 1);
    if (L < 4) return;
    // r_k[y][x] = val_k[0] + sum_{v >= 1} (val_k[v] - val_k[0]) * #{rects of v covering (x, y)}   
    std::vector<double> r((size_t)L * d);
    double rmax = 0;
    for (int k = 0; k < L; ++k) {
        const int v0 = md->val_off[k], cntval = md->val_off[k + 1] - v0;
        double* rk = &r[(size_t)k * d];
        for (int i = 0; i < d; ++i) rk[i] = md->val[v0];
        for (int v = 1; v < cntval; ++v) {
            const double dv = md->val[v0 + v] - md->val[v0];
            for (int ri = md->rec_off[v0 + v]; ri < md->rec_off[v0 + v + 1]; ++ri) {
                const uint8_t* rc = md->rects + 4 * (size_t)ri;
                for (int y = rc[1]; y <= rc[3]; ++y)
                    for (int x = rc[0]; x <= rc[2]; ++x) rk[y * pw + x] += dv;
            }
        }
        for (int i = 0; i < d; ++i) {
            if (!std::isfinite(rk[i])) return;
            rmax = std::max(rmax, std::fabs(rk[i]));
        }
    }
    if (!std::isfinite((double)md->basis_param) || !std::isfinite((double)md->bias)) return;
    int sh = 40;
    if (rmax > 0) sh = std::min(40, (int)std::floor(30.0 - std::log2(rmax)));
    if (sh < 4) return;   // grey values far outside the 0..255 domain: nothing to gain
    const double up = std::ldexp(1.0, sh);
    WvdConst C;
    std::memset(&C, 0, sizeof(C));
    m->denseScale = std::ldexp(1.0, -sh);
    // k-step ks, slot t = pixel 32 ks + t of the row-major patch (k_wvm_prefilter walks the patch as a flat run of dwords)
    const int KS = (d + 31) / 32;
    std::vector<int8_t> B((size_t)((KS + 3) / 4 * 4) * 2 * 64 * 16, 0);   // zero k-steps up to a multiple of four: k_wvm_prefilter_group's fragment ring runs over whole rounds
    const double nb2 = (double)m->dev.negBasis * 1.4426950408889634;   // log2(e) * (-basis): K = 2^(nb2 * norm)
    for (int k = 0; k < WVD_L; ++k) C.thr[k] = -INFINITY;              // levels past L never reject
    for (int k = 0; k < L; ++k) {
        double sumQ = 0;
        for (int i = 0; i < d; ++i) {
            long long Q = std::llround(r[(size_t)k * d + i] * up);   // |Q| <= 2^30
            sumQ += (double)Q;
            int dig[4];
            for (int j = 0; j < 4; ++j) {
                const int q0 = (int)(((Q + 128) & 255) - 128);
                dig[j] = q0;
                Q = (Q - q0) / 256;   // exact
            }
            if (Q != 0) return;   // cannot happen for |Q| < 2^31 - 2^23
            const int ks = i / 32, slot = i % 32, h = slot / 16, t = slot % 16;
            for (int j = 0; j < 4; ++j) {
                const int g = k + 16 * j, nt = g / 32, col = g % 32;
                B[((((size_t)ks * 2 + nt) * 64) + (h * 32 + col)) * 16 + t] = (int8_t)dig[j];
            }
        }
        // x . Q = x' . Q + 128 sum Q (x' = x - 128); xp = 2^-s x . Q; norm = sxx - 2 xp + pp
        C.cA[k] = nb2 * ((double)md->pp[k] - 2.0 * m->denseScale * (128.0 * sumQ));
        C.thr[k] = md->thresholds[k];
        for (int pidx = 0; pidx <= k; ++pidx) {
            const float w = md->hk_weights[(size_t)k * md->num_filters + pidx];
            C.w2[k][pidx][0] = w;
            C.w2[k][pidx][1] = std::fabs(w);
        }
        if (!std::isfinite(C.cA[k]) || !std::isfinite((double)C.thr[k])) return;
    }
    m->denseB.reserve(B.size());
    HIP_CHECK(hipMemcpy(m->denseB.p, B.data(), B.size(), hipMemcpyHostToDevice));
    m->denseL = L;
    {   // the same values wvd_dev_from hands to k_wvm_prefilter as kernel arguments
        WvdDev dv;
        wvd_dev_from(m, nullptr, nullptr, dv);
        C.sc.L = dv.L; C.sc.negBasis = dv.negBasis; C.sc.negBias = dv.negBias; C.sc.stretch = dv.stretch; C.sc.sxxSlack = dv.sxxSlack;
        C.sc.scale = dv.scale; C.sc.nb2 = dv.nb2; C.sc.mXq = dv.mXq;
    }
    m->denseC.reserve(sizeof(C));
    HIP_CHECK(hipMemcpy(m->denseC.p, &C, sizeof(C), hipMemcpyHostToDevice));
}

// Tables of the dense stage B (wvm_stageb.hpp), pure host part.  Returns false (the rect-lookup stage-B kernels run instead) when
// the model never reaches stage B or when more than 127 rects of one grey value overlap on a pixel.
struct WvbTables {
    int KS = 0, KSP = 0, DS = 0, Fr = 0, nphase = 0, ntile = 0, maxCnt = 1;
    int phaseGen[WVB_MAXPHASE + 1] = {};
    std::vector<int32_t> lvl, c128, rec;
    std::vector<int8_t> A;
    std::vector<float> wR;
};
static bool wvb_tables(const fd_wvm_model* md, int NU, WvbTables& T) {
    const int F = md->num_filters, NP = md->num_per_level;
    if (NU <= WVM_LCAP) return false;
    // the callers outside fd_wvm_create (test hooks) bring unvalidated models: everything the loops below index with is checked here
    if (F < 1 || NU > F || NP < 1 || md->filter_w < 1 || md->filter_h < 1 || !md->val_off || !md->rec_off || !md->rects || !md->val || !md->pp ||
        !md->hk_weights) return false;
    if (md->val_off[0] != 0) return false;
    for (int k = 0; k < NU; ++k) {
        const int cnt = md->val_off[k + 1] - md->val_off[k];
        if (cnt < 1 || cnt > WVM_MAX_VALS) return false;   // a level is one slot of at most 15 rows, its record holds 16 grey values
    }
    for (int v = 0; v < md->val_off[NU]; ++v) {
        if (md->rec_off[v] < 0 || md->rec_off[v + 1] < md->rec_off[v]) return false;
        for (int ri = md->rec_off[v]; ri < md->rec_off[v + 1]; ++ri) {
            const uint8_t* rc = md->rects + 4 * (size_t)ri;
            if (rc[0] > rc[2] || rc[1] > rc[3] || rc[2] >= md->filter_w || rc[3] >= md->filter_h) return false;
        }
    }
    const int pw = md->filter_w, ph = md->filter_h, d = pw * ph;
    const int KS = (d + 31) / 32;
    const int KSP = (KS + 7) / 8 * 8;   // k-steps per tile in the operand table: whole groups of eight (zero fragments behind the patch)
    const int G = (NU + NP - 1) / NP;
    T.KS = KS;
    T.KSP = KSP;
    T.DS = KS * 32 + 16;
    {   // phases: generations [0, 2), [2, 6), [6, G) by default; FD_WVB_PHASES="a,b,c" sets other cuts
        std::vector<int> cuts = {2, 6};
        if (const char* e = getenv("FD_WVB_PHASES")) {
            cuts.clear();
            for (const char* c = e; *c;) {
                char* end = nullptr;
                const long v = std::strtol(c, &end, 10);
                if (end == c) break;
                if (v > 0) cuts.push_back((int)v);
                c = *end ? end + 1 : end;
            }
        }
        std::sort(cuts.begin(), cuts.end());
        T.nphase = 0;
        T.phaseGen[0] = 0;
        for (int c : cuts)
            if (c > T.phaseGen[T.nphase] && c < G && T.nphase + 1 < WVB_MAXPHASE) T.phaseGen[++T.nphase] = c;
        T.phaseGen[++T.nphase] = G;
    }
    T.lvl.assign((size_t)F * 4, 0);
    T.c128.clear();
    T.A.clear();
    std::vector<int> cover((size_t)d);
    int ntile = 0;
    for (int k = 0; k < NU; ++k) T.maxCnt = std::max(T.maxCnt, md->val_off[k + 1] - md->val_off[k]);
    // Fixed slots: every level takes 7 rows of its tile (15 when a filter has more than 8 grey values), unused rows stay zero -- generation g
    // of a class is slot g % 4 (g % 2) of the class's tile g / 4 (g / 2), so k_wvb_chain2 addresses a level's rect sums with constants
    // (they stay in the accumulator registers).  The operand table grows by the padding (7 rows for a 6-value filter); it stays in L2.
    const int RPL = T.maxCnt <= 8 ? 7 : 15;
    for (int n = 0; n < NP; ++n) {
        int cur = -1, rowc = 32;
        for (int g = 0; g < G; ++g) {
            const int k = g * NP + n;
            if (k >= NU) break;
            const int v0 = md->val_off[k], cntval = md->val_off[k + 1] - v0;
            const int rows = RPL;
            // a tile may hold levels of two phases: a phase then evaluates the whole tile and uses its own rows (the contraction is cheap)
            if (cur < 0 || rowc + rows > 32) {
                cur = ntile++;
                rowc = 0;
                T.A.resize((size_t)ntile * KSP * 64 * 16, 0);
                T.c128.resize((size_t)ntile * 32, 0);
            }
            int32_t* lv = &T.lvl[4 * (size_t)k];
            lv[0] = cur; lv[1] = rowc; lv[2] = cntval; lv[3] = v0;
            for (int v = 1; v < cntval; ++v) {
                std::fill(cover.begin(), cover.end(), 0);
                for (int ri = md->rec_off[v0 + v]; ri < md->rec_off[v0 + v + 1]; ++ri) {
                    const uint8_t* rc = md->rects + 4 * (size_t)ri;
                    for (int y = rc[1]; y <= rc[3]; ++y)
                        for (int x = rc[0]; x <= rc[2]; ++x) ++cover[(size_t)y * pw + x];
                }
                const int row = rowc + v - 1;
                long sum = 0;
                for (int i = 0; i < d; ++i) {
                    if (cover[i] > 127) return false;   // does not fit the int8 operand
                    sum += cover[i];
                    const int ks = i >> 5, h = (i >> 4) & 1, t = i & 15;   // lane h * 32 + row of k-step ks, byte t
                    T.A[((((size_t)cur * KSP + ks) * 64) + (size_t)(h * 32 + row)) * 16 + t] = (int8_t)cover[i];
                }
                T.c128[(size_t)cur * 32 + row] = (int32_t)(128 * sum);
            }
            rowc += rows;
        }
    }
    T.ntile = ntile;
    {   // k_wvb_chain2 computes a level's tile instead of reading it from the record: classes 0 .. rem - 1 have G generations, tiles of a class are consecutive
        const int LPT = 32 / RPL, rem = NU - (G - 1) * NP;
        const int tf = (G + LPT - 1) / LPT, tsh = (G - 1 + LPT - 1) / LPT;
        for (int k = 0; k < NU; ++k) {
            const int n = k % NP, g = k / NP;
            const int tile = std::min(n, rem) * tf + std::max(0, n - rem) * tsh + g / LPT;
            if (T.lvl[4 * (size_t)k] != tile || T.lvl[4 * (size_t)k + 1] != (g % LPT) * RPL) return false;
        }
    }
    T.A.resize((size_t)(ntile + 1) * KSP * 64 * 16, 0);   // a spare zero tile: the operand prefetch runs one tile ahead
    // the chain's per-level records (k_wvb_chain): one dword per lane
    T.rec.assign((size_t)F * 64, 0);
    for (int k = 0; k < NU; ++k) {
        int32_t* r = &T.rec[(size_t)k * 64];
        const int32_t* lv = &T.lvl[4 * (size_t)k];
        r[0] = lv[0]; r[1] = lv[1]; r[2] = lv[2];
        std::memcpy(r + 4, &md->pp[k], 8);
        for (int v = 0; v < lv[2]; ++v) {
            std::memcpy(r + 8 + 2 * v, &md->val[lv[3] + v], 8);
            if (v >= 1) r[40 + v] = T.c128[(size_t)lv[0] * 32 + lv[1] + v - 1];
        }
    }
    T.Fr = (F + 7) / 8 * 8 + 8;
    T.wR.assign((size_t)F * T.Fr, 0.f);
    for (int k = 0; k < F; ++k)
        for (int pidx = 0; pidx <= k; ++pidx) T.wR[(size_t)pidx * T.Fr + k] = md->hk_weights[(size_t)k * F + pidx];
    return true;
}

static void wvb_build(fd_wvm* m, const fd_wvm_model* md) {
    m->wvbOk = false;
    if (const char* e = getenv("FD_WVM_STAGEB")) if (!std::strcmp(e, "old")) return;   // read per model: the tests compare both
    WvbTables T;
    if (!wvb_tables(md, m->dev.numUsed, T)) return;
    WvbDev& mv = m->wvb;
    std::memset(&mv, 0, sizeof(mv));
    auto up = [&](DevBuf& b, const void* src, size_t bytes) {
        b.reserve(std::max<size_t>(bytes, 16));
        HIP_CHECK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    };
    up(m->wvbA, T.A.data(), T.A.size());
    up(m->wvbLvl, T.lvl.data(), sizeof(int32_t) * T.lvl.size());
    up(m->wvbC128, T.c128.data(), sizeof(int32_t) * T.c128.size());
    up(m->wvbWR, T.wR.data(), sizeof(float) * T.wR.size());
    up(m->wvbRec, T.rec.data(), sizeof(int32_t) * T.rec.size());
    m->sbCnt.reserve(64);
    HIP_CHECK(hipMemset(m->sbCnt.p, 0, 64));
    mv.A = m->wvbA.as<wvb_v4i>(); mv.lvl = m->wvbLvl.as<int4>(); mv.c128 = m->wvbC128.as<int32_t>();
    mv.pp = m->pp.as<double>(); mv.val = m->val.as<double>(); mv.thr = m->thresholds.as<float>(); mv.wR = m->wvbWR.as<float>(); mv.rec = m->wvbRec.as<int32_t>();
    mv.KS = T.KS; mv.KSP = T.KSP; mv.dstride = T.DS; mv.Fr = T.Fr;
    mv.numPer = md->num_per_level; mv.numUsed = m->dev.numUsed; mv.numFilters = md->num_filters; mv.d = md->filter_w * md->filter_h;
    mv.nphase = T.nphase;
    mv.maxCnt = T.maxCnt;
    for (int i = 0; i <= WVB_MAXPHASE; ++i) mv.phaseGen[i] = T.phaseGen[i];
    mv.negBasis = m->dev.negBasis; mv.negBias = m->dev.negBias;
    m->wvbOk = true;
}

// State buffers of a stage-B run over at most `total` queued windows (grow only); fills m->sb and m->deepCap.
// Capacity: min(total, 2^18) windows, never less than what the handle already holds or than minCap (fd_wvm_finish grows the state when
// a run queued more); FD_WVM_DEEP_CAP fixes it (then an overflow is an error).
static int64_t wvb_cap_env() {   // read per run (the tests provoke the overflow report)
    if (const char* e = getenv("FD_WVM_DEEP_CAP")) if (atoll(e) > 0) return (int64_t)atoll(e);
    return 0;
}
static void wvb_reserve(fd_wvm* m, int64_t total, int64_t minCap = 0) {
    const int64_t capEnv = wvb_cap_env();
    const WvbDev& mv = m->wvb;
    int64_t cap = capEnv ? capEnv : std::max<int64_t>(std::max<int64_t>(minCap, m->sbGrown), (int64_t)1 << 18);
    cap = std::max<int64_t>(64, std::min<int64_t>(total, cap));
    const int64_t tilesCap = (cap + 63) / 64;
    WvbState& s = m->sb;
    for (int i = 0; i < 2; ++i) {
        m->sbX[i].reserve((size_t)cap * mv.dstride);
        m->sbWid[i].reserve(sizeof(int64_t) * (size_t)cap);
        m->sbAux[i].reserve(sizeof(int2) * (size_t)cap);
        m->sbU[i].reserve(sizeof(float) * (size_t)mv.numPer * 64 * tilesCap);
        m->sbK[i].reserve(sizeof(float) * (size_t)mv.numUsed * 64 * tilesCap);
        s.X[i] = m->sbX[i].as<int8_t>(); s.wid[i] = m->sbWid[i].as<int64_t>(); s.aux[i] = m->sbAux[i].as<int2>();
        s.U[i] = m->sbU[i].as<float>(); s.K[i] = m->sbK[i].as<float>();
    }
    m->sbKey.reserve(sizeof(unsigned long long) * (size_t)cap);
    s.exitKey = m->sbKey.as<unsigned long long>();
    s.cnt = m->sbCnt.as<unsigned int>();
    s.cap = cap;
    m->deepCap = cap;
}

// k_wvm_prefilter's tiles for K windows per lane: 64 (column, row group of K) tasks of a layer each
static int wvd_plan_sliding(WvdTable& t, int K) {
    t.K = K;
    int tiles = 0;
    for (int i = 0; i < t.n; ++i) {
        WvdLayer& l = t.l[i];
        l.G = (l.ny + K - 1) / K;
        l.sTileFirst = tiles;
        tiles += (int)(((int64_t)l.nx * l.G + 63) / 64);
    }
    t.sTilesPerImage = tiles;
    return tiles;
}
// How many windows a lane walks down.  More is cheaper per window (the first window of a lane pays the full histogram, ~0.45 of a
// window's other work; each later one a slide of 2 sy rows, ~0.04 each) but makes the tiles longer: the launch takes
// rounds(K) x (first + K x step) with rounds = tiles / wavefront slots rounded up.  FD_WVD_K fixes it.
static int wvd_choose_k(WvdTable t, int slots, int ph) {
    static const int forced = [] { const char* e = getenv("FD_WVD_K"); const int v = e ? atoi(e) : 0; return v < 1 ? 0 : (v > WVD_KMAX ? WVD_KMAX : v); }();
    if (2 * t.sy > ph) return 1;   // windows of a column barely overlap: nothing to slide
    if (forced) return forced;
    int best = 1;
    double bestCost = 0;
    for (int K = 1; K <= WVD_KMAX; ++K) {
        const int64_t tiles = (int64_t)wvd_plan_sliding(t, K) * t.nimg;
        const double rounds = (double)((tiles + slots - 1) / slots);
        const double cost = rounds * (0.45 + K * (1.0 + (K > 1 ? 0.04 * t.sy : 0.0)));
        if (K == 1 || cost < bestCost * 0.999) { best = K; bestCost = cost; }
    }
    return best;
}

template <int PW_, int PH_>
static void launch_prefilter_sized(fd_ctx* ctx, hipStream_t st, const uint8_t* arena, WvdTable wt, const WvdDev& dv) {
    static int perCu = 0;
    if (perCu == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_wvm_prefilter<PW_, PH_>, 256, 0) != hipSuccess || perCu < 1)) perCu = 2;
    const int slots = ctx->num_cus * perCu * 4;
    const int64_t tiles = (int64_t)wvd_plan_sliding(wt, wvd_choose_k(wt, slots, PH_)) * wt.nimg;
    // rounds of resident workgroups the tiles are dealt over (FD_WVD_ROUNDS, default 2)
    static const int rounds = [] { const char* e = getenv("FD_WVD_ROUNDS"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > 64 ? 64 : v); }();
    int grid = (int)std::min<int64_t>((tiles + 3) / 4, (int64_t)ctx->num_cus * perCu * rounds);
    if (wt.nimg >= 8 && grid >= 64) grid &= ~7;   // a multiple of the 8 XCDs: the kernel then keeps every frame on one XCD
    hipLaunchKernelGGL((k_wvm_prefilter<PW_, PH_>), dim3(grid), dim3(256), 0, st, arena, wt, dv);
}

// window table of the dense kernels from the cascade's; false when the call does not qualify
static bool wvd_table_from(const WinTable& wt, WvdTable& t) {
    if (wt.raw || wt.list || wt.total < 512) return false;
    std::memset(&t, 0, sizeof(t));
    t.n = wt.n; t.sx = wt.sx; t.sy = wt.sy;
    const int64_t perImage = wt.nimg > 1 ? wt.per_image : wt.total;
    int tiles = 0;
    for (int i = 0; i < wt.n; ++i) {
        const WinLayerDev& s = wt.l[i];
        const int64_t nwin = (i + 1 < wt.n ? wt.l[i + 1].first : perImage) - s.first;
        if (nwin <= 0 || nwin > (int64_t)INT32_MAX - 64) return false;
        WvdLayer& dl = t.l[i];
        dl.bx = s.bx; dl.by = s.by; dl.nx = s.nx; dl.lw = s.lw; dl.off = s.off; dl.magic = s.magic; dl.first = s.first;
        dl.nwin = (int32_t)nwin;
        dl.ny = s.ny;
        if ((int64_t)s.nx * s.ny != nwin) return false;
        tiles += (int)((nwin + 63) / 64);
    }
    t.nimg = wt.nimg > 1 ? wt.nimg : 1;
    t.perImage = perImage;
    t.imageStride = wt.image_stride;
    if ((int64_t)tiles * t.nimg > (int64_t)INT32_MAX) return false;
    return true;
}
static void wvd_dev_from(const fd_wvm* m, int64_t* q, unsigned int* qcount, WvdDev& dv) {
    dv.B = m->denseB.as<wvd_v4i>();
    dv.c = m->denseC.as<WvdConst>();
    dv.q = q;
    dv.qcount = qcount;
    dv.L = m->denseL;
    dv.negBasis = m->dev.negBasis; dv.negBias = m->dev.negBias; dv.stretch = m->dev.stretch;
    dv.sxxSlack = (float)(2 * m->dev.fh + 2);
    dv.scale = m->denseScale;
    dv.nb2 = (double)m->dev.negBasis * 1.4426950408889634;
    dv.mXq = -2.0 * m->denseScale * dv.nb2;
}

// queues k_wvm_prefilter over all windows of `wt`; returns false when the model / call does not qualify
static bool launch_prefilter(fd_ctx* ctx, hipStream_t st, fd_wvm* m, const uint8_t* arena, const WinTable& wt, int64_t* q, unsigned int* qcount) {
    if (m->denseL == 0) return false;
    WvdTable t;
    if (!wvd_table_from(wt, t)) return false;
    WvdDev dv;
    wvd_dev_from(m, q, qcount, dv);
#define FD_WVM_CASE(W, H) \
    if (m->dev.fw == W && m->dev.fh == H) { launch_prefilter_sized<W, H>(ctx, st, arena, t, dv); return true; }
    FD_WVM_CASE(20, 20) FD_WVM_CASE(24, 24) FD_WVM_CASE(16, 24) FD_WVM_CASE(32, 16) FD_WVM_CASE(32, 24)
#undef FD_WVM_CASE
    return false;
}

// the same for a group of detectors with one patch size on one window table (wvm_dense_group.hpp)
template <int PW_, int PH_>
static void launch_prefilter_group_sized(fd_ctx* ctx, hipStream_t st, const uint8_t* arena, WvdTable wt, const WvdGroup& g) {
    static int perCu = 0;
    if (perCu == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_wvm_prefilter_group<PW_, PH_>, 256, 0) != hipSuccess || perCu < 1)) perCu = 2;
    const int slots = ctx->num_cus * perCu * 4;
    const int64_t tiles = (int64_t)wvd_plan_sliding(wt, wvd_choose_k(wt, slots, PH_)) * wt.nimg;
    static const int rounds = [] { const char* e = getenv("FD_WVD_ROUNDS"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > 64 ? 64 : v); }();
    int grid = (int)std::min<int64_t>((tiles + 3) / 4, (int64_t)ctx->num_cus * perCu * rounds);
    if (wt.nimg >= 8 && grid >= 64) grid &= ~7;
    hipLaunchKernelGGL((k_wvm_prefilter_group<PW_, PH_>), dim3(grid), dim3(256), 0, st, arena, wt, g);
}
// patch sizes whose equalised patch fits a lane's registers (KS <= 18 k-steps)
static bool wvd_group_size(int fw, int fh) { return (fw == 20 && fh == 20) || (fw == 24 && fh == 24) || (fw == 16 && fh == 24) || (fw == 32 && fh == 16); }
static bool launch_prefilter_group(fd_ctx* ctx, hipStream_t st, fd_wvm* const* ms, int n, const uint8_t* arena, const WinTable& wt, const CascadeOut* os) {
    if (n < 2 || n > WVD_GMAX) return false;
    WvdTable t;
    if (!wvd_table_from(wt, t)) return false;
    WvdGroup g;
    std::memset(&g, 0, sizeof(g));
    g.n = n;
    for (int i = 0; i < n; ++i) {
        const fd_wvm* m = ms[i];
        if (m->denseL == 0 || m->dev.fw != ms[0]->dev.fw || m->dev.fh != ms[0]->dev.fh) return false;
        g.m[i].B = m->denseB.as<wvd_v4i>();
        g.m[i].c = m->denseC.as<WvdConst>();
        g.m[i].q = os[i].deep_q;
        g.m[i].qcount = os[i].deep_count;
    }
#define FD_WVM_CASE(W, H) \
    if (ms[0]->dev.fw == W && ms[0]->dev.fh == H) { launch_prefilter_group_sized<W, H>(ctx, st, arena, t, g); return true; }
    FD_WVM_CASE(20, 20) FD_WVM_CASE(24, 24) FD_WVM_CASE(16, 24) FD_WVM_CASE(32, 16)
#undef FD_WVM_CASE
    return false;
}

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
    wt.nimg = p->nimg;
    wt.per_image = total;
    wt.image_stride = p->image_stride;
    wt.total = total * p->nimg;
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
    bool timed = false;
};

// Asynchronous half of a WVM run: enumerates the windows, launches both cascade stages on the context's stream,
// queues the read-back of the counter + first positives into the model's own pinned buffer and records m->done.
static void wvm_launch_table(fd_ctx* ctx, hipStream_t st, fd_pyramid* p, fd_wvm* m, const WinTable& wt, bool want_all, WvmRun& run, bool time_kernel);
void fd_wvm_launch_on(fd_ctx* ctx, hipStream_t st, fd_pyramid* p, fd_wvm* m, int sx, int sy, const int* roi, bool want_all, WvmRun& run, bool time_kernel);

void fd_wvm_launch(fd_ctx* ctx, fd_pyramid* p, fd_wvm* m, int sx, int sy, const int* roi, bool want_all, WvmRun& run, bool time_kernel) {
    fd_wvm_launch_on(ctx, ctx->stream, p, m, sx, sy, roi, want_all, run, time_kernel);
}

// explicit stream; touches no mutable context state (callable from the worker threads of the batch entry points)
void fd_wvm_launch_on(fd_ctx* ctx, hipStream_t st, fd_pyramid* p, fd_wvm* m, int sx, int sy, const int* roi, bool want_all, WvmRun& run, bool time_kernel) {
    if (p->ctx != ctx || m->ctx != ctx) FD_THROW(FD_ERR_INVALID_ARGUMENT, "objects belong to different contexts");
    if (p->filter_kind != FD_LAYER_NONE)
        FD_THROW(FD_ERR_INVALID_ARGUMENT, "WVM detection needs a gray pyramid (no layer filter)");
    if (p->all.empty()) FD_THROW(FD_ERR_RUNTIME, "pyramid has not been updated with an image");
    HIP_CHECK(hipSetDevice(ctx->device));
    fd_pyramid_wait(p, st);   // the batch entry points launch on pool streams, the caller may have updated on the context's
    WinTable wt;
    fd_wvm_build_table(p, m->dev.fw, m->dev.fh, sx, sy, roi, wt, run.wls);
    wvm_launch_table(ctx, st, p, m, wt, want_all, run, time_kernel);
}

// first half of a cascade launch: buffers, header, output descriptor (everything in front of the first kernel)
struct WvmLaunch {
    CascadeOut o;
    bool zc = false;
    bool headerMemset = false;   // a memset of the device header was queued on the stream
    bool queued = false;         // the head queued anything at all on the stream (header / counter clears)
};
static bool wvm_launch_head(fd_ctx* ctx, hipStream_t st, fd_wvm* m, const WinTable& wt, bool want_all, WvmRun& run, bool time_kernel, WvmLaunch& L) {
    run.total = wt.total;
    run.pos.clear();
    run.slots.clear();
    run.timed = time_kernel;
    const bool tailWanted = m->tailWanted;
    m->tailWanted = false;
    m->tailRun = false;
    const bool specWanted = m->specWanted;
    m->specWanted = false;
    m->specRun = false;
    m->fstLastState = -1;
    if (wt.total == 0) return false;
    if (want_all) {
        m->all_level.reserve(sizeof(int32_t) * (size_t)wt.total);
        m->all_fout.reserve(sizeof(float) * (size_t)wt.total);
    }
    {   // capacity of the positive buffers (records + 1 patch each): never more than the window count, 2^18 at most unless
        // FD_WVM_POS_CAP says otherwise; grows with the largest run seen (a 640x480 FaceFrontal handle holds 6.5 MB, not 105 MB)
        static const int64_t capEnv = [] { const char* e = getenv("FD_WVM_POS_CAP"); return e && atoll(e) > 0 ? (int64_t)atoll(e) : (int64_t)0; }();
        const int64_t want = capEnv ? capEnv : std::min<int64_t>(wt.total, (int64_t)1 << 18);
        if (want > m->pos_cap) { m->pos_cap = want; m->hdrClean = false; }
    }
    // pos buffer: record 0 is the header (positive counter), records 1.. are the positives, so that the
    // counter and the first records come back in a single read
    m->pos.reserve(sizeof(PosRec) * ((size_t)m->pos_cap + 1));
    m->pos_patches.reserve((size_t)m->dev.d * (size_t)m->pos_cap);
    m->h_pos.reserve(sizeof(PosRec) * ((size_t)m->pos_cap + 1));
    if (!m->done) HIP_CHECK(hipEventCreateWithFlags(&m->done, hipEventDisableTiming));
    m->deep_q.reserve(sizeof(int64_t) * (size_t)wt.total);
    // Zero-copy read-back when the run ends in a stage-B kernel (every production run of a model with more than WVM_LCAP
    // filters): positives and their count go straight to the pinned host buffer.  FD_WVM_ZEROCOPY=0 restores copy + memset.
    static const bool zcOff = [] { const char* e = getenv("FD_WVM_ZEROCOPY"); return e && atoi(e) == 0; }();
    const bool zc = !zcOff && !want_all && m->dev.numUsed > WVM_LCAP;
    m->zcRun = zc;
    m->tailRun = tailWanted && zc && m->wvbOk && wt.total < ((int64_t)1 << 32);   // decided per run; the caller queues the tail kernels iff it is set
    L.zc = zc;
    L.headerMemset = !(zc && m->hdrClean);
    if (L.headerMemset) HIP_CHECK(hipMemsetAsync(m->pos.p, 0, sizeof(PosRec), st));
    L.queued = L.headerMemset;
    m->sbRun = m->wvbOk && m->dev.numUsed > WVM_LCAP;
    if (m->sbRun) {
        wvb_reserve(m, wt.total);
        if (L.headerMemset) HIP_CHECK(hipMemsetAsync(m->sbCnt.p, 0, 64, st));   // phase counters (a zero-copy run clears them itself)
    }
    m->hdrClean = false;   // set again by fd_wvm_finish once a zero-copy run has completed
    CascadeOut& o = L.o;
    o.all_level = want_all ? m->all_level.as<int32_t>() : nullptr;
    o.all_fout = want_all ? m->all_fout.as<float>() : nullptr;
    o.pos = ((zc && !m->tailRun) ? m->h_pos.as<PosRec>() : m->pos.as<PosRec>()) + 1;   // pinned host memory is device-accessible under the same address
    if (m->tailRun) {
        // The tail's counters (FstHdr words, positives per frame) are cleared by k_fs_oe itself, so a run normally finds them zero.  A
        // run whose cascade was queued but whose k_fs_oe never was (an error between the two) leaves k_wvb_exit's counts
        // behind: fstDirty stays set from here until fst_launch has queued k_fs_oe, and a run that finds it set clears them first.
        const bool fresh = !m->fstHdr.p || !m->fstFrameCount.p;
        m->fstHdr.reserve(64);
        m->fstFrameCount.reserve(sizeof(unsigned int) * FD_MAX_FRAMES);
        if (fresh || m->fstDirty) {
            HIP_CHECK(hipMemsetAsync(m->fstHdr.p, 0, 64, st));
            HIP_CHECK(hipMemsetAsync(m->fstFrameCount.p, 0, sizeof(unsigned int) * FD_MAX_FRAMES, st));
            L.queued = true;
        }
        m->fstDirty = true;
        o.tail_count = m->fstHdr.as<unsigned int>() + 8;   // behind the FstHdr words
        const int nimg = wt.nimg > 1 ? wt.nimg : 1;
        m->fstFrameList.reserve(sizeof(uint32_t) * (size_t)FST_NMAX * nimg);
        o.frame_count = m->fstFrameCount.as<unsigned int>();
        o.frame_list = m->fstFrameList.as<uint32_t>();
        o.frame_cap = (unsigned int)FST_NMAX;
        o.frame_n = (unsigned int)nimg;
        o.frame_per_image = (unsigned int)(wt.total / nimg);
        o.frame_magic = o.frame_per_image ? 0xffffffffu / o.frame_per_image : 0u;
    }
    m->specRun = specWanted && !m->tailRun && zc && m->wvbOk;
    if (m->specRun) {   // no clearing: every run that ends in stage B writes the word before anything behind it on the stream reads it
        m->specCnt.reserve(16);
        o.tail_count = m->specCnt.as<unsigned int>();
    }
    o.pos_patches = m->pos_patches.as<uint8_t>();
    o.pos_count = m->pos.as<unsigned int>();        // header word 0
    o.pos_cap = (unsigned int)m->pos_cap;
    o.deep_q = m->deep_q.as<int64_t>();
    o.deep_count = m->pos.as<unsigned int>() + 1;   // header word 1
    o.host_count = zc ? m->h_pos.as<unsigned int>() : nullptr;
    o.done_blocks = m->pos.as<unsigned int>() + 3;  // header word 3
    if (zc) *m->h_pos.as<unsigned int>() = 0xffffffffu;   // overwritten by the last stage-B workgroup
    if (time_kernel) HIP_CHECK(hipEventRecord(ctx->ev0, st));
    return true;
}
// second half: the exact cascade kernels on `wtq`, read-back, completion event
static void wvm_launch_tail(fd_ctx* ctx, hipStream_t st, fd_pyramid* p, fd_wvm* m, const WinTable& wt, const WinTable& wtq, const WvmLaunch& L, bool skipA,
                            bool time_kernel) {
    launch_cascade<false>(ctx, st, wt.total, m, p->arena.as<uint8_t>(), wtq, L.o, skipA);
    if (m->sbRun) {
        if (!m->relaunch) m->relaunch = std::shared_ptr<void>(new WvbRelaunch(), [](void* q) { delete static_cast<WvbRelaunch*>(q); });
        WvbRelaunch* R = static_cast<WvbRelaunch*>(m->relaunch.get());
        R->arena = p->arena.as<uint8_t>(); R->wt = wtq; R->o = L.o; R->st = st;
    }
    if (time_kernel) HIP_CHECK(hipEventRecord(ctx->ev1, st));
    HIP_CHECK(hipGetLastError());
    if (!L.zc) {
        const size_t firstChunk = (size_t)std::min<int64_t>(m->pos_cap, WVM_FIRST_CHUNK);
        HIP_CHECK(hipMemcpyAsync(m->h_pos.p, m->pos.p, sizeof(PosRec) * (firstChunk + 1), hipMemcpyDeviceToHost, st));
    }
    HIP_CHECK(hipEventRecord(m->done, st));
}

static void wvm_launch_table(fd_ctx* ctx, hipStream_t st, fd_pyramid* p, fd_wvm* m, const WinTable& wt, bool want_all, WvmRun& run, bool time_kernel) {
    WvmLaunch L;
    if (!wvm_launch_head(ctx, st, m, wt, want_all, run, time_kernel, L)) return;
    const CascadeOut& o = L.o;
    WinTable wtq = wt;
    bool skipA = false;
    // production path: the dense pre-filter drops every window the first cascade levels reject with a margin; the exact
    // cascade then only sees the queue (header word 2 = its length).  Per-window outputs need the exact path for all.
    // The pre-filter feeds stage B's queue directly (stage B evaluates a window from level 0 anyway).
    if (!want_all && m->denseL) {
        skipA = launch_prefilter(ctx, st, m, p->arena.as<uint8_t>(), wt, o.deep_q, o.deep_count);
        if (skipA && time_kernel && ctx->kernel_timing_mode == 2) {   // bench hook: the pre-filter alone
            HIP_CHECK(hipEventRecord(ctx->ev1, st));
            time_kernel = false;
        }
    }
    wvm_launch_tail(ctx, st, p, m, wt, wtq, L, skipA, time_kernel);
}

// A detector of a batch that may share its pre-filter with others on the same pyramid (five_stage_batch_begin groups them).
// FD_WVM_GROUP=0: never (read per call: the tests compare both ways).
bool fd_wvm_groupable(const fd_wvm* m) {
    if (const char* e = getenv("FD_WVM_GROUP")) if (atoi(e) == 0) return false;
    return m && m->denseL != 0 && m->wvbOk && m->dev.numUsed > WVM_LCAP && wvd_group_size(m->dev.fw, m->dev.fh);
}

// The cascades of n detectors with the same patch size on the same pyramid and window stepping (no ROI), each on its own stream:
// every member's buffers and header as in fd_wvm_launch_on, ONE k_wvm_prefilter_group on the first member's stream that fills all
// members' stage-B queues, then every member's stage B, read-back and completion event on its own stream behind the group's event.
void fd_wvm_launch_group(fd_ctx* ctx, const hipStream_t* sts, fd_pyramid* p, fd_wvm* const* ms, int n, int sx, int sy, WvmRun* const* runs) {
    if (n < 1 || n > WVD_GMAX) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_wvm_launch_group: %d members", n);
    for (int i = 0; i < n; ++i)
        if (p->ctx != ctx || ms[i]->ctx != ctx) FD_THROW(FD_ERR_INVALID_ARGUMENT, "objects belong to different contexts");
    if (p->filter_kind != FD_LAYER_NONE) FD_THROW(FD_ERR_INVALID_ARGUMENT, "WVM detection needs a gray pyramid (no layer filter)");
    if (p->all.empty()) FD_THROW(FD_ERR_RUNTIME, "pyramid has not been updated with an image");
    HIP_CHECK(hipSetDevice(ctx->device));
    fd_pyramid_wait(p, sts[0]);
    WinTable wt;
    fd_wvm_build_table(p, ms[0]->dev.fw, ms[0]->dev.fh, sx, sy, nullptr, wt, runs[0]->wls);
    for (int i = 1; i < n; ++i) runs[i]->wls = runs[0]->wls;
    WvmLaunch L[WVD_GMAX];
    CascadeOut os[WVD_GMAX];
    bool any = false;
    for (int i = 0; i < n; ++i) {
        const bool ok = wvm_launch_head(ctx, sts[i], ms[i], wt, false, *runs[i], false, L[i]);
        any = any || ok;
        os[i] = L[i].o;
        // A member's header / counter clears are queued on ITS stream: the shared kernel waits for them.  (Only then: a steady-state
        // zero-copy run queues nothing in front of its kernels, and an unconditional join made every frame's group kernel wait for all
        // pool streams to drain the previous frame -- and all of them for it.)
        if (ok && L[i].queued && i > 0 && sts[i] != sts[0]) {
            if (!ms[i]->prep) HIP_CHECK(hipEventCreateWithFlags(&ms[i]->prep, hipEventDisableTiming));
            HIP_CHECK(hipEventRecord(ms[i]->prep, sts[i]));
            HIP_CHECK(hipStreamWaitEvent(sts[0], ms[i]->prep, 0));
        }
    }
    if (!any) return;   // no windows at all (every member sees the same table)
    const uint8_t* arena = p->arena.as<uint8_t>();
    const bool timeGroup = ctx->kernel_timing && ctx->kernel_timing_mode == 2;   // bench hook: one group launched alone (fd_last_group_prefilter_ms)
    if (timeGroup) {
        for (int e = 0; e < 2; ++e)
            if (!ctx->evg[e]) HIP_CHECK(hipEventCreate(&ctx->evg[e]));
        HIP_CHECK(hipEventRecord(ctx->evg[0], sts[0]));
    }
    const bool grouped = launch_prefilter_group(ctx, sts[0], ms, n, arena, wt, os);
    if (timeGroup) {
        HIP_CHECK(hipEventRecord(ctx->evg[1], sts[0]));
        ctx->evgMembers = grouped ? n : 0;
    }
    if (grouped) {
        if (!ms[0]->grp) HIP_CHECK(hipEventCreateWithFlags(&ms[0]->grp, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(ms[0]->grp, sts[0]));
    }
    for (int i = 0; i < n; ++i) {
        bool skipA = grouped;
        if (grouped) {
            if (sts[i] != sts[0]) HIP_CHECK(hipStreamWaitEvent(sts[i], ms[0]->grp, 0));
        } else {
            if (i > 0 && sts[i] != sts[0]) fd_pyramid_wait(p, sts[i]);
            skipA = ms[i]->denseL ? launch_prefilter(ctx, sts[i], ms[i], arena, wt, os[i].deep_q, os[i].deep_count) : false;
        }
        wvm_launch_tail(ctx, sts[i], p, ms[i], wt, wt, L[i], skipA, false);
    }
}

// bench hook: duration of the timed kernel(s) of a finished run (fd_hip_bench.h)
static void wvm_read_timing(fd_ctx* ctx) {
    if (ctx->kernel_timing_mode == 3 && ctx->evxN > 0) {
        float sum = 0.f;
        for (int ph = 0; ph < ctx->evxN; ++ph) {
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, ctx->evx[2 * ph], ctx->evx[2 * ph + 1]));
            sum += ms;
        }
        ctx->last_kernel_ms = sum;
        ctx->last_kernel = "k_wvb_chain2";
        return;
    }
    HIP_CHECK(hipEventElapsedTime(&ctx->last_kernel_ms, ctx->ev0, ctx->ev1));
    ctx->last_kernel = "k_wvm_cascade";
}

// what a finished run's header tells the handle about the next one (idempotent)
static void wvm_finish_header(fd_wvm* m, const PosRec* hraw) {
    if (m->zcRun) m->hdrClean = true;   // the last stage-B workgroup has cleared the device header
    if (m->sbRun) {   // header word 1: windows queued for stage B; zero-copy runs also deliver the counts of phases 1 and 2
        m->sbDeep = (int64_t)hraw[0].wid_hi;
        if (m->zcRun) {
            uint32_t c2;
            std::memcpy(&c2, &hraw[0].fout, 4);
            const int64_t c[3] = {m->sbDeep, (int64_t)(uint32_t)hraw[0].level, (int64_t)c2};
            for (int j = 1; j <= m->sbPlanN && j <= 2; ++j) {
                const int cut = m->sbPlanCut[j - 1];
                m->sbCutAlive[cut] = c[j];
                if (c[j - 1] > 0 && c[j] * 10 >= c[j - 1] * 7) m->sbCutMask &= ~(1u << cut);
                else m->sbCutMask |= 1u << cut;
            }
        }
    }
}

// Synchronous half: waits for m->done, fetches the remaining positives and sorts them into extraction order.
void fd_wvm_finish(fd_ctx* ctx, fd_wvm* m, WvmRun& run) {
    if (run.total == 0) return;
    HIP_CHECK(hipEventSynchronize(m->done));
    PosRec* hraw = m->h_pos.as<PosRec>();
    const unsigned int cnt = hraw[0].wid_lo;
    if (run.timed) wvm_read_timing(ctx);
    if (m->zcRun && cnt == 0xffffffffu) FD_THROW(FD_ERR_HIP, "WVM stage B did not deliver its positive count");
    if (m->sbRun && (int64_t)hraw[0].wid_hi > m->deepCap && !wvb_cap_env() && m->relaunch) {
        // More windows reached stage B than its state holds (header word 1 = queue length; stage B took the first deepCap of them).
        // The queue is intact: grow the state and run stage B again over all of it (the pyramid must not have been updated in between,
        // which the entry points guarantee).  One extra run per growth; the handle keeps the larger state.
        const int64_t deep = (int64_t)hraw[0].wid_hi;
        WvbRelaunch* R = static_cast<WvbRelaunch*>(m->relaunch.get());
        m->sbGrown = deep + deep / 8 + 64;
        m->specRun = false;   // the scores queued behind the first run belong to its positives, not to the rerun's
        wvb_reserve(m, run.total, m->sbGrown);
        if (deep <= m->deepCap) {
            const uint32_t hdr[4] = {0u, (uint32_t)deep, 0u, 0u};   // positives 0, queue length, pre-filter queue, retired workgroups
            HIP_CHECK(hipMemcpyAsync(m->pos.p, hdr, sizeof(hdr), hipMemcpyHostToDevice, R->st));
            HIP_CHECK(hipMemsetAsync(m->sbCnt.p, 0, 64, R->st));
            if (m->zcRun) *m->h_pos.as<unsigned int>() = 0xffffffffu;
            m->sbDeep = deep;
            CascadeOut o2 = R->o;   // no k_fs_oe follows this run: it must not file positives under frames (the host does stages 2-5)
            o2.frame_list = nullptr; o2.frame_count = nullptr; o2.tail_count = nullptr;
            launch_cascade<false>(ctx, R->st, run.total, m, R->arena, R->wt, o2, true);   // stage B only
            HIP_CHECK(hipGetLastError());
            if (!m->zcRun) {
                const size_t firstChunk = (size_t)std::min<int64_t>(m->pos_cap, WVM_FIRST_CHUNK);
                HIP_CHECK(hipMemcpyAsync(m->h_pos.p, m->pos.p, sizeof(PosRec) * (firstChunk + 1), hipMemcpyDeviceToHost, R->st));
            }
            HIP_CHECK(hipEventRecord(m->done, R->st));
            run.timed = false;
            fd_wvm_finish(ctx, m, run);
            return;
        }
    }
    m->prevPos = (int64_t)cnt;
    if ((int64_t)cnt > m->pos_cap)
        FD_THROW(FD_ERR_DEVICE_CAPACITY, "WVM produced %u positives, device buffer holds %lld (set FD_WVM_POS_CAP)", cnt, (long long)m->pos_cap);
    wvm_finish_header(m, hraw);
    if (m->sbRun && (int64_t)hraw[0].wid_hi > m->deepCap)   // header word 1: windows queued for stage B
        FD_THROW(FD_ERR_DEVICE_CAPACITY, "WVM stage B: %u windows queued, its state holds %lld (set FD_WVM_DEEP_CAP)", hraw[0].wid_hi, (long long)m->deepCap);
    if (cnt) {
        const size_t firstChunk = m->tailRun ? (size_t)0 : (m->zcRun ? (size_t)cnt : (size_t)std::min<int64_t>(m->pos_cap, WVM_FIRST_CHUNK));
        if (cnt > firstChunk) {   // on the auxiliary stream: the main stream may already hold the next detectors' kernels
            hipStream_t ax = fd_aux_stream(ctx);
            HIP_CHECK(hipMemcpyAsync(hraw + 1 + firstChunk, m->pos.as<PosRec>() + 1 + firstChunk, sizeof(PosRec) * (cnt - firstChunk),
                                     hipMemcpyDeviceToHost, ax));
            HIP_CHECK(hipStreamSynchronize(ax));
        }
        const PosRec* raw = hraw + 1;
        // extraction order = ascending window id (ids are unique): sort compact (id, slot) keys, not the records
        std::vector<std::pair<uint64_t, uint32_t>> order(cnt);
        uint64_t widMax = 0;
        for (uint32_t i = 0; i < cnt; ++i) {
            order[i] = {((uint64_t)raw[i].wid_hi << 32) | raw[i].wid_lo, i};
            widMax = std::max(widMax, order[i].first);
        }
        if (cnt >= 4096 && widMax < ((uint64_t)1 << 33)) {
            // busy frames (config 3: up to 70 K positives per detector, 200 K per frame) spent 5 of a frame's 24 ms of host work in
            // std::sort here: the keys are unique, so an LSD radix sort (11 bits per pass) gives the same order at ~1 ns per key and pass
            std::vector<std::pair<uint64_t, uint32_t>> tmp(cnt);
            int bits = 0;
            while (((uint64_t)1 << bits) <= widMax) ++bits;
            std::pair<uint64_t, uint32_t>* src = order.data();
            std::pair<uint64_t, uint32_t>* dst = tmp.data();
            for (int sh = 0; sh < bits; sh += 11) {
                uint32_t hist[2049] = {};
                for (uint32_t i = 0; i < cnt; ++i) ++hist[((src[i].first >> sh) & 2047u) + 1];
                for (int b = 0; b < 2048; ++b) hist[b + 1] += hist[b];
                for (uint32_t i = 0; i < cnt; ++i) dst[hist[(src[i].first >> sh) & 2047u]++] = src[i];
                std::swap(src, dst);
            }
            if (src != order.data()) order.swap(tmp);
        } else {
            std::sort(order.begin(), order.end());
        }
        run.pos.resize(cnt);
        run.slots.resize(cnt);
        for (uint32_t i = 0; i < cnt; ++i) { run.pos[i] = raw[order[i].second]; run.slots[i] = order[i].second; }
    }
}

void fd_wvm_run(fd_ctx* ctx, fd_pyramid* p, fd_wvm* m, int sx, int sy, const int* roi, bool want_all, WvmRun& run,
                bool time_kernel) {
    fd_wvm_launch(ctx, p, m, sx, sy, roi, want_all, run, time_kernel);
    fd_wvm_finish(ctx, m, run);
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
        if (p->nimg > 1) wid %= run.total / p->nimg;   // multi-frame pyramid: id inside its frame (the caller groups by frame)
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
        // the offset tables index the caller's val / rects arrays: they must start at 0 and ascend, and stay inside the arrays
        // when the caller states their lengths (num_vals / num_rects; 0 = not stated)
        if (md->val_off[0] != 0) FD_THROW(FD_ERR_INVALID_ARGUMENT, "WvmClassifier: val_off[0] must be 0");
        for (int k = 0; k < F; ++k)
            if (md->val_off[k + 1] <= md->val_off[k]) FD_THROW(FD_ERR_INVALID_ARGUMENT, "WvmClassifier: val_off is not ascending at filter %d", k);
        const int nval = md->val_off[F];
        if (md->num_vals > 0 && nval > md->num_vals) FD_THROW(FD_ERR_INVALID_ARGUMENT, "WvmClassifier: val_off[num_filters] = %d exceeds num_vals = %d", nval, md->num_vals);
        if (md->rec_off[0] < 0) FD_THROW(FD_ERR_INVALID_ARGUMENT, "WvmClassifier: rec_off[0] must not be negative");
        for (int v = 0; v < nval; ++v)
            if (md->rec_off[v + 1] < md->rec_off[v]) FD_THROW(FD_ERR_INVALID_ARGUMENT, "WvmClassifier: rec_off is not ascending at grey value %d", v);
        if (md->num_rects > 0 && md->rec_off[nval] > md->num_rects)
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "WvmClassifier: rec_off[%d] = %d exceeds num_rects = %d", nval, md->rec_off[nval], md->num_rects);
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
        wvm_build_dense(m, md);
        wvb_build(m, md);
        *out = guard.release();
    });
}

void fd_wvm_destroy(fd_wvm* m) { delete m; }

int fd_detect_wvm(fd_ctx* ctx, fd_pyramid* p, const fd_wvm* wvm_, int sx, int sy, const int* roi, fd_detection* out,
                  int64_t cap, int64_t* count, int32_t* all_level, float* all_score) {
    return fd_guard(ctx, [&] {
        if (!ctx || !p || !wvm_ || !count) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_detect_wvm: NULL argument");
        fd_pyramid_require_single(p, "fd_detect_wvm");
        fd_wvm* m = const_cast<fd_wvm*>(wvm_);
        WvmRun run;
        const bool want_all = all_level || all_score;
        fd_wvm_run(ctx, p, m, sx, sy, roi, want_all, run, ctx->kernel_timing);
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
        m->deep_q.reserve(sizeof(int64_t) * (size_t)n);
        HIP_CHECK(hipMemcpyAsync(in.p, patches, bytes, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemsetAsync(m->counter.p, 0, 16, st));
        m->sbRun = m->wvbOk && m->dev.numUsed > WVM_LCAP;
        if (m->sbRun) {
            wvb_reserve(m, n, n);   // explicit patches: every one of them may reach stage B
            HIP_CHECK(hipMemsetAsync(m->sbCnt.p, 0, 64, st));
            m->hdrClean = false;   // the phase counters are left dirty: the next detect run clears them (and the header) first
        }
        WinTable wt;
        std::memset(&wt, 0, sizeof(wt));
        wt.raw = 1;
        wt.total = n;
        CascadeOut o;
        o.all_level = m->all_level.as<int32_t>();
        o.all_fout = m->all_fout.as<float>();
        o.pos = m->pos.as<PosRec>();
        o.pos_patches = m->pos_patches.as<uint8_t>();
        o.pos_count = m->counter.as<unsigned int>();
        o.pos_cap = 0u;   // positives are not collected here
        o.deep_q = m->deep_q.as<int64_t>();
        o.deep_count = m->counter.as<unsigned int>() + 1;
        launch_cascade<true>(ctx, st, n, m, in.as<uint8_t>(), wt, o);
        HIP_CHECK(hipGetLastError());
        if (out_level) HIP_CHECK(hipMemcpyAsync(out_level, m->all_level.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
        if (out_score) HIP_CHECK(hipMemcpyAsync(out_score, m->all_fout.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
    });
}

#include "five_stage.hpp"

// condensation::WvmSvmModel::evaluate(image, samples) (WvmSvmModel.cpp:69-118) on top of a DirectPyramidFeatureExtractor
// + HistEq64Filter: every sample {x, y, width, height} maps to one pyramid window (DirectPyramidFeatureExtractor::extract
// (x, y, width, height), :67-73,134-153); all windows run through the WVM cascade in one launch (explicit window list);
// the (at most 8) most probable WVM positives are re-scored by the SVM.  The reference's patch cache is keyed by
// shared_ptr identity and never hits (oracle/orc_detect.cpp), so every sample is scored on its own.
int fd_wvm_svm_evaluate_samples(fd_ctx* ctx, fd_pyramid* p, const fd_wvm* wvm_, const fd_svm* svm, int n, const int32_t* xywh,
                                uint8_t* target, double* weight) {
    return fd_guard(ctx, [&] {
        if (!ctx || !p || !wvm_ || !svm || n < 0 || (n > 0 && (!xywh || !target || !weight)))
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_wvm_svm_evaluate_samples: bad argument");
        fd_pyramid_require_single(p, "fd_wvm_svm_evaluate_samples");
        fd_wvm* m = const_cast<fd_wvm*>(wvm_);
        five_stage_check(m, svm);
        if (p->ctx != ctx || m->ctx != ctx) FD_THROW(FD_ERR_INVALID_ARGUMENT, "objects belong to different contexts");
        if (p->filter_kind != FD_LAYER_NONE) FD_THROW(FD_ERR_INVALID_ARGUMENT, "WVM evaluation needs a gray pyramid (no layer filter)");
        if (p->all.empty()) FD_THROW(FD_ERR_RUNTIME, "pyramid has not been updated with an image");
        if (p->kept.size() > (size_t)WVM_MAX_LAYERS) FD_THROW(FD_ERR_INVALID_ARGUMENT, "pyramid has %zu layers, this backend supports %d", p->kept.size(), WVM_MAX_LAYERS);
        HIP_CHECK(hipSetDevice(ctx->device));
        const int pw = m->dev.fw, ph = m->dev.fh;
        // sample -> window (ImagePyramid::getLayer(double) :307-310, getLayer(int) by index, extract bounds check)
        std::vector<int32_t> list;
        std::vector<int> sampleOf;
        const int firstIndex = p->kept.empty() ? 0 : p->all[p->kept[0]].index;
        for (int i = 0; i < n; ++i) {
            target[i] = 0;
            weight[i] = 0;
            const int x = xywh[4 * i], y = xywh[4 * i + 1], width = xywh[4 * i + 2], height = xywh[4 * i + 3];
            if (width <= 0 || p->kept.empty()) continue;
            const double power = std::log((double)pw / (double)width) / std::log(p->inc);
            const long index = std::lround(power);   // std::round, then the int cast of the reference
            const long realIndex = index - firstIndex;
            if (realIndex < 0 || realIndex >= (long)p->kept.size()) continue;
            const HostLayer& L = p->all[p->kept[realIndex]];
            const int bx = fd_cvRound((x - width / 2) * L.scale), by = fd_cvRound((y - height / 2) * L.scale);
            if (bx < 0 || by < 0 || bx + pw > L.w || by + ph > L.h) continue;
            list.push_back((int32_t)realIndex); list.push_back(bx); list.push_back(by);
            sampleOf.push_back(i);
        }
        const int64_t nv = (int64_t)sampleOf.size();
        if (nv == 0) return;
        hipStream_t st = ctx->stream;
        m->list.reserve(sizeof(int32_t) * list.size());
        int32_t* pin = (int32_t*)fd_pinned(ctx, sizeof(int32_t) * list.size());
        std::memcpy(pin, list.data(), sizeof(int32_t) * list.size());
        HIP_CHECK(hipMemcpyAsync(m->list.p, pin, sizeof(int32_t) * list.size(), hipMemcpyHostToDevice, st));
        WinTable wt;
        std::memset(&wt, 0, sizeof(wt));
        wt.n = (int32_t)p->kept.size();
        wt.sx = wt.sy = 1;
        wt.total = nv;
        wt.list = m->list.as<int32_t>();
        for (size_t i = 0; i < p->kept.size(); ++i) {
            const HostLayer& L = p->all[p->kept[i]];
            wt.l[i].lw = L.w; wt.l[i].off = L.gray_off; wt.l[i].nx = 1; wt.l[i].ny = 1; wt.l[i].magic = 0xffffffffu; wt.l[i].first = INT64_MAX;
        }
        WvmRun run;
        wvm_launch_table(ctx, ctx->stream, p, m, wt, true, run, false);
        fd_wvm_finish(ctx, m, run);
        std::vector<float> fout((size_t)nv);
        HIP_CHECK(hipMemcpy(fout.data(), m->all_fout.p, sizeof(float) * (size_t)nv, hipMemcpyDeviceToHost));
        std::vector<double> prob((size_t)nv);
        for (int64_t k = 0; k < nv; ++k) {
            prob[k] = wvm_probability(m, (double)fout[k]);
            weight[sampleOf[k]] = 0.5 * prob[k];
        }
        struct Remaining { int64_t item; double prob; uint32_t slot; };
        std::vector<Remaining> remaining;
        for (size_t i = 0; i < run.pos.size(); ++i) {   // WVM positives in sample order
            const int64_t item = (int64_t)(((uint64_t)run.pos[i].wid_hi << 32) | run.pos[i].wid_lo);
            remaining.push_back(Remaining{item, prob[item], run.slots[i]});
        }
        if (remaining.empty()) return;
        if (remaining.size() > 8) {   // sort(indirect, greater<ClassifiedPatch>()) + resize(8), WvmSvmModel.cpp:100-103
            std::sort(remaining.begin(), remaining.end(), [](const Remaining& a, const Remaining& b) { return a.prob > b.prob; });
            remaining.resize(8);
        }
        uint32_t slots[8];
        double dist[8];
        for (size_t i = 0; i < remaining.size(); ++i) slots[i] = remaining[i].slot;
        DevBuf& idx = m->all_level;   // scratch
        idx.reserve(64);
        m->all_fout.reserve(sizeof(double) * 8 + sizeof(float) * (size_t)nv);
        HIP_CHECK(hipMemcpy(idx.p, slots, sizeof(uint32_t) * remaining.size(), hipMemcpyHostToDevice));
        fd_svm_generic_launch(ctx, svm, m->pos_patches.p, idx.as<uint32_t>(), (int64_t)m->dev.d, (int64_t)remaining.size(), m->all_fout.as<double>());
        HIP_CHECK(hipStreamSynchronize(st));
        HIP_CHECK(hipMemcpy(dist, m->all_fout.p, sizeof(double) * remaining.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < remaining.size(); ++i) {
            const int sidx = sampleOf[remaining[i].item];
            target[sidx] = dist[i] >= (double)fd_svm_threshold(svm) ? 1 : 0;
            weight[sidx] = 2 * weight[sidx] * fd_svm_probability(svm, dist[i]);
        }
    });
}

#ifdef FD_WVB_PROF
// dev tool (tools/wvb_phases.py, -DFD_WVB_PROF builds only): the accumulated timestamps of the stage-B kernels
void fd_debug_wvb_prof(unsigned long long* out, int reset) {
    if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(fd_wvb_prof), sizeof(unsigned long long) * 64);
    if (reset) { unsigned long long z[64] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(fd_wvb_prof), z, sizeof(z)); }
}
// tools/wvd_residency.py: the per-wavefront records of the LAST k_wvm_prefilter launch (see wvm_dense.hpp); returns the record capacity
int fd_debug_wvd_prof(unsigned long long* out, int nwaves) {
    if (nwaves > WVD_PROF_WAVES) nwaves = WVD_PROF_WAVES;
    if (out && nwaves > 0) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(fd_wvd_prof), sizeof(unsigned long long) * 8 * (size_t)nwaves);
    return WVD_PROF_WAVES;
}
#endif

// Measurement hook (include/fd_hip_bench.h): windows the last finished run of this handle queued for stage B (-1: none yet)
int fd_last_group_prefilter_ms(fd_ctx* ctx, float* ms, int* members) {
    return fd_guard(ctx, [&] {
        if (!ctx || !ms) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_last_group_prefilter_ms: NULL argument");
        *ms = 0.f;
        if (members) *members = ctx->evgMembers;
        if (ctx->evgMembers <= 0) return;
        HIP_CHECK(hipEventSynchronize(ctx->evg[1]));
        HIP_CHECK(hipEventElapsedTime(ms, ctx->evg[0], ctx->evg[1]));
    });
}
int64_t fd_wvm_last_queue_length(const fd_wvm* m) { return m ? m->sbDeep : -1; }
int fd_wvm_last_tail_state(const fd_wvm* m) { return m ? m->fstLastState : -1; }
int fd_wvm_last_spec_state(const fd_wvm* m) { return m ? m->specLastState : -1; }
int fd_wvm_last_stage_b_plan(const fd_wvm* m, int64_t* out) {
    if (!m || !out) return FD_ERR_INVALID_ARGUMENT;
    const int n = std::min(m->sbLastN, WVB_MAXPHASE);   // out: 1 + 3 * WVB_MAXPHASE values (the hook reported 3 of up to 4 phases)
    out[0] = n;
    for (int ph = 0; ph < n; ++ph) {
        out[1 + 3 * ph] = m->sbLastGen[ph];
        out[2 + 3 * ph] = m->sbLastGen[ph + 1];
        out[3 + 3 * ph] = ph == 0 ? m->sbDeep : (ph - 1 < m->sbPlanN ? m->sbCutAlive[m->sbPlanCut[ph - 1]] : -1);
    }
    return FD_OK;
}
#ifdef FD_FST_PROF
int fd_debug_fst_prof(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(fd_fst_prof), 64) == hipSuccess ? 0 : -1; }
#endif

// Test hook (include/fd_hip_bench.h; needs no GPU): the pre-filter's plan -- K and the tile list -- for a set of layers
int fd_debug_wvd_plan(const int32_t* nx, const int32_t* ny, int n_layers, int frames, int sy, int ph, int slots, int32_t* tile_first) {
    if (!nx || !ny || n_layers < 1 || n_layers > WVM_MAX_LAYERS || frames < 1 || sy < 1 || ph < 1 || slots < 1) return FD_ERR_INVALID_ARGUMENT;
    WvdTable t;
    std::memset(&t, 0, sizeof(t));
    t.n = n_layers; t.sx = 1; t.sy = sy; t.nimg = frames;
    for (int i = 0; i < n_layers; ++i) {
        if (nx[i] < 1 || ny[i] < 1) return FD_ERR_INVALID_ARGUMENT;
        t.l[i].nx = nx[i]; t.l[i].ny = ny[i];
    }
    const int K = wvd_choose_k(t, slots, ph);
    const int tiles = wvd_plan_sliding(t, K);
    if (tile_first) {
        for (int i = 0; i < n_layers; ++i) tile_first[i] = t.l[i].sTileFirst;
        tile_first[n_layers] = tiles;
    }
    return K;
}

// Test hook (include/fd_hip_bench.h; needs no GPU): the rect sums of every used level of `md` for n equalised patches, computed from
// the stage-B tables with the operand addressing of k_wvb_chain (A fragment of lane h * 32 + row, byte t <-> pixel ks * 32 + h * 16 + t).
// out[i * ncols + c]: c runs over the levels 0 .. numUsed - 1 in order, grey values 1 .. cntval - 1 inside a level.  Returns the number
// of columns, or -1 when the model has no dense stage B.
int64_t fd_debug_wvb_rect_sums(const fd_wvm_model* md, const uint8_t* patches, int64_t n, int32_t* out, int32_t* phase_gen /* [5] or NULL */) {
    if (!md || md->num_filters < 1) return -1;
    const int NU = (md->num_used > md->num_filters || md->num_used <= 0) ? md->num_filters : md->num_used;
    WvbTables T;
    try {
        if (!wvb_tables(md, NU, T)) return -1;
    } catch (...) { return -1; }
    const int d = md->filter_w * md->filter_h;
    int64_t ncols = 0;
    for (int k = 0; k < NU; ++k) ncols += T.lvl[4 * (size_t)k + 2] - 1;
    if (phase_gen) for (int i = 0; i <= WVB_MAXPHASE; ++i) phase_gen[i] = i <= T.nphase ? T.phaseGen[i] : -1;
    if (!out || !patches) return ncols;
    std::vector<int8_t> x((size_t)T.DS);
    for (int64_t i = 0; i < n; ++i) {
        std::fill(x.begin(), x.end(), 0);
        for (int j = 0; j < d; ++j) x[j] = (int8_t)(patches[(size_t)i * d + j] ^ 0x80u);
        int64_t c = 0;
        for (int k = 0; k < NU; ++k) {
            const int tile = T.lvl[4 * (size_t)k], row0 = T.lvl[4 * (size_t)k + 1], cnt = T.lvl[4 * (size_t)k + 2];
            for (int v = 1; v < cnt; ++v, ++c) {
                const int row = row0 + v - 1;
                int32_t acc = 0;
                for (int ks = 0; ks < T.KS; ++ks)
                    for (int h = 0; h < 2; ++h)
                        for (int t = 0; t < 16; ++t)
                            acc += (int32_t)T.A[((((size_t)tile * T.KSP + ks) * 64) + (size_t)(h * 32 + row)) * 16 + t] * (int32_t)x[(size_t)ks * 32 + h * 16 + t];
                out[(size_t)i * ncols + c] = acc + T.c128[(size_t)tile * 32 + row];
            }
        }
    }
    return ncols;
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
