// featuredetection_amd/csrc/wvm_dense.hpp -- dense pre-filter of the WVM cascade on the matrix pipe (included by wvm.hip only).
//
// WvmClassifier.cpp:191-346 evaluates filter k through rectangle sums on an integral image; in exact arithmetic its dot product
// is x . r_k with the dense residual image r_k[y][x] = val_k[0] + sum_{v>=1} (val_k[v] - val_k[0]) * #{rects of v covering (x, y)}
// (SURVEY.md App. A.3), and for k < numFiltersPerLevel nothing is carried over from earlier levels (u_kernel_eval[n] is still 0).
// So the first L <= min(16, numPer) kernel values of ALL windows are one [windows x d] . [d x L] contraction:
//
//   k_wvm_prefilter: lane == window; a lane walks down its column of a layer through K windows (64 columns / row groups per wavefront).
//     1. HistEq64 (HistEq64Filter.cpp:32-125) lane-serial: private 64-bin histogram column in LDS (u16 counters, lanes l and l + 32
//        share a dword: ds_add_u32 of 1 << 16*(lane>>5), conflict-free; one v_perm_b32 per pixel address) that survives the window:
//        the next window down takes the rows that left out of it and adds the rows that entered.  The fp32 cdf is a plain 63-step
//        chain in the lane's registers (same operation order as the reference, 64 windows per instruction instead of the DPP chain's
//        one), the LUT goes to its own LDS block.  sum(x) and sum(x^2) of the equalised patch come from the histogram (exact integers);
//        the reference's fp32 sum of squares (IImg.cpp:33-47) equals the integer below 2^24 and is within 2*ph + 2 of it above.
//     2. the equalised pixels (x - 128 as int8, gathered through the LUT one patch row -- two for 16-wide patches -- per k-step) are
//        the B operand of v_mfma_i32_32x32x32_i8, built in registers with v_permlane32_swap; the A operand is the residual images
//        quantised to 32-bit integers Q = round(r * 2^s) and split into four balanced base-256 digits: integer arithmetic, so x . Q is
//        EXACT; |x . r - 2^-s x . Q| <= 2^-(s+1) * sum(x) is the only approximation.  The accumulators C[digit row][window] are folded
//        into exact doubles and brought back to lane == window with v_permlane32_swap.
//     3. per window: norm, K = exp(-basis * norm), res_k = -bias + sum_p w[k][p] K_p with a rigorous error bound eps_k on
//        |res_k - reference res_k| (quantisation, fast fp32 exp, fp32 summation order).  A window with res_k + eps_k < thr_k at ANY
//        level k < L is rejected by the reference at some level <= k, so it cannot be a WVM positive and is dropped here.
//        Everything else is appended to a queue and runs the exact stage B (wvm_stageb.hpp) from level 0, so positives, their levels
//        and their fp32 outputs stay bit-identical to the rectangle-sum formulation.
//   k_wvm_prefilter_multi: the first formulation (64 consecutive windows per wavefront, pixels staged and results transposed through
//     LDS) for several detectors on the same windows; off by default (FD_WVM_GROUP=1).
//   k_wvb_prepare_lanes: stage B's preparation of long queues with the same per-lane machinery.
//   Only used when no per-window outputs are requested (fd_detect_wvm with all_level / all_score takes the exact path for every window).
#pragma once

constexpr int WVD_L = 16;          // filters evaluated densely (16 filters x 4 digits = two tiles of 32 digit rows)

typedef int wvd_v4i __attribute__((ext_vector_type(4)));
typedef int wvd_v16i __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) unsigned int wvd_lds_u32;
typedef __attribute__((address_space(3))) unsigned short wvd_lds_u16;

struct WvdLayer {
    int32_t bx, by, nx, lw;
    uint32_t off, magic;
    int32_t nwin, tileFirst;             // k_wvm_prefilter_multi: tiles of 64 consecutive windows
    int64_t first;
    int32_t ny, G, sTileFirst, pad;      // k_wvm_prefilter: G = ceil(ny / K) row groups; tiles of 64 (column, row group) tasks
};
struct WvdTable {
    int32_t n, sx, sy, ntiles;           // ntiles = tilesPerImage * nimg
    int32_t nimg, tilesPerImage;         // multi-frame pyramid: tile -> (frame, tile inside the frame)
    int32_t K, sTilesPerImage;           // k_wvm_prefilter: windows a lane walks down (WVD_KMAX at most), its tiles per frame
    int64_t perImage;                    // windows per frame
    uint64_t imageStride;                // bytes between the frames' arenas
    WvdLayer l[WVM_MAX_LAYERS];
};

// per-level model constants of the dense stage (device memory)
struct WvdConst {
    double c128[WVD_L];        // 128 * sum_i Q_k[i]
    double pp[WVD_L];
    float thr[WVD_L];
    float w[WVD_L][WVD_L];     // hkWeights[k][p], p <= k
    float w2[WVD_L][WVD_L][2]; // {w, |w|}: the level sum and its error bound as one packed fma (k_wvm_prefilter)
};

struct WvdDev {
    const wvd_v4i* B;          // [k-step][2 N-tiles][64 lanes] 16 signed digit bytes each (k-step = patch row; two rows when pw == 16)
    const WvdConst* c;
    int64_t* q;                // windows that pass
    unsigned int* qcount;
    // scalars (kernel arguments, so that they live in SGPRs)
    int32_t L;
    float negBasis, negBias, stretch;
    float sxxSlack;            // 2 * ph + 2: |fp32 row-ordered sum of squares - exact integer| when the sum is >= 2^24
    double scale;              // 2^-s; also the error of norm per unit of sum(x): 2 * 2^-(s+1)
};

// several detectors on the same windows (same pyramid, patch size and steps: the lip / nose / eye-corner detectors of
// ffpDetectApp): k_wvm_prefilter_multi equalises a tile once and runs the contraction + levels of every detector on it
constexpr int WVD_MAXD = 8;
struct WvdMulti {
    int32_t nd;
    WvdDev d[WVD_MAXD];
};

namespace {

// -DFD_WVB_PROF (tools/build_prof_lib.sh, tools/wvd_residency.py): every wavefront of k_wvm_prefilter leaves {start, end, HW_ID | XCC_ID << 32,
// tiles, ticks in: histogram, cdf + LUT, equalise + MFMA, transposes + levels + queue} (s_memtime ticks of the shader clock)
#ifdef FD_WVB_PROF
constexpr int WVD_PROF_WAVES = 16384;
__device__ unsigned long long fd_wvd_prof[WVD_PROF_WAVES * 8];
#define WVD_T(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#else
#define WVD_T(x)
#endif

template <int PW_>
struct WvdGeo {
#ifndef FD_WVD_RPS16
#define FD_WVD_RPS16 2
#endif
    // patch rows per k-step (32 k-slots).  Two 16-wide rows fill a step; one row per step (FD_WVD_RPS16=1) needs fewer
    // registers (4 instead of 3 wavefronts per SIMD) but twice the steps -- measured equal (0.806 vs 0.794 ms, 16x24 ear
    // detector at 1080p), so the two-row form stays
    static constexpr int RPS = PW_ == 16 ? FD_WVD_RPS16 : 1;
};

// LDS of a workgroup (4 wavefronts).  Histogram rows are 256 B: [wave pair][bin][wave of the pair][lane & 31] dwords, the u16 counters
// of lanes l (low half) and l + 32 (high half) in one dword; LUT rows are 256 B too: [bin][lane][wave] bytes -- the byte of lane l sits
// in dword l of its bin row, so the 32 lanes of a ds_read_u8 lane group hit 32 different banks whatever their bins are (round 3's
// [bin][wave][lane] order put lanes 4j..4j+3 into one dword: a 4-way conflict whenever their bins differed, 63 % of the kernel's LDS
// cycles by SQ_LDS_BANK_CONFLICT).  With every 16 KB block
// 16 KB-aligned, the LDS address of (bin, lane) is {byte 3: 0, byte 2: lane word, byte 1: bin | block bits, byte 0: lane word}: ONE
// v_perm_b32 per pixel takes the bin byte out of a dword of four pre-shifted pixels and drops it into the lane's address word
// (wvd_bins / wvd_addr; 1.5 VALU per pixel instead of 2).  The histogram survives the window (the next one down slides it), so the
// LUT has a block of its own.
struct WvdLds {   // the variable is 16 KB-aligned (not the type: 48 KB must stay 48 KB, three workgroups per CU)
    unsigned int hist[2][64][2][32];                 // counters
    unsigned char lut[64][64][4];                    // the lanes' LUTs: e - 128 as int8, [bin][lane][wave] (see below)
};
static_assert(sizeof(unsigned int[64][2][32]) == 16384 && sizeof(unsigned char[64][64][4]) == 16384, "16 KB blocks, 256 B per bin");

typedef __attribute__((address_space(3))) unsigned char wvd_lds_u8;
typedef unsigned short wvd_u16x2 __attribute__((ext_vector_type(2)));
// the four bins (pixel >> 2) of a dword of pixels, each OR-ed with the block bits of the address byte
__device__ __forceinline__ unsigned int wvd_bins(unsigned int w4, unsigned int blk4) { return ((w4 >> 2) & 0x3F3F3F3Fu) | blk4; }
// LDS address of (bin of pixel B_, this lane): laneWord has byte 1 clear
template <int B_>
__device__ __forceinline__ unsigned int wvd_addr(unsigned int bins, unsigned int laneWord) {
    return __builtin_amdgcn_perm(bins, laneWord, 0x0c020000u | ((4u + B_) << 8));   // {0, laneWord.b2, bins.b[B_], laneWord.b0}
}
// Equalise N_ dwords of pixels (N_ = 3, 4, 5) through the lane's LUT: out[j] = the four LUT bytes of w[j].  Written out as one block so
// that all 4 N_ ds_read_u8 are in flight together (the compiler's schedule under this kernel's register pressure waited after every
// second read); the LDS answers in order, so the first dwords are packed while the last reads are still on their way; the address
// registers double as the read destinations.
// Per dword: shift, and-or, 4 address perms, 4 reads, 3 packing perms.
#define WVD_EQ_ISSUE(j)                                                                                   \
    "v_lshrrev_b32 %[t], 2, %[w" #j "]\n v_and_or_b32 %[t], %[t], %[mask], %[blk]\n"                     \
    "v_perm_b32 %[a" #j "], %[t], %[lane], %[s0]\n v_perm_b32 %[b" #j "], %[t], %[lane], %[s1]\n"          \
    "v_perm_b32 %[c" #j "], %[t], %[lane], %[s2]\n v_perm_b32 %[d" #j "], %[t], %[lane], %[s3]\n"          \
    "ds_read_u8 %[a" #j "], %[a" #j "]\n ds_read_u8 %[b" #j "], %[b" #j "]\n"                              \
    "ds_read_u8 %[c" #j "], %[c" #j "]\n ds_read_u8 %[d" #j "], %[d" #j "]\n"
#define WVD_EQ_PACK(j)                                                                                    \
    "v_perm_b32 %[a" #j "], %[b" #j "], %[a" #j "], %[sp]\n v_perm_b32 %[c" #j "], %[d" #j "], %[c" #j "], %[sp]\n" \
    "v_perm_b32 %[a" #j "], %[c" #j "], %[a" #j "], %[sq]\n"
#define WVD_EQ_OUT(j) [a##j] "=&v"(out[j]), [b##j] "=&v"(tb[j]), [c##j] "=&v"(tc[j]), [d##j] "=&v"(td[j])
#define WVD_EQ_IN [lane] "v"(laneWord), [blk] "v"(blk4), [mask] "s"(0x3F3F3F3Fu), [s0] "s"(0x0c020400u), [s1] "s"(0x0c020500u), \
                  [s2] "s"(0x0c020600u), [s3] "s"(0x0c020700u), [sp] "s"(0x0c0c0400u), [sq] "s"(0x05040100u)
template <int N_>
__device__ __forceinline__ void wvd_equalise(const unsigned int* w, unsigned int* out, unsigned int laneWord, unsigned int blk4) {
    static_assert(N_ >= 3 && N_ <= 5, "block sizes");
    unsigned int t, tb[N_], tc[N_], td[N_];
    if constexpr (N_ == 3)
        asm volatile(WVD_EQ_ISSUE(0) WVD_EQ_ISSUE(1) WVD_EQ_ISSUE(2) "s_waitcnt lgkmcnt(4)\n" WVD_EQ_PACK(0) WVD_EQ_PACK(1) "s_waitcnt lgkmcnt(0)\n" WVD_EQ_PACK(2)
                     : [t] "=&v"(t), WVD_EQ_OUT(0), WVD_EQ_OUT(1), WVD_EQ_OUT(2)
                     : [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), WVD_EQ_IN);
    if constexpr (N_ == 4)
        asm volatile(WVD_EQ_ISSUE(0) WVD_EQ_ISSUE(1) WVD_EQ_ISSUE(2) WVD_EQ_ISSUE(3) "s_waitcnt lgkmcnt(8)\n" WVD_EQ_PACK(0) WVD_EQ_PACK(1) "s_waitcnt lgkmcnt(0)\n" WVD_EQ_PACK(2) WVD_EQ_PACK(3)
                     : [t] "=&v"(t), WVD_EQ_OUT(0), WVD_EQ_OUT(1), WVD_EQ_OUT(2), WVD_EQ_OUT(3)
                     : [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), WVD_EQ_IN);
    if constexpr (N_ == 5)
        asm volatile(WVD_EQ_ISSUE(0) WVD_EQ_ISSUE(1) WVD_EQ_ISSUE(2) WVD_EQ_ISSUE(3) WVD_EQ_ISSUE(4) "s_waitcnt lgkmcnt(8)\n"
                     WVD_EQ_PACK(0) WVD_EQ_PACK(1) WVD_EQ_PACK(2) "s_waitcnt lgkmcnt(0)\n" WVD_EQ_PACK(3) WVD_EQ_PACK(4)
                     : [t] "=&v"(t), WVD_EQ_OUT(0), WVD_EQ_OUT(1), WVD_EQ_OUT(2), WVD_EQ_OUT(3), WVD_EQ_OUT(4)
                     : [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[4]), WVD_EQ_IN);
}

__device__ __forceinline__ unsigned int wvd_load_u32(const uint8_t* p) {
    unsigned int v;
    __builtin_memcpy(&v, p, 4);   // unaligned global_load_dword
    return v;
}
// LDS byte address of (bin of byte b of w4, this lane): two VALU instructions per pixel
template <int B_>
__device__ __forceinline__ unsigned int wvd_slot(unsigned int w4, unsigned int laneOff) {
    unsigned int bin, a;
    asm("v_bfe_u32 %0, %1, %2, 6" : "=v"(bin) : "v"(w4), "n"(8 * B_ + 2));
    asm("v_lshl_or_b32 %0, %1, 7, %2" : "=v"(a) : "v"(bin), "v"(laneOff));
    return a;
}
__device__ __forceinline__ void wvd_count(unsigned int ldsAddr, unsigned int inc) {   // ds_add_u32, no return value
    __hip_atomic_fetch_add((wvd_lds_u32*)(uintptr_t)ldsAddr, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ void wvd_uncount(unsigned int ldsAddr, unsigned int inc) {   // ds_sub_u32: the counter is >= 1, no borrow into the other half
    __hip_atomic_fetch_sub((wvd_lds_u32*)(uintptr_t)ldsAddr, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ unsigned int wvd_lut(unsigned int ldsAddr) { return *(wvd_lds_u16*)(uintptr_t)ldsAddr; }   // ds_read_u16
__device__ __forceinline__ unsigned int wvd_lshl_or(unsigned int a, int sh, unsigned int b) {
    unsigned int r;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(sh), "v"(b));
    return r;
}

// One wavefront = 64 (column, row group) tasks of a layer: lane == task, and a lane walks DOWN its column through up to K windows
// (wt.K, WVD_KMAX at most).  Windows one step apart share all but `sy` rows, so after the first window a lane takes the rows that
// left its window out of its private histogram and adds the ones that entered: 2 * sy * PW_ LDS atomics per window instead of
// PW_ * PH_ (the histogram pass was a third of the kernel, bound by the LDS pipe).
// The contraction runs with the roles swapped against k_wvm_prefilter_multi -- A = the digit matrix, B = the pixels -- so that the
// accumulators come out as C[digit row][window]: a lane holds 16 of the 32 digit rows of windows lane & 31 and 32 + (lane & 31), i.e. all
// four digits of 8 of the 16 filters.  v_permlane32_swap builds the pixel operands (a lane's 32 row bytes -> the k-halves the two
// N-tiles want) and, after the digits are folded into exact doubles, brings the two halves of a window's filters together:
// lane == window again without the LDS round trips (pixel staging, 8 KB transpose) of the other formulation.
constexpr int WVD_KMAX = 16;

// one patch row into (ADD_) or out of the lane's histogram
template <int NW_, bool ADD_>
__device__ __forceinline__ void wvd_hist_row(const unsigned int* w4, unsigned int blk4, unsigned int laneOff32, unsigned int inc) {
#pragma unroll
    for (int j = 0; j < NW_; ++j) {
        const unsigned int bins = wvd_bins(w4[j], blk4);
        if (ADD_) {
            wvd_count(wvd_addr<0>(bins, laneOff32), inc); wvd_count(wvd_addr<1>(bins, laneOff32), inc);
            wvd_count(wvd_addr<2>(bins, laneOff32), inc); wvd_count(wvd_addr<3>(bins, laneOff32), inc);
        } else {
            wvd_uncount(wvd_addr<0>(bins, laneOff32), inc); wvd_uncount(wvd_addr<1>(bins, laneOff32), inc);
            wvd_uncount(wvd_addr<2>(bins, laneOff32), inc); wvd_uncount(wvd_addr<3>(bins, laneOff32), inc);
        }
    }
}
// v0's lanes 32..63 <-> v1's lanes 0..31
__device__ __forceinline__ void wvd_swap32(unsigned int& v0, unsigned int& v1) {
    const auto r = __builtin_amdgcn_permlane32_swap(v0, v1, false, false);
    v0 = r[0]; v1 = r[1];
}

// The fp32 cdf of a lane's histogram in the reference's order (cdf[0] = pdf[0]; cdf[b] = cdf[b-1] + pdf[b]) -> the lane's LUT, and
// the exact integer sum / sum of squares of the equalised patch.  cntLds / lutLds: LDS addresses of the lane's u16 counter / LUT byte of
// bin 0 (bin b: + 256 b).  The LUT keeps e + 128 (low byte = e - 128 as int8, what the contractions want); the sums are taken over
// e + 128 too and corrected once.  e <= 255 without the reference's (uchar) cast: the counts add up to N_ and stretch = 255 / N_, so
// cdf <= 255 (1 + 66 * 2^-24) < 255.5.
template <unsigned int N_>
__device__ __forceinline__ void wvd_cdf_lut(unsigned int cntLds, unsigned int lutLds, float stretch, unsigned int& sumx, unsigned int& sumxx) {
    float cdf = 0.f;
    unsigned int s1 = 0, s2 = 0;   // sum cnt * (e + 128), sum cnt * (e + 128)^2 <= 768 * 383^2 < 2^27
    wvd_lds_u16* cnt0 = (wvd_lds_u16*)(uintptr_t)cntLds;   // bin b: + b * 256 bytes (an instruction offset)
    wvd_lds_u8* lut0 = (wvd_lds_u8*)(uintptr_t)lutLds;
#pragma unroll
    for (int bb = 0; bb < 64; bb += 16) {
        unsigned int cnt[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) cnt[j] = cnt0[(bb + j) * 128];   // one LDS round trip per 16 bins, not per bin
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float pdf = (float)cnt[j] * stretch;
            cdf = (bb + j) == 0 ? pdf : cdf + pdf;
            // (uchar)floor((double)cdf + 0.5): cdf < 2^9 has at most 24 significant bits, so cdf + 0.5 is exact in double;
            // floor(cdf) + (frac >= 0.5) is the same value without leaving fp32 (cdf + 0.5f itself can round up to an integer)
            const unsigned int e1 = (unsigned int)cdf + 128u + (__builtin_amdgcn_fractf(cdf) >= 0.5f ? 1u : 0u);
            lut0[(bb + j) * 256] = (unsigned char)e1;
            const unsigned int ce = __umul24(cnt[j], e1);   // <= 768 * 383: all three products are 24-bit multiplies
            s1 = ce + s1;
            s2 = __umul24(ce, e1) + s2;
            asm("" : "+v"(s1), "+v"(s2));   // accumulate here (sunk to their use, the 128 products spill)
        }
    }
    sumx = s1 - 128u * N_;
    sumxx = s2 - 256u * s1 + 16384u * N_;   // sum cnt (e1 - 128)^2
}

template <int PW_, int PH_>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_wvm_prefilter(const uint8_t* __restrict__ arena, WvdTable wt, WvdDev dv) {
    static_assert(PW_ % 4 == 0 && PW_ >= 16 && PW_ <= 32, "rows are read as dwords; a row (or two 16-wide rows) fills one k-step");
    constexpr int RPS = WvdGeo<PW_>::RPS;
    static_assert(PH_ % RPS == 0, "whole k-steps");
    constexpr int KS = PH_ / RPS;
    constexpr int NW = PW_ / 4;          // dwords per patch row
    constexpr int ND = RPS * NW;         // dwords per k-step: 4, 5, 6 or 8
    __shared__ __attribute__((aligned(16384))) WvdLds S;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // constant address space: scalar loads (SMEM) even though the kernel also stores to global memory
    const __attribute__((address_space(4))) WvdConst& C = *(const __attribute__((address_space(4))) WvdConst*)(uintptr_t)dv.c;
    const int L = dv.L;
    const int K = wt.K;
    int li = 0;
    unsigned char* histPtr = reinterpret_cast<unsigned char*>(&S.hist[wave >> 1][0][wave & 1][0]);   // this wavefront's half rows, 256 B apart
    const unsigned int histLds = (unsigned int)(uintptr_t)(wvd_lds_u16*)histPtr;   // LDS byte address: 16 KB block + (wave & 1) * 128
    const unsigned int blkH4 = ((histLds >> 8) & 0xC0u) * 0x01010101u;             // block bits of the address byte, for all four pixels
    // lanes l and l + 32 share a dword of every bin row: they are served in different LDS cycles, so nothing conflicts
    const unsigned int laneOff32 = (histLds & ~0xFF00u) + (unsigned int)(lane & 31) * 4u;
    const unsigned int cntLds = histLds + (unsigned int)(lane & 31) * 4u + (unsigned int)(lane >> 5) * 2u;   // this lane's u16 counter of bin 0
    const unsigned int inc = 1u << (16 * (lane >> 5));
    const unsigned int lutLds = (unsigned int)(uintptr_t)(wvd_lds_u8*)&S.lut[0][lane][wave];   // this lane's LUT byte of bin 0
    const unsigned int blkL4 = ((lutLds >> 8) & 0xC0u) * 0x01010101u;
    const unsigned int lutWord = lutLds & ~0xFF00u;

    int lastImg = -1;
#ifdef FD_WVB_PROF
    const unsigned long long pT0 = __builtin_amdgcn_s_memtime();
    unsigned long long pAcc[4] = {0, 0, 0, 0}, pTiles = 0;
#endif
    // Multi-frame pyramids: workgroup b runs on XCD b % 8 (observed dispatch order; only speed depends on it), and every XCD has its own
    // L2.  With the plain round-robin every L2 pulled the layers of ALL frames across the fabric (162 MB per 64-frame call against
    // 2.5 MB of kept layers); here XCD x takes the frames x, x + 8, ...: each frame's layers live in one L2.
    const bool byXcd = wt.nimg >= 8 && (gridDim.x & 7u) == 0;
    const int xcd = blockIdx.x & 7, vStride = byXcd ? (int)(gridDim.x >> 3) * 4 : (int)gridDim.x * 4;
    const int vEnd = byXcd ? ((wt.nimg - xcd + 7) >> 3) * wt.sTilesPerImage : wt.sTilesPerImage * wt.nimg;
    for (int v = (byXcd ? (int)(blockIdx.x >> 3) : (int)blockIdx.x) * 4 + wave; v < vEnd; v += vStride) {
        const int fi = wt.nimg > 1 ? v / wt.sTilesPerImage : 0;
        const int img = byXcd ? xcd + 8 * fi : fi;   // frame of a multi-frame pyramid
        const int tile = v - fi * wt.sTilesPerImage;
        if (img != lastImg) { li = 0; lastImg = img; }
        while (li + 1 < wt.n && tile >= wt.l[li + 1].sTileFirst) ++li;   // tiles ascend per wavefront inside a frame
        const WvdLayer& wl = wt.l[li];
        const int ntask = wl.nx * wl.G;
        const int task0 = (tile - wl.sTileFirst) * 64 + lane;
        const unsigned int task = (unsigned int)(task0 < ntask ? task0 : ntask - 1);
        unsigned int g = __umulhi(task, wl.magic);   // floor(task / nx) or one less
        unsigned int ix = task - g * (unsigned int)wl.nx;
        if (ix >= (unsigned int)wl.nx) { ix -= wl.nx; ++g; }
        const int iy0 = (int)g * K;
        const int rows = task0 < ntask ? min(K, wl.ny - iy0) : 0;   // windows of this lane; 0: a lane past the layer's last task (it repeats that task, unseen)
        const int lw = wl.lw;
        const size_t rowStep = (size_t)wt.sy * lw;   // bytes between the windows of a column
        const uint8_t* src0 = arena + (size_t)img * wt.imageStride + wl.off + (size_t)(wl.by + iy0 * wt.sy) * lw + (wl.bx + (int)ix * wt.sx);
        const int64_t wid0 = (int64_t)img * wt.perImage + wl.first + (int64_t)iy0 * wl.nx + ix;

        WVD_T(pa);
        // ---- 1. histogram of the lane's first window: 64 bins x 64 lanes of u16 counters
        {
            unsigned char* z = histPtr + (lane >> 3) * 256 + (lane & 7) * 16;   // 8 rows of 128 B per step
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(z + i * 2048) = make_uint4(0, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        {
            unsigned int wn[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j) wn[j] = wvd_load_u32(src0 + 4 * j);
#pragma unroll 2
            for (int r = 0; r < PH_; ++r) {   // one patch row per iteration, the next row's loads in flight
                unsigned int w4[NW];
#pragma unroll
                for (int j = 0; j < NW; ++j) w4[j] = wn[j];
                const uint8_t* nsrc = src0 + (size_t)(r + 1 < PH_ ? r + 1 : r) * lw;
#pragma unroll
                for (int j = 0; j < NW; ++j) wn[j] = wvd_load_u32(nsrc + 4 * j);
                wvd_hist_row<NW, true>(w4, blkH4, laneOff32, inc);
            }
        }
        WVD_T(pb0);
#ifdef FD_WVB_PROF
        pAcc[0] += pb0 - pa;
#endif
#pragma unroll 1
        for (int step = 0; step < K; ++step) {
        const bool active = step < rows;
        if (__ballot(active) == 0) break;
        WVD_T(ps);
        if (step > 0 && active) {   // ---- 1'. slide the histogram down by one window: rows leave at the top, rows enter at the bottom
            const uint8_t* out0 = src0 + (size_t)(step - 1) * rowStep;
            for (int q = 0; q < wt.sy; ++q) {
                unsigned int wo[NW], wi[NW];
#pragma unroll
                for (int j = 0; j < NW; ++j) { wo[j] = wvd_load_u32(out0 + (size_t)q * lw + 4 * j); wi[j] = wvd_load_u32(out0 + (size_t)(q + PH_) * lw + 4 * j); }
                wvd_hist_row<NW, false>(wo, blkH4, laneOff32, inc);
                wvd_hist_row<NW, true>(wi, blkH4, laneOff32, inc);
            }
        }
        // the window this lane evaluates now (a lane that has run out of windows repeats its last one, unseen)
        const uint8_t* src = src0 + (size_t)(active ? step : (rows > 0 ? rows - 1 : 0)) * rowStep;
        wave_sync();
        WVD_T(pb);
        // ---- 2. the fp32 cdf, the LUT, and the exact integer sum / sum of squares of the equalised patch
        unsigned int sumx, sumxx;
        wvd_cdf_lut<PW_ * PH_>(cntLds, lutLds, dv.stretch, sumx, sumxx);
        wave_sync();
        WVD_T(pc);
        // ---- 3. equalise, exact dot products on the matrix pipe: one k-step per patch row (two rows when the patch is 16 wide).
        //         acc[M][N]: digit tile M (rows f + 16 j', digits 2 M + j') x window tile N (windows 32 N + (lane & 31))
        wvd_v16i acc00 = {}, acc01 = {}, acc10 = {}, acc11 = {};
        {
            unsigned int wn[RPS][NW];
#pragma unroll
            for (int rr = 0; rr < RPS; ++rr)
#pragma unroll
                for (int j = 0; j < NW; ++j) wn[rr][j] = wvd_load_u32(src + (size_t)rr * lw + 4 * j);
            wvd_v4i an0 = dv.B[lane], an1 = dv.B[64 + lane];
#pragma unroll 2
            for (int ks = 0; ks < KS; ++ks) {
                const wvd_v4i a0 = an0, a1 = an1;
                unsigned int wl4[ND];
#pragma unroll
                for (int rr = 0; rr < RPS; ++rr)
#pragma unroll
                    for (int j = 0; j < NW; ++j) wl4[rr * NW + j] = wn[rr][j];
                {   // next k-step's rows and digits (the last step re-reads its own)
                    const int kn = ks + 1 < KS ? ks + 1 : ks;
                    const uint8_t* nsrc = src + (size_t)(kn * RPS) * lw;
#pragma unroll
                    for (int rr = 0; rr < RPS; ++rr)
#pragma unroll
                        for (int j = 0; j < NW; ++j) wn[rr][j] = wvd_load_u32(nsrc + (size_t)rr * lw + 4 * j);
                    an0 = dv.B[(kn * 2 + 0) * 64 + lane];
                    an1 = dv.B[(kn * 2 + 1) * 64 + lane];
                }
                unsigned int pk[8];
                if constexpr (ND <= 5) wvd_equalise<ND>(wl4, pk, lutWord, blkL4);
                else { wvd_equalise<ND / 2>(wl4, pk, lutWord, blkL4); wvd_equalise<ND / 2>(wl4 + ND / 2, pk + ND / 2, lutWord, blkL4); }
#pragma unroll
                for (int j = ND; j < 8; ++j) pk[j] = 0;   // k-slots past the row(s): zero pixels against zero digits
                // this lane's 32 row bytes -> the k-half each N-tile wants from it: lanes 0..31 give bytes 0..15 of windows 0..31 (tile 0) /
                // 32..63 (tile 1), lanes 32..63 bytes 16..31
                wvd_swap32(pk[0], pk[4]); wvd_swap32(pk[1], pk[5]); wvd_swap32(pk[2], pk[6]); wvd_swap32(pk[3], pk[7]);
                const wvd_v4i b0 = {(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]}, b1 = {(int)pk[4], (int)pk[5], (int)pk[6], (int)pk[7]};
                acc00 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, acc11, 0, 0, 0);
            }
        }
        WVD_T(pd);
        // ---- 4. digits -> exact integer dot products, lane == window.  Register r of acc[M][N] is digit row (r & 3) + 8 (r >> 2) + 4 h
        //         (h = lane >> 5) of tile M: filter fs(i) = (i & 3) + 8 (i >> 2) + 4 h for i = r & 7, digit 2 M + (r >> 3).  After the
        //         swap, xq[0][i] is filter (i & 3) + 8 (i >> 2) of THIS lane's window and xq[1][i] filter (i & 3) + 8 (i >> 2) + 4.
        double xq[2][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // exact: |digit sums| < 2^24, the result < 2^53
            const double v0 = ((double)acc00[i] + 65536.0 * (double)acc10[i]) + 256.0 * ((double)acc00[8 + i] + 65536.0 * (double)acc10[8 + i]);
            const double v1 = ((double)acc01[i] + 65536.0 * (double)acc11[i]) + 256.0 * ((double)acc01[8 + i] + 65536.0 * (double)acc11[8 + i]);
            uint2 u0 = __builtin_bit_cast(uint2, v0), u1 = __builtin_bit_cast(uint2, v1);
            wvd_swap32(u0.x, u1.x);
            wvd_swap32(u0.y, u1.y);
            xq[0][i] = __builtin_bit_cast(double, u0);
            xq[1][i] = __builtin_bit_cast(double, u1);
        }
        // ---- 5. the first L cascade levels of this lane's window with error bounds
        bool undecided = active;
        {
            typedef float wvd_v2f __attribute__((ext_vector_type(2)));
            wvd_v2f KK[WVD_L];   // {K_p, bound of its error}: the level sum and its bound run as ONE v_pk_fma_f32 per term
            // the reference's fp32 sum of squares: row totals (exact ints) added in fp32, so it IS the integer below 2^24
            const float sxx = (float)sumxx;
            // |norm - reference norm|: 2 * quantisation error of xp, the fp32 sum of squares above 2^24, slack for the reference's
            // fp64 roundings (< 1e-5)
            const double dn = dv.scale * (double)sumx + (sumxx >= (1u << 24) ? (double)dv.sxxSlack : 0.0) + 1e-4;
            const float relDn = (float)(-(double)dv.negBasis * dn) * 1.0001f;
            const __attribute__((address_space(4))) wvd_v2f* W2 = (const __attribute__((address_space(4))) wvd_v2f*)&C.w2[0][0][0];
#pragma unroll
            for (int k = 0; k < WVD_L; ++k) {
                if (k < L) {
                    const double xp = (xq[(k >> 2) & 1][(k & 3) + 4 * (k >> 3)] + C.c128[k]) * dv.scale;
                    double norm = (double)sxx;
                    norm = norm - 2 * xp;
                    norm = norm + C.pp[k];
                    const float arg = (float)((double)dv.negBasis * norm);
                    // relative error of K: exponent error (quantisation, float cast of the argument, x * log2e, 2^x), the final
                    // rounding, and -- folded in here -- the fp32 summation-order term (4k + 16) 2^-24 <= 4.6e-6 of the level sums.
                    // Branch-free: the exponential is taken of every argument and replaced where it is out of range (selects, not the
                    // three exec-masked blocks per level the if / else chain compiled to)
                    const float Kraw = __expf(arg);
                    const float rho = (relDn + fabsf(arg) * 2.4e-7f + 6.0e-7f) * 1.01f + 4.6e-6f;
                    const bool lo = arg < -80.0f, hi = arg > 80.0f;   // lo: true K <= e^-80 (1 + tiny); hi cannot happen for a sane model: never reject
                    const float Kk = (lo || hi) ? 0.f : Kraw;
                    const float Kerr = hi ? 3.0e38f : (lo ? 2e-35f : Kraw * rho + 1e-37f);
                    KK[k] = wvd_v2f{Kk, Kerr};
                    wvd_v2f RE = {dv.negBias, fabsf(dv.negBias) * 4.6e-6f + 1e-37f};
#pragma unroll
                    for (int p = 0; p <= k; ++p) RE = __builtin_elementwise_fma(W2[k * WVD_L + p], KK[p], RE);   // R += w K_p, E += |w| dK_p
                    // the reference leaves at the first level with res < thr, and res_ref <= R + E
                    if (undecided && (RE.x + RE.y < C.thr[k])) undecided = false;
                }
            }
        }
        // ---- 6. survivors -> queue of the exact cascade (wave-aggregated)
        {
            const unsigned long long mask = __ballot(undecided);
            if (mask) {
                unsigned int base = 0;
                if (lane == 0) base = atomicAdd(dv.qcount, (unsigned int)__popcll(mask));
                base = __builtin_amdgcn_readfirstlane(base);
                if (undecided) dv.q[base + __popcll(mask & ((1ull << lane) - 1ull))] = wid0 + (int64_t)step * wl.nx;
            }
        }
#ifdef FD_WVB_PROF
        {
            const unsigned long long pe = __builtin_amdgcn_s_memtime();
            pAcc[0] += pb - ps; pAcc[1] += pc - pb; pAcc[2] += pd - pc; pAcc[3] += pe - pd; ++pTiles;
        }
#endif
        }   // windows of the column
        wave_sync();
    }
#ifdef FD_WVB_PROF
    if (lane == 0) {
        const unsigned int gw = blockIdx.x * 4 + wave;
        if (gw < (unsigned int)WVD_PROF_WAVES) {
            unsigned int hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* r = fd_wvd_prof + (size_t)gw * 8;
            r[0] = pT0; r[1] = __builtin_amdgcn_s_memtime(); r[2] = (unsigned long long)hw | ((unsigned long long)xcc << 32); r[3] = pTiles;
            r[4] = pAcc[0]; r[5] = pAcc[1]; r[6] = pAcc[2]; r[7] = pAcc[3];
        }
    }
#endif
}

// ---- stage B's k_wvb_prepare with lane == window ---------------------------------------------------------------------------
// HistEq64 + sums of the windows queued for stage B -> state set 0 (wvm_stageb.hpp), 64 queued windows per wavefront with the
// pre-filter's machinery (private histogram columns, register cdf chain, LUT gather) instead of one window per wavefront: models
// that keep rejecting deep into the cascade queue 28-64 % of all windows, and the wave == window kernel was 18 % of such a call.
// A lane locates its window from the id (frame, layer, row, column: a table of the layers in LDS), equalises it and writes the patch
// as x - 128 bytes, the window id, sum(x) (exact) and the reference's fp32 row-ordered sum of squares (IImg.cpp:33-47: the row totals,
// exact integers, added in fp32 from the top row down).
struct WvdPrepLds {
    unsigned int hist[2][64][2][32];
    unsigned char lut[64][64][4];        // [bin][lane][wave], like WvdLds
    int4 layer[WVM_MAX_LAYERS][2];   // {bx, by, nx, lw}, {off, magic, first, 0}
};
template <int PW_, int PH_>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_wvb_prepare_lanes(const uint8_t* __restrict__ arena, WinTable wt, float stretch,
                                                                                                       WvbDev mv, WvbState s, const int64_t* q, const unsigned int* qcount) {
    constexpr int NW = PW_ / 4;
    __shared__ __attribute__((aligned(16384))) WvdPrepLds S;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if ((int)threadIdx.x < wt.n) {
        const WinLayerDev& l = wt.l[threadIdx.x];
        S.layer[threadIdx.x][0] = make_int4(l.bx, l.by, l.nx, l.lw);
        S.layer[threadIdx.x][1] = make_int4((int)l.off, (int)l.magic, (int)l.first, 0);
    }
    __syncthreads();
    unsigned char* histPtr = reinterpret_cast<unsigned char*>(&S.hist[wave >> 1][0][wave & 1][0]);
    const unsigned int histLds = (unsigned int)(uintptr_t)(wvd_lds_u16*)histPtr;
    const unsigned int blkH4 = ((histLds >> 8) & 0xC0u) * 0x01010101u;
    const unsigned int laneOff32 = (histLds & ~0xFF00u) + (unsigned int)(lane & 31) * 4u;
    const unsigned int cntLds = histLds + (unsigned int)(lane & 31) * 4u + (unsigned int)(lane >> 5) * 2u;
    const unsigned int inc = 1u << (16 * (lane >> 5));
    const unsigned int lutLds = (unsigned int)(uintptr_t)(wvd_lds_u8*)&S.lut[0][lane][wave];
    const unsigned int blkL4 = ((lutLds >> 8) & 0xC0u) * 0x01010101u;
    const unsigned int lutWord = lutLds & ~0xFF00u;
    const unsigned int perImage = (unsigned int)wt.per_image;                          // the launcher checks that all ids fit 32 bits
    const unsigned int magicPI = wt.nimg > 1 ? (unsigned int)(0xffffffffu / perImage) : 0u;   // mulhi(id, magic) = id / perImage or one less
    const unsigned int n = wvb_count(qcount, s);
    for (unsigned int tile = blockIdx.x * 4 + wave; tile * 64 < n; tile += gridDim.x * 4) {
        const unsigned int pos = tile * 64 + lane;
        const bool valid = pos < n;
        const int64_t wid = q[valid ? pos : n - 1];
        // ---- locate
        unsigned int local = (unsigned int)wid, img = 0;
        if (wt.nimg > 1) {
            img = __umulhi(local, magicPI);
            local -= img * perImage;
            if (local >= perImage) { local -= perImage; ++img; }
        }
        int li = 0;
        for (int l = 1; l < wt.n; ++l) li += local >= (unsigned int)wt.l[l].first ? 1 : 0;   // layer starts: scalar operands
        const int4 la = S.layer[li][0], lb = S.layer[li][1];
        const unsigned int idx = local - (unsigned int)lb.z;
        unsigned int iy = __umulhi(idx, (unsigned int)lb.y);
        unsigned int ix = idx - iy * (unsigned int)la.z;
        if (ix >= (unsigned int)la.z) { ix -= la.z; ++iy; }
        const int lw = la.w;
        const uint8_t* src = arena + (size_t)img * wt.image_stride + (unsigned int)lb.x + (size_t)(la.y + (int)iy * wt.sy) * lw + (la.x + (int)ix * wt.sx);
        // ---- histogram, cdf, LUT
        {
            unsigned char* z = histPtr + (lane >> 3) * 256 + (lane & 7) * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(z + i * 2048) = make_uint4(0, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        {
            unsigned int wn[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j) wn[j] = wvd_load_u32(src + 4 * j);
#pragma unroll 2
            for (int r = 0; r < PH_; ++r) {
                unsigned int w4[NW];
#pragma unroll
                for (int j = 0; j < NW; ++j) w4[j] = wn[j];
                const uint8_t* nsrc = src + (size_t)(r + 1 < PH_ ? r + 1 : r) * lw;
#pragma unroll
                for (int j = 0; j < NW; ++j) wn[j] = wvd_load_u32(nsrc + 4 * j);
                wvd_hist_row<NW, true>(w4, blkH4, laneOff32, inc);
            }
        }
        wave_sync();
        unsigned int sumx, sumxx;
        wvd_cdf_lut<PW_ * PH_>(cntLds, lutLds, stretch, sumx, sumxx);
        wave_sync();
        // ---- equalise row by row: the patch as x - 128 bytes, the fp32 sum of squares in row order
        int8_t* xr = s.X[0] + (size_t)pos * mv.dstride;
        float sxx = 0.f;
        {
            unsigned int wn[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j) wn[j] = wvd_load_u32(src + 4 * j);
#pragma unroll 2
            for (int r = 0; r < PH_; ++r) {
                unsigned int w4[NW], pk[NW];
#pragma unroll
                for (int j = 0; j < NW; ++j) w4[j] = wn[j];
                const uint8_t* nsrc = src + (size_t)(r + 1 < PH_ ? r + 1 : r) * lw;
#pragma unroll
                for (int j = 0; j < NW; ++j) wn[j] = wvd_load_u32(nsrc + 4 * j);
                if constexpr (NW <= 5) wvd_equalise<NW>(w4, pk, lutWord, blkL4);
                else { wvd_equalise<NW / 2>(w4, pk, lutWord, blkL4); wvd_equalise<NW / 2>(w4 + NW / 2, pk + NW / 2, lutWord, blkL4); }
                unsigned int rowsq = 0;
#pragma unroll
                for (int j = 0; j < NW; ++j) {
                    const unsigned int x4 = pk[j] ^ 0x80808080u;
                    rowsq = __builtin_amdgcn_udot4(x4, x4, rowsq, false);
                }
                sxx = r == 0 ? (float)rowsq : sxx + (float)rowsq;
                if (valid) {
#pragma unroll
                    for (int j = 0; j < NW; ++j) *reinterpret_cast<unsigned int*>(xr + r * PW_ + 4 * j) = pk[j];
                }
            }
        }
        if (valid) {
            for (int i = PW_ * PH_; i < mv.dstride; i += 4) *reinterpret_cast<unsigned int*>(xr + i) = 0;
            s.wid[0][pos] = wid;
            s.aux[0][pos] = make_int2((int)sumx, __float_as_int(sxx));
        }
        (void)sumxx;
        wave_sync();
    }
}

// ---- the same for several detectors that share their windows ------------------------------------------------------------
// HistEq64 depends on the window only, so the histogram, the cdf chain and the LUT (steps 1-2, ~45 % of the single-detector
// kernel) are done once per tile; the equalise + contraction + levels (steps 3-6) run once per detector against its own digit
// matrix, constants and queue.  The LUT therefore has to outlive the transposes that reuse the histogram block: it is kept as
// bytes in its own 4 KB block per wavefront ([bin][lane], 4 KB-aligned so that (bin << 6) | (block + lane) is a complete
// address), and the k-step pixel chunk lives in the (then dead) histogram block.  48 KB of LDS per workgroup = 3 workgroups per
// CU; three wavefronts per SIMD were measured equal to four for this kernel (it is bound by instruction issue).
struct __attribute__((aligned(8192))) WvdLdsM {
    unsigned short hist[4][64][64];                  // [wave][bin][slot of the lane]: counters; then the k-step chunk (first 2 KB); then the transposes
    unsigned char lut[4][64][64];                    // [wave][bin][lane]
};
template <int B_>
__device__ __forceinline__ unsigned int wvd_slot_lut(unsigned int w4, unsigned int laneOffLut) {
    unsigned int bin, a;
    asm("v_bfe_u32 %0, %1, %2, 6" : "=v"(bin) : "v"(w4), "n"(8 * B_ + 2));
    asm("v_lshl_or_b32 %0, %1, 6, %2" : "=v"(a) : "v"(bin), "v"(laneOffLut));
    return a;
}
__device__ __forceinline__ unsigned int wvd_lut8(unsigned int ldsAddr) { return *(__attribute__((address_space(3))) unsigned char*)(uintptr_t)ldsAddr; }   // ds_read_u8

template <int PW_, int PH_>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_wvm_prefilter_multi(const uint8_t* __restrict__ arena, WvdTable wt, WvdMulti mv) {
    static_assert(PW_ % 4 == 0 && PW_ >= 16 && PW_ <= 32, "rows are read as dwords; a row (or two 16-wide rows) fills one k-step");
    constexpr int RPS = WvdGeo<PW_>::RPS;
    static_assert(PH_ % RPS == 0, "whole k-steps");
    constexpr int KS = PH_ / RPS;
    constexpr int NW = PW_ / 4;          // dwords per patch row
    __shared__ WvdLdsM S;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float stretch = mv.d[0].stretch, sxxSlack = mv.d[0].sxxSlack;   // the same for all detectors of a group (patch size)
    const int ntiles = wt.ntiles;
    int li = 0;
    unsigned char* histPtr = reinterpret_cast<unsigned char*>(&S.hist[wave][0][0]);
    unsigned char* xPtr = histPtr;   // the k-step chunk lives in the histogram block (dead between the LUT and the transposes)
    const unsigned int lutLds = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&S.lut[wave][0][0];   // 4 KB-aligned
    const unsigned int laneOffLut = lutLds + (unsigned int)lane;
    double* trPtr = reinterpret_cast<double*>(histPtr);
    const unsigned int histLds = (unsigned int)(uintptr_t)(wvd_lds_u16*)&S.hist[wave][0][0];   // LDS byte address, 8 KB-aligned
    // lanes l and l + 32 share a dword of every bin row: they are served in different LDS cycles, so nothing conflicts
    const unsigned int laneOff32 = histLds + (unsigned int)(lane & 31) * 4u;
    const unsigned int laneOff16 = laneOff32 + (unsigned int)(lane >> 5) * 2u;
    const unsigned int inc = 1u << (16 * (lane >> 5));

    int lastImg = 0;
    for (int gtile = blockIdx.x * 4 + wave; gtile < ntiles; gtile += gridDim.x * 4) {
        const int img = wt.nimg > 1 ? gtile / wt.tilesPerImage : 0;   // frame of a multi-frame pyramid
        const int tile = gtile - img * wt.tilesPerImage;
        if (img != lastImg) { li = 0; lastImg = img; }
        while (li + 1 < wt.n && tile >= wt.l[li + 1].tileFirst) ++li;   // tiles ascend per wavefront inside a frame
        const WvdLayer& wl = wt.l[li];
        const int local0 = (tile - wl.tileFirst) * 64 + lane;
        const bool valid = local0 < wl.nwin;
        const unsigned int local = (unsigned int)(valid ? local0 : wl.nwin - 1);
        unsigned int iy = __umulhi(local, wl.magic);   // floor(local / nx) or one less
        unsigned int ix = local - iy * (unsigned int)wl.nx;
        if (ix >= (unsigned int)wl.nx) { ix -= wl.nx; ++iy; }
        const int lw = wl.lw;
        const uint8_t* src = arena + (size_t)img * wt.imageStride + wl.off + (size_t)(wl.by + (int)iy * wt.sy) * lw + (wl.bx + (int)ix * wt.sx);
        const int64_t wid = (int64_t)img * wt.perImage + wl.first + local;

        // ---- 1. histogram: 64 bins x 64 lanes of u16 counters
        {
            uint4* z = reinterpret_cast<uint4*>(histPtr);
#pragma unroll
            for (int i = 0; i < 8; ++i) z[i * 64 + lane] = make_uint4(0, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        {
            unsigned int wn[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j) wn[j] = wvd_load_u32(src + 4 * j);
#pragma unroll 1
            for (int r = 0; r < PH_; ++r) {   // one patch row per iteration, the next row's loads in flight
                unsigned int w4[NW];
#pragma unroll
                for (int j = 0; j < NW; ++j) w4[j] = wn[j];
                const uint8_t* nsrc = src + (size_t)(r + 1 < PH_ ? r + 1 : r) * lw;
#pragma unroll
                for (int j = 0; j < NW; ++j) wn[j] = wvd_load_u32(nsrc + 4 * j);
#pragma unroll
                for (int j = 0; j < NW; ++j) {
                    wvd_count(wvd_slot<0>(w4[j], laneOff32), inc);
                    wvd_count(wvd_slot<1>(w4[j], laneOff32), inc);
                    wvd_count(wvd_slot<2>(w4[j], laneOff32), inc);
                    wvd_count(wvd_slot<3>(w4[j], laneOff32), inc);
                }
            }
        }
        wave_sync();
        // ---- 2. the fp32 cdf in the reference's order (cdf[0] = pdf[0]; cdf[b] = cdf[b-1] + pdf[b]), the LUT, and the exact
        //         integer sum / sum of squares of the equalised patch
        unsigned int sumx = 0, sumxx = 0;
        {
            float cdf = 0.f;
#pragma unroll
            for (int b = 0; b < 64; ++b) {
                wvd_lds_u16* slot = (wvd_lds_u16*)(uintptr_t)((unsigned int)(b << 7) + laneOff16);   // laneOff16 is a complete LDS address
                const unsigned int cnt = *slot;
                const float pdf = (float)cnt * stretch;
                cdf = b == 0 ? pdf : cdf + pdf;
                // (uchar)floor((double)cdf + 0.5): cdf < 2^9 has at most 24 significant bits, so cdf + 0.5 is exact in double;
                // floor(cdf) + (frac >= 0.5) is the same value without leaving fp32 (cdf + 0.5f itself can round up to an integer)
                const float fl = floorf(cdf);
                const float up = (cdf - fl >= 0.5f) ? fl + 1.0f : fl;
                const unsigned int e = (unsigned int)up & 255u;
                *(__attribute__((address_space(3))) unsigned char*)(uintptr_t)(lutLds + (unsigned int)(b << 6) + (unsigned int)lane) = (unsigned char)e;
                const unsigned int ce = cnt * e;   // <= 768 * 255
                sumx += ce;
                sumxx += ce * e;                   // <= 768 * 65025 < 2^26
                asm("" : "+v"(sumx), "+v"(sumxx));   // accumulate here (sunk to their use, the 128 products spill)
            }
        }
        wave_sync();
#pragma unroll 1
        for (int det = 0; det < mv.nd; ++det) {
        const WvdDev& dv = mv.d[det];
        // constant address space: scalar loads (SMEM) even though the kernel also stores to global memory
        const __attribute__((address_space(4))) WvdConst& C = *(const __attribute__((address_space(4))) WvdConst*)(uintptr_t)dv.c;
        const int L = dv.L;
        // ---- 3. equalise, exact dot products on the matrix pipe: one k-step per patch row (two rows when the patch is 16 wide)
        wvd_v16i acc00 = {}, acc01 = {}, acc10 = {}, acc11 = {};
        {
            unsigned int wn[RPS][NW];
#pragma unroll
            for (int rr = 0; rr < RPS; ++rr)
#pragma unroll
                for (int j = 0; j < NW; ++j) wn[rr][j] = wvd_load_u32(src + (size_t)rr * lw + 4 * j);
            wvd_v4i bn0 = dv.B[lane], bn1 = dv.B[64 + lane];
#pragma unroll 1
            for (int ks = 0; ks < KS; ++ks) {
                const wvd_v4i b0 = bn0, b1 = bn1;
                unsigned int w4[RPS][NW];
#pragma unroll
                for (int rr = 0; rr < RPS; ++rr)
#pragma unroll
                    for (int j = 0; j < NW; ++j) w4[rr][j] = wn[rr][j];
                {   // next k-step's rows and digits (the last step re-reads its own)
                    const int kn = ks + 1 < KS ? ks + 1 : ks;
                    const uint8_t* nsrc = src + (size_t)(kn * RPS) * lw;
#pragma unroll
                    for (int rr = 0; rr < RPS; ++rr)
#pragma unroll
                        for (int j = 0; j < NW; ++j) wn[rr][j] = wvd_load_u32(nsrc + (size_t)rr * lw + 4 * j);
                    bn0 = dv.B[(kn * 2 + 0) * 64 + lane];
                    bn1 = dv.B[(kn * 2 + 1) * 64 + lane];
                }
                unsigned int pk[RPS * NW];
#pragma unroll
                for (int rr = 0; rr < RPS; ++rr) {
#pragma unroll
                    for (int j = 0; j < NW; ++j) {
                        const unsigned int w = w4[rr][j];
                        const unsigned int e0 = wvd_lut8(wvd_slot_lut<0>(w, laneOffLut));
                        const unsigned int e1 = wvd_lut8(wvd_slot_lut<1>(w, laneOffLut));
                        const unsigned int e2 = wvd_lut8(wvd_slot_lut<2>(w, laneOffLut));
                        const unsigned int e3 = wvd_lut8(wvd_slot_lut<3>(w, laneOffLut));
                        pk[rr * NW + j] = wvd_lshl_or(e3, 24, wvd_lshl_or(e2, 16, wvd_lshl_or(e1, 8, e0))) ^ 0x80808080u;   // x - 128 as int8
                    }
                }
                // slots RPS * PW_ .. 31 of the k-step are never written: their digits are zero, so stale bytes multiply into nothing
                unsigned char* xrow = xPtr + lane * 16;
                constexpr int ND = RPS * NW;   // 4, 5, 6 or 8 dwords
                *reinterpret_cast<uint4*>(xrow) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                if constexpr (ND == 5) *reinterpret_cast<unsigned int*>(xrow + 1024) = pk[4];
                if constexpr (ND == 6) *reinterpret_cast<uint2*>(xrow + 1024) = make_uint2(pk[4], pk[5]);
                if constexpr (ND == 8) *reinterpret_cast<uint4*>(xrow + 1024) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
                __builtin_amdgcn_wave_barrier();   // LDS operations of a wavefront execute in order: the reads below see these writes
                const wvd_v4i a0 = *reinterpret_cast<const wvd_v4i*>(xPtr + (lane >> 5) * 1024 + (lane & 31) * 16);
                const wvd_v4i a1 = *reinterpret_cast<const wvd_v4i*>(xPtr + (lane >> 5) * 1024 + (32 + (lane & 31)) * 16);
                __builtin_amdgcn_wave_barrier();
                acc00 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, acc11, 0, 0, 0);
            }
        }
        // ---- 4. digits -> exact integer dot products, transposed to lane == window.  Column g = f + 16 j of N-tile g / 32 holds
        //         digit j of filter f: this lane (column lane & 31) has digit j0 = (lane >> 4) & 1 in tile 0 and digit j0 + 2 in tile 1
        wave_sync();   // the k-step chunk is dead: the histogram block becomes the transpose buffer
        {
            int laneT = lane;
            asm volatile("" : "+v"(laneT));   // the 32 slot addresses are cheap: computed here, not hoisted out of the tile loop and spilled
            const bool lowDigit = (laneT & 16) == 0;
            const int h4 = 4 * (laneT >> 5);
            const int fh = (laneT & 15) ^ h4;   // row & 15 = (rowc & 15) | h4 (rowc & 15 has bit 2 clear), so f ^ (row & 15) = fh ^ (rowc & 15)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int s0 = mt == 0 ? acc00[rg] : acc10[rg];
                    const int s2 = mt == 0 ? acc01[rg] : acc11[rg];
                    const double part = (double)s0 + 65536.0 * (double)s2;          // exact: |s| < 2^24
                    const double other = __shfl_xor(part, 16);
                    const int rowc = mt * 32 + (rg & 3) + 8 * (rg >> 2);             // window row = rowc + h4
                    if (lowDigit) trPtr[(rowc + h4) * 16 + (fh ^ (rowc & 15))] = part + 256.0 * other;   // exact: < 2^53
                }
            }
        }
        wave_sync();
        // ---- 5. the first L cascade levels of this lane's window with error bounds
        bool undecided = valid;
        {
            int laneC = lane;
            asm volatile("" : "+v"(laneC));   // as above, for the 16 read addresses
            float Kv[WVD_L], Ke[WVD_L];
            // the reference's fp32 sum of squares: row totals (exact ints) added in fp32, so it IS the integer below 2^24
            const float sxx = (float)sumxx;
            // |norm - reference norm|: 2 * quantisation error of xp, the fp32 sum of squares above 2^24, slack for the reference's
            // fp64 roundings (< 1e-5)
            const double dn = dv.scale * (double)sumx + (sumxx >= (1u << 24) ? (double)sxxSlack : 0.0) + 1e-4;
            const float relDn = (float)(-(double)dv.negBasis * dn) * 1.0001f;
#pragma unroll
            for (int k = 0; k < WVD_L; ++k) {
                if (k < L) {
                    const double xp = (trPtr[laneC * 16 + (k ^ (laneC & 15))] + C.c128[k]) * dv.scale;
                    double norm = (double)sxx;
                    norm = norm - 2 * xp;
                    norm = norm + C.pp[k];
                    const float arg = (float)((double)dv.negBasis * norm);
                    float Kk, Kerr;
                    if (arg < -80.0f) { Kk = 0.f; Kerr = 2e-35f; }           // true K <= e^-80 (1 + tiny)
                    else if (arg > 80.0f) { Kk = 0.f; Kerr = 3.0e38f; }       // cannot happen for a sane model: never reject
                    else {
                        Kk = __expf(arg);
                        // relative error of K: exponent error (quantisation, float cast of the argument, x * log2e, 2^x), the final
                        // rounding, and -- folded in here -- the fp32 summation-order term (4k + 16) 2^-24 <= 4.6e-6 of the level sums
                        const float rho = (relDn + fabsf(arg) * 2.4e-7f + 6.0e-7f) * 1.01f + 4.6e-6f;
                        Kerr = Kk * rho + 1e-37f;
                    }
                    Kv[k] = Kk;
                    Ke[k] = Kerr;
                    float R = dv.negBias, E = fabsf(dv.negBias) * 4.6e-6f + 1e-37f;
#pragma unroll
                    for (int p = 0; p <= k; ++p) {
                        const float w = C.w[k][p];
                        R = fmaf(w, Kv[p], R);
                        E = fmaf(fabsf(w), Ke[p], E);
                    }
                    // the reference leaves at the first level with res < thr, and res_ref <= R + E
                    if (undecided && (R + E < C.thr[k])) undecided = false;
                }
            }
        }
        // ---- 6. survivors -> queue of the exact cascade (wave-aggregated)
        {
            const unsigned long long mask = __ballot(undecided);
            if (mask) {
                unsigned int base = 0;
                if (lane == 0) base = atomicAdd(dv.qcount, (unsigned int)__popcll(mask));
                base = __builtin_amdgcn_readfirstlane(base);
                if (undecided) dv.q[base + __popcll(mask & ((1ull << lane) - 1ull))] = wid;
            }
        }
        wave_sync();
        }   // detectors
    }
}


}  // namespace
