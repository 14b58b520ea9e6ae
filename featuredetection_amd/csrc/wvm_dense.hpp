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
//     2. the equalised pixels (x - 128 as int8, gathered through the LUT; a k-step is 32 consecutive pixels of the row-major patch,
//        whatever the row length: 13 k-steps for 20x20, not 20) are the B operand of v_mfma_i32_32x32x32_i8, built in registers with v_permlane32_swap; the A operand is the residual images
//        quantised to 32-bit integers Q = round(r * 2^s) and split into four balanced base-256 digits: integer arithmetic, so x . Q is
//        EXACT; |x . r - 2^-s x . Q| <= 2^-(s+1) * sum(x) is the only approximation.  The accumulators C[digit row][window] are folded
//        into exact doubles and brought back to lane == window with v_permlane32_swap.
//     3. per window: norm, K = exp(-basis * norm), res_k = -bias + sum_p w[k][p] K_p with a rigorous error bound eps_k on
//        |res_k - reference res_k| (quantisation, fast fp32 exp, fp32 summation order).  A window with res_k + eps_k < thr_k at ANY
//        level k < L is rejected by the reference at some level <= k, so it cannot be a WVM positive and is dropped here.
//        Everything else is appended to a queue and runs the exact stage B (wvm_stageb.hpp) from level 0, so positives, their levels
//        and their fp32 outputs stay bit-identical to the rectangle-sum formulation.
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
    int32_t nwin, pad0;
    int64_t first;
    int32_t ny, G, sTileFirst, pad;      // G = ceil(ny / K) row groups; tiles of 64 (column, row group) tasks
};
struct WvdTable {
    int32_t n, sx, sy, pad0;
    int32_t nimg, pad1;                  // multi-frame pyramid: tile -> (frame, tile inside the frame)
    int32_t K, sTilesPerImage;           // windows a lane walks down (WVD_KMAX at most), tiles per frame
    int64_t perImage;                    // windows per frame
    uint64_t imageStride;                // bytes between the frames' arenas
    WvdLayer l[WVM_MAX_LAYERS];
};

// per-level model constants of the dense stage (device memory, read through the scalar cache)
// the scalars k_wvm_prefilter takes as kernel arguments (WvdDev), for k_wvm_prefilter_group, which reads them per detector
struct WvdScal {
    int32_t L;
    float negBasis, negBias, stretch, sxxSlack;
    int32_t pad;
    double scale, nb2, mXq;
};
struct WvdConst {
    double cA[WVD_L];          // log2(e) * (-basis) * (pp_k - 2 * 2^-s * 128 * sum_i Q_k[i]): the window-independent part of level k's exponent
    float thr[WVD_L];          // -inf from level L on (those levels never reject)
    float w2[WVD_L][WVD_L][2]; // {hkWeights[k][p], |hkWeights[k][p]|}, p <= k: the level sum and its error bound as one packed fma
    WvdScal sc;
};

struct WvdDev {
    const wvd_v4i* B;          // [k-step][2 M-tiles][64 lanes] 16 signed digit bytes each; k-step ks, slot t = pixel 32 ks + t of the row-major patch
    const WvdConst* c;
    int64_t* q;                // windows that pass
    unsigned int* qcount;
    // scalars (kernel arguments, so that they live in SGPRs)
    int32_t L;
    float negBasis, negBias, stretch;
    float sxxSlack;            // 2 * ph + 2: |fp32 row-ordered sum of squares - exact integer| when the sum is >= 2^24
    double scale;              // 2^-s; also the error of norm per unit of sum(x): 2 * 2^-(s+1)
    double nb2;                // log2(e) * (-basis): exponent of 2 per unit of norm
    double mXq;                // -2 * 2^-s * nb2: exponent of 2 per unit of the integer dot product x' . Q
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
// A patch row of NW dwords from an arbitrary byte address with as few instructions as possible (dwordx4 / x2 / x1: unaligned vector
// loads are legal on this part).  A lane that reads ITS OWN window (k_wvb_prepare_lanes) makes every load instruction touch 64 different
// cache lines, so the number of instructions, not the bytes, is what the texture addresser is busy with.
template <int NW>
__device__ __forceinline__ void wvd_load_row(const uint8_t* p, unsigned int (&w)[NW]) {
    int j = 0;
#pragma unroll
    for (; j + 4 <= NW; j += 4) { uint4 v; __builtin_memcpy(&v, p + 4 * j, 16); w[j] = v.x; w[j + 1] = v.y; w[j + 2] = v.z; w[j + 3] = v.w; }
    if (j + 2 <= NW) { uint2 v; __builtin_memcpy(&v, p + 4 * j, 8); w[j] = v.x; w[j + 1] = v.y; j += 2; }
    if (j < NW) w[j] = wvd_load_u32(p + 4 * j);
}
// the same with a wave-uniform base and a 32-bit lane offset: global_load_dword v, v_off, s[base] offset:imm -- no address arithmetic
// on the vector unit (the uniform part of an address advances on the scalar unit)
__device__ __forceinline__ unsigned int wvd_load_u32(const uint8_t* ubase, unsigned int voff) {
    unsigned int v;
    __builtin_memcpy(&v, ubase + voff, 4);
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
// bin 0 (bin b: + 256 b).  (uchar)floor((double)cdf + 0.5) (HistEq64Filter.cpp:118) is ONE instruction: v_cvt_rpi_i32_f32 is
// floor(x + 0.5) without an intermediate rounding (a float cdf + 0.5f can round up to an integer; tools/microbench/rpi_check.hip
// compares it with floor((double)x + 0.5) for every float in [0, 1024) on the device: no mismatch).  e <= 255 without the
// reference's (uchar) cast: the counts add up to N_ and stretch = 255 / N_, so cdf <= 255 (1 + 66 * 2^-24) < 255.5.
// The LUT keeps e ^ 0x80 = e - 128 as int8, what the contractions want.  8 VALU per bin (round 3: 11).
__device__ __forceinline__ unsigned int wvd_rpi(float x) {
    unsigned int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ unsigned int wvd_mul24(unsigned int a, unsigned int b) {   // by name: __umul24 of opaque operands becomes v_and + quarter-rate v_mul_lo
    unsigned int r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned int wvd_mad24(unsigned int a, unsigned int b, unsigned int c) {
    unsigned int r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <unsigned int N_>
__device__ __forceinline__ void wvd_cdf_lut(unsigned int cntLds, unsigned int lutLds, float stretch, unsigned int& sumx, unsigned int& sumxx) {
    static_assert(N_ <= 1024, "count * e and count * e * e are 24-bit multiplies");
    float cdf = 0.f;
    unsigned int s1 = 0, s2 = 0;   // sum cnt * e <= 255 N_, sum cnt * e^2 <= 65025 N_ < 2^26
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
            const unsigned int e = wvd_rpi(cdf);
            lut0[(bb + j) * 256] = (unsigned char)(e ^ 0x80u);
            const unsigned int ce = wvd_mul24(cnt[j], e);
            s1 = ce + s1;
            s2 = wvd_mad24(ce, e, s2);
            asm("" : "+v"(s1), "+v"(s2));   // accumulate here (sunk to their use, the 128 products spill)
        }
    }
    sumx = s1;
    sumxx = s2;
}

template <int PW_, int PH_>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_wvm_prefilter(const uint8_t* __restrict__ arena, WvdTable wt, WvdDev dv) {
    static_assert(PW_ % 4 == 0 && PW_ >= 16 && PW_ <= 32, "rows are read as dwords");
    constexpr int NW = PW_ / 4;          // dwords per patch row
    constexpr int D4 = NW * PH_;         // dwords of the row-major patch; a k-step takes 8 of them (32 pixels), whatever the row length
    static_assert(D4 % 4 == 0, "the equalise blocks take 4 dwords");
    constexpr int KS = (D4 + 7) / 8;
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
        const unsigned int rowStep = (unsigned int)(wt.sy * lw);   // bytes between the windows of a column
        // addresses = a wave-uniform base (scalar registers) + a 32-bit lane offset
        const uint8_t* ubase = arena + (size_t)img * wt.imageStride + wl.off;
        const unsigned int lo0 = (unsigned int)((wl.by + iy0 * wt.sy) * lw + (wl.bx + (int)ix * wt.sx));
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
            for (int j = 0; j < NW; ++j) wn[j] = wvd_load_u32(ubase + 4 * j, lo0);
#pragma unroll 2
            for (int r = 0; r < PH_; ++r) {   // one patch row per iteration, the next row's loads in flight
                unsigned int w4[NW];
#pragma unroll
                for (int j = 0; j < NW; ++j) w4[j] = wn[j];
                const uint8_t* nsrc = ubase + (size_t)(r + 1 < PH_ ? r + 1 : r) * lw;
#pragma unroll
                for (int j = 0; j < NW; ++j) wn[j] = wvd_load_u32(nsrc + 4 * j, lo0);
                wvd_hist_row<NW, true>(w4, blkH4, laneOff32, inc);
            }
        }
        WVD_T(pb0);
#ifdef FD_WVB_PROF
        pAcc[0] += pb0 - pa;
#endif
        unsigned int survBits = 0;   // bit s: the lane's window s of this column goes to the exact cascade
#pragma unroll 1
        for (int step = 0; step < K; ++step) {
        const bool active = step < rows;
        if (__ballot(active) == 0) break;
        WVD_T(ps);
        if (step > 0 && active) {   // ---- 1'. slide the histogram down by one window: rows leave at the top, rows enter at the bottom
            const uint8_t* out0 = ubase + (size_t)(step - 1) * rowStep;
            for (int q = 0; q < wt.sy; ++q) {
                unsigned int wo[NW], wi[NW];
#pragma unroll
                for (int j = 0; j < NW; ++j) { wo[j] = wvd_load_u32(out0 + (size_t)q * lw + 4 * j, lo0); wi[j] = wvd_load_u32(out0 + (size_t)(q + PH_) * lw + 4 * j, lo0); }
                wvd_hist_row<NW, false>(wo, blkH4, laneOff32, inc);
                wvd_hist_row<NW, true>(wi, blkH4, laneOff32, inc);
            }
        }
        // the window this lane evaluates now (a lane that has run out of windows repeats its last one, unseen)
        const unsigned int lo = lo0 + (unsigned int)(active ? step : (rows > 0 ? rows - 1 : 0)) * rowStep;
        wave_sync();
        WVD_T(pb);
        // ---- 2. the fp32 cdf, the LUT, and the exact integer sum / sum of squares of the equalised patch
        unsigned int sumx, sumxx;
        wvd_cdf_lut<PW_ * PH_>(cntLds, lutLds, dv.stretch, sumx, sumxx);
        wave_sync();
        WVD_T(pc);
        // ---- 3. equalise, exact dot products on the matrix pipe.  k-step ks takes the dwords 8 ks .. 8 ks + 7 of the row-major patch
        //         (dword q = row q / NW, columns 4 (q % NW) ..): the loop is unrolled completely, so every row / column is a constant.
        //         acc[M][N]: digit tile M (rows f + 16 j', digits 2 M + j') x window tile N (windows 32 N + (lane & 31))
        wvd_v16i acc00 = {}, acc01 = {}, acc10 = {}, acc11 = {};
        {
            constexpr int PF = 1;   // k-steps the pixel dwords are requested ahead of their use (2: measured 4 % slower, the registers cost more)
            unsigned int nxq[PF][8];
            wvd_v4i an0, an1;
            const char* Bb = reinterpret_cast<const char*>(dv.B);   // uniform base + lane * 16
            unsigned int lane16 = (unsigned int)lane * 16u;
            asm volatile("" : "+v"(lane16));   // stays an offset register (hoisted out of the loops as 2 KS address pairs, the fragment addresses spill)
            auto fetchPx = [&](int ks) {   // pixel dwords of k-step ks
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int q = 8 * ks + j;
                    if (q < D4) nxq[ks % PF][j] = wvd_load_u32(ubase + (size_t)(q / NW) * lw + 4 * (q % NW), lo);
                }
            };
            auto fetch = [&](int ks) {   // digit fragments of k-step ks
                an0 = *reinterpret_cast<const wvd_v4i*>(Bb + (size_t)(ks * 2 + 0) * 1024 + lane16);
                an1 = *reinterpret_cast<const wvd_v4i*>(Bb + (size_t)(ks * 2 + 1) * 1024 + lane16);
            };
#pragma unroll
            for (int i = 0; i < PF; ++i)
                if (i < KS) fetchPx(i);
            fetch(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const wvd_v4i a0 = an0, a1 = an1;
                unsigned int cur[8], pk[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) cur[j] = nxq[ks % PF][j];
                if (ks + PF < KS) fetchPx(ks + PF);
                if (ks + 1 < KS) fetch(ks + 1);
                wvd_equalise<4>(cur, pk, lutWord, blkL4);
                if (8 * ks + 4 < D4) wvd_equalise<4>(cur + 4, pk + 4, lutWord, blkL4);
                else pk[4] = pk[5] = pk[6] = pk[7] = 0;   // k-slots past the patch: zero pixels against zero digits
                // this lane's 32 bytes -> the k-half each N-tile wants from it: lanes 0..31 give bytes 0..15 of windows 0..31 (tile 0) /
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
            // exact (explicit fma: -ffp-contract=off): |digit sums| < 2^24, every intermediate is an integer below 2^53
            const double v0 = __builtin_fma(256.0, __builtin_fma(65536.0, (double)acc10[8 + i], (double)acc00[8 + i]), __builtin_fma(65536.0, (double)acc10[i], (double)acc00[i]));
            const double v1 = __builtin_fma(256.0, __builtin_fma(65536.0, (double)acc11[8 + i], (double)acc01[8 + i]), __builtin_fma(65536.0, (double)acc11[i], (double)acc01[i]));
            uint2 u0 = __builtin_bit_cast(uint2, v0), u1 = __builtin_bit_cast(uint2, v1);
            wvd_swap32(u0.x, u1.x);
            wvd_swap32(u0.y, u1.y);
            xq[0][i] = __builtin_bit_cast(double, u0);
            xq[1][i] = __builtin_bit_cast(double, u1);
        }
        // ---- 5. the first L cascade levels of this lane's window with error bounds.  Branch-free per window (selects became plain
        //         arithmetic: an exponent below the float range gives K = 0 within the absolute slack, one above it makes the sums
        //         inf / NaN, which never compare below a threshold); levels come in pairs behind one uniform test of L, the level
        //         constants of a pair are scalar loads the scheduler can issue ahead of the arithmetic.
        bool undecided = active;
        {
            typedef float wvd_v2f __attribute__((ext_vector_type(2)));
            wvd_v2f KK[WVD_L];   // {K_p, bound of its error}: the level sum and its bound run as ONE v_pk_fma_f32 per term
            // the reference's fp32 sum of squares: row totals (exact ints) added in fp32, so it IS the integer below 2^24
            const float sxx = (float)sumxx;
            // |norm - reference norm|: 2 * quantisation error of xp, the fp32 sum of squares above 2^24, slack for the fp64 roundings on
            // both sides (the reference's chain and the regrouped one below: < 1e-5)
            const double dn = dv.scale * (double)sumx + (sumxx >= (1u << 24) ? (double)dv.sxxSlack : 0.0) + 1e-4;
            const float relDn = (float)(-(double)dv.negBasis * dn) * 1.0001f;
            // relative error of K: exponent error (quantisation; float cast of the exponent and 2^x: 1.69e-7 |log2 K| + 6e-7), the final
            // rounding, and -- folded in here -- the fp32 summation-order term (4k + 16) 2^-24 <= 4.6e-6 of the level sums
            const float rho0 = (relDn + 6.0e-7f) * 1.01f + 4.6e-6f;
            const double e0 = dv.nb2 * (double)sxx;   // log2 K = nb2 (sxx - 2 xp + pp) = e0 + cA[k] + mXq (x' . Q)
            const wvd_v2f RE0 = {dv.negBias, fabsf(dv.negBias) * 4.6e-6f + 1e-37f};
            const __attribute__((address_space(4))) wvd_v2f* W2 = (const __attribute__((address_space(4))) wvd_v2f*)&C.w2[0][0][0];
            unsigned long long und = __ballot(undecided);
            // NL levels as ONE basic block (no test of L between levels: the scheduler issues the scalar loads of the level constants
            // ahead of the arithmetic; with a test per level pair every pair waited for its own loads: 132 -> 127 us per headline launch).
            // Levels L .. NL - 1 are padding (thr = -inf, weights 0): evaluated, never rejecting.
            auto levels = [&](auto nl) {
                constexpr int NL = decltype(nl)::value;
#pragma unroll
                for (int k = 0; k < NL; ++k) {
                    const double ex = __builtin_fma(xq[(k >> 2) & 1][(k & 3) + 4 * (k >> 3)], dv.mXq, e0 + C.cA[k]);
                    const float lg = (float)ex;
                    const float Kraw = __builtin_amdgcn_exp2f(lg);   // <= 2^-115 where the old form tested arg < -80: inside the 3e-35 below
                    const float rho = __builtin_fmaf(fabsf(lg), 1.69e-7f, rho0);
                    KK[k] = wvd_v2f{Kraw, __builtin_fmaf(Kraw, rho, 3e-35f)};
                    wvd_v2f RE = RE0;
#pragma unroll
                    for (int p = 0; p <= k; ++p) RE = __builtin_elementwise_fma(W2[k * WVD_L + p], KK[p], RE);   // R += w K_p, E += |w| dK_p
                    // the reference leaves at the first level with res < thr, and res_ref <= R + E
                    und &= ~__ballot(RE.x + RE.y < C.thr[k]);
                }
            };
            if (L > 14) levels(std::integral_constant<int, 16>());
            else if (L > 12) levels(std::integral_constant<int, 14>());
            else if (L > 8) levels(std::integral_constant<int, 12>());
            else levels(std::integral_constant<int, 8>());
            undecided = (und >> lane) & 1ull;
        }
        survBits |= undecided ? 1u << step : 0u;
#ifdef FD_WVB_PROF
        {
            const unsigned long long pe = __builtin_amdgcn_s_memtime();
            pAcc[0] += pb - ps; pAcc[1] += pc - pb; pAcc[2] += pd - pc; pAcc[3] += pe - pd; ++pTiles;
        }
#endif
        }   // windows of the column
        // ---- 6. survivors -> queue of the exact cascade: ONE returning atomic per wavefront and column walk (it was one per window step:
        // with a model that lets a quarter of the windows through, 48 K of them per 64-frame launch and each waited for on the spot --
        // the launch took 231 instead of 126 us)
        if (__ballot(survBits != 0)) {
            unsigned int total = 0;
            for (int st = 0; st < K; ++st) total += (unsigned int)__popcll(__ballot((survBits >> st) & 1u));
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(dv.qcount, total);
            base = __builtin_amdgcn_readfirstlane(base);
            for (int st = 0; st < K; ++st) {
                const bool mine = (survBits >> st) & 1u;
                const unsigned long long mask = __ballot(mine);
                if (mine) dv.q[base + __popcll(mask & ((1ull << lane) - 1ull))] = wid0 + (int64_t)st * wl.nx;
                base += (unsigned int)__popcll(mask);
            }
        }
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
    const WvbXcd X((int)((n + 63u) >> 6));   // tile -> XCD like the kernels that read this state (wvm_stageb.hpp)
    for (int lt = X.wg * 4 + wave; lt < X.ntl; lt += X.nwg * 4) {
        const unsigned int tile = (unsigned int)X.tile(lt);
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
            wvd_load_row<NW>(src, wn);
#pragma unroll 2
            for (int r = 0; r < PH_; ++r) {
                unsigned int w4[NW];
#pragma unroll
                for (int j = 0; j < NW; ++j) w4[j] = wn[j];
                wvd_load_row<NW>(src + (size_t)(r + 1 < PH_ ? r + 1 : r) * lw, wn);
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
            // G rows (G PW bytes = whole 16-byte slots, 16-byte aligned: the state's row stride and G PW are multiples of 16) are packed
            // in registers and stored as uint4: every lane writes its own window's row, so a store instruction is 64 scattered
            // segments whatever its width -- a quarter of the instructions (round 4 stored dword by dword: 100 per 20 x 20 window)
            constexpr int G = (PW_ % 16 == 0) ? 1 : ((PW_ % 8 == 0) ? 2 : 4);   // rows per group: the fewest whose bytes are whole 16-byte slots
            static_assert(PH_ % G == 0 && (G * PW_) % 16 == 0, "row groups of the state stores");
            unsigned int wn[NW];
            wvd_load_row<NW>(src, wn);
#pragma unroll 1
            for (int r0 = 0; r0 < PH_; r0 += G) {
                unsigned int grp[G * NW];
#pragma unroll
                for (int rr = 0; rr < G; ++rr) {
                    const int r = r0 + rr;
                    unsigned int w4[NW], pk[NW];
#pragma unroll
                    for (int j = 0; j < NW; ++j) w4[j] = wn[j];
                    wvd_load_row<NW>(src + (size_t)(r + 1 < PH_ ? r + 1 : r) * lw, wn);
                    if constexpr (NW <= 5) wvd_equalise<NW>(w4, pk, lutWord, blkL4);
                    else { wvd_equalise<NW / 2>(w4, pk, lutWord, blkL4); wvd_equalise<NW / 2>(w4 + NW / 2, pk + NW / 2, lutWord, blkL4); }
                    unsigned int rowsq = 0;
#pragma unroll
                    for (int j = 0; j < NW; ++j) {
                        const unsigned int x4 = pk[j] ^ 0x80808080u;
                        rowsq = __builtin_amdgcn_udot4(x4, x4, rowsq, false);
                        grp[rr * NW + j] = pk[j];
                    }
                    sxx = r == 0 ? (float)rowsq : sxx + (float)rowsq;
                }
                if (valid) {
                    uint4* dst = reinterpret_cast<uint4*>(xr + r0 * PW_);
#pragma unroll
                    for (int q4 = 0; q4 < G * NW / 4; ++q4) dst[q4] = make_uint4(grp[4 * q4], grp[4 * q4 + 1], grp[4 * q4 + 2], grp[4 * q4 + 3]);
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


}  // namespace
