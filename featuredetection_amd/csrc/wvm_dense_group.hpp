// featuredetection_amd/csrc/wvm_dense_group.hpp -- the dense pre-filter for a GROUP of detectors that scan the same windows
// (included by wvm.hip only, behind wvm_dense.hpp).
//
// ffpDetectApp.cpp:557-600 runs its detectors one after the other over the same frame; seven of the fifteen *.cfg detectors share one
// pyramid and a 24 x 24 patch, two more the 16 x 24 ear patch, two the 20 x 20 profile-face patch.  HistEq64 (HistEq64Filter.cpp:32-125)
// depends on the window only, so for such a group everything up to the equalised pixels -- histogram slide, cdf, LUT, the LUT gather:
// 86 % of k_wvm_prefilter's instructions -- is the same work done once per detector.  k_wvm_prefilter_group does it ONCE per window:
//   1. / 2. as k_wvm_prefilter (lane == window, private histogram column that slides down, register cdf chain, LUT in LDS);
//   3a. the equalised patch of the lane's window is gathered once and kept in REGISTERS as the finished B operands of all k-steps
//       (KS x 8 dwords per lane: 144 for 24 x 24);
//   3b. per detector of the group: the exact digit contraction against that detector's table (the only per-detector memory traffic:
//       its operand fragments, 2 KB per k-step from L2), the fold to doubles, the first L cascade levels with their error bounds --
//       the same arithmetic in the same order as k_wvm_prefilter, so a detector's queue holds the same windows;
//   6. survivors per detector (bits in LDS while the lane walks down its column) -> that detector's stage-B queue.
// Round 2's k_wvm_prefilter_multi shared only the histogram / cdf / LUT (13 % less kernel time) and was dropped; this one shares the gather
// as well, which is what the time goes into.
#pragma once

constexpr int WVD_GMAX = 8;   // detectors per group launch
#ifndef WVD_GRP_RD
#define WVD_GRP_RD 2
#endif

struct WvdMember {
    const wvd_v4i* B;          // the detector's digit table
    const WvdConst* c;         // its level constants and scalars (WvdConst::sc)
    int64_t* q;                // its stage-B queue
    unsigned int* qcount;
};
struct WvdGroup {
    int32_t n, pad;
    WvdMember m[WVD_GMAX];
};

namespace {

template <int PW_, int PH_>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_wvm_prefilter_group(const uint8_t* __restrict__ arena, WvdTable wt, WvdGroup ga) {
    static_assert(PW_ % 4 == 0 && PW_ >= 16 && PW_ <= 32, "rows are read as dwords");
    constexpr int NW = PW_ / 4;
    constexpr int D4 = NW * PH_;
    static_assert(D4 % 4 == 0, "the equalise blocks take 4 dwords");
    constexpr int KS = (D4 + 7) / 8;
    static_assert(KS <= 18, "the equalised patch stays in registers: KS x 8 dwords per lane");
    __shared__ __attribute__((aligned(16384))) WvdLds S;
    __shared__ unsigned int survL[4][WVD_GMAX][64];   // bit s of [wave][detector][lane]: the lane's window s goes to that detector's exact cascade
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K = wt.K;
    const int ND = ga.n;
    int li = 0;
    unsigned char* histPtr = reinterpret_cast<unsigned char*>(&S.hist[wave >> 1][0][wave & 1][0]);
    const unsigned int histLds = (unsigned int)(uintptr_t)(wvd_lds_u16*)histPtr;
    const unsigned int blkH4 = ((histLds >> 8) & 0xC0u) * 0x01010101u;
    const unsigned int laneOff32 = (histLds & ~0xFF00u) + (unsigned int)(lane & 31) * 4u;
    const unsigned int cntLds = histLds + (unsigned int)(lane & 31) * 4u + (unsigned int)(lane >> 5) * 2u;
    const unsigned int inc = 1u << (16 * (lane >> 5));
    const unsigned int lutLds = (unsigned int)(uintptr_t)(wvd_lds_u8*)&S.lut[0][lane][wave];
    const unsigned int blkL4 = ((lutLds >> 8) & 0xC0u) * 0x01010101u;
    const unsigned int lutWord = lutLds & ~0xFF00u;
    typedef __attribute__((address_space(4))) WvdConst wvd_cconst;
    const float stretch = ((const wvd_cconst*)(uintptr_t)ga.m[0].c)->sc.stretch;   // 255 / (PW_ PH_): the same for every member

    // Operand fragments (2 x 16 bytes per lane and k-step) are requested RD k-steps ahead of their MFMAs and wrap into the next detector's
    // table: with nothing but MFMAs between the requests an L2 round trip is ~4 k-steps of one wavefront; the SIMD's second wavefront
    // covers part of it.  (The equalised patch + the accumulators are 208 of the 256 registers two wavefronts per SIMD leave.)
    constexpr int RD = KS <= 13 ? 4 : WVD_GRP_RD;   // (16 x 24 and 20 x 20 patches have the registers for four k-steps: 396 against 415, 106 against 113 us per 1080p frame)
    static_assert(RD == 1 || RD == 2 || RD == 4, "the digit tables are padded with zero k-steps to a multiple of four");
    constexpr int KSR = (KS + RD - 1) / RD * RD;   // k-steps contracted per detector: whole rounds of the ring (the pad steps add 0 x anything)
    wvd_v4i frag[RD][2];
    unsigned int lane16 = (unsigned int)lane * 16u;
    asm volatile("" : "+v"(lane16));
    auto loadFrag = [&](const wvd_v4i* B, int ks, int mt) {
        return *reinterpret_cast<const wvd_v4i*>(reinterpret_cast<const char*>(B) + (size_t)(ks * 2 + mt) * 1024 + lane16);
    };
#pragma unroll
    for (int ks = 0; ks < RD; ++ks) { frag[ks][0] = loadFrag(ga.m[0].B, ks, 0); frag[ks][1] = loadFrag(ga.m[0].B, ks, 1); }

    int lastImg = -1;
    const bool byXcd = wt.nimg >= 8 && (gridDim.x & 7u) == 0;
    const int xcd = blockIdx.x & 7, vStride = byXcd ? (int)(gridDim.x >> 3) * 4 : (int)gridDim.x * 4;
    const int vEnd = byXcd ? ((wt.nimg - xcd + 7) >> 3) * wt.sTilesPerImage : wt.sTilesPerImage * wt.nimg;
    for (int v = (byXcd ? (int)(blockIdx.x >> 3) : (int)blockIdx.x) * 4 + wave; v < vEnd; v += vStride) {
        const int fi = wt.nimg > 1 ? v / wt.sTilesPerImage : 0;
        const int img = byXcd ? xcd + 8 * fi : fi;
        const int tile = v - fi * wt.sTilesPerImage;
        if (img != lastImg) { li = 0; lastImg = img; }
        while (li + 1 < wt.n && tile >= wt.l[li + 1].sTileFirst) ++li;
        const WvdLayer& wl = wt.l[li];
        const int ntask = wl.nx * wl.G;
        const int task0 = (tile - wl.sTileFirst) * 64 + lane;
        const unsigned int task = (unsigned int)(task0 < ntask ? task0 : ntask - 1);
        unsigned int g = __umulhi(task, wl.magic);
        unsigned int ix = task - g * (unsigned int)wl.nx;
        if (ix >= (unsigned int)wl.nx) { ix -= wl.nx; ++g; }
        const int iy0 = (int)g * K;
        const int rows = task0 < ntask ? min(K, wl.ny - iy0) : 0;
        const int lw = wl.lw;
        const unsigned int rowStep = (unsigned int)(wt.sy * lw);
        const uint8_t* ubase = arena + (size_t)img * wt.imageStride + wl.off;
        const unsigned int lo0 = (unsigned int)((wl.by + iy0 * wt.sy) * lw + (wl.bx + (int)ix * wt.sx));
        const int64_t wid0 = (int64_t)img * wt.perImage + wl.first + (int64_t)iy0 * wl.nx + ix;

        // ---- 1. histogram of the lane's first window
        {
            unsigned char* z = histPtr + (lane >> 3) * 256 + (lane & 7) * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(z + i * 2048) = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int d = 0; d < WVD_GMAX; ++d) survL[wave][d][lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        {
            unsigned int wn[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j) wn[j] = wvd_load_u32(ubase + 4 * j, lo0);
#pragma unroll 2
            for (int r = 0; r < PH_; ++r) {
                unsigned int w4[NW];
#pragma unroll
                for (int j = 0; j < NW; ++j) w4[j] = wn[j];
                const uint8_t* nsrc = ubase + (size_t)(r + 1 < PH_ ? r + 1 : r) * lw;
#pragma unroll
                for (int j = 0; j < NW; ++j) wn[j] = wvd_load_u32(nsrc + 4 * j, lo0);
                wvd_hist_row<NW, true>(w4, blkH4, laneOff32, inc);
            }
        }
#pragma unroll 1
        for (int step = 0; step < K; ++step) {
            const bool active = step < rows;
            if (__ballot(active) == 0) break;
            if (step > 0 && active) {   // ---- 1'. slide the histogram down by one window
                const uint8_t* out0 = ubase + (size_t)(step - 1) * rowStep;
                for (int q = 0; q < wt.sy; ++q) {
                    unsigned int wo[NW], wi[NW];
#pragma unroll
                    for (int j = 0; j < NW; ++j) { wo[j] = wvd_load_u32(out0 + (size_t)q * lw + 4 * j, lo0); wi[j] = wvd_load_u32(out0 + (size_t)(q + PH_) * lw + 4 * j, lo0); }
                    wvd_hist_row<NW, false>(wo, blkH4, laneOff32, inc);
                    wvd_hist_row<NW, true>(wi, blkH4, laneOff32, inc);
                }
            }
            const unsigned int lo = lo0 + (unsigned int)(active ? step : (rows > 0 ? rows - 1 : 0)) * rowStep;
            wave_sync();
            // ---- 2. cdf, LUT, exact integer sums of the equalised patch
            unsigned int sumx, sumxx;
            wvd_cdf_lut<PW_ * PH_>(cntLds, lutLds, stretch, sumx, sumxx);
            wave_sync();
            // ---- 3a. the equalised patch as the B operands of all k-steps (k_wvm_prefilter builds them k-step by k-step in front of its
            //          MFMAs; here they stay: every detector of the group contracts the same pixels)
            unsigned int pkAll[KS][8];
            {
                unsigned int nxq[8];
                auto fetchPx = [&](int ks) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int q = 8 * ks + j;
                        if (q < D4) nxq[j] = wvd_load_u32(ubase + (size_t)(q / NW) * lw + 4 * (q % NW), lo);
                    }
                };
                fetchPx(0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    unsigned int cur[8], pk[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) cur[j] = nxq[j];
                    if (ks + 1 < KS) fetchPx(ks + 1);
                    wvd_equalise<4>(cur, pk, lutWord, blkL4);
                    if (8 * ks + 4 < D4) wvd_equalise<4>(cur + 4, pk + 4, lutWord, blkL4);
                    else pk[4] = pk[5] = pk[6] = pk[7] = 0;
                    wvd_swap32(pk[0], pk[4]); wvd_swap32(pk[1], pk[5]); wvd_swap32(pk[2], pk[6]); wvd_swap32(pk[3], pk[7]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) pkAll[ks][j] = pk[j];
                }
            }
            const float sxx = (float)sumxx;
            // ---- 3b. per detector: contraction, fold, levels
#pragma unroll 1
            for (int d = 0; d < ND; ++d) {
                const WvdMember mb = ga.m[d];
                const wvd_v4i* Bnext = ga.m[d + 1 < ND ? d + 1 : 0].B;
                const wvd_cconst& C = *(const wvd_cconst*)(uintptr_t)mb.c;
                const int L = C.sc.L;
                const double mXq = C.sc.mXq, nb2 = C.sc.nb2, scale = C.sc.scale;
                const float negBasis = C.sc.negBasis, negBias = C.sc.negBias, sxxSlack = C.sc.sxxSlack;
                wvd_v16i acc00 = {}, acc01 = {}, acc10 = {}, acc11 = {};
                {
#pragma unroll
                    for (int ks = 0; ks < KSR; ++ks) {
                        const wvd_v4i a0 = frag[ks % RD][0], a1 = frag[ks % RD][1];
                        {   // k-step ks + RD of this detector, or the first ones of the next
                            const wvd_v4i* Bf = ks + RD < KSR ? mb.B : Bnext;
                            const int kf = ks + RD < KSR ? ks + RD : ks + RD - KSR;
                            frag[ks % RD][0] = loadFrag(Bf, kf, 0);
                            frag[ks % RD][1] = loadFrag(Bf, kf, 1);
                        }
                        const int kp = ks < KS ? ks : KS - 1;   // a pad step: zero digits against any pixels
                        const wvd_v4i b0 = {(int)pkAll[kp][0], (int)pkAll[kp][1], (int)pkAll[kp][2], (int)pkAll[kp][3]};
                        const wvd_v4i b1 = {(int)pkAll[kp][4], (int)pkAll[kp][5], (int)pkAll[kp][6], (int)pkAll[kp][7]};
                        acc00 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, acc00, 0, 0, 0);
                        acc01 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, acc01, 0, 0, 0);
                        acc10 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, acc10, 0, 0, 0);
                        acc11 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, acc11, 0, 0, 0);
                    }
                }
                // ---- 4. digits -> exact integer dot products, lane == window (k_wvm_prefilter's fold)
                double xq[2][8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const double v0 = __builtin_fma(256.0, __builtin_fma(65536.0, (double)acc10[8 + i], (double)acc00[8 + i]), __builtin_fma(65536.0, (double)acc10[i], (double)acc00[i]));
                    const double v1 = __builtin_fma(256.0, __builtin_fma(65536.0, (double)acc11[8 + i], (double)acc01[8 + i]), __builtin_fma(65536.0, (double)acc11[i], (double)acc01[i]));
                    uint2 u0 = __builtin_bit_cast(uint2, v0), u1 = __builtin_bit_cast(uint2, v1);
                    wvd_swap32(u0.x, u1.x);
                    wvd_swap32(u0.y, u1.y);
                    xq[0][i] = __builtin_bit_cast(double, u0);
                    xq[1][i] = __builtin_bit_cast(double, u1);
                }
                // ---- 5. the first L cascade levels with error bounds (k_wvm_prefilter's, with this detector's constants)
                bool undecided = active;
                {
                    typedef float wvd_v2f __attribute__((ext_vector_type(2)));
                    wvd_v2f KK[WVD_L];
                    const double dn = scale * (double)sumx + (sumxx >= (1u << 24) ? (double)sxxSlack : 0.0) + 1e-4;
                    const float relDn = (float)(-(double)negBasis * dn) * 1.0001f;
                    const float rho0 = (relDn + 6.0e-7f) * 1.01f + 4.6e-6f;
                    const double e0 = nb2 * (double)sxx;
                    const wvd_v2f RE0 = {negBias, fabsf(negBias) * 4.6e-6f + 1e-37f};
                    const __attribute__((address_space(4))) wvd_v2f* W2 = (const __attribute__((address_space(4))) wvd_v2f*)&C.w2[0][0][0];
                    unsigned long long und = __ballot(undecided);
                    auto levels = [&](auto nl) {
                        constexpr int NL = decltype(nl)::value;
#pragma unroll
                        for (int k = 0; k < NL; ++k) {
                            const double ex = __builtin_fma(xq[(k >> 2) & 1][(k & 3) + 4 * (k >> 3)], mXq, e0 + C.cA[k]);
                            const float lg = (float)ex;
                            const float Kraw = __builtin_amdgcn_exp2f(lg);
                            const float rho = __builtin_fmaf(fabsf(lg), 1.69e-7f, rho0);
                            KK[k] = wvd_v2f{Kraw, __builtin_fmaf(Kraw, rho, 3e-35f)};
                            wvd_v2f RE = RE0;
#pragma unroll
                            for (int p = 0; p <= k; ++p) RE = __builtin_elementwise_fma(W2[k * WVD_L + p], KK[p], RE);
                            und &= ~__ballot(RE.x + RE.y < C.thr[k]);
                        }
                    };
                    if (L > 14) levels(std::integral_constant<int, 16>());
                    else if (L > 12) levels(std::integral_constant<int, 14>());
                    else if (L > 8) levels(std::integral_constant<int, 12>());
                    else levels(std::integral_constant<int, 8>());
                    undecided = (und >> lane) & 1ull;
                }
                if (undecided) survL[wave][d][lane] |= 1u << step;   // the lane's own word
            }
        }   // windows of the column
        // ---- 6. survivors -> each detector's stage-B queue, one returning atomic per wavefront, column walk and detector
        wave_sync();
#pragma unroll 1
        for (int d = 0; d < ND; ++d) {
            const unsigned int survBits = survL[wave][d][lane];
            if (__ballot(survBits != 0)) {
                const WvdMember mb = ga.m[d];
                unsigned int total = 0;
                for (int st = 0; st < K; ++st) total += (unsigned int)__popcll(__ballot((survBits >> st) & 1u));
                unsigned int base = 0;
                if (lane == 0) base = atomicAdd(mb.qcount, total);
                base = __builtin_amdgcn_readfirstlane(base);
                for (int st = 0; st < K; ++st) {
                    const bool mine = (survBits >> st) & 1u;
                    const unsigned long long mask = __ballot(mine);
                    if (mine) mb.q[base + __popcll(mask & ((1ull << lane) - 1ull))] = wid0 + (int64_t)st * wl.nx;
                    base += (unsigned int)__popcll(mask);
                }
            }
        }
        wave_sync();
    }
}

}  // namespace
