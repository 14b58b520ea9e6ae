// featuredetection_amd/csrc/wvm_dense.hpp -- dense pre-filter of the WVM cascade on the matrix pipe (included by wvm.hip only).
//
// WvmClassifier.cpp:191-346 evaluates filter k through rectangle sums on an integral image; in exact arithmetic its dot product
// is x . r_k with the dense residual image r_k[y][x] = val_k[0] + sum_{v>=1} (val_k[v] - val_k[0]) * #{rects of v covering (x, y)}
// (SURVEY.md App. A.3), and for k < numFiltersPerLevel nothing is carried over from earlier levels (u_kernel_eval[n] is still 0).
// So the first L <= min(16, numPer) kernel values of ALL windows are one [windows x d] . [d x L] contraction:
//
//   k_wvm_prefilter: lane == window (64 consecutive windows of a layer per wavefront).
//     1. HistEq64 (HistEq64Filter.cpp:32-125) lane-serial: private 64-bin histogram column in LDS (u16 counters, two lanes share a
//        dword: ds_add_u32 of 1 << 16*(lane&1)), the fp32 cdf as a plain 63-step chain in the lane's registers (same operation
//        order as the reference, 64 windows per instruction instead of the DPP chain's one), LUT written back over the counters.
//     2. the equalised pixels go to LDS as signed bytes (x - 128), 32 per k-step, and are multiplied on v_mfma_i32_32x32x32_i8
//        against the residual images quantised to 32-bit integers Q = round(r * 2^s) and split into four balanced base-256 digits:
//        integer arithmetic, so x . Q is EXACT; |x . r - 2^-s x . Q| <= 2^-(s+1) * sum(x) is the only approximation.
//        The fp32 sum of squares is accumulated in the reference's own order (IImg.cpp:33-47), i.e. it is the reference's value.
//     3. per window: norm, K = exp(-basis * norm), res_k = -bias + sum_p w[k][p] K_p with a rigorous error bound eps_k on
//        |res_k - reference res_k| (quantisation, fast fp32 exp, fp32 summation order).  A window with res_k + eps_k < thr_k at ANY
//        level k < L is rejected by the reference at some level <= k, so it cannot be a WVM positive and is dropped here.
//        Everything else is appended to a queue and runs the exact cascade kernels (k_wvm_cascade4/2 -> k_wvm_deep4) unchanged,
//        so positives, their levels and their fp32 outputs stay bit-identical to the rectangle-sum formulation.
//   Only used when no per-window outputs are requested (fd_detect_wvm with all_level / all_score takes the exact path for every window).
#pragma once

constexpr int WVD_L = 16;          // filters evaluated densely (columns: 16 filters x 4 digits = 2 N-tiles of 32)
constexpr int WVD_XSTRIDE = 48;    // bytes per window row of the LDS pixel chunk: 32 pixels + 16 pad (16 x odd: conflict-free b128)
constexpr int WVD_TSTRIDE = 17;    // doubles per window row of the transposed dot products (odd: conflict-free b64)

typedef int wvd_v4i __attribute__((ext_vector_type(4)));
typedef int wvd_v16i __attribute__((ext_vector_type(16)));

struct WvdLayer {
    int32_t bx, by, nx, lw;
    uint32_t off, magic;
    int32_t nwin, tileFirst;
    int64_t first;
};
struct WvdTable {
    int32_t n, sx, sy, ntiles;
    WvdLayer l[WVM_MAX_LAYERS];
};

// model constants of the dense stage (device memory, read through scalar loads)
struct WvdConst {
    int32_t L, pad0;
    double scale;              // 2^-s
    double c128[WVD_L];        // 128 * sum_i Q_k[i]
    double pp[WVD_L];
    float thr[WVD_L];
    float w[WVD_L][WVD_L];     // hkWeights[k][p], p <= k
    float negBasis, negBias, stretch, pad1;
    double dnScale;            // 2^-s: error of the quantised dot product per unit of sum(x) is 2^-(s+1); norm uses 2 * xp
};

struct WvdDev {
    const wvd_v4i* B;          // [KS][2 N-tiles][64 lanes] 16 signed digit bytes each
    const WvdConst* c;
    int64_t* q;                // windows that pass
    unsigned int* qcount;
};

namespace {

template <int PW_, int PH_>
struct __attribute__((aligned(16))) WvdLds {
    union {
        unsigned short hist[64][64];                 // [bin][lane]: counters, then the lane's LUT
        double tr[64 * WVD_TSTRIDE];                 // [window][filter] exact dot products (after the MFMA loop)
    };
    unsigned char x[64 * WVD_XSTRIDE];               // current k-step: 32 equalised pixels of every window, as x - 128
};

__device__ __forceinline__ unsigned int wvd_load_u32(const uint8_t* p) {
    unsigned int v;
    __builtin_memcpy(&v, p, 4);   // unaligned global_load_dword
    return v;
}

// 64 consecutive windows of one layer per wavefront; 4 wavefronts per workgroup, persistent grid over the tiles.
template <int PW_, int PH_>
__global__ __launch_bounds__(256) void k_wvm_prefilter(const uint8_t* __restrict__ arena, WvdTable wt, WvdDev dv) {
    static_assert(PW_ % 4 == 0 && PW_ >= 4 && PW_ <= 32, "rows are read as dwords");
    constexpr int d = PW_ * PH_;
    constexpr int KS = (d + 31) / 32;
    __shared__ WvdLds<PW_, PH_> lds[4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WvdLds<PW_, PH_>& S = lds[wave];
    const WvdConst& C = *dv.c;
    const int L = C.L;
    const int ntiles = wt.ntiles;
    int li = 0;
    unsigned char* histBase = reinterpret_cast<unsigned char*>(&S.hist[0][0]);
    const unsigned int laneOff16 = (unsigned int)lane * 2u;             // byte offset of this lane's u16 slot inside a bin row
    const unsigned int laneOff32 = (unsigned int)(lane >> 1) * 4u;      // dword holding it
    const unsigned int inc = 1u << (16 * (lane & 1));

    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        while (li + 1 < wt.n && tile >= wt.l[li + 1].tileFirst) ++li;   // tiles ascend per wavefront
        const WvdLayer& wl = wt.l[li];
        const int local0 = (tile - wl.tileFirst) * 64 + lane;
        const bool valid = local0 < wl.nwin;
        const unsigned int local = (unsigned int)(valid ? local0 : wl.nwin - 1);
        unsigned int iy = __umulhi(local, wl.magic);   // floor(local / nx) or one less
        unsigned int ix = local - iy * (unsigned int)wl.nx;
        if (ix >= (unsigned int)wl.nx) { ix -= wl.nx; ++iy; }
        const int lw = wl.lw;
        const uint8_t* src = arena + wl.off + (size_t)(wl.by + (int)iy * wt.sy) * lw + (wl.bx + (int)ix * wt.sx);
        const int64_t wid = wl.first + local;

        // ---- 1. histogram: 64 bins x 64 lanes of u16 counters
        {
            uint4* z = reinterpret_cast<uint4*>(histBase);
#pragma unroll
            for (int i = 0; i < 8; ++i) z[i * 64 + lane] = make_uint4(0, 0, 0, 0);
        }
        wave_sync();
#pragma unroll
        for (int r = 0; r < PH_; ++r) {
#pragma unroll
            for (int j = 0; j < PW_ / 4; ++j) {
                const unsigned int w4 = wvd_load_u32(src + (size_t)r * lw + 4 * j);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const unsigned int bin = __builtin_amdgcn_ubfe(w4, 8 * b + 2, 6);
                    atomicAdd(reinterpret_cast<unsigned int*>(histBase + (bin << 7) + laneOff32), inc);
                }
            }
        }
        wave_sync();
        // ---- 2. the fp32 cdf in the reference's order (cdf[0] = pdf[0]; cdf[b] = cdf[b-1] + pdf[b]) and the LUT
        {
            float cdf = 0.f;
#pragma unroll
            for (int b = 0; b < 64; ++b) {
                unsigned short* slot = reinterpret_cast<unsigned short*>(histBase + (b << 7) + laneOff16);
                const float pdf = (float)(unsigned int)*slot * C.stretch;
                cdf = b == 0 ? pdf : cdf + pdf;
                // (uchar)floor((double)cdf + 0.5): cdf < 2^9 has at most 24 significant bits, so cdf + 0.5 is exact in double;
                // floor(cdf) + (frac >= 0.5) is the same value without leaving fp32
                const float fl = floorf(cdf);
                const float up = (cdf - fl >= 0.5f) ? fl + 1.0f : fl;
                *slot = (unsigned short)((unsigned int)up & 255u);
            }
        }
        wave_sync();
        // ---- 3. equalise, sum of squares in the reference's order, exact dot products on the matrix pipe
        wvd_v16i acc00 = {}, acc01 = {}, acc10 = {}, acc11 = {};
        float sxx = 0.f;
        unsigned int rowq = 0, sumx = 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const wvd_v4i b0 = dv.B[(ks * 2 + 0) * 64 + lane];
            const wvd_v4i b1 = dv.B[(ks * 2 + 1) * 64 + lane];
#pragma unroll
            for (int qd = 0; qd < 2; ++qd) {
                unsigned int pk[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int p0 = ks * 32 + qd * 16 + g * 4;   // first pixel of this dword (compile time)
                    unsigned int packed = 0;
                    if (p0 < d) {
                        const int r = p0 / PW_, c0 = p0 % PW_;
                        const unsigned int w4 = wvd_load_u32(src + (size_t)r * lw + c0);
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const unsigned int bin = __builtin_amdgcn_ubfe(w4, 8 * b + 2, 6);
                            const unsigned int e = *reinterpret_cast<unsigned short*>(histBase + (bin << 7) + laneOff16);
                            rowq += e * e;
                            packed |= e << (8 * b);
                        }
                        sumx += __builtin_amdgcn_sad_u8(packed, 0u, 0u);
                        if (c0 + 4 == PW_) {   // end of patch row r: IImg.cpp:33-47 adds the row's exact int sum in fp32
                            sxx = r == 0 ? (float)rowq : sxx + (float)rowq;
                            rowq = 0;
                        }
                        packed ^= 0x80808080u;   // x - 128 as int8
                    }
                    pk[g] = packed;
                }
                *reinterpret_cast<uint4*>(&S.x[lane * WVD_XSTRIDE + qd * 16]) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
            wave_sync();
            const wvd_v4i a0 = *reinterpret_cast<const wvd_v4i*>(&S.x[(lane & 31) * WVD_XSTRIDE + (lane >> 5) * 16]);
            const wvd_v4i a1 = *reinterpret_cast<const wvd_v4i*>(&S.x[(32 + (lane & 31)) * WVD_XSTRIDE + (lane >> 5) * 16]);
            wave_sync();
            acc00 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, acc11, 0, 0, 0);
        }
        // ---- 4. digits -> exact integer dot products, transposed to lane == window.  Column g = f + 16 j of N-tile g / 32 holds
        //         digit j of filter f: this lane (column lane & 31) has digit j0 = (lane >> 4) & 1 in tile 0 and digit j0 + 2 in tile 1
        wave_sync();   // the LUT is dead: the region becomes the transpose buffer
        {
            const int f = lane & 15;
            const bool lowDigit = (lane & 16) == 0;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int s0 = mt == 0 ? acc00[rg] : acc10[rg];
                    const int s2 = mt == 0 ? acc01[rg] : acc11[rg];
                    const double part = (double)s0 + 65536.0 * (double)s2;          // exact: |s| < 2^24
                    const double other = __shfl_xor(part, 16);
                    const int row = (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5);
                    if (lowDigit) S.tr[(mt * 32 + row) * WVD_TSTRIDE + f] = part + 256.0 * other;   // exact: < 2^53
                }
            }
        }
        wave_sync();
        // ---- 5. the first L cascade levels of this lane's window with error bounds
        bool undecided = valid;
        {
            float Kv[WVD_L], Ke[WVD_L];
            const double dn = C.dnScale * (double)sumx + 1e-5;   // |norm - reference norm| (2 * quantisation error + fp64 slack)
            const float relDn = (float)(-(double)C.negBasis * dn) * 1.0001f;
#pragma unroll
            for (int k = 0; k < WVD_L; ++k) {
                if (k < L) {
                    const double xp = (S.tr[lane * WVD_TSTRIDE + k] + C.c128[k]) * C.scale;
                    double norm = (double)sxx;
                    norm = norm - 2 * xp;
                    norm = norm + C.pp[k];
                    const float arg = (float)((double)C.negBasis * norm);
                    float Kk, Kerr;
                    if (arg < -80.0f) { Kk = 0.f; Kerr = 2e-35f; }           // true K <= e^-80 (1 + tiny)
                    else if (arg > 80.0f) { Kk = 0.f; Kerr = 3.0e38f; }       // cannot happen for a sane model: never reject
                    else {
                        Kk = __expf(arg);
                        // relative: exponent error (quantisation, float cast of the argument, x*log2e, 2^x) + final rounding
                        const float rho = relDn + fabsf(arg) * 2.4e-7f + 6.0e-7f;
                        Kerr = Kk * rho * 1.01f + 1e-37f;
                    }
                    Kv[k] = Kk;
                    Ke[k] = Kerr;
                    float R = C.negBias, A = fabsf(C.negBias), E = 0.f;
#pragma unroll
                    for (int p = 0; p <= k; ++p) {
                        const float w = C.w[k][p];
                        R = fmaf(w, Kv[p], R);
                        A = fmaf(fabsf(w), Kv[p], A);
                        E = fmaf(fabsf(w), Ke[p], E);
                    }
                    const float eps = E + A * ((float)(4 * k + 16) * 5.97e-8f) + 1e-37f;
                    // the reference leaves at the first level with res < thr; res_ref <= R + eps
                    if (undecided && (R + eps < C.thr[k])) undecided = false;
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
    }
}

}  // namespace
