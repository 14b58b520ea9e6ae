// featuredetection_amd/csrc/sdm.hip -- supervised descent landmark fitting on the GPU.
//
// superviseddescent::SdmLandmarkModelFitting::optimize (SdmLandmarkModel.hpp:199-256) for a batch of
// faces: per cascade step
//   k_sdm_prepare      dynamic face size, window half size (:212-229), integer patch origins and the
//                      zero-border quirk of DescriptorExtractor.hpp:156-178,
//   k_sdm_descriptors  one wavefront per (face, landmark): crop -> fp32 bilinear resize to 30x30
//                      (cv::resize semantics) -> VLFeat HOG (hog.c:596-721,858-1063).  Lane e owns one
//                      (cell, orientation) accumulator and walks the pixels in the reference's scan
//                      order, so descriptors are bit-identical to hog.c,
//   k_sdm_regress      delta = F * R[0:-1] + R[-1] (:241) as a batch x regressor contraction on the f64
//                      MFMA pipe (v_mfma_f64_16x16x4_f64): fp32 inputs are exact in f64 and the sum is
//                      rounded to fp32 once, like OpenCV's gemm (double accumulator), which keeps the
//                      next step's cvRound(landmark) decisions identical to the CPU path,
//   k_sdm_update       shape += delta * d (:243).
// MFMA-bound stage: 2*B*F*2L flop per step; HBM: R (F+1)*2L*4 B + descriptors.
#include "fd_internal.hpp"
#include <algorithm>
#include <cstring>
#include <memory>
#include <type_traits>

namespace {

constexpr int SDM_IMG = 48;          // max working image side in LDS
constexpr int SDM_MAX_CELLS = 12;    // per dimension
constexpr int SDM_MAX_ORI = 16;
// hog.c:630 tests the fp32 gradient magnitude as a double against 1e-10.  T = (float)1e-10 lies above 1e-10 and the float below it
// below (T is the nearest float: |T - 1e-10| <= ulp / 2), so (double)g > 1e-10 exactly when g >= T: one fp32 compare
constexpr float SDM_TINY = 1e-10f;
static_assert((double)SDM_TINY > 1e-10, "the fp32 form of the 1e-10 test needs (float)1e-10 > 1e-10");

struct DescParams {
    int32_t W, H;              // gray image size
    int32_t L;                 // landmarks per face
    int32_t adaptive;          // 1: resize patch to 30x30, cell 10, 9 orientations
    int32_t cellSize, nori, variant;
    int32_t iw, ih;            // working image size (30x30 when adaptive, side x side otherwise)
    int32_t hogW, hogH, dim, len;
    int64_t image_stride;      // bytes between consecutive face images
    int32_t ldsPerWave;        // bytes of LDS one wavefront needs (host-computed)
    float oX[SDM_MAX_ORI], oY[SDM_MAX_ORI];  // hog.c:195-204, computed on the host with libm like the reference
};

struct DescLds {   // per-wave views carved out of dynamic LDS
    float* img;                   // working image; SMALL: the gradient magnitudes overwrite it, later the features
    float* grad;
    float *wx1, *wx2;
    int* binx;
    float *hog, *norm, *feat;
    void* masks;                  // [2*nori][ih] words (32 bits SMALL, else 64): bit x set iff pixel (y, x) voted for that orientation
    float* wsel;                  // [max(hogW, hogH)][m]: interpolation weight of column / row p towards cell column / row c
    float2* oxy;                  // [SDM_MAX_ORI] orientation unit vectors (a lane picks three of them by index)
    double *fac, *hc;             // block factors [ncell][4], clamped undirected terms [ncell*nori][4] (reuse the mask region)
    unsigned long long* colmask;  // [hogW]: columns x contributing to cell column cx
};
constexpr int SDM_SMALL_ITERS = 16;   // "small" working images: up to 64 * 16 pixels and at most 32 columns (32-bit orientation masks)
#ifndef FD_SDM_REGGRAD
#define FD_SDM_REGGRAD 1
#endif
// 1: small working images turn into their gradient magnitudes in place, one block of 64 pixels behind the block being computed (3.6 KB
// of LDS less per wave: 18 instead of 12 waves per CU); 0: the magnitudes go to their own LDS block through the plain loop
constexpr bool SDM_REGGRAD = FD_SDM_REGGRAD != 0;
__host__ __device__ inline int align16i(int v) { return (v + 15) & ~15; }
__host__ __device__ inline bool desc_small(int iw, int ih) { return iw * ih <= 64 * SDM_SMALL_ITERS && iw <= 32; }
__host__ __device__ inline int desc_region_img(int iw, int ih, int ncell, int dim) {
    const int a = align16i(iw * ih * 4), b = align16i(ncell * dim * 4);
    return a > b ? a : b;
}
__host__ __device__ inline int desc_region_masks(int iw, int ih, int ncell, int nori) {
    const int a = align16i(2 * nori * ih * (desc_small(iw, ih) ? 4 : 8)), b = align16i(ncell * 4 * 8) + align16i(ncell * nori * 4 * 8);
    const int c = align16i(6 * (iw > ih ? iw : ih) * 4);   // the six resize tables
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
}
__host__ __device__ inline int desc_lds_bytes(int iw, int ih, int ncell, int nori, int dim, int hogMax) {
    const int m = iw > ih ? iw : ih;
    return desc_region_img(iw, ih, ncell, dim) + ((desc_small(iw, ih) && SDM_REGGRAD) ? 0 : align16i(iw * ih * 4)) + 3 * align16i(m * 4) +
           align16i(ncell * nori * 2 * 4) + align16i(ncell * 4) + desc_region_masks(iw, ih, ncell, nori) + align16i(m * 8) +
           align16i(hogMax * m * 4) + align16i(SDM_MAX_ORI * 8);
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// per (face, landmark): integer patch origin in image coordinates (may be negative: zero border),
// side length, validity.  origin[.] = {ox, oy, side, valid}
__global__ void k_sdm_prepare(const float* __restrict__ shapes, int B, int L, int W, int H, int adaptive, int fixedHalf,
                              int maxSide, double stepFactor, int32_t* __restrict__ origin, float* __restrict__ dist_out,
                              int32_t* __restrict__ status) {
    // one thread per (face, landmark); every thread of a face derives the face's window half size itself
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * L) return;
    const int f = t / L, i = t - f * L;
    const float* s = shapes + (size_t)f * 2 * L;
    int pwh = fixedHalf;
    float dist = 1.f;   // non-adaptive: modelShape + deltaShape.t() (SdmLandmarkModel.hpp:246-248); delta * 1.0f is exact
    if (adaptive) {
        // SdmLandmarkModel.hpp:212-229
        const float a1x = (s[8] + s[9]) / 2.0f, a1y = (s[8 + L] + s[9 + L]) / 2.0f;
        const float a2x = (s[11] + s[12]) / 2.0f, a2y = (s[11 + L] + s[12 + L]) / 2.0f;
        const double dx = (double)(a1x - a2x), dy = (double)(a1y - a2y);
        dist = (float)sqrt(dx * dx + dy * dy);
        const float windowSize = dist / 2.0f;
        float windowSizeHalf = windowSize / 2;
        windowSizeHalf = (float)round((double)windowSizeHalf * stepFactor);
        const int wi = (int)windowSizeHalf;
        pwh = wi + 3 - (wi % 3);
    }
    if (i == 0) dist_out[f] = dist;
    const int side = 2 * pwh;
    const int x = __float2int_rn(s[i]), y = __float2int_rn(s[i + L]);
    int ox = x - pwh, oy = y - pwh, valid = 1;
    if (x - pwh < 0 || y - pwh < 0 || x + pwh >= W || y + pwh >= H) {
        const int bl = (x - pwh) < 0 ? abs(x - pwh) : 0;
        const int bt = (y - pwh) < 0 ? abs(y - pwh) : 0;
        const int br = (x + pwh) >= W ? abs(W - (x + pwh)) : 0;
        const int bb = (y + pwh) >= H ? abs(H - (y + pwh)) : 0;
        const int rx = (x - pwh) + bl, ry = (y - pwh) + br;  // reference quirk: y uses borderRight (:171)
        const int EW = W + bl + br, EH = H + bt + bb;
        if (rx < 0 || ry < 0 || rx + side > EW || ry + side > EH) valid = 0;  // cv::Mat roi assertion
        ox = rx - bl;
        oy = ry - bt;
    }
    if (side < 4 || side > maxSide) valid = 0;
    int32_t* o = origin + 4 * (size_t)t;
    o[0] = ox; o[1] = oy; o[2] = side; o[3] = valid;
    if (!valid) status[f] = 1;
}

// zero outside the image.  Unsigned compares fold the two-sided tests; the offset is a 24-bit multiply + add (images are far
// below 2^24 pixels per side): the 64-bit multiply-add of the plain index runs at a quarter of the rate.
__device__ __forceinline__ float src_px(const uint8_t* __restrict__ img, int W, int H, int x, int y) {
    return ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) ? (float)img[(unsigned int)(__mul24(y, W) + x)] : 0.f;
}

#ifdef FD_SDM_PROF
__device__ unsigned long long fd_sdm_prof[8];
extern "C" void fd_debug_sdm_prof(unsigned long long* out, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(fd_sdm_prof), sizeof(fd_sdm_prof));
    if (reset) { unsigned long long z[8] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(fd_sdm_prof), z, sizeof(z)); }
}
#define SDM_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define SDM_ADD(i, x) pacc[i] += (unsigned long long)(x)
#else
#define SDM_T(v)
#define SDM_ADD(i, x)
#endif

// one wavefront per (face, landmark).  LDS per wave is what bounds the occupancy of this latency-bound kernel, so regions
// are reused: the gradient magnitudes overwrite the working image (SMALL: in place, one 64-pixel block behind the
// block being computed), the features overwrite it once the votes are in, the block factors reuse the orientation masks.
#ifndef FD_SDM_WPE
#define FD_SDM_WPE 5
#endif
template <bool SMALL>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(FD_SDM_WPE, 8))) void k_sdm_descriptors(const uint8_t* __restrict__ images, const int32_t* __restrict__ origin,
                                                         DescParams p, int64_t nitems, float* __restrict__ out, int64_t out_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: the per-wave LDS views below live in SGPRs
#ifdef FD_SDM_PROF
    unsigned long long pacc[8] = {};
#endif
    const int iw = p.iw, ih = p.ih, cs = p.cellSize, nori = p.nori;
    const int hogW = p.hogW, hogH = p.hogH;
    const int ncell = hogW * hogH;
    const int npix = iw * ih;
    const int m_ = iw > ih ? iw : ih;
    using mask_t = typename std::conditional<SMALL, unsigned int, unsigned long long>::type;   // SMALL: iw <= 32
    // i / iw for 0 <= i < 4096, iw <= 64 (exhaustively checked): three VALU instead of an integer division
    const float invW = 1.0f / (float)iw;
    auto divw = [&](int i) { return (int)(((float)i + 0.5f) * invW); };
    DescLds S;
    {
        unsigned char* b = smem + (size_t)wave * p.ldsPerWave;
        const int m = iw > ih ? iw : ih;
        S.img = (float*)b; S.feat = (float*)b; b += desc_region_img(iw, ih, ncell, p.dim);
        if (SMALL && SDM_REGGRAD) S.grad = S.img;
        else { S.grad = (float*)b; b += align16i(npix * 4); }
        S.wx1 = (float*)b; b += align16i(m * 4);
        S.wx2 = (float*)b; b += align16i(m * 4);
        S.binx = (int*)b; b += align16i(m * 4);
        S.hog = (float*)b; b += align16i(ncell * nori * 2 * 4);
        S.norm = (float*)b; b += align16i(ncell * 4);
        S.masks = (void*)b; S.fac = (double*)b; S.hc = (double*)(b + align16i(ncell * 4 * 8));
        b += desc_region_masks(iw, ih, ncell, nori);
        S.colmask = (unsigned long long*)b; b += align16i(m * 8);
        S.wsel = (float*)b; b += align16i((hogW > hogH ? hogW : hogH) * m * 4);
        S.oxy = (float2*)b;
    }
    mask_t* const masks = (mask_t*)S.masks;
    const int hogMax = hogW > hogH ? hogW : hogH;
    // ---- per-wave tables that only depend on the geometry.  Column/row interpolation: hx = (x + 0.5) / cellSize - 0.5
    // (hog.c:697-704); rows use the same table
    for (int x = lane; x < max(iw, ih); x += 64) {
        const float hx = (float)((x + 0.5) / cs - 0.5);
        int b = (int)hx;
        if (!(hx >= 0 || (float)b == hx)) b -= 1;  // vl_floor_f
        const float w2 = hx - b;
        const float w1 = (float)(1.0 - (double)w2);
        S.binx[x] = b; S.wx1[x] = w1; S.wx2[x] = w2;
    }
#pragma unroll
    for (int k = 0; k < SDM_MAX_ORI; ++k)
        if (lane == k) S.oxy[k] = make_float2(p.oX[k], p.oY[k]);
    wave_sync();
    for (int i = lane; i < hogMax * m_; i += 64) {   // weight of pixel column / row p towards cell column / row c
        const int c = i / m_, p_ = i - c * m_;
        S.wsel[i] = S.binx[p_] == c ? S.wx1[p_] : S.wx2[p_];
    }
    // per cell column: bit mask of the interior columns that vote into it
    for (int c = lane; c < hogW; c += 64) {
        unsigned long long mk = 0ull;
        for (int x = 1; x < iw - 1; ++x)
            if (S.binx[x] == c || S.binx[x] + 1 == c) mk |= 1ull << x;
        S.colmask[c] = mk;
    }
    wave_sync();
    // where the transposing store (DescriptorExtractor.hpp:198-205) of output element lane + 64 t reads: geometry only, so the two
    // integer divisions per element are done once per wavefront instead of once per item (descriptors of up to 64 * SDM_ST values)
    constexpr int SDM_ST = 5;
    int srcIdx[SDM_ST];
    const bool stFast = p.dim * ncell <= 64 * SDM_ST;
#pragma unroll
    for (int t = 0; t < SDM_ST; ++t) {
        const int i = lane + 64 * t;
        const int j = i / ncell, rem = i - j * ncell;
        const int cc = rem / hogH, r = rem - cc * hogH;
        srcIdx[t] = i < p.dim * ncell ? j * ncell + r * hogW + cc : -1;
    }
    // rows at which the cell row index moves on (bit y: binx[y] == binx[y - 1] + 1; it never moves by more), and binx[1]: scalars of the
    // voting loop
    unsigned long long rowStep = 0ull;
    for (int y = 2; y < ih - 1; ++y)
        if (S.binx[y] != S.binx[y - 1]) rowStep |= 1ull << y;
    rowStep = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned int)(rowStep >> 32)) << 32) | (unsigned long long)__builtin_amdgcn_readfirstlane((unsigned int)rowStep);
    const int binx1 = __builtin_amdgcn_readfirstlane(S.binx[ih > 1 ? 1 : 0]);
    // XCD-aware item order: workgroup b runs on XCD b % 8 (round-robin dispatch), and every XCD has its own 4 MB L2.  With the
    // plain order every XCD touches every face image (256 x 64 KB = 16.8 MB: the crops stream through the L2s, 263 MB of fabric
    // traffic per launch by the FETCH_SIZE counter); here XCD x only works on the faces f = x (mod 8), whose images stay resident.
    const bool byXcd = (gridDim.x % 8 == 0) && (nitems % p.L == 0);
    const int64_t nfaces = nitems / p.L;
    const int xcd = blockIdx.x & 7;
    const int64_t myFaces = byXcd ? (nfaces - xcd + 7) / 8 : 0;
    const int64_t jEnd = byXcd ? myFaces * p.L : nitems;
    const int64_t jStep = byXcd ? (int64_t)(gridDim.x / 8) * 2 : (int64_t)gridDim.x * 2;
    for (int64_t j = byXcd ? (int64_t)(blockIdx.x / 8) * 2 + wave : (int64_t)blockIdx.x * 2 + wave; j < jEnd; j += jStep) {
        const int64_t face = byXcd ? (j / p.L) * 8 + xcd : j / p.L;
        const int lm = (int)(j % p.L);
        const int64_t item = face * p.L + lm;
        const int32_t* org = origin + 4 * item;
        const int ox = org[0], oy = org[1], side = org[2], valid = org[3];
        float* dst = out + face * out_stride + (size_t)lm * p.len;
        if (!valid) {
            for (int i = lane; i < p.len; i += 64) dst[i] = 0.f;
            continue;
        }
        const uint8_t* img = images + face * p.image_stride;
        SDM_T(t0);
        // ---- working image: crop (+ fp32 bilinear resize to 30x30 when adaptive)
        if (p.adaptive && side != iw) {
            // cv::resize coordinates per destination column / row (the orientation-mask region is free until the gradients)
            int* rsx = (int*)S.masks;
            int* rsx1 = rsx + m_;
            int* ry0 = rsx1 + m_;
            int* ry1 = ry0 + m_;
            float* rfx = (float*)(ry1 + m_);
            float* rfy = rfx + m_;
            const double scale = 1. / ((double)iw / side);  // cv::resize: scale = 1/inv_scale
            for (int d = lane; d < m_; d += 64) {
                float fx = (float)((d + 0.5) * scale - 0.5);
                int sx = (int)floorf(fx);
                fx -= sx;
                float fy = fx;
                const int sy = sx;
                if (sx < 0) { fx = 0; sx = 0; }
                if (sx >= side - 1) { fx = 0; sx = side - 1; }
                rsx[d] = sx; rsx1[d] = sx + 1 < side ? sx + 1 : sx; rfx[d] = fx;
                ry0[d] = sy < 0 ? 0 : (sy >= side ? side - 1 : sy);
                ry1[d] = sy + 1 < 0 ? 0 : (sy + 1 >= side ? side - 1 : sy + 1);
                rfy[d] = fy;
            }
            // The source square goes to LDS first, row by row (64 consecutive bytes of the crop per load instruction: two or three cache
            // lines; the resize's own taps are four scattered bytes per lane, ~30 lines per instruction -- the phase was 27 % of the kernel,
            // nearly all of it the texture path).  The bytes sit behind the six tables in the (still unused) orientation-mask region.
            uint8_t* cropB = reinterpret_cast<uint8_t*>(rfy + m_);
            const int cropCap = desc_region_masks(iw, ih, ncell, nori) - 6 * m_ * 4;
            const bool staged = side * side <= cropCap;
            if (staged) {
                // four bytes per lane and load, lane = (row of a group of 64 / ndw rows, dword of the row): the column part of every address
                // and test is fixed per lane, a load instruction fetches 64 / ndw crop rows (a row of the crop is ndw = ceil(side / 4)
                // dwords; rows keep their stride of `side` bytes, so a row's last dword is stored bytewise when side is not a multiple of
                // four): a quarter of the load instructions of the byte-per-lane form and a tenth of its index arithmetic.  Dwords that touch the image border are put together from single bytes
                // (zero outside, DescriptorExtractor.hpp:156-178)
                const int ndw = (side + 3) >> 2;                 // dwords per crop row
                const int rpi = 64 / ndw;                        // crop rows per load instruction
                const int lr = (int)(((float)lane + 0.5f) * (1.0f / (float)ndw)), lc = (lane - __mul24(lr, ndw)) << 2;   // this lane's row in the group, byte column
                const bool act = lr < rpi;
                const int x = ox + lc;
                const bool xin = x >= 0 && x + 3 < p.W, whole = lc + 4 <= side;
                constexpr int CU = 4;
                for (int r0 = lr; r0 < side; r0 += rpi * CU) {   // (lanes behind the last row group leave the loop early: no barrier inside)
                    unsigned int v[CU];
#pragma unroll
                    for (int u = 0; u < CU; ++u) {
                        const int r = r0 + u * rpi, y = oy + r;
                        v[u] = 0u;
                        if (act && r < side) {
                            if (xin && (unsigned)y < (unsigned)p.H) {
                                __builtin_memcpy(&v[u], img + (unsigned int)(__mul24(y, p.W) + x), 4);
                            } else {
                                auto px = [&](int xx) -> unsigned int { return ((unsigned)xx < (unsigned)p.W && (unsigned)y < (unsigned)p.H) ? (unsigned int)img[(unsigned int)(__mul24(y, p.W) + xx)] : 0u; };
                                v[u] = px(x) | (px(x + 1) << 8) | (px(x + 2) << 16) | (px(x + 3) << 24);
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < CU; ++u) {
                        const int r = r0 + u * rpi;
                        if (act && r < side) {
                            uint8_t* dstB = cropB + __mul24(r, side) + lc;
                            if (whole) __builtin_memcpy(dstB, &v[u], 4);
                            else for (int k = 0; k < side - lc; ++k) dstB[k] = (uint8_t)(v[u] >> (8 * k));
                        }
                    }
                }
            }
            wave_sync();
            if (staged && SMALL) {
                // lane = (row parity, column): a lane keeps its column's taps and weights in registers and walks every other row
                const int half = lane >> 5, dx = lane & 31;
                if (dx < iw) {
                    const float fx = rfx[dx];
                    const int sx = rsx[dx], sx1 = rsx1[dx];
                    const float a0 = 1.f - fx, a1 = fx;
                    for (int dy = half; dy < ih; dy += 2) {
                        const float fy = rfy[dy];
                        const int y0 = __mul24(ry0[dy], side), y1 = __mul24(ry1[dy], side);
                        const float b0 = 1.f - fy, b1 = fy;
                        const float r0 = (float)cropB[y0 + sx] * a0 + (float)cropB[y0 + sx1] * a1;
                        const float r1 = (float)cropB[y1 + sx] * a0 + (float)cropB[y1 + sx1] * a1;
                        S.img[__mul24(dy, iw) + dx] = r0 * b0 + r1 * b1;
                    }
                }
            } else if (staged) {
                for (int i = lane; i < npix; i += 64) {
                    const int dy = divw(i), dx = i - __mul24(dy, iw);
                    const float fx = rfx[dx], fy = rfy[dy];
                    const int sx = rsx[dx], sx1 = rsx1[dx], y0 = __mul24(ry0[dy], side), y1 = __mul24(ry1[dy], side);
                    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
                    const float r0 = (float)cropB[y0 + sx] * a0 + (float)cropB[y0 + sx1] * a1;
                    const float r1 = (float)cropB[y1 + sx] * a0 + (float)cropB[y1 + sx1] * a1;
                    S.img[i] = r0 * b0 + r1 * b1;
                }
            } else
            for (int i = lane; i < npix; i += 64) {
                const int dy = divw(i), dx = i - __mul24(dy, iw);
                const float fx = rfx[dx], fy = rfy[dy];
                const int sx = rsx[dx], sx1 = rsx1[dx], y0 = ry0[dy], y1 = ry1[dy];
                const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
                const float r0 = src_px(img, p.W, p.H, ox + sx, oy + y0) * a0 + src_px(img, p.W, p.H, ox + sx1, oy + y0) * a1;
                const float r1 = src_px(img, p.W, p.H, ox + sx, oy + y1) * a0 + src_px(img, p.W, p.H, ox + sx1, oy + y1) * a1;
                S.img[i] = r0 * b0 + r1 * b1;
            }
            wave_sync();
        } else {
            for (int i = lane; i < npix; i += 64) {
                const int dy = divw(i), dx = i - __mul24(dy, iw);
                S.img[i] = src_px(img, p.W, p.H, ox + dx, oy + dy);
            }
        }
        for (int i = lane; i < ncell * nori * 2; i += 64) S.hog[i] = 0.f;
        for (int i = lane; i < 2 * nori * ih; i += 64) masks[i] = (mask_t)0;
        wave_sync();
        SDM_T(t1);
        // ---- gradient magnitude and hard orientation assignment per interior pixel (hog.c:612-665)
        // (the orientation unit vectors come from the per-wave LDS table S.oxy)
        auto gradient = [&](int i, float& gout) -> int {   // returns the orientation bin or -1, -2 for border pixels
            const int y = divw(i), x = i - __mul24(y, iw);
            if (x < 1 || y < 1 || x >= iw - 1 || y >= ih - 1) return -2;
            const float* it = S.img + i;
            float gradx = *(it + 1) - *(it - 1);
            float grady = *(it + iw) - *(it - iw);
            float grad2 = gradx * gradx + grady * grady;
            if (!(grad2 > 0.f)) { gradx = 0.f; grady = 0.f; grad2 = 0.f; }
            const float grad = sqrtf(grad2);
            float w0 = 0.f;
            int b0 = -1;
            bool exact = true;
            if (nori == 9) {
                // Nine orientations (the reference's SDM models): the winner of hog.c:640-655 from THREE candidates.  The undirected angle
                // of the gradient is estimated through its "diamond angle" |gy| / (|gx| + |gy|) (off by at most 4.1 degrees); the direction
                // of the 20-degree fan nearest to the estimate and its two neighbours contain the two directions nearest to the true
                // angle, and every other direction scores at least 4 % lower.  The three dot products are taken on the unnormalised
                // gradient; they differ from the reference's (normalised, separately rounded) scores by less than 1.2e-6 |g|, so a winner
                // that leads by more than 4e-6 of its score is the reference's winner.  Anything closer, a zero or a tiny gradient takes
                // the exact path below (nine normalised scores in order, ties to the first): the two IEEE divisions and six of the nine
                // evaluations are gone for all but a handful of pixels.
                exact = !(grad >= SDM_TINY);
                if (grad2 > 0.f && !exact) {
                    const float ax = fabsf(gradx), ay = fabsf(grady);
                    const float d = ay * __builtin_amdgcn_rcpf(ax + ay);
                    const float u = ((gradx < 0.f) == (grady < 0.f) || ay == 0.f) ? 4.5f * d : 9.0f - 4.5f * d;   // angle / 20 degrees
                    int ke = (int)(u + 0.5f);
                    ke = ke >= 9 ? ke - 9 : ke;
                    const int ka = ke == 0 ? 8 : ke - 1, kc = ke == 8 ? 0 : ke + 1;
                    const float2 oa = S.oxy[ka], ob = S.oxy[ke], oc = S.oxy[kc];
                    const float da = gradx * oa.x + grady * oa.y, db = gradx * ob.x + grady * ob.y, dc = gradx * oc.x + grady * oc.y;
                    const float sa = fabsf(da), sb = fabsf(db), sc = fabsf(dc);
                    const float m1 = fmaxf(fmaxf(sa, sb), sc), m2 = __builtin_amdgcn_fmed3f(sa, sb, sc);
                    exact = !(m1 - m2 > m1 * 4.0e-6f);
                    const int kb = sa == m1 ? ka : (sb == m1 ? ke : kc);
                    const float dbest = sa == m1 ? da : (sb == m1 ? db : dc);
                    b0 = kb + (dbest < 0.f ? 9 : 0);
                }
            }
            if (exact) {
                b0 = -1;
                if (grad >= SDM_TINY) {
                    // (float)((double)a / (double)b) == a / b for fp32 a, b: rounding the fp64 quotient (53 >= 2 * 24 + 2 bits) to
                    // fp32 cannot double-round, and the device's fp32 division is correctly rounded
                    gradx = gradx / grad;
                    grady = grady / grad;
                } else {
                    asm volatile("" ::: "memory");   // a real branch: the fp64 divisions must not be if-converted into every pixel
                    gradx = (float)((double)gradx / 1e-10);
                    grady = (float)((double)grady / 1e-10);
                }
                // hog.c:640-655 without branches: |score| through the sign bit, "score > best" as a compare + two selects (v_max keeps
                // the first maximum like the reference's strict >)
                auto consider = [&](int k) {
                    const float2 ok = S.oxy[k];
                    const float dot = gradx * ok.x + grady * ok.y;
                    const float score = fabsf(dot);
                    const int bin = dot < 0.f ? k + nori : k;
                    const bool gt = score > w0;
                    b0 = gt ? bin : b0;
                    w0 = fmaxf(w0, score);
                };
                if (nori == 9) {
                    asm volatile("" ::: "memory");   // a real branch (rare lanes): not if-converted into every pixel
#pragma unroll
                    for (int k = 0; k < 9; ++k) consider(k);
                } else {
#pragma unroll
                    for (int k = 0; k < SDM_MAX_ORI; ++k)
                        if (k < nori) consider(k);
                }
            }
            gout = grad;
            return b0;
        };
        if (SMALL && SDM_REGGRAD) {
            // In place, one block of 64 pixels behind: the votes go to their orientation masks at once (a region of their own); a
            // block's magnitudes wait in ONE register until the next block has been computed.  Pixel i reads i +- 1 and i +- iw with
            // iw <= 32: block t reads nothing below 64 t - 32, so block t - 1 may be overwritten as soon as every lane has finished
            // block t, and block t + 1 never looks at it.  -1 marks a border pixel (a magnitude is never negative).  (Round 5: the
            // 16-fold unrolled form that kept all magnitudes and bins in registers spilled 53 VGPRs at the 128 of four wavefronts
            // per SIMD -- 240 MB of scratch traffic per launch.)
            float gprev = -1.f;
#pragma unroll 1
            for (int i = lane; i < npix + 64; i += 64) {
                float g = -1.f;
                if (i < npix) {
                    float gg = 0.f;
                    const int b0 = gradient(i, gg);
                    if (b0 > -2) {
                        g = gg;
                        if (b0 >= 0) { const int y = divw(i); atomicOr(&masks[__mul24(b0, ih) + y], (mask_t)1 << (i - __mul24(y, iw))); }
                    }
                }
                wave_sync();   // every lane has read its neighbours in the previous block
                if (gprev >= 0.f) S.grad[i - 64] = gprev;
                gprev = g;
            }
        } else {
            for (int i = lane; i < npix; i += 64) {
                float g = 0.f;
                const int b0 = gradient(i, g);
                if (b0 > -2) {
                    S.grad[i] = g;
                    if (b0 >= 0) { const int y = divw(i); atomicOr(&masks[__mul24(b0, ih) + y], (mask_t)1 << (i - __mul24(y, iw))); }
                }
            }
        }
        wave_sync();
        SDM_T(t2);
        // ---- spatial voting (hog.c:697-721): lane e = (orientation, cell COLUMN) walks the interior rows once, in the reference's scan
        // order (y outer, x inner), and carries the accumulators of the two cell rows a pixel row votes into: A = cell row binx[y]
        // (weight wx1[y]), B = binx[y] + 1 (weight wx2[y]).  binx never decreases and grows by at most one per row, so when it moves
        // on, A is complete (stored), B becomes A and a fresh B starts.  Every accumulator still receives exactly its own terms
        // (g * wx) * wy in scan order with sequential fp32 adds -- bit-identical to hog.c -- but the rows are visited once by 2 * nori *
        // hogW lanes (54 of 64 for the 3 x 3 x 18 descriptor: one pass) instead of three passes of (orientation, cell) lanes over 20
        // rows each, the row loop is wave-uniform, and g * wx is shared by the two cell rows.  Accumulators of cell rows outside the
        // descriptor (binx = -1, binx + 1 = hogH) collect terms nobody stores, like the reference's bounds checks drop them.
        for (int e = lane; e < 2 * nori * hogW; e += 64) {
            const int o = e / hogW, cx = e - o * hogW;
            const mask_t cm = (mask_t)S.colmask[cx];
            const float* __restrict__ wxs = S.wsel + cx * m_;
            const int oih = __mul24(o, ih);
            float* __restrict__ hcol = S.hog + o * ncell + cx;   // + cy * hogW
            float accA = 0.f, accB = 0.f;
            int b = binx1;
            mask_t mkN = ih > 2 ? masks[oih + 1] : (mask_t)0;
            for (int y = 1; y < ih - 1; ++y) {
                if (y > 1 && (rowStep >> y & 1ull)) {   // wave-uniform, scalar: binx[y] = binx[y - 1] + 1
                    if (b >= 0 && b < hogH) hcol[b * hogW] = accA;
                    accA = accB;
                    accB = 0.f;
                    ++b;
                }
                const float wyA = S.wx1[y], wyB = S.wx2[y];
                mask_t mk = mkN & cm;
                if (y + 1 < ih - 1) mkN = masks[oih + y + 1];   // the next row's mask is requested before this row's votes (and not looked at)
                if (!mk) continue;
                const float* __restrict__ grow = S.grad + __mul24(y, iw);
                int x = (SMALL ? __ffs((int)mk) : __ffsll((long long)mk)) - 1;
                mk &= mk - 1;
                float g = grow[x], wx = wxs[x];
                while (mk) {   // the next vote's two LDS operands are requested before the current vote is added
                    x = (SMALL ? __ffs((int)mk) : __ffsll((long long)mk)) - 1;
                    mk &= mk - 1;
                    const float gn = grow[x], wxn = wxs[x];
                    const float t = g * wx;
                    accA = accA + t * wyA;
                    accB = accB + t * wyB;
                    g = gn;
                    wx = wxn;
                }
                const float t = g * wx;
                accA = accA + t * wyA;
                accB = accB + t * wyB;
            }
            if (b >= 0 && b < hogH) hcol[b * hogW] = accA;
            if (b + 1 >= 0 && b + 1 < hogH) hcol[(b + 1) * hogW] = accB;
        }
        wave_sync();
        SDM_T(t3);
        // ---- undirected squared norms (hog.c:878-893)
        for (int c = lane; c < ncell; c += 64) {
            float nrm = 0.f;
            for (int k = 0; k < nori; ++k) {
                const float h = S.hog[c + k * ncell] + S.hog[c + (k + nori) * ncell];
                nrm = nrm + h * h;
            }
            S.norm[c] = nrm;
        }
        wave_sync();
        // ---- block normalisation and feature assembly (hog.c:925-1060).  Lane (cell, q): block factor f_q of the cell;
        // lane (cell, k): the clamped terms of orientation k; lane cell: the four texture sums in orientation order.
        const int dim = p.dim;
        for (int j = lane; j < ncell * 4; j += 64) {
            const int c = j >> 2, q = j & 3;
            const int y = c / hogW, x = c - y * hogW;
            const int xm = max(x - 1, 0), xp = min(x + 1, hogW - 1), ym = max(y - 1, 0), yp = min(y + 1, hogH - 1);
            // f1 = n1+n2+n4+n5, f2 = n2+n3+n5+n6, f3 = n4+n5+n7+n8, f4 = n5+n6+n8+n9 over the 3x3 neighbourhood n1..n9
            const int xa = (q & 1) ? x : xm, xb = (q & 1) ? xp : x, ya = (q & 2) ? y : ym, yb = (q & 2) ? yp : y;
            const double na = S.norm[xa + ya * hogW], nb = S.norm[xb + ya * hogW], nc = S.norm[xa + yb * hogW], nd = S.norm[xb + yb * hogW];
            S.fac[j] = 1.0 / sqrt(na + nb + nc + nd + 1e-4);
        }
        wave_sync();
        for (int j = lane; j < ncell * nori; j += 64) {
            const int k = j / ncell, c = j - k * ncell;
            const double f1 = S.fac[4 * c], f2 = S.fac[4 * c + 1], f3 = S.fac[4 * c + 2], f4 = S.fac[4 * c + 3];
            const double ha = S.hog[c + k * ncell], hb = S.hog[c + (k + nori) * ncell];
            double ha1 = f1 * ha, ha2 = f2 * ha, ha3 = f3 * ha, ha4 = f4 * ha;
            double hb1 = f1 * hb, hb2 = f2 * hb, hb3 = f3 * hb, hb4 = f4 * hb;
            double hc1 = ha1 + hb1, hc2 = ha2 + hb2, hc3 = ha3 + hb3, hc4 = ha4 + hb4;
            ha1 = fmin(0.2, ha1); ha2 = fmin(0.2, ha2); ha3 = fmin(0.2, ha3); ha4 = fmin(0.2, ha4);
            hb1 = fmin(0.2, hb1); hb2 = fmin(0.2, hb2); hb3 = fmin(0.2, hb3); hb4 = fmin(0.2, hb4);
            hc1 = fmin(0.2, hc1); hc2 = fmin(0.2, hc2); hc3 = fmin(0.2, hc3); hc4 = fmin(0.2, hc4);
            double* hcp = S.hc + 4 * (size_t)(c * nori + k);
            hcp[0] = hc1; hcp[1] = hc2; hcp[2] = hc3; hcp[3] = hc4;
            if (p.variant == 1) {
                S.feat[c + k * ncell] = (float)(0.5 * (ha1 + ha2 + ha3 + ha4));
                S.feat[c + (k + nori) * ncell] = (float)(0.5 * (hb1 + hb2 + hb3 + hb4));
                S.feat[c + (k + 2 * nori) * ncell] = (float)(0.5 * (hc1 + hc2 + hc3 + hc4));
            } else {
                S.feat[c + k * ncell] = (float)hc1;
                S.feat[c + (k + nori) * ncell] = (float)hc2;
                S.feat[c + (k + 2 * nori) * ncell] = (float)hc3;
                S.feat[c + (k + 3 * nori) * ncell] = (float)hc4;
            }
        }
        wave_sync();
        if (p.variant == 1) {
            for (int j = lane; j < ncell * 4; j += 64) {
                const int c = j >> 2, q4 = j & 3;
                double t = 0;
                for (int k = 0; k < nori; ++k) t = t + S.hc[4 * (size_t)(c * nori + k) + q4];
                const float q = 1.0f / sqrtf(18.0f);
                S.feat[c + (3 * nori + q4) * ncell] = (float)((double)q * t);
            }
        }
        wave_sync();
        SDM_T(t4);
        // ---- per-plane transpose and stack (DescriptorExtractor.hpp:198-205): out[j][c][r] = feat[j][r][c]
        if (stFast) {
#pragma unroll
            for (int t = 0; t < SDM_ST; ++t)
                if (srcIdx[t] >= 0) dst[lane + 64 * t] = S.feat[srcIdx[t]];
        } else
        for (int i = lane; i < dim * ncell; i += 64) {
            const int j = i / ncell, rem = i - j * ncell;
            const int cc = rem / hogH, r = rem - cc * hogH;
            dst[i] = S.feat[j * ncell + r * hogW + cc];
        }
        wave_sync();
#ifdef FD_SDM_PROF
        { SDM_T(t5); SDM_ADD(0, 1); SDM_ADD(1, t1 - t0); SDM_ADD(2, t2 - t1); SDM_ADD(3, t3 - t2); SDM_ADD(4, t4 - t3); SDM_ADD(5, t5 - t4); SDM_ADD(6, t5 - t0); }
#endif
    }
#ifdef FD_SDM_PROF
    if (lane == 0) for (int q_ = 0; q_ < 8; ++q_) atomicAdd(&fd_sdm_prof[q_], pacc[q_]);
#endif
}

typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int RG_KCHUNK = 256;

// delta partial sums: one wavefront per (16-face tile, RG_NT 16-output tiles, K chunk) on the f64 MFMA pipe; the descriptor
// fragment is loaded once for the RG_NT independent accumulator chains
constexpr int RG_NT = 3;
__global__ __launch_bounds__(64) void k_sdm_regress(const float* __restrict__ D, int B, int F, const float* __restrict__ R, int N,
                                                    double* __restrict__ partial, int nchunks) {
    const int lane = threadIdx.x;
    const int mt = blockIdx.x, ng = blockIdx.y, ch = blockIdx.z;
    const int i = mt * 16 + (lane & 15), kq = lane >> 4;
    const int k0 = ch * RG_KCHUNK, k1 = min(F, k0 + RG_KCHUNK);
    f64x4 acc[RG_NT];
    int j[RG_NT];
    bool jok[RG_NT];
#pragma unroll
    for (int t = 0; t < RG_NT; ++t) {
        acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
        j[t] = (ng * RG_NT + t) * 16 + (lane & 15);
        jok[t] = j[t] < N;
    }
    const bool iok = i < B;
    const float* __restrict__ drow = D + (size_t)(iok ? i : 0) * F;
    // RG_U k-steps per iteration: their loads are issued together, then the dependent MFMA chains run
    constexpr int RG_U = 4;
    for (int k = k0; k < k1; k += 4 * RG_U) {
        float a[RG_U], b[RG_U][RG_NT];
#pragma unroll
        for (int u = 0; u < RG_U; ++u) {
            const int kk = k + 4 * u + kq;
            const bool kok = kk < k1;
            const int kc = kok ? kk : k0;   // in-range address for the masked lanes
            a[u] = drow[kc];
            a[u] = (iok && kok) ? a[u] : 0.f;
#pragma unroll
            for (int t = 0; t < RG_NT; ++t) {
                const float v = R[(size_t)kc * N + (jok[t] ? j[t] : 0)];
                b[u][t] = (jok[t] && kok) ? v : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < RG_U; ++u)
#pragma unroll
            for (int t = 0; t < RG_NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a[u], (double)b[u][t], acc[t], 0, 0, 0);
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int t = 0; t < RG_NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = mt * 16 + (lane >> 4) + 4 * r;
            if (row < B && jok[t]) partial[((size_t)ch * B + row) * N + j[t]] = acc[t][r];
        }
}

// shape += (float)(sum of partials + bias row) * dist   (SdmLandmarkModel.hpp:241-243)
__global__ void k_sdm_update(float* __restrict__ shapes, const double* __restrict__ partial, int nchunks, const float* __restrict__ biasRow,
                             const float* __restrict__ dist, int B, int N) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * N) return;
    const int f = idx / N, j = idx - f * N;
    double acc = 0.0;
    // the chunks' partial sums, sixteen loads in flight at a time, added in chunk order (one load per round trip made this 20 us of a 330 us step)
    const double* pp = partial + (size_t)f * N + j;
    const size_t cs = (size_t)B * N;
    int c = 0;
    for (; c + 16 <= nchunks; c += 16) {
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = pp[(size_t)(c + i) * cs];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = acc + v[i];
    }
    for (; c < nchunks; ++c) acc = acc + pp[(size_t)c * cs];
    acc = acc + (double)biasRow[j];
    const float delta = (float)acc;
    const float t = delta * dist[f];
    shapes[idx] = shapes[idx] + t;
}

}  // namespace

// device + pinned scratch of one batch in flight (fd_sdm_fit_batch_begin / _end: several batches of one model can be queued)
struct SdmScratch {
    DevBuf images, shapes, origin, dist, status, desc, partial;
    HostBuf hshapes, hstatus;
    hipEvent_t done = nullptr;
    ~SdmScratch() { if (done) (void)hipEventDestroy(done); }
};
struct fd_sdm {
    fd_ctx* ctx;
    int L, S, variant;
    std::vector<int> descParams;   // empty: adaptive; else 3 per step {numCells, cellSize, numBins}
    std::vector<float> mean;
    std::vector<int> Rrows;
    std::vector<std::unique_ptr<DevBuf>> R;
    std::vector<std::unique_ptr<SdmScratch>> idle;   // scratch sets not in use (handles are single-threaded: no lock)
    unsigned int launches = 0;
    hipEvent_t order = nullptr;   // orders the auxiliary stream of odd tickets behind the context's stream (device-resident images)
    ~fd_sdm() { if (order) (void)hipEventDestroy(order); }
};
struct fd_sdm_ticket {
    fd_sdm* m = nullptr;
    std::unique_ptr<SdmScratch> s;
    int B = 0;
    bool timed = false;
};

namespace {

void fill_desc_params(DescParams& p, int W, int H, int L, bool adaptive, int variant, int numCells, int cellSize, int numBins,
                      int side) {
    std::memset(&p, 0, sizeof(p));
    p.W = W; p.H = H; p.L = L;
    p.adaptive = adaptive ? 1 : 0;
    p.variant = variant;
    if (adaptive) { p.cellSize = 10; p.nori = 9; p.iw = p.ih = 30; }
    else { p.cellSize = cellSize; p.nori = numBins; p.iw = p.ih = side; }
    (void)numCells;
    if (p.cellSize < 1 || p.nori < 1 || p.nori > SDM_MAX_ORI) FD_THROW(FD_ERR_INVALID_ARGUMENT, "VlHog: unsupported cellSize/numBins");
    if (p.iw < 4 || p.iw > SDM_IMG) FD_THROW(FD_ERR_INVALID_ARGUMENT, "VlHog: patch side %d outside 4..%d", p.iw, SDM_IMG);
    p.hogW = (p.iw + p.cellSize / 2) / p.cellSize;
    p.hogH = (p.ih + p.cellSize / 2) / p.cellSize;
    if (p.hogW < 1 || p.hogH < 1 || p.hogW > SDM_MAX_CELLS || p.hogH > SDM_MAX_CELLS)
        FD_THROW(FD_ERR_INVALID_ARGUMENT, "VlHog: %dx%d cells unsupported", p.hogW, p.hogH);
    p.dim = variant == 1 ? 3 * p.nori + 4 : 4 * p.nori;
    p.len = p.hogW * p.hogH * p.dim;
    for (int o = 0; o < p.nori; ++o) {
        double angle = o * 3.141592653589793 / p.nori;
        p.oX[o] = (float)std::cos(angle);
        p.oY[o] = (float)std::sin(angle);
    }
    p.ldsPerWave = desc_lds_bytes(p.iw, p.ih, p.hogW * p.hogH, p.nori, p.dim, p.hogW > p.hogH ? p.hogW : p.hogH);
    if (2 * p.ldsPerWave > 64 * 1024) FD_THROW(FD_ERR_INVALID_ARGUMENT, "VlHog: patch too large for the LDS budget");
}

// resident workgroups per CU of the descriptor kernel (LDS-bound): the persistent grid is sized to them so that the per-wave
// geometry tables are built once per resident wave
int descriptor_blocks_per_cu(const DescParams& p) {
    int nb = 0;
    const hipError_t e = desc_small(p.iw, p.ih)
        ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_sdm_descriptors<true>, 128, (size_t)2 * p.ldsPerWave)
        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_sdm_descriptors<false>, 128, (size_t)2 * p.ldsPerWave);
    return e == hipSuccess && nb > 0 ? nb : 4;
}

void launch_descriptors(dim3 grid, hipStream_t st, const uint8_t* dimg, const int32_t* origin, const DescParams& p, int64_t nitems, float* out,
                        int64_t out_stride) {
    if (desc_small(p.iw, p.ih))
        hipLaunchKernelGGL(k_sdm_descriptors<true>, grid, dim3(128), 2 * p.ldsPerWave, st, dimg, origin, p, nitems, out, out_stride);
    else
        hipLaunchKernelGGL(k_sdm_descriptors<false>, grid, dim3(128), 2 * p.ldsPerWave, st, dimg, origin, p, nitems, out, out_stride);
}

}  // namespace

extern "C" {

int fd_sdm_create(fd_ctx* ctx, const fd_sdm_model* md, fd_sdm** out) {
    return fd_guard(ctx, [&] {
        if (!ctx || !md || !out || !md->mean || !md->R || !md->R_rows) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_sdm_create: NULL argument");
        if (md->num_landmarks < 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "SdmLandmarkModel: no landmarks");
        if (!md->desc_params && md->num_landmarks < 13)
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "SdmLandmarkModelFitting needs landmarks 8,9,11,12 (SdmLandmarkModel.hpp:212-216)");
        if (md->num_steps < 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "SdmLandmarkModel: no cascade steps");
        if (md->hog_variant != 0 && md->hog_variant != 1)
            FD_THROW(FD_ERR_LOGIC, "descriptorType does not match 'vlhog-dt' or 'vlhog-uoctti'");
        HIP_CHECK(hipSetDevice(ctx->device));
        std::unique_ptr<fd_sdm> m(new fd_sdm());
        m->ctx = ctx;
        m->L = md->num_landmarks;
        m->S = md->num_steps;
        m->variant = md->hog_variant;
        m->mean.assign(md->mean, md->mean + 2 * m->L);
        const int dim = m->variant == 1 ? 31 : 36;
        if (md->desc_params) m->descParams.assign(md->desc_params, md->desc_params + 3 * (size_t)m->S);
        for (int s = 0; s < m->S; ++s) {
            if (md->desc_params) {
                // the non-adaptive branch: the descriptor length follows from the step's own {numCells, cellSize, numBins}
                const int32_t* dp = md->desc_params + 3 * s;
                if (dp[0] < 1 || dp[1] < 1 || dp[2] < 1) FD_THROW(FD_ERR_LOGIC, "descriptorParameters must contain numCells, cellSize and numBins.");
                //  bounds of what the descriptor kernel's LDS layout holds; also catches a struct that was not zero-initialised
                if (dp[0] > 8 || dp[1] < 2 || dp[1] > 64 || dp[2] > 32)
                    FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_sdm_model.desc_params of step %d out of range: numCells %d (1..8), cellSize %d (2..64), numBins %d (1..32)",
                             s, dp[0], dp[1], dp[2]);
                DescParams p;
                fill_desc_params(p, 64, 64, m->L, false, m->variant, dp[0], dp[1], dp[2], 2 * (dp[0] * (dp[1] / 2)));
                if (md->R_rows[s] != m->L * p.len + 1)
                    FD_THROW(FD_ERR_INVALID_ARGUMENT, "regressor %d has %d rows, expected %d (%dx%dx%d descriptor per landmark + bias)",
                             s, md->R_rows[s], m->L * p.len + 1, p.hogW, p.hogH, p.dim);
            } else if (md->R_rows[s] != m->L * 9 * dim + 1)
                FD_THROW(FD_ERR_INVALID_ARGUMENT, "regressor %d has %d rows, expected %d (adaptive 3x3x%d descriptor per landmark + bias)",
                         s, md->R_rows[s], m->L * 9 * dim + 1, dim);
            m->Rrows.push_back(md->R_rows[s]);
            std::unique_ptr<DevBuf> b(new DevBuf());
            const size_t bytes = sizeof(float) * (size_t)md->R_rows[s] * 2 * m->L;
            b->reserve(bytes);
            HIP_CHECK(hipMemcpy(b->p, md->R[s], bytes, hipMemcpyHostToDevice));
            m->R.push_back(std::move(b));
        }
        *out = m.release();
    });
}

void fd_sdm_destroy(fd_sdm* m) { delete m; }

int fd_sdm_descriptors(fd_ctx* ctx, const uint8_t* gray, int W, int H, const float* px, const float* py, int n, int wsh,
                       int variant, int numCells, int cellSize, int numBins, float* out, int* len) {
    return fd_guard(ctx, [&] {
        if (!ctx || !gray || !px || !py || n < 0 || W < 1 || H < 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_sdm_descriptors: bad argument");
        if (variant != 0 && variant != 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_sdm_descriptors: unknown HOG variant");
        HIP_CHECK(hipSetDevice(ctx->device));
        const bool adaptive = wsh > 0;
        const int pwh = adaptive ? wsh : numCells * (cellSize / 2);
        DescParams p;
        fill_desc_params(p, W, H, n, adaptive, variant, numCells, cellSize, numBins, 2 * pwh);
        p.image_stride = 0;
        if (len) *len = p.len;
        if (!out || n == 0) return;
        hipStream_t st = ctx->stream;
        DevBuf dimg, dshape, dorigin, ddist, dstatus, ddesc;
        dimg.reserve((size_t)W * H);
        dshape.reserve(sizeof(float) * 2 * (size_t)n);
        dorigin.reserve(sizeof(int32_t) * 4 * (size_t)n);
        ddist.reserve(16);
        dstatus.reserve(16);
        ddesc.reserve(sizeof(float) * (size_t)n * p.len);
        HIP_CHECK(hipMemcpyAsync(dimg.p, gray, (size_t)W * H, hipMemcpyHostToDevice, st));
        std::vector<float> shape(2 * (size_t)n);
        for (int i = 0; i < n; ++i) { shape[i] = px[i]; shape[i + n] = py[i]; }
        HIP_CHECK(hipMemcpyAsync(dshape.p, shape.data(), sizeof(float) * shape.size(), hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemsetAsync(dstatus.p, 0, 4, st));
        // one "face" with n landmarks and a fixed half window
        hipLaunchKernelGGL(k_sdm_prepare, dim3((n + 63) / 64), dim3(64), 0, st, dshape.as<float>(), 1, n, W, H, 0, pwh, adaptive ? (1 << 20) : SDM_IMG, 1.0, dorigin.as<int32_t>(),
                           ddist.as<float>(), dstatus.as<int32_t>());
        launch_descriptors(dim3((n + 1) / 2), st, dimg.as<uint8_t>(), dorigin.as<int32_t>(), p, (int64_t)n, ddesc.as<float>(), (int64_t)n * p.len);
        HIP_CHECK(hipGetLastError());
        int32_t status = 0;
        HIP_CHECK(hipMemcpyAsync(out, ddesc.p, sizeof(float) * (size_t)n * p.len, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(&status, dstatus.p, 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (status) FD_THROW(FD_ERR_RUNTIME, "VlHogDescriptorExtractor: patch window leaves the zero-extended image (cv::Mat roi assertion in the reference)");
    });
}

// Queues one batch (S cascade steps: prepare -> descriptors -> regress -> update, then the read-back into the scratch set's pinned
// buffers) on `st` and records sc.done; nothing waits.  sc.hshapes holds the start shapes on entry.
static void sdm_launch(fd_ctx* ctx, fd_sdm* m, SdmScratch& sc, hipStream_t st, const uint8_t* gray_images, int W, int H, int B, int images_on_device,
                       bool timeIt) {
    HIP_CHECK(hipSetDevice(ctx->device));
    const int L = m->L, N = 2 * L;
    const uint8_t* dimg = gray_images;
    if (!images_on_device) {
        sc.images.reserve((size_t)W * H * B);
        HIP_CHECK(hipMemcpyAsync(sc.images.p, gray_images, (size_t)W * H * B, hipMemcpyHostToDevice, st));
        dimg = sc.images.as<uint8_t>();
    }
    const bool adaptive = m->descParams.empty();
    // per step: the descriptor geometry (one for all steps when adaptive) and the half window of the non-adaptive branch
    std::vector<DescParams> ps(adaptive ? 1 : m->S);
    std::vector<int> half(ps.size(), 0);
    int Fmax = 0;
    for (size_t s = 0; s < ps.size(); ++s) {
        if (adaptive) fill_desc_params(ps[s], W, H, L, true, m->variant, 3, 10, 9, 30);
        else {
            const int* dp = &m->descParams[3 * s];
            half[s] = dp[0] * (dp[1] / 2);   // patchWidthHalf = numCells * (cellSize / 2), DescriptorExtractor.hpp:142
            fill_desc_params(ps[s], W, H, L, false, m->variant, dp[0], dp[1], dp[2], 2 * half[s]);
        }
        ps[s].image_stride = (int64_t)W * H;
        Fmax = std::max(Fmax, L * ps[s].len);
    }
    const int nchunksMax = (Fmax + RG_KCHUNK - 1) / RG_KCHUNK;
    const size_t nshape = (size_t)B * N;
    sc.shapes.reserve(sizeof(float) * nshape);
    sc.origin.reserve(sizeof(int32_t) * 4 * (size_t)B * L);
    sc.dist.reserve(sizeof(float) * B);
    sc.status.reserve(sizeof(int32_t) * B);
    sc.desc.reserve(sizeof(float) * (size_t)B * Fmax);
    sc.partial.reserve(sizeof(double) * (size_t)nchunksMax * B * N);
    sc.hstatus.reserve(sizeof(int32_t) * B);
    HIP_CHECK(hipMemcpyAsync(sc.shapes.p, sc.hshapes.p, sizeof(float) * nshape, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemsetAsync(sc.status.p, 0, sizeof(int32_t) * B, st));
    const int64_t nitems = (int64_t)B * L;
    for (int step = 0; step < m->S; ++step) {
        const double stepFactor = 1 / (1 + std::exp((double)((step + 1) - m->S)));  // :226, double on the host
        const DescParams& p = ps[adaptive ? 0 : step];
        const int F = L * p.len;
        const int nchunks = (F + RG_KCHUNK - 1) / RG_KCHUNK;
        hipLaunchKernelGGL(k_sdm_prepare, dim3((B * L + 255) / 256), dim3(256), 0, st, sc.shapes.as<float>(), B, L, W, H, adaptive ? 1 : 0,
                           adaptive ? 0 : half[step], adaptive ? (1 << 20) : SDM_IMG, stepFactor,
                           sc.origin.as<int32_t>(), sc.dist.as<float>(), sc.status.as<int32_t>());
        int grid = (int)std::min<int64_t>((nitems + 1) / 2, (int64_t)ctx->num_cus * descriptor_blocks_per_cu(p));
        if (grid >= 16) grid &= ~7;   // a multiple of the 8 XCDs: the kernel then keeps every face on one XCD
        const bool timeThis = timeIt && step + 1 == m->S;   // fd_hip_bench.h: the last step's descriptor launch
        if (timeThis) HIP_CHECK(hipEventRecord(ctx->ev0, st));
        launch_descriptors(dim3(grid), st, dimg, sc.origin.as<int32_t>(), p, nitems, sc.desc.as<float>(), (int64_t)F);
        if (timeThis) HIP_CHECK(hipEventRecord(ctx->ev1, st));
        const float* R = m->R[step]->as<float>();
        hipLaunchKernelGGL(k_sdm_regress, dim3((B + 15) / 16, ((N + 15) / 16 + RG_NT - 1) / RG_NT, nchunks), dim3(64), 0, st, sc.desc.as<float>(), B, F, R, N,
                           sc.partial.as<double>(), nchunks);
        hipLaunchKernelGGL(k_sdm_update, dim3((B * N + 255) / 256), dim3(256), 0, st, sc.shapes.as<float>(), sc.partial.as<double>(), nchunks,
                           R + (size_t)F * N, sc.dist.as<float>(), B, N);
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(sc.hshapes.p, sc.shapes.p, sizeof(float) * nshape, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(sc.hstatus.p, sc.status.p, sizeof(int32_t) * B, hipMemcpyDeviceToHost, st));
    if (!sc.done) HIP_CHECK(hipEventCreateWithFlags(&sc.done, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(sc.done, st));
}

static std::unique_ptr<SdmScratch> sdm_take_scratch(fd_sdm* m) {
    if (m->idle.empty()) return std::unique_ptr<SdmScratch>(new SdmScratch());
    std::unique_ptr<SdmScratch> s = std::move(m->idle.back());
    m->idle.pop_back();
    return s;
}

// begin: start shapes -> pinned buffer, kernels queued; end: wait, copy out, scratch back to the model
static void sdm_begin(fd_ctx* ctx, fd_sdm* m, fd_sdm_ticket& t, const uint8_t* gray_images, int W, int H, int B, int images_on_device,
                      const float* start_shapes, const int32_t* face_boxes, hipStream_t st, bool timeIt) {
    const int L = m->L, N = 2 * L;
    t.m = m;
    t.B = B;
    t.timed = timeIt;
    t.s = sdm_take_scratch(m);
    t.s->hshapes.reserve(sizeof(float) * (size_t)B * N);
    float* shapes = t.s->hshapes.as<float>();
    if (start_shapes) {
        std::memcpy(shapes, start_shapes, sizeof(float) * (size_t)B * N);
    } else {
        // alignRigid (SdmLandmarkModel.hpp:156-192) on the host, exactly as the cv::MatExpr evaluates it:
        // x * float(w) + float(0.5 * w + bx)
        for (int f = 0; f < B; ++f) {
            const int32_t* fb = face_boxes + 4 * f;
            const float ax = (float)(double)fb[2], bx = (float)(0.5 * fb[2] + fb[0]);
            const float ay = (float)(double)fb[3], by = (float)(0.5 * fb[3] + fb[1]);
            for (int i = 0; i < L; ++i) {
                shapes[(size_t)f * N + i] = m->mean[i] * ax + bx;
                shapes[(size_t)f * N + i + L] = m->mean[i + L] * ay + by;
            }
        }
    }
    sdm_launch(ctx, m, *t.s, st, gray_images, W, H, B, images_on_device, timeIt);
}
static void sdm_end(fd_ctx* ctx, fd_sdm_ticket& t, float* shapes_out, int32_t* status_out) {
    if (!t.s) return;
    HIP_CHECK(hipEventSynchronize(t.s->done));
    const int N = 2 * t.m->L;
    if (shapes_out) std::memcpy(shapes_out, t.s->hshapes.p, sizeof(float) * (size_t)t.B * N);
    if (status_out) std::memcpy(status_out, t.s->hstatus.p, sizeof(int32_t) * t.B);
    if (t.timed && t.m->S > 0) {
        HIP_CHECK(hipEventElapsedTime(&ctx->last_kernel_ms, ctx->ev0, ctx->ev1));
        ctx->last_kernel = "k_sdm_descriptors";
    }
    t.m->idle.push_back(std::move(t.s));
}

int fd_sdm_fit_batch(fd_ctx* ctx, const fd_sdm* m_, const uint8_t* gray_images, int W, int H, int batch, const int32_t* face_boxes,
                     int images_on_device, float* shapes_out, int32_t* status_out) {
    return fd_guard(ctx, [&] {
        if (!ctx || !m_ || !gray_images || !face_boxes || !shapes_out || batch < 0 || W < 1 || H < 1)
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_sdm_fit_batch: bad argument");
        if (batch == 0) return;
        fd_sdm_ticket t;
        sdm_begin(ctx, const_cast<fd_sdm*>(m_), t, gray_images, W, H, batch, images_on_device, nullptr, face_boxes, ctx->stream, ctx->kernel_timing);
        sdm_end(ctx, t, shapes_out, status_out);
    });
}

// Asynchronous form: _begin queues the whole fit of a batch and returns; _end waits for it and delivers the shapes.  Several
// batches of one model can be in flight (each has its own scratch set); consecutive tickets alternate between two streams so that
// the small regress / update kernels of one batch run beside the descriptor kernel of the next.  The images (and face boxes) of a
// batch must stay valid until its _end when they are host memory.
int fd_sdm_fit_batch_begin(fd_ctx* ctx, const fd_sdm* m_, const uint8_t* gray_images, int W, int H, int batch, const int32_t* face_boxes,
                           int images_on_device, fd_sdm_ticket** ticket) {
    if (ticket) *ticket = nullptr;
    return fd_guard(ctx, [&] {
        if (!ctx || !m_ || !gray_images || !face_boxes || !ticket || batch < 1 || W < 1 || H < 1)
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_sdm_fit_batch_begin: bad argument");
        fd_sdm* m = const_cast<fd_sdm*>(m_);
        std::unique_ptr<fd_sdm_ticket> t(new fd_sdm_ticket());
        hipStream_t st = (m->launches++ & 1u) ? fd_aux_stream(ctx) : ctx->stream;
        if (st != ctx->stream && images_on_device) {
            // device-resident images may still be being written by work the caller queued on the context's stream: the auxiliary stream
            // starts behind everything queued there so far
            if (!m->order) HIP_CHECK(hipEventCreateWithFlags(&m->order, hipEventDisableTiming));
            HIP_CHECK(hipEventRecord(m->order, ctx->stream));
            HIP_CHECK(hipStreamWaitEvent(st, m->order, 0));
        }
        sdm_begin(ctx, m, *t, gray_images, W, H, batch, images_on_device, nullptr, face_boxes, st, false);
        *ticket = t.release();
    });
}
int fd_sdm_fit_batch_end(fd_ctx* ctx, fd_sdm_ticket* ticket, float* shapes_out, int32_t* status_out) {
    std::unique_ptr<fd_sdm_ticket> t(ticket);
    return fd_guard(ctx, [&] {
        if (!ctx || !t) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_sdm_fit_batch_end: bad argument");
        sdm_end(ctx, *t, shapes_out, status_out);
    });
}

int fd_sdm_optimize_batch(fd_ctx* ctx, const fd_sdm* m_, const uint8_t* gray_images, int W, int H, int batch, int images_on_device,
                          float* shapes_inout, int32_t* status_out) {
    return fd_guard(ctx, [&] {
        if (!ctx || !m_ || !gray_images || !shapes_inout || batch < 0 || W < 1 || H < 1)
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_sdm_optimize_batch: bad argument");
        if (batch == 0) return;
        fd_sdm_ticket t;
        sdm_begin(ctx, const_cast<fd_sdm*>(m_), t, gray_images, W, H, batch, images_on_device, shapes_inout, nullptr, ctx->stream, ctx->kernel_timing);
        sdm_end(ctx, t, shapes_inout, status_out);
    });
}

}  // extern "C"
