// featuredetection_amd/csrc/pyramid.hip -- image pyramid on the GPU (gfx950).
//
// Restates imageprocessing::ImagePyramid (ImagePyramid.cpp:67-92,116-128,170-198) with the
// GrayscaleFilter image filter and optional layer filters.  All arithmetic is the integer-exact
// OpenCV 2.4 fixed-point arithmetic (cvtColor, resize INTER_LINEAR, pyrDown, Sobel), so layers are
// bit-identical to the CPU path.  Layout in HBM: ONE arena per pyramid holding the full-resolution
// gray image, every computed layer (kept or only a pyrDown source) and the filtered kept layers,
// each 256-byte aligned, rows densely packed.  The whole pyramid of a 640x480 frame is ~1.6 MB and
// stays L2/MALL resident for the scoring kernels that follow.
//
// HBM-bound stage: algorithmic bytes = 3WH (BGR read) + WH (gray write) + per layer (src read +
// dst write); see DESIGN.md.  One launch covers all chains of a pyramid depth (blockIdx.y = chain).
#include "fd_internal.hpp"
#include <algorithm>
#include <cstring>

namespace {

constexpr int MAXJ = 16;

struct ResizeJob {
    int dw, dh;
    uint32_t dst_off;
    double scale_x, scale_y;
};
struct ResizeJobs {
    int n;
    ResizeJob j[MAXJ];
};
struct DownJob {
    int sw, sh;
    uint32_t src_off, dst_off;
};
// one first-octave chain of k_resize_down: cv::resize of the frame to dw0 x dh0 and its pyrDown to dw1 x dh1
struct FusedJob {
    int dw0, dh0, dw1, dh1;
    uint32_t dst0_off;      // resized layer, written only when it is a kept layer (0xffffffff: stays in LDS)
    uint32_t dst1_off;      // its pyrDown
    uint32_t xtab, ytab;    // offsets (in int2 entries) into the coordinate tables
};
struct FusedJobs {
    int n;
    FusedJob j[MAXJ];
};
struct DownJobs {
    int n;
    int tile0[MAXJ + 1];   // k_pyrdown_tiled: first tile of every job in the launch's flat tile list (exact grids: no empty workgroups)
    DownJob j[MAXJ];
};
struct FilterJob {
    int w, h;
    uint32_t src_off, dst_off;
};
struct FilterJobs {
    int n;
    FilterJob j[MAXJ];
};

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}
// BORDER_REFLECT_101 for a coordinate at most one reflection away; anything further out is clamped (the tiled kernels only meet
// such coordinates under outputs that lie outside the layer and are not stored)
__device__ __forceinline__ int reflect1(int p, int len) {
    if (len < 3) return reflect101(p, len);   // the 5-tap halo reaches two pixels out: one reflection needs three pixels
    const int q = p < 0 ? -p : (p >= len ? 2 * len - 2 - p : p);
    return min(max(q, 0), len - 1);
}

// cv::cvtColor(BGR2GRAY), 8U: (B*1868 + G*9617 + R*4899 + 8192) >> 14
struct FramePtrs {
    const uint8_t* p[FD_MAX_FRAMES];
};
// 24-bit multiplies by name: __umul24 became v_and + v_mul_lo_u32 (quarter rate) wherever the compiler could not see the operand ranges
__device__ __forceinline__ unsigned int mul24(unsigned int a, unsigned int b) {
    unsigned int r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned int mul24_s(unsigned int a_uniform, unsigned int b) {   // a wave-uniform factor stays in its SGPR
    unsigned int r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "s"(a_uniform), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned int mad24(unsigned int a, unsigned int b, unsigned int c) {
    unsigned int r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned int mulhi24_s(unsigned int a_uniform, unsigned int b) {   // (a * b) >> 32 of two 24-bit factors
    unsigned int r;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "s"(a_uniform), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t ld_u32_unaligned(const uint8_t* p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint32_t gray_of(uint32_t b, uint32_t g, uint32_t r) { return (b * 1868u + g * 9617u + r * 4899u + 8192u) >> 14; }

// the frames of a multi-frame pyramid: BGR -> gray (ch == 3) or copy (ch == 1) into frame blockIdx.y's arena.  Four pixels per
// thread: three dword loads, one dword store (the stage is bound by the number of memory instructions, not by their bytes).
__global__ void k_frames_to_gray(FramePtrs frames, uint8_t* __restrict__ grayBase, size_t imageStride, int n, int ch) {
    const uint8_t* __restrict__ src = frames.p[blockIdx.y];
    uint8_t* __restrict__ gray = grayBase + (size_t)blockIdx.y * imageStride;
    const int nq = n >> 2;
    const bool aligned = ((uintptr_t)src & 3) == 0;
    const int stride = gridDim.x * blockDim.x;
    auto load3 = [&](int q, uint32_t (&w)[3]) {
        if (aligned) {
            const uint32_t* s3 = reinterpret_cast<const uint32_t*>(src) + 3 * (size_t)q;
            w[0] = s3[0]; w[1] = s3[1]; w[2] = s3[2];
        } else {
            const uint8_t* s1 = src + 12 * (size_t)q;
            w[0] = ld_u32_unaligned(s1); w[1] = ld_u32_unaligned(s1 + 4); w[2] = ld_u32_unaligned(s1 + 8);
        }
    };
    auto gray4 = [&](const uint32_t (&w)[3]) {
        // bytes: B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
        const uint32_t g0 = gray_of(w[0] & 255u, (w[0] >> 8) & 255u, (w[0] >> 16) & 255u);
        const uint32_t g1 = gray_of(w[0] >> 24, w[1] & 255u, (w[1] >> 8) & 255u);
        const uint32_t g2 = gray_of((w[1] >> 16) & 255u, w[1] >> 24, w[2] & 255u);
        const uint32_t g3 = gray_of((w[2] >> 8) & 255u, (w[2] >> 16) & 255u, w[2] >> 24);
        return g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
    };
    // GQ quads per thread and pass, their loads in flight together: with one 12-byte load per thread the stage moved 4.5 TB/s -- the
    // bytes a full chip of wavefronts keeps in flight (6 MB) over the memory latency -- not what the memory can deliver
    constexpr int GQ = 4;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += GQ * stride) {
        if (ch == 3) {
            uint32_t w[GQ][3];
#pragma unroll
            for (int i = 0; i < GQ; ++i) load3(min(q + i * stride, nq - 1), w[i]);
#pragma unroll
            for (int i = 0; i < GQ; ++i)
                if (q + i * stride < nq) reinterpret_cast<uint32_t*>(gray)[q + i * stride] = gray4(w[i]);
        } else {
            uint32_t w[GQ];
#pragma unroll
            for (int i = 0; i < GQ; ++i) w[i] = ld_u32_unaligned(src + 4 * (size_t)min(q + i * stride, nq - 1));
#pragma unroll
            for (int i = 0; i < GQ; ++i)
                if (q + i * stride < nq) reinterpret_cast<uint32_t*>(gray)[q + i * stride] = w[i];
        }
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < (n & 3)) {   // tail
        const int i = (nq << 2) + threadIdx.x;
        gray[i] = ch == 3 ? (uint8_t)gray_of(src[3 * (size_t)i], src[3 * (size_t)i + 1], src[3 * (size_t)i + 2]) : src[i];
    }
}

__global__ void k_bgr2gray(const uint8_t* __restrict__ bgr, uint8_t* __restrict__ gray, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int stride = gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        int b = bgr[3 * (size_t)i], g = bgr[3 * (size_t)i + 1], r = bgr[3 * (size_t)i + 2];
        gray[i] = (uint8_t)((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14);
    }
}

// cv::resize INTER_LINEAR 8UC1 -> all first-octave layers from the full-resolution gray image.  A thread owns one destination
// column of a band of RS_ROWS rows: the horizontal coordinates / fixed-point weights are computed once per thread, the vertical
// ones are workgroup-uniform (scalar); per pixel that leaves four byte loads and a dozen integer operations.
constexpr int RS_ROWS = 8;
__global__ __launch_bounds__(256) void k_resize_linear(const uint8_t* __restrict__ arena, uint8_t* __restrict__ out, uint32_t src_off, int sw,
                                                       int sh, ResizeJobs jobs, size_t imageStride) {
    const ResizeJob jb = jobs.j[blockIdx.y];
    const uint8_t* src = arena + (size_t)blockIdx.z * imageStride + src_off;   // blockIdx.z = frame of a multi-frame pyramid
    uint8_t* dst = out + (size_t)blockIdx.z * imageStride + jb.dst_off;
    const int colBlocks = (jb.dw + 255) / 256, rowBands = (jb.dh + RS_ROWS - 1) / RS_ROWS;
    for (int t = blockIdx.x; t < colBlocks * rowBands; t += gridDim.x) {
        const int band = t / colBlocks, cb = t - band * colBlocks;
        const int dx = cb * 256 + threadIdx.x;
        if (dx >= jb.dw) continue;
        float fx = (float)((dx + 0.5) * jb.scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        const int a0 = __float2int_rn((1.f - fx) * 2048), a1 = __float2int_rn(fx * 2048);
        const int sx1 = sx + 1 < sw ? sx + 1 : sx;
        const int dyEnd = min(jb.dh, (band + 1) * RS_ROWS);
        for (int dy = band * RS_ROWS; dy < dyEnd; ++dy) {
            float fy = (float)((dy + 0.5) * jb.scale_y - 0.5);
            int sy = (int)floorf(fy);
            fy -= sy;
            const int b0 = __float2int_rn((1.f - fy) * 2048), b1 = __float2int_rn(fy * 2048);
            const int y0 = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
            const int y1 = sy + 1 < 0 ? 0 : (sy + 1 >= sh ? sh - 1 : sy + 1);
            const uint8_t* S0 = src + (size_t)y0 * sw;
            const uint8_t* S1 = src + (size_t)y1 * sw;
            const int r0 = S0[sx] * a0 + S0[sx1] * a1;
            const int r1 = S1[sx] * a0 + S1[sx1] * a1;
            dst[(size_t)dy * jb.dw + dx] = (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
        }
    }
}

// cv::pyrDown 8UC1: separable [1 4 6 4 1], (sum + 128) >> 8, BORDER_REFLECT_101.  A thread computes PD_ROWS vertically adjacent
// outputs of one column: the 2 * PD_ROWS + 3 horizontal 5-tap sums it needs are formed once each (55 loads for 4 outputs instead
// of 100).
constexpr int PD_ROWS = 4;
__global__ __launch_bounds__(256) void k_pyrdown(uint8_t* __restrict__ arena, DownJobs jobs, size_t imageStride) {
    const DownJob jb = jobs.j[blockIdx.y];
    const int dw = (jb.sw + 1) / 2, dh = (jb.sh + 1) / 2;
    arena += (size_t)blockIdx.z * imageStride;   // blockIdx.z = frame of a multi-frame pyramid
    const uint8_t* src = arena + jb.src_off;
    uint8_t* dst = arena + jb.dst_off;
    const int bands = (dh + PD_ROWS - 1) / PD_ROWS;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dw * bands; i += gridDim.x * blockDim.x) {
        const int band = i / dw, x = i - band * dw;
        const int y0 = band * PD_ROWS;
        int xs[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) xs[k] = reflect101(2 * x - 2 + k, jb.sw);
        int h[2 * PD_ROWS + 3];
#pragma unroll
        for (int r = 0; r < 2 * PD_ROWS + 3; ++r) {
            const uint8_t* S = src + (size_t)reflect101(2 * y0 - 2 + r, jb.sh) * jb.sw;
            h[r] = S[xs[2]] * 6 + (S[xs[1]] + S[xs[3]]) * 4 + S[xs[0]] + S[xs[4]];
        }
#pragma unroll
        for (int j = 0; j < PD_ROWS; ++j) {
            if (y0 + j < dh) {
                const int v = h[2 * j + 2] * 6 + (h[2 * j + 1] + h[2 * j + 3]) * 4 + h[2 * j] + h[2 * j + 4];
                dst[(size_t)(y0 + j) * dw + x] = (uint8_t)((v + 128) >> 8);
            }
        }
    }
}

// ---- LDS-tiled versions of the two kernels above -------------------------------------------------------------------------
// The direct kernels issue four (resize) / 14 (pyrDown) byte loads per output pixel and are bound by the number of memory
// instructions.  Here a workgroup stages the source rectangle of a 64 x 32 output tile in LDS with dword loads (unaligned where
// the row start is) and every tap is an LDS byte read; the arithmetic is the same, value for value.
constexpr int TL_W = 64, TL_H = 32, TL_PITCH = 136;   // 64 x 32 output pixels per tile, 8 rows per thread: the per-thread column set-up is the
                                                        // larger part of the work of a row
constexpr int PD_TH = 16;                               // pyrDown: 64 x 16 tiles (its layers are small: 32-row tiles leave CUs idle)
constexpr int TL_ROWS = 72;                             // resize: 31 * 2.05 + 3 source rows; pyrDown: 2 * 32 + 3

// resize: valid while a tile's source rectangle fits the stage from its conservative origin, i.e. 1 <= scale_x, scale_y <= 2.05
// (first-octave layers: 1 <= scale < 2)
__global__ __launch_bounds__(256) void k_resize_tiled(const uint8_t* __restrict__ arena, uint8_t* __restrict__ out, uint32_t src_off, int sw,
                                                      int sh, ResizeJobs jobs, size_t imageStride) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[TL_ROWS * TL_PITCH];
    __shared__ int4 rowTab[TL_H];   // per dst row of the tile: LDS offsets of its two source rows, vertical weights
    const ResizeJob jb = jobs.j[blockIdx.y];
    const uint8_t* src = arena + (size_t)blockIdx.z * imageStride + src_off;   // blockIdx.z = frame of a multi-frame pyramid
    uint8_t* dst = out + (size_t)blockIdx.z * imageStride + jb.dst_off;
    const int tilesX = (jb.dw + TL_W - 1) / TL_W, tilesY = (jb.dh + TL_H - 1) / TL_H;
    auto srcX = [&](int dx, float& fx) {   // cv::resize: left source column of dst column dx and its fraction
        fx = (float)((dx + 0.5) * jb.scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        return sx;
    };
    auto srcY = [&](int dy, float& fy) {
        fy = (float)((dy + 0.5) * jb.scale_y - 0.5);
        const int sy = (int)floorf(fy);
        fy -= sy;
        return sy;
    };
    auto clampY = [&](int y) { return y < 0 ? 0 : (y >= sh ? sh - 1 : y); };
    const int c = threadIdx.x & 63, rq = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // rq scalar: row offsets on the scalar unit
    for (int t = blockIdx.x; t < tilesX * tilesY; t += gridDim.x) {
        const int ty = t / tilesX, tx = t - ty * tilesX;
        const int dx0 = tx * TL_W, dy0 = ty * TL_H;
        // source rectangle of the tile: a conservative origin (one below the smallest coordinate any of its pixels can ask for:
        // sx >= floor(dx * scale) for scale >= 1) and the full stage, TL_ROWS x TL_PITCH bytes -- cheaper than the exact extents,
        // which need the fp64 coordinate of cv::resize per tile and thread
        const int X0 = max(0, (int)((float)dx0 * (float)jb.scale_x) - 1), Y0 = max(0, (int)((float)dy0 * (float)jb.scale_y) - 1);
        {   // lane == dword of a source row (34 per row), wave + 4 k == row: no index arithmetic, all loads of a thread in flight together
            const int x = X0 + 4 * c;
            const bool colok = c < TL_PITCH / 4;
            uint32_t v[TL_ROWS / 4];
#pragma unroll
            for (int k = 0; k < TL_ROWS / 4; ++k) {
                const int sy = min(Y0 + rq + 4 * k, sh - 1);
                v[k] = 0;
                if (colok && x + 3 < sw) v[k] = ld_u32_unaligned(src + (uint32_t)(sy * sw) + x);
            }
            if (colok && x + 3 >= sw) {   // right edge of the image: never past the row
#pragma unroll 1
                for (int k = 0; k < TL_ROWS / 4; ++k) {
                    const uint8_t* row = src + (uint32_t)(min(Y0 + rq + 4 * k, sh - 1) * sw);
                    uint32_t w = 0;
                    for (int b = 0; b < 4; ++b) w |= (uint32_t)row[min(x + b, sw - 1)] << (8 * b);
                    v[k] = w;
                }
            }
            if (colok) {
#pragma unroll
                for (int k = 0; k < TL_ROWS / 4; ++k) *reinterpret_cast<uint32_t*>(&tile[(rq + 4 * k) * TL_PITCH + 4 * c]) = v[k];
            }
        }
        if (threadIdx.x < TL_H) {   // vertical taps of the tile's rows, once per tile instead of once per thread and row
            float fy;
            const int sy = srcY(dy0 + threadIdx.x, fy);
            rowTab[threadIdx.x] = make_int4((clampY(sy) - Y0) * TL_PITCH, (clampY(sy + 1) - Y0) * TL_PITCH, __float2int_rn((1.f - fy) * 2048),
                                            __float2int_rn(fy * 2048));
        }
        __syncthreads();
        const int dx = dx0 + c;
        if (dx < jb.dw) {
            float fx;
            const int sx = srcX(dx, fx);
            const int a0 = __float2int_rn((1.f - fx) * 2048), a1 = __float2int_rn(fx * 2048);
            const int lx = sx - X0, lx1 = (sx + 1 < sw ? sx + 1 : sx) - X0;
            // 24-bit multiplies (full rate; every factor is below 2^16) and a running destination pointer: the 32-bit multiplies /
            // 64-bit multiply-adds the plain expressions compile to run at a quarter of the rate and made this kernel VALU-bound
            uint8_t* dp = dst + (uint32_t)((dy0 + rq * (TL_H / 4)) * jb.dw) + dx;
#pragma unroll
            for (int k = 0; k < TL_H / 4; ++k, dp += jb.dw) {
                const int rr = rq * (TL_H / 4) + k;
                if (dy0 + rr < jb.dh) {
                    const int4 rt = rowTab[rr];
                    const uint8_t* S0 = tile + rt.x;
                    const uint8_t* S1 = tile + rt.y;
                    const unsigned int r0 = __umul24(S0[lx], a0) + __umul24(S0[lx1], a1);
                    const unsigned int r1 = __umul24(S1[lx], a0) + __umul24(S1[lx1], a1);
                    *dp = (uint8_t)(((__umul24(rt.z, r0 >> 4) >> 16) + (__umul24(rt.w, r1 >> 4) >> 16) + 2) >> 2);
                }
            }
        }
        __syncthreads();
    }
}

// ---- cv::resize of the frame + the first cv::pyrDown of the result in one kernel ---------------------------------------------
// Most first-octave layers of a detection pyramid are not kept themselves (FaceFrontal keeps scales 0.05 .. 0.16): they only exist as the
// source of their pyrDown chain, and writing 1.2 Mpixels per frame to memory only to read them back in the next launch is what the
// pyramid stage spent its time on.  A workgroup computes the resized pixels under one 62 x 16 tile of the pyrDown layer (127 x 35, the
// 5-tap halo included; BORDER_REFLECT_101 columns / rows are the resized pixels at the reflected coordinates) into LDS and takes the
// pyrDown from there; the resized layer goes to memory only when it is a kept layer.  Same integer arithmetic as k_resize_tiled /
// k_pyrdown_tiled, value for value (ImagePyramid.cpp:177,186), with two savings per resized pixel:
//   * the cv::resize coordinates and fixed-point weights come from per-layer tables built on the host with the kernel's own float
//     expressions (xtab[dx] = {left source column, a0 | a1 << 16}, ytab[dy] = {y0 | y1 << 16, b0 | b1 << 16});
//   * a thread walks down one column and keeps the horizontally interpolated value of the last two source rows: consecutive
//     destination rows share them (1 <= scale < 2), so a destination pixel costs ~1.4 horizontal interpolations instead of 2.
// -DFD_PYR_PROF (tools/build_prof_lib.sh, tools/pyr_phases.py): ticks thread 0 of every workgroup spends in the phases of k_resize_down
#ifdef FD_PYR_PROF
constexpr int PYR_PROF_WGS = 65536;
__device__ unsigned long long fd_pyr_prof[PYR_PROF_WGS * 8];   // one record per workgroup: same-address atomics would serialise the launch
#define PYR_T(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#else
#define PYR_T(x)
#endif
constexpr int FT_W1 = 62, FT_H1 = 16;                    // pyrDown tile
constexpr int G0_W = 2 * FT_W1 + 3, G0_H = 2 * FT_H1 + 3, G0_PITCH = 128;   // resized pixels under it: 127 x 35
constexpr int FS_PITCH = 256, FS_ROWS = 76;             // source stage: 126 * 2.0 + 3 columns (64 dwords: one per lane), 34 * 2.05 + 3 rows
// Persistent workgroups walk a host-built list of tiles ({source rectangle, tile position, chain} per entry: the layout is static) and keep
// the NEXT tile's global loads -- ~20 source dwords, the column's xtab entry, a row's ytab entry per thread -- in flight in registers
// while they resize the current one: a tile used to spend 45 % of its 7.6 us waiting for them (in-kernel timestamps, tools/pyr_phases.py).
// Multi-frame pyramids: workgroup b runs on XCD b % 8 and takes the frames b % 8, b % 8 + 8, ...: a frame's gray image is read by one L2.
constexpr int FS_LOADS = (FS_ROWS + 3) / 4;   // source rows per wavefront: wavefront w holds rows w, w + 4, ..., a lane one dword of each
struct FusedFetch {   // what a thread holds for a tile before it is in LDS
    uint32_t v[FS_LOADS];
    int2 ex, ey;   // ex: thread = column of the tile; ey: lane = resized row of the tile (every wavefront its own copy)
};
__device__ __forceinline__ void fused_issue(FusedFetch& f, const uint8_t* __restrict__ src, int sw, const int2* __restrict__ tabs, const FusedJob& jb,
                                            const int4 d) {
    const int X0 = d.x & 0xffff, Y0 = (int)((uint32_t)d.x >> 16), ncol = d.y & 0xffff, nrow = (int)((uint32_t)d.y >> 16);
    const int gx0 = 2 * (d.z & 0xffff) * FT_W1 - 2;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // A dword that hangs over the right edge of the image takes its last bytes from the next row (or, in the last row, from the arena
    // behind the gray image): they stand for columns >= sw, which no tap reads (the last column's right neighbour has weight 0).
    // No predication: lanes right of the rectangle and rows below it load its last dword / last row again (valid addresses, values
    // nobody reads) -- with a test per load the 19 loads were 19 basic blocks of exec-mask bookkeeping, 400 instructions per tile.
    const uint32_t voff = (uint32_t)(4 * min(lane, (ncol - 1) >> 2) + X0);
    const int rlast = nrow - 1;
#pragma unroll
    for (int q = 0; q < FS_LOADS; ++q) {
        const int r = min(wave + 4 * q, rlast);   // scalar
        f.v[q] = ld_u32_unaligned(src + (uint32_t)((Y0 + r) * sw) + voff);
    }
    f.ex = tabs[jb.xtab + reflect101(gx0 + (int)(threadIdx.x & 127), jb.dw0)];
    {   // BORDER_REFLECT_101 of the resized rows the kept pyrDown rows reach (one reflection); rows further out only feed pyrDown
        // rows past the layer's end, any valid row will do for them
        const int py = 2 * (int)((uint32_t)d.z >> 16) * FT_H1 - 2 + min(lane, G0_H - 1), pr = py < 0 ? -py : (py >= jb.dh0 ? 2 * jb.dh0 - 2 - py : py);
        f.ey = tabs[jb.ytab + min(max(pr, 0), jb.dh0 - 1)];
    }
}

__global__ __launch_bounds__(256) void k_resize_down(uint8_t* __restrict__ arena0, uint32_t src_off, int sw, int sh, const int2* __restrict__ tabs,
                                                     FusedJobs jobs, uint32_t tileTab, int tilesPerFrame, int nimg, size_t imageStride) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[FS_ROWS * FS_PITCH];
    __shared__ __attribute__((aligned(16))) uint8_t g0[G0_H * G0_PITCH];
    const int4* tiles = reinterpret_cast<const int4*>(tabs + tileTab);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool byXcd = nimg >= 8 && (gridDim.x & 7u) == 0;
    const int xcd = blockIdx.x & 7;
    const int slot = byXcd ? (int)(blockIdx.x >> 3) : (int)blockIdx.x, nslot = byXcd ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    const int items = (byXcd ? (nimg - xcd + 7) >> 3 : nimg) * tilesPerFrame;
    auto frame_of = [&](int item) { const int fi = item / tilesPerFrame; return byXcd ? xcd + 8 * fi : fi; };
    auto desc_of = [&](int item) { return item < items ? tiles[item % tilesPerFrame] : make_int4(0, 0, 0, 0); };

    int item = slot;
    if (item >= items) return;
    int4 d = desc_of(item);
    FusedFetch f;
    fused_issue(f, arena0 + (size_t)frame_of(item) * imageStride + src_off, sw, tabs, jobs.j[d.w], d);
    int itemN = item + nslot;
    int4 dN = desc_of(itemN);

#ifdef FD_PYR_PROF
    unsigned long long pacc[6] = {0, 0, 0, 0, 0, 0}, ptiles = 0;
#endif
    for (; item < items; item = itemN, d = dN, itemN += nslot, dN = desc_of(itemN)) {
        PYR_T(p0);
        const FusedJob jb = jobs.j[d.w];
        uint8_t* arena = arena0 + (size_t)frame_of(item) * imageStride;
        const int X0 = d.x & 0xffff, Y0 = (int)((uint32_t)d.x >> 16);
        const int x1 = (d.z & 0xffff) * FT_W1, y1 = (int)((uint32_t)d.z >> 16) * FT_H1;   // first pyrDown pixel of the tile
        const int gx0 = 2 * x1 - 2, gy0 = 2 * y1 - 2;                      // resized pixel of tile entry (0, 0), before the border reflection
        {   // the fetched source rectangle and row table -> LDS
            // every lane stores its dword of every row (the stage has a dword per lane and FS_ROWS rows; what lies outside the
            // rectangle is never read)
            static_assert(4 * (FS_LOADS - 1) + 3 < FS_ROWS && FS_PITCH == 256, "stage holds a dword per lane for rows wave + 4 q");
#pragma unroll
            for (int q = 0; q < FS_LOADS; ++q) *reinterpret_cast<uint32_t*>(&stage[(wave + 4 * q) * FS_PITCH + 4 * lane]) = f.v[q];
        }
        const int2 ex = f.ex, ey = f.ey;
        PYR_T(p1);
        __syncthreads();
        PYR_T(p2);
        if (itemN < items)   // the next tile's loads fly while this one is resized
            fused_issue(f, arena0 + (size_t)frame_of(itemN) * imageStride + src_off, sw, tabs, jobs.j[dN.w], dN);
        PYR_T(p3);
        {   // ---- resize: thread = one column of the tile, half of its rows.  A row's vertical taps (ytab) are the same for the whole
            //      wavefront (fetched with the tile, lane = row; v_readlane): scalar address arithmetic; every row reads its four source bytes whether or not the
            //      previous row shared one (reusing them saved 0.6 interpolations per pixel but chained every row behind an LDS round trip)
            const int c = threadIdx.x & 127, half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);
            if (c < G0_W) {
                // the right neighbour is the next byte except in the image's last column, where its weight a1 is 0
                const uint8_t* sp = stage + (ex.x - X0);
                const unsigned int a0 = ex.y & 0xffff, a1 = (unsigned int)ex.y >> 16;
                uint8_t* gp = g0 + c;
                // A row's record in ONE register (lane == row): stage row of y0 (7 bits), b0, b1 (12 bits each).  The lower tap's row is the
                // next stage row -- the table builder only fuses layers where y1 == y0 + 1 or b1 == 0 (every down-scaling layer) -- so a row
                // costs one v_readlane, a handful of scalar instructions and one address add (round 5: two v_readlane, nine scalar
                // instructions and two address computations per row and wavefront: as many scalar as vector instructions).
                const unsigned int rowRec = (unsigned int)((ey.x & 0xffff) - Y0) | (((unsigned int)ey.y & 0xfffu) << 7) | ((((unsigned int)ey.y >> 16) & 0xfffu) << 19);
                // three rows at a time: their twelve byte loads first, then the arithmetic (one LDS round trip per three rows).  The 35 rows
                // are dealt as 0..17 / 17..34: row 17 is computed by both halves (the same value, stored twice) and nothing is predicated
                const int rBase = half * (G0_H - 18);
#pragma unroll 2
                for (int i = 0; i < 18; i += 3) {
                    unsigned int s00[3], s01[3], s10[3], s11[3], b0s[3], b1s[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const int r = rBase + i + q;
                        const unsigned int e = (unsigned int)__builtin_amdgcn_readlane((int)rowRec, r);
                        b0s[q] = ((e >> 7) & 0xfffu) << 12;   // b << 12 (<= 2^23): (b << 12) * (h & ~15) >> 32 == (b * (h >> 4)) >> 16
                        b1s[q] = (e >> 19) << 12;
                        const uint8_t* sr = sp + ((e & 127u) << 8);
                        s00[q] = sr[0]; s01[q] = sr[1]; s10[q] = sr[FS_PITCH]; s11[q] = sr[FS_PITCH + 1];
                    }
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const unsigned int h0 = mad24(s01[q], a1, mul24(s00[q], a0)) & ~15u;   // cv::resize's horizontal intermediate, times 16
                        const unsigned int h1 = mad24(s11[q], a1, mul24(s10[q], a0)) & ~15u;
                        const unsigned int v = (mulhi24_s(b0s[q], h0) + mulhi24_s(b1s[q], h1) + 2) >> 2;
                        gp[(rBase + i + q) * G0_PITCH] = (uint8_t)v;
                    }
                }
            }
        }
        PYR_T(p4);
        __syncthreads();
        PYR_T(p5);
        {   // ---- pyrDown of the tile (k_pyrdown_tiled's arithmetic on the LDS copy)
            const int c1 = lane;
            const int x = x1 + c1;
            if (c1 < FT_W1 && x < jb.dw1) {
                constexpr int PR = FT_H1 / 4;   // output rows per thread
                const uint8_t* T = g0 + (2 * wave * PR) * G0_PITCH + 2 * c1;
                int h[2 * PR + 3];
#pragma unroll
                for (int r = 0; r < 2 * PR + 3; ++r) {
                    const uint8_t* S = T + r * G0_PITCH;
                    const uint32_t p01 = *reinterpret_cast<const uint16_t*>(S), p23 = *reinterpret_cast<const uint16_t*>(S + 2);
                    h[r] = (int)__builtin_amdgcn_udot4(p01 | (p23 << 16), 0x04060401u, (uint32_t)S[4], false);
                }
                uint8_t* dp = arena + jb.dst1_off + (uint32_t)((y1 + wave * PR) * jb.dw1) + x;
#pragma unroll
                for (int j = 0; j < PR; ++j, dp += jb.dw1) {
                    if (y1 + wave * PR + j < jb.dh1) {
                        const int v = h[2 * j + 2] * 6 + (h[2 * j + 1] + h[2 * j + 3]) * 4 + h[2 * j] + h[2 * j + 4];
                        *dp = (uint8_t)((v + 128) >> 8);
                    }
                }
            }
        }
        if (jb.dst0_off != 0xffffffffu) {   // the resized layer is a kept layer: the tile's own 124 x 32 pixels of it
            uint8_t* d0 = arena + jb.dst0_off;
            for (int i = threadIdx.x; i < 32 * 128; i += 256) {
                const int r = 2 + (i >> 7), c = 2 + (i & 127);
                const int gx = gx0 + c, gy = gy0 + r;
                if (c < 2 + 2 * FT_W1 && gx < jb.dw0 && gy < jb.dh0) d0[(uint32_t)(gy * jb.dw0) + gx] = g0[r * G0_PITCH + c];
            }
        }
#ifdef FD_PYR_PROF
        {
            const unsigned long long p6 = __builtin_amdgcn_s_memtime();
            pacc[0] += p1 - p0; pacc[1] += p2 - p1; pacc[2] += p3 - p2; pacc[3] += p4 - p3; pacc[4] += p5 - p4; pacc[5] += p6 - p5; ++ptiles;
        }
#endif
        // no barrier here: the next tile's stage is free since the barrier after the resize, and its resize writes g0 only behind the
        // next barrier, which every wavefront reaches after its pyrDown reads above
    }
#ifdef FD_PYR_PROF
    if (threadIdx.x == 0 && blockIdx.x < (unsigned int)PYR_PROF_WGS) {
        for (int i = 0; i < 6; ++i) fd_pyr_prof[blockIdx.x * 8 + i] = pacc[i];
        fd_pyr_prof[blockIdx.x * 8 + 6] = ptiles;
    }
#endif
}

// BORDER_REFLECT_101 without branches for the coordinates a 5-tap pyrDown USES (two pixels out on either side): |p|, one reflection
// at the far end, then a clamp.  Exact for every len >= 1 there (len 1: everything is 0; len 2: -2 -> 0, -1 -> 1, 2 -> 0); coordinates
// further out only feed outputs outside the layer and may be anything valid.
__device__ __forceinline__ int reflect_cf(int p, int len) {
    int q = p < 0 ? -p : p;
    q = q >= len ? 2 * len - 2 - q : q;
    return min(max(q, 0), len - 1);
}
constexpr int PD_W = 62;   // pyrDown tile: 62 x 16 outputs = 127 x 35 source bytes = 32 dwords per row: two rows per wavefront load
__global__ __launch_bounds__(256) void k_pyrdown_tiled(uint8_t* __restrict__ arena, DownJobs jobs, size_t imageStride) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[TL_ROWS * TL_PITCH];
    arena += (size_t)blockIdx.z * imageStride;   // blockIdx.z = frame of a multi-frame pyramid
    const int c = threadIdx.x & 63, rq = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // rq scalar: row offsets on the scalar unit
    constexpr int NROWS = 2 * PD_TH + 3, PD_LD = (NROWS + 7) / 8;   // 35 source rows; a wavefront stages rows 2 (rq + 4 k) and the next
    static_assert(2 * (3 + 4 * (PD_LD - 1)) + 1 < TL_ROWS && 128 <= TL_PITCH && 2 * PD_W + 3 <= 128, "pyrDown stage");
    for (int g = blockIdx.x; g < jobs.tile0[jobs.n]; g += gridDim.x) {   // flat list of the tiles of all jobs
        int ji = 0;
        while (ji + 1 < jobs.n && g >= jobs.tile0[ji + 1]) ++ji;
        const DownJob jb = jobs.j[ji];
        const int t = g - jobs.tile0[ji];
        const int sw = jb.sw, sh = jb.sh;
        const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
        const uint8_t* src = arena + jb.src_off;
        uint8_t* dst = arena + jb.dst_off;
        const int tilesX = (dw + PD_W - 1) / PD_W;
        const int ty = t / tilesX, tx = t - ty * tilesX;
        const int dx0 = tx * PD_W, dy0 = ty * PD_TH;
        const int X0 = 2 * dx0 - 2, Y0 = 2 * dy0 - 2;
        {   // stage the source rectangle: lane = dword j of row 2 (rq + 4 k) + (lane >> 5).  No tests in the common path: the row is
            // reflected in closed form, the column clamped into the row; only tiles on the left / right edge of the layer rebuild the
            // dwords that hang over it from reflected bytes.  (Round 3 staged a flat element list -- a division by 33, two range
            // tests and a loop-form reflection per element: 170 branches in the kernel, the staging was most of a tile's time.)
            const int j4 = 4 * (c & 31), half = c >> 5;
            const int xs = X0 + j4;
            const int xc = min(max(xs, 0), max(sw - 4, 0));
            const uint32_t lastDw = (uint32_t)max(sw * sh - 4, 0);
            uint32_t v[PD_LD];
#pragma unroll
            for (int k = 0; k < PD_LD; ++k) {
                const int r = 2 * (rq + 4 * k) + half;
                v[k] = ld_u32_unaligned(src + min((uint32_t)(reflect_cf(Y0 + r, sh) * sw + xc), lastDw));
            }
            if (X0 < 0 || X0 + 128 > sw) {   // wave-uniform: an edge tile
                if (!(xs >= 0 && xs + 3 < sw)) {
#pragma unroll
                    for (int k = 0; k < PD_LD; ++k) {
                        const int r = 2 * (rq + 4 * k) + half;
                        const uint8_t* row = src + (uint32_t)(reflect_cf(Y0 + r, sh) * sw);
                        uint32_t w = 0;
#pragma unroll
                        for (int b = 0; b < 4; ++b) w |= (uint32_t)row[reflect_cf(xs + b, sw)] << (8 * b);
                        v[k] = w;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < PD_LD; ++k) *reinterpret_cast<uint32_t*>(&tile[(2 * (rq + 4 * k) + half) * TL_PITCH + j4]) = v[k];
        }
        __syncthreads();
        const int x = dx0 + c;
        if (c < PD_W && x < dw) {
            constexpr int PR = PD_TH / 4;   // output rows per thread
            const uint8_t* T = tile + (2 * rq * PR) * TL_PITCH + 2 * c;
            int h[2 * PR + 3];
#pragma unroll
            for (int r = 0; r < 2 * PR + 3; ++r) {
                const uint8_t* S = T + r * TL_PITCH;
                // [1 4 6 4] . (s0 s1 s2 s3) + s4 as one byte dot product (v_dot4_u32_u8): exact integers, a third of the instructions
                const uint32_t p01 = *reinterpret_cast<const uint16_t*>(S), p23 = *reinterpret_cast<const uint16_t*>(S + 2);
                h[r] = (int)__builtin_amdgcn_udot4(p01 | (p23 << 16), 0x04060401u, (uint32_t)S[4], false);
            }
            uint8_t* dp = dst + (uint32_t)((dy0 + rq * PR) * dw) + x;
#pragma unroll
            for (int j = 0; j < PR; ++j, dp += dw) {
                if (dy0 + rq * PR + j < dh) {
                    const int v = h[2 * j + 2] * 6 + (h[2 * j + 1] + h[2 * j + 3]) * 4 + h[2 * j] + h[2 * j + 4];
                    *dp = (uint8_t)((v + 128) >> 8);
                }
            }
        }
        __syncthreads();
    }
}

// cv::Sobel derivatives of GradientFilter (GradientFilter.cpp:16-59; taps of getDerivKernels, see oracle/orc_image.cpp): ksize 1, 3
// take the short forms; 5, 7 and CV_SCHARR (-1) the separable sums.  Returns the scale 1 / 2^(2 ksize - 3) (1/2, 1/32 for ksize 1 /
// Scharr).  Everything is an exact integer; 127 + scale * g is exact in float.
__device__ __forceinline__ float grad_pair(const uint8_t* __restrict__ src, int w, int h, int x, int y, int ksize, int& gx, int& gy) {
    if (ksize == 1 || ksize == 3) {
        const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
        const int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
        const uint8_t *S0 = src + (size_t)ym * w, *S1 = src + (size_t)y * w, *S2 = src + (size_t)yp * w;
        if (ksize == 1) {
            gx = S1[xp] - S1[xm];
            gy = S2[x] - S0[x];
            return 0.5f;
        }
        gx = (S0[xp] - S0[xm]) + 2 * (S1[xp] - S1[xm]) + (S2[xp] - S2[xm]);
        gy = (S2[xm] - S0[xm]) + 2 * (S2[x] - S0[x]) + (S2[xp] - S0[xp]);
        return 0.125f;
    }
    // derivative taps d (odd symmetry) and smoothing taps s (even symmetry), both of length n = 2 a + 1
    int a, d1, d2, d3, s0, s1, s2, s3;
    float scale;
    if (ksize == 5) { a = 2; d1 = 2; d2 = 1; d3 = 0; s0 = 6; s1 = 4; s2 = 1; s3 = 0; scale = 1.f / 128.f; }
    else if (ksize == 7) { a = 3; d1 = 5; d2 = 4; d3 = 1; s0 = 20; s1 = 15; s2 = 6; s3 = 1; scale = 1.f / 2048.f; }
    else { a = 1; d1 = 1; d2 = 0; d3 = 0; s0 = 10; s1 = 3; s2 = 0; s3 = 0; scale = 1.f / 32.f; }   // CV_SCHARR
    const int dk[4] = {0, d1, d2, d3}, sk[4] = {s0, s1, s2, s3};
    int cx[7], cy[7];
    for (int t = -a; t <= a; ++t) { cx[t + a] = reflect101(x + t, w); cy[t + a] = reflect101(y + t, h); }
    gx = 0;
    gy = 0;
    for (int j = -a; j <= a; ++j) {
        const uint8_t* S = src + (size_t)cy[j + a] * w;
        int rd = 0, rs = 0;   // derivative / smoothing along x of row y + j
        for (int i = 1; i <= a; ++i) {
            const int hi = S[cx[a + i]], lo = S[cx[a - i]];
            rd += dk[i] * (hi - lo);
            rs += sk[i] * (hi + lo);
        }
        rs += sk[0] * S[cx[a]];
        const int aj = j < 0 ? -j : j;
        gx += sk[aj] * rd;
        gy += (j < 0 ? -dk[aj] : dk[aj]) * rs;
    }
    return scale;
}

// cv::blur(image, Size(k, k)) of GradientFilter's optional blur (GradientFilter.cpp:43-46): normalised box filter, anchor k / 2,
// BORDER_REFLECT_101, saturate_cast<uchar>(sum * (1.0 / (k * k))) in double
__global__ void k_box_blur(uint8_t* __restrict__ arena, int k, FilterJobs jobs) {
    const FilterJob jb = jobs.j[blockIdx.y];
    const uint8_t* src = arena + jb.src_off;
    uint8_t* dst = arena + jb.dst_off;
    const int w = jb.w, h = jb.h, a = k / 2;
    const double scale = 1.0 / ((double)k * k);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < w * h; i += gridDim.x * blockDim.x) {
        const int y = i / w, x = i - y * w;
        int s = 0;
        for (int j = 0; j < k; ++j) {
            const uint8_t* S = src + (size_t)reflect101(y - a + j, h) * w;
            for (int q = 0; q < k; ++q) s += S[reflect101(x - a + q, w)];
        }
        dst[i] = (uint8_t)min(255, max(0, __double2int_rn((double)s * scale)));
    }
}

// GradientFilter (delta 127, 8U saturate + cvRound) fused with the GradientBinningFilter 64K-entry look-up (index = gx | gy << 8).
template <int E>  // bytes per LUT entry: 2 (one bin + weight) or 4 (two bins + weights)
__global__ void k_gradbin(uint8_t* __restrict__ arena, const uint8_t* __restrict__ lut, int ksize, FilterJobs jobs) {
    const FilterJob jb = jobs.j[blockIdx.y];
    const uint8_t* src = arena + jb.src_off;
    uint8_t* dst = arena + jb.dst_off;
    const int w = jb.w, h = jb.h;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < w * h; i += gridDim.x * blockDim.x) {
        int y = i / w, x = i - y * w;
        int gx, gy;
        const float scale = grad_pair(src, w, h, x, y, ksize, gx, gy);
        // exact in float; cvRound = round-half-even; saturate to 0..255
        int vx = __float2int_rn(127.f + scale * gx), vy = __float2int_rn(127.f + scale * gy);
        vx = min(255, max(0, vx));
        vy = min(255, max(0, vy));
        uint32_t idx = (uint32_t)vx | ((uint32_t)vy << 8);
        if (E == 2) {
            *(uint16_t*)(dst + 2 * (size_t)i) = *(const uint16_t*)(lut + 2 * (size_t)idx);
        } else {
            *(uint32_t*)(dst + 4 * (size_t)i) = *(const uint32_t*)(lut + 4 * (size_t)idx);
        }
    }
}

// GradientFilter::applyTo alone (GradientFilter.cpp:38-59): the CV_8UC2 gradient image (x, y), same arithmetic as k_gradbin
__global__ void k_gradient_image(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int w, int h, int ksize) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < w * h; i += gridDim.x * blockDim.x) {
        int y = i / w, x = i - y * w;
        int gx, gy;
        const float scale = grad_pair(src, w, h, x, y, ksize, gx, gy);
        int vx = __float2int_rn(127.f + scale * gx), vy = __float2int_rn(127.f + scale * gy);
        dst[2 * (size_t)i] = (uint8_t)min(255, max(0, vx));
        dst[2 * (size_t)i + 1] = (uint8_t)min(255, max(0, vy));
    }
}
// GradientBinningFilter::applyTo alone (GradientBinningFilter.cpp:62-93): LUT look-up of a CV_8UC2 gradient image
template <int E>
__global__ void k_binning_image(const uint8_t* __restrict__ grad, const uint8_t* __restrict__ lut, uint8_t* __restrict__ dst, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t idx = (uint32_t)grad[2 * (size_t)i] | ((uint32_t)grad[2 * (size_t)i + 1] << 8);
        for (int e = 0; e < E; ++e) dst[(size_t)E * i + e] = lut[(size_t)E * idx + e];
    }
}

// LbpFilter 3x3 codes with BORDER_REPLICATE (LbpFilter.hpp:88-180), optional uniform map
__global__ void k_lbp(uint8_t* __restrict__ arena, int type, FilterJobs jobs) {
    const FilterJob jb = jobs.j[blockIdx.y];
    const uint8_t* src = arena + jb.src_off;
    uint8_t* dst = arena + jb.dst_off;
    const int w = jb.w, h = jb.h;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < w * h; i += gridDim.x * blockDim.x) {
        int y = i / w, x = i - y * w;
        int xm = max(x - 1, 0), xp = min(x + 1, w - 1), ym = max(y - 1, 0), yp = min(y + 1, h - 1);
        const uint8_t *P = src + (size_t)ym * w, *C = src + (size_t)y * w, *N = src + (size_t)yp * w;
        int c = C[x];
        int code = 0;
        if (type == FD_LBP8 || type == FD_LBP8_UNIFORM) {
            code |= (P[xm] > c) << 7;
            code |= (P[x] > c) << 6;
            code |= (P[xp] > c) << 5;
            code |= (C[xp] > c) << 4;
            code |= (N[xp] > c) << 3;
            code |= (N[x] > c) << 2;
            code |= (N[xm] > c) << 1;
            code |= (C[xm] > c) << 0;
            if (type == FD_LBP8_UNIFORM) {
                // LbpFilter.cpp:20-44: uniform patterns (<= 2 circular transitions) get indices 1..58 in
                // increasing code order, all others 0.  index = 1 + #uniform codes below this one.
                int rot = ((code << 1) | (code >> 7)) & 0xff;  // bit pos compared with bit pos-1 (pos 0 with 7)
                int transitions = __popc((code ^ rot) & 0xff);
                if (transitions > 2) code = 0;
                else {
                    int cnt = 0;
                    for (int q = 0; q < code; ++q) {
                        int rq = ((q << 1) | (q >> 7)) & 0xff;
                        cnt += __popc((q ^ rq) & 0xff) <= 2;
                    }
                    code = 1 + cnt;
                }
            }
        } else if (type == FD_LBP4) {
            code |= (P[x] > c) << 3;
            code |= (C[xp] > c) << 2;
            code |= (N[x] > c) << 1;
            code |= (C[xm] > c) << 0;
        } else {
            code |= (P[xm] > c) << 3;
            code |= (P[xp] > c) << 2;
            code |= (N[xp] > c) << 1;
            code |= (N[xm] > c) << 0;
        }
        dst[i] = (uint8_t)code;
    }
}

// GreyWorldNormalizationFilter.cpp:20-71 -- pass 1: per-channel sum and max
__global__ void k_greyworld_stats(const uint8_t* __restrict__ bgr, int n, unsigned long long* sums, unsigned int* maxs) {
    unsigned long long s[3] = {0, 0, 0};
    unsigned int m[3] = {0, 0, 0};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int c = 0; c < 3; ++c) {
            unsigned int v = bgr[3 * (size_t)i + c];
            s[c] += v;
            m[c] = max(m[c], v);
        }
    for (int c = 0; c < 3; ++c) {
        for (int o = 32; o > 0; o >>= 1) {
            s[c] += __shfl_down(s[c], o, 64);
            m[c] = max(m[c], (unsigned int)__shfl_down((int)m[c], o, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&sums[c], s[c]);
            atomicMax(&maxs[c], m[c]);
        }
    }
}
__global__ void k_greyworld_apply(const uint8_t* __restrict__ bgr, uint8_t* __restrict__ dst, int n, double s0, double s1,
                                  double s2) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double sc[3] = {s0, s1, s2};
        for (int c = 0; c < 3; ++c) {
            int v = __double2int_rn(sc[c] * (double)bgr[3 * (size_t)i + c]);
            dst[3 * (size_t)i + c] = (uint8_t)min(255, max(0, v));
        }
    }
}

inline uint32_t align256(size_t v) { return (uint32_t)((v + 255) & ~(size_t)255); }

// GradientBinningFilter.cpp:18-60 -- built on the host with libm, exactly like the reference ctor
void build_gradient_lut(int bins, bool signedGradients, bool interpolate, std::vector<uint8_t>& lut) {
    const double PI = 3.1415926535897932384626433832795;
    const int E = interpolate ? 4 : 2;
    lut.resize((size_t)65536 * E);
    for (int x = 0; x < 256; ++x) {
        double gradientX = ((double)x - 127) / 255;
        for (int y = 0; y < 256; ++y) {
            double gradientY = ((double)y - 127) / 255;
            double direction = std::atan2(gradientY, gradientX);
            double magnitude = std::sqrt(gradientX * gradientX + gradientY * gradientY);
            double bin;
            if (signedGradients) {
                direction += PI;
                bin = direction * bins / (2 * PI);
            } else {
                if (direction < 0) direction += PI;
                bin = direction * bins / PI;
            }
            auto sat = [](double v) { int r = fd_cvRound(v); return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r)); };
            size_t index = (size_t)x | ((size_t)y << 8);
            if (!interpolate) {
                lut[2 * index] = (uint8_t)((uint8_t)std::round(bin) % (unsigned)bins);
                lut[2 * index + 1] = sat(255 * magnitude);
            } else {
                uint8_t w1 = sat(255 * magnitude * (bin - std::floor(bin)));
                lut[4 * index] = (uint8_t)((uint8_t)std::floor(bin) % (unsigned)bins);
                lut[4 * index + 1] = sat(255 * magnitude - w1);
                lut[4 * index + 2] = (uint8_t)((uint8_t)std::ceil(bin) % (unsigned)bins);
                lut[4 * index + 3] = w1;
            }
        }
    }
}

// cv::resize coordinate tables of the first-octave layers that feed a pyrDown chain (k_resize_down): the kernel's own float
// expressions (k_resize_tiled's srcX / srcY), evaluated once per geometry on the host (this file is built with -ffp-contract=off)
void build_resize_tables(fd_pyramid* p, int W, int H) {
    p->rtab_x.assign(p->all.size(), ~0u);
    p->rtab_y.assign(p->all.size(), ~0u);
    static const int mode = [] { const char* e = getenv("FD_PYR_FUSED"); return e ? atoi(e) : 1; }();   // 0: never, 1: default, 2: kept layers too
    if (mode == 0) return;
    std::vector<int2> tab;
    // tiles of k_resize_down, one list per launch (MAXJ chains): {X0 | Y0 << 16, ncol | nrow << 16, tx | ty << 16, chain of the launch}
    std::vector<std::vector<int4>> tiles;
    p->rtile_off.clear();
    p->rtile_cnt.clear();
    int nfused = 0;
    for (size_t k = 0; k + 1 < p->all.size(); ++k) {
        const HostLayer& L = p->all[k];
        const HostLayer& D = p->all[k + 1];
        if (L.depth != 0 || D.depth != 1 || D.chain != L.chain) continue;
        // the scale-1 layer IS the gray image: its pyrDown stays with k_pyrdown_tiled (through this kernel with identity tables it
        // costs +35 us per 64-frame call against 15 us saved: the resize arithmetic is not free)
        if (L.w == W && L.h == H && L.gray_off == p->gray_full_off) continue;
        if (L.w < 3 || L.h < 3 || W > 65535 || H > 65535) continue;
        // A first-octave layer that is itself a kept layer (config 2's pyramid: scales up to 1) has to be written anyway: the fusion saves
        // nothing there and the fused kernel is the slower resize (round 3: 699 us per 640x480 frame; k_resize_tiled + k_pyrdown_tiled:
        // 101 + 4 x 68 us, round 4).  FD_PYR_FUSED=2 fuses those too (A/B).  Also measured in round 4 and dropped: persistent
        // k_pyrdown_tiled workgroups with the next tile's loads in flight (69 vs 63 us per 64-frame headline call).
        if (L.kept && mode != 2) continue;
        const double scale_x = 1. / ((double)L.w / W), scale_y = 1. / ((double)L.h / H);
        std::vector<int2> xt((size_t)L.w), yt((size_t)L.h);
        for (int dx = 0; dx < L.w; ++dx) {
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            int sx = (int)floorf(fx);
            fx -= sx;
            if (sx < 0) { fx = 0; sx = 0; }
            if (sx >= W - 1) { fx = 0; sx = W - 1; }
            const int a0 = (int)nearbyintf((1.f - fx) * 2048), a1 = (int)nearbyintf(fx * 2048);
            xt[(size_t)dx] = make_int2(sx, a0 | (a1 << 16));
        }
        for (int dy = 0; dy < L.h; ++dy) {
            float fy = (float)((dy + 0.5) * scale_y - 0.5);
            const int sy = (int)floorf(fy);
            fy -= sy;
            const int b0 = (int)nearbyintf((1.f - fy) * 2048), b1 = (int)nearbyintf(fy * 2048);
            const int y0 = sy < 0 ? 0 : (sy >= H ? H - 1 : sy), y1 = sy + 1 < 0 ? 0 : (sy + 1 >= H ? H - 1 : sy + 1);
            yt[(size_t)dy] = make_int2(y0 | (y1 << 16), b0 | (b1 << 16));
        }
        // every tile's source rectangle must fit the kernel's stage
        bool fits = true;
        // the kernel reads a row's lower tap from the stage row behind its upper tap, and packs b0 / b1 into 12 bits each
        for (int dy = 0; dy < L.h && fits; ++dy) {
            const int y0 = yt[(size_t)dy].x & 0xffff, y1 = (int)((uint32_t)yt[(size_t)dy].x >> 16), b0 = yt[(size_t)dy].y & 0xffff, b1 = (int)((uint32_t)yt[(size_t)dy].y >> 16);
            fits = (y1 == y0 + 1 || b1 == 0) && b0 >= 0 && b0 <= 2048 && b1 >= 0 && b1 <= 2048 && y1 >= y0;
        }
        for (int x1 = 0; x1 < D.w && fits; x1 += FT_W1) {
            const int cLo = std::max(0, 2 * x1 - 2), cHi = std::min(L.w - 1, 2 * x1 - 2 + G0_W - 1);
            fits = std::min(W - 1, xt[(size_t)cHi].x + 1) - xt[(size_t)cLo].x + 1 <= FS_PITCH;
        }
        for (int y1 = 0; y1 < D.h && fits; y1 += FT_H1) {
            const int rLo = std::max(0, 2 * y1 - 2), rHi = std::min(L.h - 1, 2 * y1 - 2 + G0_H - 1);
            fits = (yt[(size_t)rHi].x >> 16) - (yt[(size_t)rLo].x & 0xffff) + 1 <= FS_ROWS;
        }
        if (!fits) continue;
        {   // the source rectangle of every tile, exactly as the kernel's addressing expects it
            if (nfused % MAXJ == 0) tiles.emplace_back();
            std::vector<int4>& tl = tiles.back();
            for (int ty = 0; ty * FT_H1 < D.h; ++ty)
                for (int tx = 0; tx * FT_W1 < D.w; ++tx) {
                    const int gx0 = 2 * tx * FT_W1 - 2, gy0 = 2 * ty * FT_H1 - 2;
                    const int cLo = std::max(0, gx0), cHi = std::min(L.w - 1, gx0 + G0_W - 1), rLo = std::max(0, gy0), rHi = std::min(L.h - 1, gy0 + G0_H - 1);
                    const int X0 = xt[(size_t)cLo].x, Y0 = yt[(size_t)rLo].x & 0xffff;
                    const int ncol = std::min(W - 1, xt[(size_t)cHi].x + 1) - X0 + 1, nrow = (yt[(size_t)rHi].x >> 16) - Y0 + 1;
                    tl.push_back(make_int4((int)((uint32_t)X0 | (uint32_t)Y0 << 16), (int)((uint32_t)ncol | (uint32_t)nrow << 16),
                                           (int)((uint32_t)tx | (uint32_t)ty << 16), nfused % MAXJ));
                }
            ++nfused;
        }
        p->rtab_x[k] = (uint32_t)tab.size();
        tab.insert(tab.end(), xt.begin(), xt.end());
        p->rtab_y[k] = (uint32_t)tab.size();
        tab.insert(tab.end(), yt.begin(), yt.end());
    }
    if (tab.empty()) return;
    for (const std::vector<int4>& tl : tiles) {   // 16-byte entries behind the 8-byte ones, 16-byte aligned
        if (tab.size() & 1) tab.push_back(make_int2(0, 0));
        p->rtile_off.push_back((uint32_t)tab.size());
        p->rtile_cnt.push_back((int)tl.size());
        for (const int4& t : tl) { tab.push_back(make_int2(t.x, t.y)); tab.push_back(make_int2(t.z, t.w)); }
    }
    p->rtab.reserve(sizeof(int2) * tab.size());
    HIP_CHECK(hipMemcpy(p->rtab.p, tab.data(), sizeof(int2) * tab.size(), hipMemcpyHostToDevice));
}

void build_layout(fd_pyramid* p, int W, int H) {
    p->img_w = W;
    p->img_h = H;
    p->all.clear();
    p->kept.clear();
    size_t off = 0;
    p->gray_full_off = 0;
    off = align256((size_t)W * H);
    const int fch = p->filter_kind == FD_LAYER_GRADBIN ? (p->interpolate ? 4 : 2) : 1;
    for (size_t i = 0; i < p->octl; ++i) {
        double scaleFactor = std::pow(p->inc, (double)i);
        int w = fd_cvRound(W * scaleFactor), h = fd_cvRound(H * scaleFactor);
        if (w < 1 || h < 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "ImagePyramid: layer %zu would be empty", i);
        int depth = 0;
        // a first-octave layer of the image's own size (scale 1) IS the gray image: cv::resize to the same size copies every pixel
        // (weights 2048 / 0: ((2048 * (S * 128)) >> 16) + 2 >> 2 == S), so it shares the gray image's storage and costs no launch
        const bool isFull = (w == W && h == H);
        HostLayer L{(int)i, scaleFactor, w, h, 1, (uint32_t)(isFull ? p->gray_full_off : off), 0, scaleFactor <= p->maxS && scaleFactor >= p->minS, (int)i, depth};
        if (!isFull) off = align256(off + (size_t)w * h);
        p->all.push_back(L);
        int pw = w, ph = h;
        scaleFactor *= 0.5;
        for (size_t j = 1; scaleFactor >= p->minS && pw > 1; ++j, scaleFactor *= 0.5) {
            int dw = (pw + 1) / 2, dh = (ph + 1) / 2;
            HostLayer D{(int)(i + j * p->octl), scaleFactor, dw, dh, 1, (uint32_t)off, 0, scaleFactor <= p->maxS, (int)i, (int)j};
            off = align256(off + (size_t)dw * dh);
            p->all.push_back(D);
            pw = dw;
            ph = dh;
        }
    }
    for (size_t k = 0; k < p->all.size(); ++k) {
        HostLayer& L = p->all[k];
        if (!L.kept) continue;
        L.ch = fch;
        if (p->filter_kind == FD_LAYER_NONE) L.filt_off = L.gray_off;
        else {
            L.filt_off = (uint32_t)off;
            off = align256(off + (size_t)L.w * L.h * fch);
            if (p->filter_kind == FD_LAYER_GRADBIN && p->grad_blur > 0) {
                L.blur_off = (uint32_t)off;
                off = align256(off + (size_t)L.w * L.h);
            }
        }
        p->kept.push_back((int)k);
    }
    if (off > 0xfffffff0ull) FD_THROW(FD_ERR_INVALID_ARGUMENT, "ImagePyramid: pyramid exceeds 4 GB arena");
    std::sort(p->kept.begin(), p->kept.end(), [&](int a, int b) { return p->all[a].index < p->all[b].index; });
    p->arena_bytes = off + 256;
    p->image_stride = (p->arena_bytes + 255) & ~(size_t)255;
    if (p->nimg > 1 && p->image_stride * (size_t)p->nimg > 0xfffffff0ull * 16) FD_THROW(FD_ERR_INVALID_ARGUMENT, "ImagePyramid: multi-frame pyramid too large");
    p->arena.reserve(p->image_stride * (size_t)p->nimg);
    p->h_layer_table.clear();
    for (int k : p->kept) {
        const HostLayer& L = p->all[k];
        p->h_layer_table.push_back(LayerDesc{L.w, L.h, L.ch, 0, L.gray_off, L.filt_off});
    }
    p->layer_table.reserve(sizeof(LayerDesc) * std::max<size_t>(1, p->h_layer_table.size()));
    if (!p->h_layer_table.empty())
        HIP_CHECK(hipMemcpyAsync(p->layer_table.p, p->h_layer_table.data(), sizeof(LayerDesc) * p->h_layer_table.size(),
                                 hipMemcpyHostToDevice, p->ctx->stream));
    build_resize_tables(p, W, H);
}

int grid_for(int npix) { return std::max(1, std::min(1024, (npix + 255) / 256)); }
int tile_grid_for(int ntiles) { return std::max(1, std::min(1024, ntiles)); }

// (Measured and dropped in round 5: single-image updates replayed as a hipGraph -- captured with hipStreamBeginCapture around the launch
// code below on the second update with the same layout, the image address patched into the k_bgr2gray node with
// hipGraphExecKernelNodeSetParams.  Bit-exact and the call returned after 10 instead of 18 us, but a blocking 640x480 frame took
// 153.5 us p50 against 146 with plain launches (ROCm 7.2: the graph's first node starts later than a plain kernel would), and the
// 15-detector batch 5723 against 5871 Mpatches/s.  The rocprofv3 timeline that motivated it: k_bgr2gray 3.3 + k_resize_down 5.9 + 4 x
// k_pyrdown_tiled ~4.5 us = 27 us of kernels spread over 48 us, the host's ~3.5 us per launch between them.)
void pyramid_update(fd_pyramid* p, const uint8_t* image, int W, int H, int ch, int is_device, hipStream_t st, const uint8_t* const* frames = nullptr) {
    if (!image && !frames) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_update: image is NULL");
    if (p->nimg > 1 && !frames) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_update: the pyramid holds %d frames, use fd_pyramid_update_frames", p->nimg);
    if (p->nimg > 1 && p->filter_kind != FD_LAYER_NONE) FD_THROW(FD_ERR_INVALID_ARGUMENT, "multi-frame pyramids have no layer filters");
    if (W < 1 || H < 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_update: empty image");
    if (ch != 1 && ch != 3) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_update: image must have 1 or 3 channels");
    if (W != p->img_w || H != p->img_h || p->all.empty()) build_layout(p, W, H);
    uint8_t* arena = p->arena.as<uint8_t>();
    const size_t npix = (size_t)W * H;
    const int NI = frames ? p->nimg : 1;
    const size_t IS = p->image_stride;
    // everything behind the gray image: the resizes, the pyrDown chains, the layer filters
    auto enqueue_rest = [&]() {
    // depth 0: resize from the full-resolution gray image
    int maxDepth = 0;
    for (const HostLayer& L : p->all) maxDepth = std::max(maxDepth, L.depth);
    {
        ResizeJobs jobs;
        jobs.n = 0;
        int maxpix = 0, maxtiles = 0;
        auto flush = [&]() {
            if (!jobs.n) return;
            bool fits = true;   // the tiled kernel stages at most TL_ROWS x TL_PITCH source bytes per 64 x 32 tile
            for (int q = 0; q < jobs.n; ++q)
                fits = fits && jobs.j[q].scale_x >= 1.0 && jobs.j[q].scale_x <= 2.05 && jobs.j[q].scale_y >= 1.0 && jobs.j[q].scale_y <= 2.05;
            if (fits)
                hipLaunchKernelGGL(k_resize_tiled, dim3(tile_grid_for(maxtiles), jobs.n, NI), dim3(256), 0, st, arena, arena, p->gray_full_off, W, H, jobs, IS);
            else
                hipLaunchKernelGGL(k_resize_linear, dim3(grid_for(maxpix), jobs.n, NI), dim3(256), 0, st, arena, arena, p->gray_full_off, W, H, jobs, IS);
            jobs.n = 0;
            maxpix = 0;
            maxtiles = 0;
        };
        for (size_t k = 0; k < p->all.size(); ++k) {
            const HostLayer& L = p->all[k];
            if (L.depth != 0) continue;
            if (L.w == W && L.h == H && L.gray_off == p->gray_full_off) continue;   // the gray image itself (build_layout)
            if (p->rtab_x[k] != ~0u) continue;   // resized and pyrDown'ed by k_resize_down below
            ResizeJob& j = jobs.j[jobs.n++];
            j.dw = L.w; j.dh = L.h; j.dst_off = L.gray_off;
            j.scale_x = 1. / ((double)L.w / W);
            j.scale_y = 1. / ((double)L.h / H);
            maxpix = std::max(maxpix, L.w * L.h);
            maxtiles = std::max(maxtiles, ((L.w + TL_W - 1) / TL_W) * ((L.h + TL_H - 1) / TL_H));
            if (jobs.n == MAXJ) flush();
        }
        flush();
    }
    {   // first-octave layers with a pyrDown chain: resize + first pyrDown in one kernel, the resized pixels stay in LDS
        FusedJobs jobs;
        jobs.n = 0;
        size_t group = 0;
        auto flush = [&]() {
            if (!jobs.n) return;
            static int perCu = 0;
            if (perCu == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_resize_down, 256, 0) != hipSuccess || perCu < 1)) perCu = 4;
            const int tilesPerFrame = p->rtile_cnt[group];
            // persistent workgroups; a multiple of the 8 XCDs for a multi-frame pyramid (every frame is then resized on one XCD)
            int grid = (int)std::min<int64_t>((int64_t)tilesPerFrame * NI, (int64_t)p->ctx->num_cus * perCu);
            if (NI >= 8 && grid >= 64) grid &= ~7;
            hipLaunchKernelGGL(k_resize_down, dim3(grid), dim3(256), 0, st, arena, p->gray_full_off, W, H, p->rtab.as<int2>(), jobs, p->rtile_off[group], tilesPerFrame, NI, IS);
            jobs.n = 0;
            ++group;
        };
        for (size_t k = 0; k + 1 < p->all.size(); ++k) {
            if (p->rtab_x[k] == ~0u) continue;
            const HostLayer& L = p->all[k];
            const HostLayer& D = p->all[k + 1];
            FusedJob& j = jobs.j[jobs.n++];
            j.dw0 = L.w; j.dh0 = L.h; j.dw1 = D.w; j.dh1 = D.h;
            j.dst0_off = (L.kept && L.gray_off != p->gray_full_off) ? L.gray_off : 0xffffffffu;   // the scale-1 layer IS the gray image
            j.dst1_off = D.gray_off;
            j.xtab = p->rtab_x[k]; j.ytab = p->rtab_y[k];
            if (jobs.n == MAXJ) flush();
        }
        flush();
    }
    {   // generation 1 of the chains k_resize_down does not cover (the scale-1 chain: pyrDown of the gray image itself)
        DownJobs dj;
        dj.n = 0;
        dj.tile0[0] = 0;
        auto flush1 = [&]() {
            if (!dj.n) return;
            hipLaunchKernelGGL(k_pyrdown_tiled, dim3(tile_grid_for(dj.tile0[dj.n]), 1, NI), dim3(256), 0, st, arena, dj, IS);
            dj.n = 0;
        };
        for (size_t k = 1; k < p->all.size(); ++k) {
            const HostLayer& L = p->all[k];
            if (L.depth != 1 || p->rtab_x[k - 1] != ~0u) continue;
            const HostLayer& S = p->all[k - 1];
            DownJob& j = dj.j[dj.n++];
            j.sw = S.w; j.sh = S.h; j.src_off = S.gray_off; j.dst_off = L.gray_off;
            dj.tile0[dj.n] = dj.tile0[dj.n - 1] + ((L.w + PD_W - 1) / PD_W) * ((L.h + PD_TH - 1) / PD_TH);
            if (dj.n == MAXJ) flush1();
        }
        flush1();
    }
    // (Measured and dropped, twice.  Round 3: one workgroup walking all deeper generations of a chain tile by tile -- 160 us per 64-frame
    // call against 57 us for the per-generation launches, 32 serial tiles per workgroup.  Round 5: k_pyrdown_chain, a workgroup owning a
    // 16 x 16 tile of the deepest generation and computing the 35 / 73 / 149-pixel boxes above it in LDS (halo recomputed, reflected
    // borders materialised; bit-exact on every pyramid test) -- ONE launch of 121 us instead of three of 11.4 us, headline 3270 -> 2690
    // Mpatches/s: 1.8x the outputs (halo) at ~60 instructions each, against ~12 per output of the column walk below, which shares
    // the row dot products between vertically adjacent outputs.  The per-generation launches are latency-bound but cheap.)
    for (int d = 2; d <= maxDepth; ++d) {
        DownJobs jobs;
        jobs.n = 0;
        jobs.tile0[0] = 0;
        auto flush = [&]() {
            if (!jobs.n) return;
            hipLaunchKernelGGL(k_pyrdown_tiled, dim3(tile_grid_for(jobs.tile0[jobs.n]), 1, NI), dim3(256), 0, st, arena, jobs, IS);
            jobs.n = 0;
        };
        for (size_t k = 0; k < p->all.size(); ++k) {
            const HostLayer& L = p->all[k];
            if (L.depth != d) continue;
            const HostLayer& S = p->all[k - 1];  // previous entry of the same chain
            DownJob& j = jobs.j[jobs.n++];
            j.sw = S.w; j.sh = S.h; j.src_off = S.gray_off; j.dst_off = L.gray_off;
            jobs.tile0[jobs.n] = jobs.tile0[jobs.n - 1] + ((L.w + PD_W - 1) / PD_W) * ((L.h + PD_TH - 1) / PD_TH);
            if (jobs.n == MAXJ) flush();
        }
        flush();
    }
    if (p->filter_kind != FD_LAYER_NONE) {
        FilterJobs jobs, bjobs;
        jobs.n = 0;
        bjobs.n = 0;
        int maxpix = 0;
        const bool blur = p->filter_kind == FD_LAYER_GRADBIN && p->grad_blur > 0;
        auto flush = [&]() {
            if (!jobs.n) return;
            dim3 g(grid_for(maxpix), jobs.n);
            // GradientFilter's blur: gray layer -> blurred copy; the gradients are then taken of the copy
            if (blur) hipLaunchKernelGGL(k_box_blur, g, dim3(256), 0, st, arena, p->grad_blur, bjobs);
            bjobs.n = 0;
            if (p->filter_kind == FD_LAYER_GRADBIN) {
                if (p->interpolate)
                    hipLaunchKernelGGL(k_gradbin<4>, g, dim3(256), 0, st, arena, p->lut.as<uint8_t>(), p->grad_kernel, jobs);
                else
                    hipLaunchKernelGGL(k_gradbin<2>, g, dim3(256), 0, st, arena, p->lut.as<uint8_t>(), p->grad_kernel, jobs);
            } else {
                hipLaunchKernelGGL(k_lbp, g, dim3(256), 0, st, arena, p->lbp_type, jobs);
            }
            jobs.n = 0;
            maxpix = 0;
        };
        for (int k : p->kept) {
            const HostLayer& L = p->all[k];
            FilterJob& j = jobs.j[jobs.n++];
            j.w = L.w; j.h = L.h; j.src_off = blur ? L.blur_off : L.gray_off; j.dst_off = L.filt_off;
            if (blur) {
                FilterJob& b = bjobs.j[bjobs.n++];
                b.w = L.w; b.h = L.h; b.src_off = L.gray_off; b.dst_off = L.blur_off;
            }
            maxpix = std::max(maxpix, L.w * L.h);
            if (jobs.n == MAXJ) flush();
        }
        flush();
    }
    };   // enqueue_rest
    if (frames) {   // one launch converts / copies all frames into their arenas
        FramePtrs fp;
        if (!is_device) {
            p->input.reserve(npix * ch * (size_t)NI);
            for (int f = 0; f < NI; ++f) {
                HIP_CHECK(hipMemcpyAsync(p->input.as<uint8_t>() + (size_t)f * npix * ch, frames[f], npix * ch, hipMemcpyHostToDevice, st));
                fp.p[f] = p->input.as<uint8_t>() + (size_t)f * npix * ch;
            }
        } else {
            for (int f = 0; f < NI; ++f) fp.p[f] = frames[f];
        }
        hipLaunchKernelGGL(k_frames_to_gray, dim3(grid_for((int)(npix / 16 + 1)), NI), dim3(256), 0, st, fp, arena + p->gray_full_off, IS, (int)npix, ch);   // four quads per thread
    } else {
        const uint8_t* dimg = image;
        if (!is_device) {
            p->input.reserve(npix * ch);
            HIP_CHECK(hipMemcpyAsync(p->input.p, image, npix * ch, hipMemcpyHostToDevice, st));
            dimg = p->input.as<uint8_t>();
        }
        if (ch == 3) {
            hipLaunchKernelGGL(k_bgr2gray, dim3(grid_for((int)npix)), dim3(256), 0, st, dimg, arena + p->gray_full_off, (int)npix);
        } else {
            HIP_CHECK(hipMemcpyAsync(arena + p->gray_full_off, dimg, npix, hipMemcpyDeviceToDevice, st));
        }
    }
    enqueue_rest();
    HIP_CHECK(hipGetLastError());
    if (!p->ready) HIP_CHECK(hipEventCreateWithFlags(&p->ready, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(p->ready, st));
    p->readyStream = st;
    p->version++;
}

}  // namespace

// fd_pyramid_update on an explicit stream (worker threads of the batch entry points); throws FdError
void fd_pyramid_update_on(fd_pyramid* p, const uint8_t* image, int w, int h, int ch, int is_device, hipStream_t st) {
    if (!p) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_update: NULL pyramid");
    HIP_CHECK(hipSetDevice(p->ctx->device));
    pyramid_update(p, image, w, h, ch, is_device, st);
}

// DirectPyramidFeatureExtractor::extract(stepX, stepY, roi) window grid, :75-123
void fd_enumerate_layers(const fd_pyramid* p, int pw, int ph, int sx, int sy, const int* roiIn,
                         std::vector<WindowLayer>& out, int64_t& total) {
    if (sx < 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "DirectPyramidFeatureExtractor: stepX has to be greater than zero");
    if (sy < 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "DirectPyramidFeatureExtractor: stepY has to be greater than zero");
    if (pw < 1 || ph < 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "DirectPyramidFeatureExtractor: empty patch size");
    int rx = 0, ry = 0, rw = 0, rh = 0;
    if (!roiIn && p->sel_has_roi) roiIn = p->sel_roi;   // fd_pyramid_select
    if (roiIn) { rx = roiIn[0]; ry = roiIn[1]; rw = roiIn[2]; rh = roiIn[3]; }
    if (rx == 0 && ry == 0 && rw == 0 && rh == 0) {
        rw = p->img_w;
        rh = p->img_h;
    } else {
        int nx = std::max(0, rx), ny = std::max(0, ry);
        rw = std::min(p->img_w, rw + nx) - nx;
        rh = std::min(p->img_h, rh + ny) - ny;
        rx = nx;
        ry = ny;
    }
    out.clear();
    total = 0;
    size_t viewCount = 0;
    for (size_t li = 0; li < p->kept.size(); ++li) {
        const HostLayer& L = p->all[p->kept[li]];
        auto scaled = [&](int v) { return fd_cvRound(v * L.scale); };     // ImagePyramidLayer.hpp:65-67
        auto original = [&](int v) { return fd_cvRound(v / L.scale); };   // :98-100
        WindowLayer wl;
        wl.layer = (int)li;
        wl.ow = original(pw);
        wl.oh = original(ph);
        wl.bx = scaled(rx);
        wl.by = scaled(ry);
        int ex = scaled(rx + rw), ey = scaled(ry + rh);
        // positions x = bx + k*sx with x + pw < ex  (strict)
        long spanx = (long)ex - pw - wl.bx, spany = (long)ey - ph - wl.by;
        wl.nx = spanx > 0 ? (int)((spanx - 1) / sx + 1) : 0;
        wl.ny = spany > 0 ? (int)((spany - 1) / sy + 1) : 0;
        // layer selection (DirectPyramidFeatureExtractor.cpp:99-107): every sel_step-th layer counted from the first one,
        // skipping indices below sel_first, stopping above sel_last
        // The step walks pyramid->getLayers() from its begin(): for a pyramid built on another one that is the first layer of its
        // scale range (fd_pyramid_select_view), not the source's first layer.
        const bool inView = (p->view_first < 0 || L.index >= p->view_first) && (p->view_last < 0 || L.index <= p->view_last);
        const size_t viewPos = viewCount;
        if (inView) ++viewCount;
        const bool selected = inView && (viewPos % (size_t)p->sel_step == 0) && (p->sel_first < 0 || L.index >= p->sel_first) &&
                              (p->sel_last < 0 || L.index <= p->sel_last);
        if (!selected) wl.nx = wl.ny = 0;
        // windows must lie inside the layer (cv::Mat(image, bounds) would assert otherwise)
        if (wl.nx > 0 && wl.ny > 0) {
            if (wl.bx < 0 || wl.by < 0 || wl.bx + (wl.nx - 1) * sx + pw > L.w || wl.by + (wl.ny - 1) * sy + ph > L.h)
                FD_THROW(FD_ERR_RUNTIME, "DirectPyramidFeatureExtractor: window outside of pyramid layer %d", L.index);
        } else {
            wl.nx = wl.ny = 0;
        }
        wl.first = total;
        total += (int64_t)wl.nx * wl.ny;
        out.push_back(wl);
    }
}

extern "C" {

int fd_pyramid_create(fd_ctx* ctx, int octl, double minS, double maxS, fd_pyramid** out) {
    return fd_guard(ctx, [&] {
        if (!ctx || !out) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_create: NULL argument");
        if (octl <= 0) FD_THROW(FD_ERR_INVALID_ARGUMENT, "the number of layers per octave must be greater than zero");
        if (minS <= 0) FD_THROW(FD_ERR_INVALID_ARGUMENT, "the minimum scale factor must be greater than zero");
        if (maxS > 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "the maximum scale factor must not exceed one");
        fd_pyramid* p = new fd_pyramid();
        p->ctx = ctx;
        p->octl = (size_t)octl;
        p->inc = std::pow(0.5, 1. / octl);
        p->minS = minS;
        p->maxS = maxS;
        *out = p;
    });
}

int fd_pyramid_create_inc(fd_ctx* ctx, double inc, double minS, double maxS, fd_pyramid** out) {
    return fd_guard(ctx, [&] {
        if (inc <= 0 || inc >= 1)
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "the incremental scale factor must be greater than zero and smaller than one");
    }) ?: fd_pyramid_create(ctx, (int)std::round(std::log(0.5) / std::log(inc)), minS, maxS, out);
}

void fd_pyramid_destroy(fd_pyramid* p) { delete p; }

int fd_pyramid_set_layer_filter(fd_pyramid* p, int kind, int bins, int signed_gradients, int interpolate, int grad_kernel,
                                int lbp_type) {
    return fd_guard(p ? p->ctx : nullptr, [&] {
        if (!p) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_set_layer_filter: NULL pyramid");
        if (kind == FD_LAYER_GRADBIN) {
            if (grad_kernel != 1 && grad_kernel != 3 && grad_kernel != 5 && grad_kernel != 7 && grad_kernel != FD_GRAD_SCHARR)
                FD_THROW(FD_ERR_INVALID_ARGUMENT, "GradientFilter: the kernel size must be 1, 3, 5, 7 or CV_SCHARR");
            if (bins < 1 || bins > 255) FD_THROW(FD_ERR_INVALID_ARGUMENT, "GradientBinningFilter: bins must be in 1..255");
            std::vector<uint8_t> lut;
            build_gradient_lut(bins, signed_gradients != 0, interpolate != 0, lut);
            p->lut.reserve(lut.size());
            HIP_CHECK(hipMemcpyAsync(p->lut.p, lut.data(), lut.size(), hipMemcpyHostToDevice, p->ctx->stream));
            HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
        } else if (kind == FD_LAYER_LBP) {
            if (lbp_type < 0 || lbp_type > 3) FD_THROW(FD_ERR_INVALID_ARGUMENT, "LbpFilter: invalid type");
        } else if (kind != FD_LAYER_NONE) {
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_set_layer_filter: unknown kind %d", kind);
        }
        p->filter_kind = kind;
        p->bins = bins;
        p->signed_gradients = signed_gradients;
        p->interpolate = interpolate;
        p->grad_kernel = grad_kernel;
        p->lbp_type = lbp_type;
        p->all.clear();  // force a new layout
        p->img_w = p->img_h = 0;
    });
}

int fd_pyramid_update(fd_pyramid* p, const uint8_t* image, int w, int h, int ch, int is_device) {
    return fd_guard(p ? p->ctx : nullptr, [&] {
        if (!p) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_update: NULL pyramid");
        HIP_CHECK(hipSetDevice(p->ctx->device));
        pyramid_update(p, image, w, h, ch, is_device, p->ctx->stream);
    });
}

int fd_pyramid_set_frames(fd_pyramid* p, int frames) {
    return fd_guard(p ? p->ctx : nullptr, [&] {
        if (!p) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_set_frames: NULL pyramid");
        if (frames < 1 || frames > FD_MAX_FRAMES) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_set_frames: 1..%d frames", FD_MAX_FRAMES);
        if (frames > 1 && p->filter_kind != FD_LAYER_NONE) FD_THROW(FD_ERR_INVALID_ARGUMENT, "multi-frame pyramids have no layer filters");
        if (frames != p->nimg) { p->nimg = frames; p->all.clear(); p->img_w = p->img_h = 0; }   // new layout at the next update
    });
}

int fd_pyramid_update_frames(fd_pyramid* p, const uint8_t* const* images, int n, int w, int h, int ch, int is_device) {
    return fd_guard(p ? p->ctx : nullptr, [&] {
        if (!p || !images) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_update_frames: NULL argument");
        if (n != p->nimg) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_update_frames: %d images for a pyramid of %d frames", n, p->nimg);
        for (int f = 0; f < n; ++f)
            if (!images[f]) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_update_frames: image %d is NULL", f);
        HIP_CHECK(hipSetDevice(p->ctx->device));
        pyramid_update(p, nullptr, w, h, ch, is_device, p->ctx->stream, images);
    });
}

int fd_pyramid_select(fd_pyramid* p, int first_layer, int last_layer, int step_layer, const int* roi) {
    return fd_guard(p ? p->ctx : nullptr, [&] {
        if (!p) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_select: NULL pyramid");
        if (step_layer < 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "DirectPyramidFeatureExtractor: stepLayer has to be greater than zero");
        p->sel_first = first_layer < 0 ? -1 : first_layer;
        p->sel_last = last_layer < 0 ? -1 : last_layer;
        p->sel_step = step_layer;
        p->sel_has_roi = roi != nullptr;
        if (roi) std::memcpy(p->sel_roi, roi, sizeof(p->sel_roi));
    });
}

int fd_pyramid_set_gradient_blur(fd_pyramid* p, int blur_kernel) {
    return fd_guard(p ? p->ctx : nullptr, [&] {
        if (!p) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_set_gradient_blur: NULL pyramid");
        if (blur_kernel < 0 || blur_kernel > 31) FD_THROW(FD_ERR_INVALID_ARGUMENT, "GradientFilter: the blur kernel size must be in 0..31");
        if (blur_kernel != p->grad_blur) {
            p->grad_blur = blur_kernel;
            p->all.clear();  // force a new layout
            p->img_w = p->img_h = 0;
        }
    });
}

int fd_pyramid_select_view(fd_pyramid* p, int first_layer, int last_layer) {
    return fd_guard(p ? p->ctx : nullptr, [&] {
        if (!p) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_select_view: NULL pyramid");
        p->view_first = first_layer < 0 ? -1 : first_layer;
        p->view_last = last_layer < 0 ? -1 : last_layer;
    });
}

int fd_pyramid_octave_layer_count(const fd_pyramid* p) { return p ? (int)p->octl : 0; }
double fd_pyramid_incremental_scale(const fd_pyramid* p) { return p ? p->inc : 0.0; }
int fd_pyramid_layer_count(const fd_pyramid* p) { return p ? (int)p->kept.size() : 0; }

int fd_pyramid_layer_info(const fd_pyramid* p, int i, int* index, double* scale, int* w, int* h, int* ch) {
    if (!p || i < 0 || i >= (int)p->kept.size()) return FD_ERR_INVALID_ARGUMENT;
    const HostLayer& L = p->all[p->kept[i]];
    if (index) *index = L.index;
    if (scale) *scale = L.scale;
    if (w) *w = L.w;
    if (h) *h = L.h;
    if (ch) *ch = L.ch;
    return FD_OK;
}

int fd_pyramid_layer_download(fd_pyramid* p, int i, uint8_t* host_dst) {
    return fd_guard(p ? p->ctx : nullptr, [&] {
        if (!p || !host_dst || i < 0 || i >= (int)p->kept.size()) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_layer_download: bad argument");
        if (p->nimg > 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_layer_download: the pyramid holds %d frames; use fd_pyramid_frame_layer_download", p->nimg);
        const HostLayer& L = p->all[p->kept[i]];
        HIP_CHECK(hipMemcpyAsync(host_dst, p->arena.as<uint8_t>() + L.filt_off, (size_t)L.w * L.h * L.ch, hipMemcpyDeviceToHost, p->ctx->stream));
        HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    });
}

int fd_pyramid_frame_layer_download(fd_pyramid* p, int frame, int i, uint8_t* host_dst) {
    return fd_guard(p ? p->ctx : nullptr, [&] {
        if (!p || !host_dst || i < 0 || i >= (int)p->kept.size() || frame < 0 || frame >= p->nimg)
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_frame_layer_download: bad argument");
        const HostLayer& L = p->all[p->kept[i]];
        HIP_CHECK(hipMemcpyAsync(host_dst, p->arena.as<uint8_t>() + (size_t)frame * p->image_stride + L.filt_off, (size_t)L.w * L.h * L.ch,
                                 hipMemcpyDeviceToHost, p->ctx->stream));
        HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    });
}

int fd_pyramid_window_count(const fd_pyramid* p, int pw, int ph, int sx, int sy, const int* roi, int64_t* count) {
    return fd_guard(p ? p->ctx : nullptr, [&] {
        if (!p || !count) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_window_count: NULL argument");
        std::vector<WindowLayer> wl;
        fd_enumerate_layers(p, pw, ph, sx, sy, roi, wl, *count);
    });
}

int fd_pyramid_windows(const fd_pyramid* p, int pw, int ph, int sx, int sy, const int* roi, int32_t* out, int64_t cap,
                       int64_t* count) {
    return fd_guard(p ? p->ctx : nullptr, [&] {
        if (!p || !count) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_windows: NULL argument");
        std::vector<WindowLayer> wl;
        int64_t total;
        fd_enumerate_layers(p, pw, ph, sx, sy, roi, wl, total);
        *count = total;
        if (!out) return;
        int64_t n = 0;
        for (const WindowLayer& w : wl) {
            const HostLayer& L = p->all[p->kept[w.layer]];
            for (int iy = 0; iy < w.ny; ++iy)
                for (int ix = 0; ix < w.nx; ++ix, ++n) {
                    if (n >= cap) continue;
                    int x = w.bx + ix * sx, y = w.by + iy * sy;
                    int32_t* o = out + 7 * n;
                    o[0] = w.layer; o[1] = x; o[2] = y;
                    o[3] = fd_cvRound(x / L.scale) + w.ow / 2;
                    o[4] = fd_cvRound(y / L.scale) + w.oh / 2;
                    o[5] = w.ow; o[6] = w.oh;
                }
        }
        if (total > cap) FD_THROW(FD_ERR_CAPACITY, "fd_pyramid_windows: %lld windows, capacity %lld", (long long)total, (long long)cap);
    });
}

int fd_greyworld(fd_ctx* ctx, const uint8_t* bgr, int w, int h, uint8_t* dst, int is_device) {
    return fd_guard(ctx, [&] {
        if (!ctx || !bgr || !dst || w < 1 || h < 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_greyworld: bad argument");
        HIP_CHECK(hipSetDevice(ctx->device));
        const int n = w * h;
        DevBuf in, out, stats;
        const uint8_t* din = bgr;
        uint8_t* dout = dst;
        if (!is_device) {
            in.reserve((size_t)n * 3);
            out.reserve((size_t)n * 3);
            HIP_CHECK(hipMemcpyAsync(in.p, bgr, (size_t)n * 3, hipMemcpyHostToDevice, ctx->stream));
            din = in.as<uint8_t>();
            dout = out.as<uint8_t>();
        }
        stats.reserve(64);
        HIP_CHECK(hipMemsetAsync(stats.p, 0, 64, ctx->stream));
        unsigned long long* sums = stats.as<unsigned long long>();
        unsigned int* maxs = (unsigned int*)(sums + 3);
        hipLaunchKernelGGL(k_greyworld_stats, dim3(grid_for(n)), dim3(256), 0, ctx->stream, din, n, sums, maxs);
        HIP_CHECK(hipGetLastError());
        unsigned long long hs[5] = {0, 0, 0, 0, 0};
        HIP_CHECK(hipMemcpyAsync(hs, stats.p, 40, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        unsigned int hm[3];
        std::memcpy(hm, (char*)hs + 24, 12);
        // scalar part exactly as GreyWorldNormalizationFilter.cpp:45-60 (double, host)
        double mean[3], maxNew[3], scale[3];
        for (int c = 0; c < 3; ++c) { mean[c] = (double)hs[c] / n; maxNew[c] = (uint8_t)hm[c] / mean[c]; }
        double mx = maxNew[0];
        if (maxNew[1] > mx) mx = maxNew[1];
        if (maxNew[2] > mx) mx = maxNew[2];
        for (int c = 0; c < 3; ++c) scale[c] = 255.0 / (mean[c] * mx);
        hipLaunchKernelGGL(k_greyworld_apply, dim3(grid_for(n)), dim3(256), 0, ctx->stream, din, dout, n, scale[0], scale[1], scale[2]);
        HIP_CHECK(hipGetLastError());
        if (!is_device) HIP_CHECK(hipMemcpyAsync(dst, dout, (size_t)n * 3, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

// ---- stand-alone ImageFilter::applyTo(Mat) forms of the layer filters (ImageFilter.hpp:18-57) on one host image ----------
int fd_gradient_filter_image(fd_ctx* ctx, const uint8_t* gray, int w, int h, int grad_kernel, int blur_kernel, uint8_t* dst2ch) {
    return fd_guard(ctx, [&] {
        if (!ctx || !gray || !dst2ch || w < 1 || h < 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_gradient_filter_image: bad argument");
        if (grad_kernel != 1 && grad_kernel != 3 && grad_kernel != 5 && grad_kernel != 7 && grad_kernel != FD_GRAD_SCHARR)
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "GradientFilter: the kernel size must be 1, 3, 5, 7 or CV_SCHARR");
        if (blur_kernel < 0 || blur_kernel > 31) FD_THROW(FD_ERR_INVALID_ARGUMENT, "GradientFilter: the blur kernel size must be in 0..31");
        HIP_CHECK(hipSetDevice(ctx->device));
        const size_t n = (size_t)w * h, boff = (n + 255) & ~(size_t)255;
        DevBuf in, out;   // in: [gray | blurred]: k_box_blur addresses both through one base pointer
        in.reserve(boff + n);
        out.reserve(2 * n);
        HIP_CHECK(hipMemcpyAsync(in.p, gray, n, hipMemcpyHostToDevice, ctx->stream));
        const uint8_t* src = in.as<uint8_t>();
        if (blur_kernel > 0) {
            FilterJobs jobs;
            jobs.n = 1;
            jobs.j[0].w = w; jobs.j[0].h = h; jobs.j[0].src_off = 0; jobs.j[0].dst_off = (uint32_t)boff;
            hipLaunchKernelGGL(k_box_blur, dim3(grid_for((int)n), 1), dim3(256), 0, ctx->stream, in.as<uint8_t>(), blur_kernel, jobs);
            src += boff;
        }
        hipLaunchKernelGGL(k_gradient_image, dim3(grid_for((int)n)), dim3(256), 0, ctx->stream, src, out.as<uint8_t>(), w, h, grad_kernel);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(dst2ch, out.p, 2 * n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}
int fd_gradient_image(fd_ctx* ctx, const uint8_t* gray, int w, int h, int grad_kernel, uint8_t* dst2ch) {
    return fd_gradient_filter_image(ctx, gray, w, h, grad_kernel, 0, dst2ch);
}

int fd_gradient_binning_image(fd_ctx* ctx, const uint8_t* grad2ch, int w, int h, int bins, int signed_gradients, int interpolate, uint8_t* dst) {
    return fd_guard(ctx, [&] {
        if (!ctx || !grad2ch || !dst || w < 1 || h < 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_gradient_binning_image: bad argument");
        if (bins < 1 || bins > 255) FD_THROW(FD_ERR_INVALID_ARGUMENT, "GradientBinningFilter: bins must be in 1..255");
        HIP_CHECK(hipSetDevice(ctx->device));
        const size_t n = (size_t)w * h;
        const int E = interpolate ? 4 : 2;
        std::vector<uint8_t> lut;
        build_gradient_lut(bins, signed_gradients != 0, interpolate != 0, lut);
        DevBuf in, out, dl;
        in.reserve(2 * n);
        out.reserve((size_t)E * n);
        dl.reserve(lut.size());
        HIP_CHECK(hipMemcpyAsync(dl.p, lut.data(), lut.size(), hipMemcpyHostToDevice, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(in.p, grad2ch, 2 * n, hipMemcpyHostToDevice, ctx->stream));
        if (E == 2) hipLaunchKernelGGL(k_binning_image<2>, dim3(grid_for((int)n)), dim3(256), 0, ctx->stream, in.as<uint8_t>(), dl.as<uint8_t>(), out.as<uint8_t>(), (int)n);
        else hipLaunchKernelGGL(k_binning_image<4>, dim3(grid_for((int)n)), dim3(256), 0, ctx->stream, in.as<uint8_t>(), dl.as<uint8_t>(), out.as<uint8_t>(), (int)n);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(dst, out.p, (size_t)E * n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int fd_lbp_image(fd_ctx* ctx, const uint8_t* gray, int w, int h, int lbp_type, uint8_t* dst) {
    return fd_guard(ctx, [&] {
        if (!ctx || !gray || !dst || w < 1 || h < 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_lbp_image: bad argument");
        if (lbp_type < 0 || lbp_type > 3) FD_THROW(FD_ERR_INVALID_ARGUMENT, "LbpFilter: invalid type");
        HIP_CHECK(hipSetDevice(ctx->device));
        const size_t n = (size_t)w * h, doff = (n + 255) & ~(size_t)255;
        DevBuf buf;   // [gray | codes]: k_lbp addresses both through one base pointer
        buf.reserve(doff + n);
        HIP_CHECK(hipMemcpyAsync(buf.p, gray, n, hipMemcpyHostToDevice, ctx->stream));
        FilterJobs jobs;
        jobs.n = 1;
        jobs.j[0].w = w; jobs.j[0].h = h; jobs.j[0].src_off = 0; jobs.j[0].dst_off = (uint32_t)doff;
        hipLaunchKernelGGL(k_lbp, dim3(grid_for((int)n), 1), dim3(256), 0, ctx->stream, buf.as<uint8_t>(), lbp_type, jobs);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(dst, buf.as<uint8_t>() + doff, n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

}  // extern "C"

#ifdef FD_PYR_PROF
extern "C" int fd_debug_pyr_prof(unsigned long long* out) {   // the records of the last k_resize_down launch; returns the capacity
    if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(fd_pyr_prof), sizeof(unsigned long long) * 8 * PYR_PROF_WGS);
    return PYR_PROF_WGS;
}
#endif
