// featuredetection_amd/csrc/fhog.hip -- imageprocessing::filtering::FhogFilter (FhogFilter.cpp:20-132,
// FhogFilter.hpp:120-207) + FhogAggregationFilter::computeDescriptors (FhogAggregationFilter.cpp:38-168) on gray
// images / gray pyramid layers: the cell descriptors (2B signed + B unsigned orientation features + 4 energy features)
// the AggregatedFeaturesDetector family convolves its linear SVM over.  SURVEY.md 8(f) row 2, second piece.
//
// All layers of a launch (one image, or every layer of a pyramid) go through the kernels together over a device layer table:
// k_fhog_coeff (bilinear cell-interpolation tables), k_fhog_grad (per-pixel (bin, weight) entries from the same 511 x 511
// gradient look-up table the reference builds with host libm atan2 / sqrt: FhogFilter.cpp:35-57), k_fhog_hist (lane == cell:
// a lane walks the pixels that contribute to its cell in the reference's row-major scan order -- with bilinear cell
// interpolation a pixel feeds up to four cells, so a cell sees a 2c x 2c neighbourhood -- and accumulates into its private
// 2B-bin histogram in LDS, so every fp32 accumulator receives its addends in the reference order), k_fhog_desc (the four
// neighbourhood normalisers, truncation at alpha, the 0.5 / 0.2357 factors) and, for the detector, k_fhog_score.
#include "fd_internal.hpp"
#include "fd_device.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>

struct FhogLut {   // (bin, weight) of a gradient: what k_fhog_hist consumes per pixel
    uint8_t index1, index2;
    uint16_t pad;
    float weight1, weight2;
};
struct FhogLutEntry {   // one entry per (dy, dx) gradient code (FhogFilter.hpp:108-111: bins + magnitude)
    FhogLut bins;
    float magnitude;
};
struct FhogCoeffDev { int32_t index1, index2; float weight1, weight2; };

struct FhogParamsDev {
    int32_t cell, ubins, sbins, D, interpBins, interpCells;
    float alpha;
    const FhogLutEntry* lut;       // [512 * 512], index dy * 512 + dx
    const FhogCoeffDev* coeff;     // all layers: rows of layer 0, columns of layer 0, rows of layer 1, ...
};
struct FhogLayerDev {              // one gray image / pyramid layer of a launch
    const uint8_t* img;
    int32_t w, h, stride;          // stride in bytes
    int32_t channels;              // 1 or 3 (interleaved)
    int32_t rows, cols;            // cells
    int32_t cellBase, coeffBase;   // first cell / first coefficient of the layer in the launch-wide arrays
    int32_t vw, vh, posBase;       // window positions of the score map (aggregated detector only)
    int32_t cellBlockBase;         // first 64-cell block of the layer
    int32_t pixBase, pixBlockBase; // first covered pixel (rows*cell x cols*cell per layer) / first 256-pixel block
    int32_t posBlockBase;          // first 8-position block of the layer
};

struct FhogLayoutTotals { int cells = 0, coeffs = 0, cellBlocks = 0, positions = 0, posBlocks = 0, pixels = 0, pixBlocks = 0; };

struct fd_aggregated {
    fd_ctx* ctx;
    fd_aggregated_params prm;
    std::vector<float> weights;
    DevBuf dweights, scores;
    fd_pyramid* pyr = nullptr;
    int pyrW = 0, pyrH = 0;
    std::vector<FhogLayerDev> layerTable;   // of the current pyramid geometry
    FhogLayoutTotals layout;
    DevBuf dlayers;
    void* arenaAt = nullptr;
    ~fd_aggregated() { if (pyr) fd_pyramid_destroy(pyr); }
};

namespace {

using namespace fd_dev;

constexpr int FHOG_MAX_SBINS = 36;
constexpr int FHOG_CH = 6;         // pixels whose loads are in flight together in k_fhog_hist

__device__ __forceinline__ int layer_of_block(const FhogLayerDev* __restrict__ layers, int nLayers, int block, bool cells) {
    int l = 0;
    for (int i = 1; i < nLayers; ++i)
        if (block >= (cells ? layers[i].cellBlockBase : layers[i].posBlockBase)) l = i;
    return l;
}

// (bin, weight) entries of every pixel the cells cover: the gradient of FhogFilter.hpp:120-160 (central differences,
// replicated border) as a code into the reference's look-up table, so that k_fhog_hist reads each entry from a dense map
// instead of chasing the table once per neighbouring cell.  The map is stored phase-major, entry (y, x) at
// [(y * cell + x % cell) * cols + x / cell]: in k_fhog_hist lane == cell, so the 64 lanes of a wavefront, which look at the
// same in-cell offset of consecutive cells, read consecutive entries (a lane-per-cell walk over a row-major map touches 64
// cache lines per load and is bound by the L1 tag rate).  Thread order == storage order: coalesced writes.
__global__ __launch_bounds__(256) void k_fhog_grad(const FhogLayerDev* __restrict__ layers, int nLayers, FhogParamsDev d, FhogLut* __restrict__ grad) {
    int l = 0;
    for (int i = 1; i < nLayers; ++i)
        if ((int)blockIdx.x >= layers[i].pixBlockBase) l = i;
    const FhogLayerDev L = layers[l];
    const int W = L.cols * d.cell, H = L.rows * d.cell;
    const int i = (blockIdx.x - L.pixBlockBase) * 256 + threadIdx.x;
    if (i >= W * H) return;
    const int y = i / W, rem = i - y * W;
    const int o = rem / L.cols, cx = rem - o * L.cols;
    const int x = cx * d.cell + o;
    const int py = max(y - 1, 0), ny = min(y + 1, L.h - 1), px = max(x - 1, 0), nx = min(x + 1, L.w - 1);
    const uint8_t* rowp = L.img + (size_t)y * L.stride;
    FhogLut e;
    if (L.channels == 1) {
        const int dx = (int)rowp[nx] - (int)rowp[px] + 256;
        const int dy = (int)L.img[(size_t)ny * L.stride + x] - (int)L.img[(size_t)py * L.stride + x] + 256;
        e = d.lut[dy * 512 + dx].bins;
    } else {   // getBinCoefficients<false> (FhogFilter.hpp:144-172): the channel with the largest gradient magnitude
        const uint8_t* up = L.img + (size_t)py * L.stride + 3 * x;
        const uint8_t* dn = L.img + (size_t)ny * L.stride + 3 * x;
        const uint8_t* lf = rowp + 3 * px;
        const uint8_t* rt = rowp + 3 * nx;
        const FhogLutEntry e1 = d.lut[((int)dn[0] - (int)up[0] + 256) * 512 + ((int)rt[0] - (int)lf[0] + 256)];
        const FhogLutEntry e2 = d.lut[((int)dn[1] - (int)up[1] + 256) * 512 + ((int)rt[1] - (int)lf[1] + 256)];
        const FhogLutEntry e3 = d.lut[((int)dn[2] - (int)up[2] + 256) * 512 + ((int)rt[2] - (int)lf[2] + 256)];
        if (e1.magnitude > e2.magnitude) e = e1.magnitude > e3.magnitude ? e1.bins : e3.bins;
        else e = e2.magnitude > e3.magnitude ? e2.bins : e3.bins;
    }
    grad[(size_t)L.pixBase + i] = e;
}

// createInterpolationCoefficients (FhogFilter.cpp:74-98), one thread per pixel row / column of every layer.
// fp32 add / divide / floor are correctly rounded on the device (no fast-math), so the table equals the host's.
__global__ __launch_bounds__(256) void k_fhog_coeff(const FhogLayerDev* __restrict__ layers, int nLayers, int total, FhogParamsDev d,
                                                    FhogCoeffDev* __restrict__ coeff) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int l = 0;
    for (int k = 1; k < nLayers; ++k)
        if (i >= layers[k].coeffBase) l = k;
    const FhogLayerDev L = layers[l];
    const int j = i - L.coeffBase, rowPixels = L.rows * d.cell;
    const int pixel = j < rowPixels ? j : j - rowPixels;
    const int sizeInCells = j < rowPixels ? L.rows : L.cols;
    FhogCoeffDev c;
    if (d.interpCells) {
        const float realCellIndex = (pixel + 0.5f) / d.cell - 0.5f;
        int index1 = (int)floorf(realCellIndex);
        int index2 = index1 + 1;
        float weight2 = realCellIndex - index1;
        float weight1 = index2 - realCellIndex;
        if (index1 < 0) { index1 = index2; weight1 = 0; }
        else if (index2 >= sizeInCells) { index2 = index1; weight2 = 0; }
        c = FhogCoeffDev{index1, index2, weight1, weight2};
    } else {
        c = FhogCoeffDev{pixel / d.cell, -1, 1.f, 0.f};
    }
    // rows in pixel order; columns phase-major ([x % cell][x / cell]) like the gradient map
    coeff[j < rowPixels ? i : L.coeffBase + rowPixels + (pixel % d.cell) * L.cols + pixel / d.cell] = c;
}

__global__ __launch_bounds__(64) void k_fhog_hist(const FhogLayerDev* __restrict__ layers, int nLayers, FhogParamsDev d,
                                                  const FhogLut* __restrict__ gradAll, float* __restrict__ rawHist, float* __restrict__ energies) {
    __shared__ float hist[FHOG_MAX_SBINS][64];
    const FhogLayerDev L = layers[layer_of_block(layers, nLayers, blockIdx.x, true)];
    const int lane = threadIdx.x;
    const int cellId = (blockIdx.x - L.cellBlockBase) * 64 + lane;
    const bool valid = cellId < L.rows * L.cols;
    const int r = valid ? cellId / L.cols : 0, c = valid ? cellId - r * L.cols : 0;
    for (int b = 0; b < d.sbins; ++b) hist[b][lane] = 0.f;
    if (valid) {
        const FhogCoeffDev* __restrict__ rowCoeff = d.coeff + L.coeffBase;
        const FhogCoeffDev* __restrict__ colCoeff = rowCoeff + L.rows * d.cell;
        // pixel box feeding this cell: non-interpolated [r*cell, (r+1)*cell); interpolated: every pixel whose index1 or index2
        // is r lies within `off` pixels around it
        const int cs = d.cell;
        const int H = L.rows * cs;
        const int off = d.interpCells ? (cs + 1) / 2 + 1 : 0, box = cs + 2 * off;
        // A pixel reaches a cell through at most one row role and one column role with a non-zero weight: where index1 ==
        // index2 (clamped borders, FhogFilter.cpp:88-95) one of the two weights is exactly 0, and the reference's add of
        // e.weight * 0 * w = +0 leaves the (non-negative) accumulator unchanged.  So the effective weight of a row / column is
        // (index1 hit ? weight1 : 0) + (index2 hit ? weight2 : 0) -- exact, one addend is 0 -- and every pixel costs one
        // read-modify-write per bin, in scan order.
        const FhogLut* __restrict__ gbase = gradAll + (size_t)L.pixBase;
        for (int ii = 0; ii < box; ++ii) {
            const int y = r * cs - off + ii;
            if (y < 0 || y >= H) continue;
            const FhogCoeffDev rc = rowCoeff[y];
            const bool r1 = rc.index1 == r, r2 = d.interpCells && rc.index2 == r;
            if (!r1 && !r2) continue;
            const float wr = (r1 ? rc.weight1 : 0.f) + (r2 ? rc.weight2 : 0.f);
            const size_t yrow = (size_t)y * cs;
            for (int jb = 0; jb < box; jb += FHOG_CH) {
                FhogCoeffDev cc[FHOG_CH];
                FhogLut e[FHOG_CH];
                bool in[FHOG_CH];
#pragma unroll
                for (int j = 0; j < FHOG_CH; ++j) {   // the loads of FHOG_CH pixels are issued together
                    // x = c*cs - off + jb + j = (c + q - 2) * cs + o with wave-uniform q, o
                    const int t = 2 * cs - off + jb + j, q = t / cs, o = t - q * cs;
                    const int cx = c + q - 2;
                    in[j] = jb + j < box && cx >= 0 && cx < L.cols;
                    const int cxs = in[j] ? cx : c;
                    cc[j] = colCoeff[o * L.cols + cxs];
                    e[j] = gbase[(yrow + o) * L.cols + cxs];
                }
#pragma unroll
                for (int j = 0; j < FHOG_CH; ++j) {
                    const bool c1 = cc[j].index1 == c, c2 = d.interpCells && cc[j].index2 == c;
                    if (!in[j] || (!c1 && !c2)) continue;
                    if (d.interpCells) {
                        const float wc = (c1 ? cc[j].weight1 : 0.f) + (c2 ? cc[j].weight2 : 0.f);
                        hist[e[j].index1][lane] = hist[e[j].index1][lane] + e[j].weight1 * wr * wc;
                        if (d.interpBins) hist[e[j].index2][lane] = hist[e[j].index2][lane] + e[j].weight2 * wr * wc;
                    } else {
                        hist[e[j].index1][lane] = hist[e[j].index1][lane] + e[j].weight1;
                        if (d.interpBins) hist[e[j].index2][lane] = hist[e[j].index2][lane] + e[j].weight2;
                    }
                }
            }
        }
        float energy = 0.f;   // computeGradientEnergy, FhogAggregationFilter.cpp:53-61
        for (int b = 0; b < d.ubins; ++b) {
            const float u = hist[b][lane] + hist[b + d.ubins][lane];
            energy = energy + u * u;
        }
        energies[L.cellBase + cellId] = energy;
    }
    // raw histograms [cell][2B] of the block's 64 consecutive cells, written as one contiguous run
    wave_sync();
    const int cellsHere = min(64, L.rows * L.cols - (blockIdx.x - L.cellBlockBase) * 64);
    float* out = rawHist + ((size_t)L.cellBase + (size_t)(blockIdx.x - L.cellBlockBase) * 64) * d.sbins;
    for (int i = lane; i < cellsHere * d.sbins; i += 64) {
        const int cl = i / d.sbins, b = i - cl * d.sbins;
        out[i] = hist[b][cl];
    }
}

// FhogAggregationFilter::computeDescriptors (:63-148) from the raw histograms: LPC (32 or 64) lanes per cell, lane == output
// feature, so that the [cell][2B] reads and the [cell][3B+4] writes are contiguous; every lane derives the cell's four
// normalisers itself (16 cached energy reads).
template <int LPC>
__global__ __launch_bounds__(256) void k_fhog_desc(const FhogLayerDev* __restrict__ layers, int nLayers, int totalCells, FhogParamsDev d,
                                                   const float* __restrict__ energiesAll, const float* __restrict__ rawHist, float* __restrict__ descAll) {
    const int g = (blockIdx.x * 256 + threadIdx.x) / LPC, f = threadIdx.x & (LPC - 1);
    if (g >= totalCells || f >= d.D) return;
    int l = 0;
    for (int i = 1; i < nLayers; ++i)
        if (g >= layers[i].cellBase) l = i;
    const FhogLayerDev L = layers[l];
    const int cellId = g - L.cellBase;
    const float* energies = energiesAll + L.cellBase;
    const int r = cellId / L.cols, c = cellId - r * L.cols;
    const int pr = max(r - 1, 0), nr = min(r + 1, L.rows - 1), pc = max(c - 1, 0), nc = min(c + 1, L.cols - 1);
    auto E = [&](int rr, int cc) { return energies[rr * L.cols + cc]; };
    const float eps = 1e-4f;
    float n[4];   // computeNormalizers, FhogAggregationFilter.cpp:77-99
    n[0] = 1.f / sqrtf(E(pr, pc) + E(pr, c) + E(r, pc) + E(r, c) + eps);
    n[1] = 1.f / sqrtf(E(pr, c) + E(pr, nc) + E(r, c) + E(r, nc) + eps);
    n[2] = 1.f / sqrtf(E(r, pc) + E(r, c) + E(nr, pc) + E(nr, c) + eps);
    n[3] = 1.f / sqrtf(E(r, c) + E(r, nc) + E(nr, c) + E(nr, nc) + eps);
    const float* h = rawHist + (size_t)g * d.sbins;
    float out;   // computeDescriptor, :101-148 (0.5 and 0.2357 are double literals)
    if (f < d.sbins) {
        const float v = h[f];
        const float v0 = fminf(d.alpha, n[0] * v), v1 = fminf(d.alpha, n[1] * v), v2 = fminf(d.alpha, n[2] * v), v3 = fminf(d.alpha, n[3] * v);
        out = (float)(0.5 * (double)(v0 + v1 + v2 + v3));
    } else if (f < d.sbins + d.ubins) {
        const int b = f - d.sbins;
        const float v = h[b] + h[b + d.ubins];
        const float s = fminf(d.alpha, n[0] * v) + fminf(d.alpha, n[1] * v) + fminf(d.alpha, n[2] * v) + fminf(d.alpha, n[3] * v);
        out = (float)(0.5 * (double)s);
    } else {
        const float ni = n[f - d.sbins - d.ubins];
        float energy = 0.f;
        for (int b = 0; b < d.sbins; ++b) energy = energy + fminf(d.alpha, ni * h[b]);
        out = (float)(0.2357 * (double)energy);
    }
    descAll[(size_t)g * d.D + f] = out;
}

struct FhogScratch {
    DevBuf lut, coeff, img, desc, energies, layers, grad, hist;
    fd_fhog_params lutFor;
    bool lutValid = false;
};
FhogScratch& scratch(fd_ctx* ctx) { return fd_scratch<FhogScratch>(ctx); }

// gradient look-up table of FhogFilter::createGradientLut (FhogFilter.cpp:35-57), host libm like the reference
void build_lut(fd_ctx* ctx, FhogScratch& S, const fd_fhog_params& fp) {
    if (S.lutValid && S.lutFor.unsigned_bins == fp.unsigned_bins && S.lutFor.interpolate_bins == fp.interpolate_bins) return;
    const int signedBinCount = 2 * fp.unsigned_bins;
    const float TWO_PI = (float)(2 * M_PI);
    const float value2bin = signedBinCount / TWO_PI;
    std::vector<FhogLutEntry> lut((size_t)512 * 512);
    std::memset(lut.data(), 0, sizeof(FhogLutEntry) * lut.size());
    for (int gradientCodeX = 1; gradientCodeX < 512; ++gradientCodeX) {
        const float gradientX = (gradientCodeX - 256) / (255.0f * 2.0f);
        for (int gradientCodeY = 1; gradientCodeY < 512; ++gradientCodeY) {
            const float gradientY = (gradientCodeY - 256) / (255.0f * 2.0f);
            const float magnitude = std::sqrt(gradientX * gradientX + gradientY * gradientY);
            float orientation = std::atan2(gradientY, gradientX);
            if (orientation < 0) orientation += TWO_PI;
            FhogLut e;
            std::memset(&e, 0, sizeof(e));
            if (fp.interpolate_bins) {
                const float bin = orientation * value2bin;
                int i1 = (int)bin, i2 = i1 + 1;
                if (i2 == signedBinCount) i2 = 0;
                e.index1 = (uint8_t)i1; e.index2 = (uint8_t)i2;
                e.weight2 = magnitude * (bin - i1);
                e.weight1 = magnitude - e.weight2;
            } else {
                int bin = (int)(orientation * value2bin + 0.5f);
                if (bin == signedBinCount) bin = 0;
                e.index1 = (uint8_t)bin; e.weight1 = magnitude;
            }
            lut[(size_t)gradientCodeY * 512 + gradientCodeX] = FhogLutEntry{e, magnitude};
        }
    }
    S.lut.reserve(sizeof(FhogLutEntry) * lut.size());
    HIP_CHECK(hipMemcpy(S.lut.p, lut.data(), sizeof(FhogLutEntry) * lut.size(), hipMemcpyHostToDevice));
    S.lutFor = fp;
    S.lutValid = true;
}

void check_fhog_params(const fd_fhog_params& fp) {
    if (fp.cell_size < 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "FhogFilter: cellSize must be bigger than zero");
    if (fp.unsigned_bins < 1 || 2 * fp.unsigned_bins > FHOG_MAX_SBINS)
        FD_THROW(FD_ERR_INVALID_ARGUMENT, "FhogFilter: unsignedBinCount must be bigger than zero, but was: %d (this backend: <= %d)", fp.unsigned_bins,
                 FHOG_MAX_SBINS / 2);
    if (!(fp.alpha > 0)) FD_THROW(FD_ERR_INVALID_ARGUMENT, "FhogAggregationFilter: alpha must be bigger than zero, but was: %g", (double)fp.alpha);
}

// fills the launch-wide offsets of a layer list (img, w, h, stride, vw, vh set by the caller); returns total cells
using FhogLayout = FhogLayoutTotals;
FhogLayout layout_layers(std::vector<FhogLayerDev>& layers, const fd_fhog_params& fp) {
    FhogLayout t;
    for (FhogLayerDev& L : layers) {
        L.rows = L.h / fp.cell_size;
        L.cols = L.w / fp.cell_size;
        L.cellBase = t.cells; L.coeffBase = t.coeffs; L.cellBlockBase = t.cellBlocks; L.posBase = t.positions; L.posBlockBase = t.posBlocks;
        L.pixBase = t.pixels; L.pixBlockBase = t.pixBlocks;
        const int64_t npix = (int64_t)L.rows * L.cols * fp.cell_size * fp.cell_size;
        if (t.pixels + npix > (int64_t)0x7fffff00) FD_THROW(FD_ERR_INVALID_ARGUMENT, "FhogFilter: the layers of one launch exceed 2^31 pixels");
        t.pixels += (int)npix;
        t.pixBlocks += (int)((npix + 255) / 256);
        t.cells += L.rows * L.cols;
        t.coeffs += (L.rows + L.cols) * fp.cell_size;
        t.cellBlocks += (L.rows * L.cols + 63) / 64;
        t.positions += L.vw * L.vh;
        t.posBlocks += (((L.vw + 3) / 4) * L.vh + 7) / 8;   // k_fhog_score: 8 groups of FHOG_SP = 4 positions per block
    }
    return t;
}

// descriptors of every layer of the table at dlayers (device copy of `layers`, laid out by layout_layers) into S.desc
FhogParamsDev run_fhog(fd_ctx* ctx, FhogScratch& S, const FhogLayerDev* dlayers, int nLayers, const FhogLayout& t, const fd_fhog_params& fp) {
    check_fhog_params(fp);
    build_lut(ctx, S, fp);
    FhogParamsDev d;
    std::memset(&d, 0, sizeof(d));
    d.cell = fp.cell_size; d.ubins = fp.unsigned_bins; d.sbins = 2 * fp.unsigned_bins; d.D = 3 * fp.unsigned_bins + 4;
    d.interpBins = fp.interpolate_bins != 0; d.interpCells = fp.interpolate_cells != 0; d.alpha = fp.alpha;
    if (t.cells == 0) return d;
    S.coeff.reserve(sizeof(FhogCoeffDev) * (size_t)t.coeffs);
    S.desc.reserve(sizeof(float) * (size_t)t.cells * d.D);
    S.energies.reserve(sizeof(float) * (size_t)t.cells);
    S.grad.reserve(sizeof(FhogLut) * (size_t)t.pixels);
    S.hist.reserve(sizeof(float) * (size_t)t.cells * d.sbins);
    d.lut = S.lut.as<FhogLutEntry>();
    d.coeff = S.coeff.as<FhogCoeffDev>();
    hipLaunchKernelGGL(k_fhog_coeff, dim3((t.coeffs + 255) / 256), dim3(256), 0, ctx->stream, dlayers, nLayers, t.coeffs, d, S.coeff.as<FhogCoeffDev>());
    hipLaunchKernelGGL(k_fhog_grad, dim3(t.pixBlocks), dim3(256), 0, ctx->stream, dlayers, nLayers, d, S.grad.as<FhogLut>());
    hipLaunchKernelGGL(k_fhog_hist, dim3(t.cellBlocks), dim3(64), 0, ctx->stream, dlayers, nLayers, d, S.grad.as<FhogLut>(), S.hist.as<float>(),
                       S.energies.as<float>());
    if (d.D <= 32)
        hipLaunchKernelGGL(k_fhog_desc<32>, dim3((unsigned)(((int64_t)t.cells * 32 + 255) / 256)), dim3(256), 0, ctx->stream, dlayers, nLayers, t.cells, d,
                           S.energies.as<float>(), S.hist.as<float>(), S.desc.as<float>());
    else
        hipLaunchKernelGGL(k_fhog_desc<64>, dim3((unsigned)(((int64_t)t.cells * 64 + 255) / 256)), dim3(256), 0, ctx->stream, dlayers, nLayers, t.cells, d,
                           S.energies.as<float>(), S.hist.as<float>(), S.desc.as<float>());
    HIP_CHECK(hipGetLastError());
    return d;
}

// one gray image already on the device
void run_fhog_single(fd_ctx* ctx, FhogScratch& S, const uint8_t* dimg, int w, int h, int stride, const fd_fhog_params& fp, int& rows, int& cols,
                     int channels = 1) {
    check_fhog_params(fp);
    std::vector<FhogLayerDev> layers(1);
    std::memset(&layers[0], 0, sizeof(FhogLayerDev));
    layers[0].img = dimg; layers[0].w = w; layers[0].h = h; layers[0].stride = stride; layers[0].channels = channels;
    const FhogLayout t = layout_layers(layers, fp);
    rows = layers[0].rows; cols = layers[0].cols;
    if (rows == 0 || cols == 0) return;
    S.layers.reserve(sizeof(FhogLayerDev));
    HIP_CHECK(hipMemcpyAsync(S.layers.p, layers.data(), sizeof(FhogLayerDev), hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));   // `layers` is pageable host memory
    run_fhog(ctx, S, S.layers.as<FhogLayerDev>(), 1, t, fp);
}

// ConvolutionFilter(CV_32F) of AggregatedFeaturesDetector (ConvolutionFilter.cpp:27-43 with anchor (0, 0), delta = -bias):
// score(y, x) = delta + sum over channels c of [sum over the kernel window, row-major, of K[ky][kx][c] * F[y+ky][x+kx][c]]:
// the nesting and the fp32 accumulation order of the per-channel cv::filter2D + channel sum.  All layers in one launch;
// a group of 32 lanes (lane == channel, D <= 32: coalesced reads of a cell's descriptor) owns FHOG_SP horizontally adjacent
// window positions and slides a register window over the descriptor row, so a kernel column costs one K and one F load for
// FHOG_SP multiply-adds; every position's per-channel sum still runs in kernel row-major order.  The channel sums are then
// added in channel order onto delta by one lane per position.
constexpr int FHOG_SP = 4;
__global__ __launch_bounds__(256) void k_fhog_score(const FhogLayerDev* __restrict__ layers, int nLayers, const float* __restrict__ descAll, int D,
                                                    const float* __restrict__ K, int kh, int kw, float delta, float* __restrict__ scores) {
    __shared__ float part[8][FHOG_SP][33];
    const FhogLayerDev L = layers[layer_of_block(layers, nLayers, blockIdx.x, false)];
    const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int gpr = (L.vw + FHOG_SP - 1) / FHOG_SP;   // groups per score row
    const int g = (blockIdx.x - L.posBlockBase) * 8 + grp;
    const bool valid = g < gpr * L.vh;
    const int y = valid ? g / gpr : 0, x0 = valid ? (g - y * gpr) * FHOG_SP : 0;
    const float* F = descAll + (size_t)L.cellBase * D;
    float sacc[FHOG_SP];
#pragma unroll
    for (int p = 0; p < FHOG_SP; ++p) sacc[p] = 0.f;
    if (valid && lane < D) {
        const int lastCol = L.cols - 1;
        for (int ky = 0; ky < kh; ++ky) {
            const float* frow = F + (size_t)(y + ky) * L.cols * D + lane;
            const float* krow = K + (size_t)ky * kw * D + lane;
            float w[FHOG_SP];
#pragma unroll
            for (int p = 0; p < FHOG_SP; ++p) w[p] = frow[(size_t)min(x0 + p, lastCol) * D];
            for (int kx = 0; kx < kw; ++kx) {
                const float k = krow[(size_t)kx * D];
                const float nxt = frow[(size_t)min(x0 + kx + FHOG_SP, lastCol) * D];
#pragma unroll
                for (int p = 0; p < FHOG_SP; ++p) sacc[p] = sacc[p] + k * w[p];
#pragma unroll
                for (int p = 0; p + 1 < FHOG_SP; ++p) w[p] = w[p + 1];
                w[FHOG_SP - 1] = nxt;
            }
        }
    }
#pragma unroll
    for (int p = 0; p < FHOG_SP; ++p) part[grp][p][lane] = sacc[p];
    __syncthreads();
    if (valid && lane < FHOG_SP && x0 + lane < L.vw) {
        float score = delta;
        for (int c = 0; c < D; ++c) score = score + part[grp][lane][c];
        scores[L.posBase + y * L.vw + x0 + lane] = score;
    }
}

// descriptors wider than 32 channels (more than 9 unsigned bins): one lane per position, one block row per layer
__global__ __launch_bounds__(256) void k_fhog_score_wide(const FhogLayerDev* __restrict__ layers, const float* __restrict__ descAll, int D,
                                                         const float* __restrict__ K, int kh, int kw, float delta, float* __restrict__ scores) {
    const FhogLayerDev L = layers[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.vw * L.vh) return;
    const int y = i / L.vw, x = i - y * L.vw;
    const float* F = descAll + (size_t)L.cellBase * D;
    float score = delta;
    for (int c = 0; c < D; ++c) {
        float sacc = 0.f;
        for (int ky = 0; ky < kh; ++ky) {
            const float* frow = F + ((size_t)(y + ky) * L.cols + x) * D + c;
            const float* krow = K + (size_t)ky * kw * D + c;
            for (int kx = 0; kx < kw; ++kx) sacc = sacc + krow[(size_t)kx * D] * frow[(size_t)kx * D];
        }
        score = score + sacc;
    }
    scores[L.posBase + i] = score;
}

}  // namespace

extern "C" {

int fd_fhog_size(const fd_fhog_params* fp, int width, int height, int* rows, int* cols, int* channels) {
    if (!fp || fp->cell_size < 1 || fp->unsigned_bins < 1) return FD_ERR_INVALID_ARGUMENT;
    if (rows) *rows = height / fp->cell_size;
    if (cols) *cols = width / fp->cell_size;
    if (channels) *channels = 3 * fp->unsigned_bins + 4;
    return FD_OK;
}

int fd_fhog_image(fd_ctx* ctx, const uint8_t* gray, int width, int height, const fd_fhog_params* fp, float* out) {
    return fd_fhog_image_channels(ctx, gray, width, height, 1, fp, out);
}

int fd_fhog_image_channels(fd_ctx* ctx, const uint8_t* image, int width, int height, int channels, const fd_fhog_params* fp, float* out) {
    return fd_guard(ctx, [&] {
        if (!ctx || !image || !fp || !out || width < 1 || height < 1) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_fhog_image: bad argument");
        if (channels != 1 && channels != 3) FD_THROW(FD_ERR_INVALID_ARGUMENT, "FhogFilter: the image type must be CV_8UC1 or CV_8UC3 (%d channels)", channels);
        HIP_CHECK(hipSetDevice(ctx->device));
        FhogScratch& S = scratch(ctx);
        const size_t bytes = (size_t)width * height * channels;
        S.img.reserve(bytes);
        HIP_CHECK(hipMemcpyAsync(S.img.p, image, bytes, hipMemcpyHostToDevice, ctx->stream));
        int rows, cols;
        run_fhog_single(ctx, S, S.img.as<uint8_t>(), width, height, width * channels, *fp, rows, cols, channels);
        if (rows && cols)
            HIP_CHECK(hipMemcpyAsync(out, S.desc.p, sizeof(float) * (size_t)rows * cols * (3 * fp->unsigned_bins + 4), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int fd_pyramid_fhog_layer(fd_ctx* ctx, fd_pyramid* p, int layer, const fd_fhog_params* fp, float* out) {
    return fd_guard(ctx, [&] {
        if (!ctx || !p || !fp || !out) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_pyramid_fhog_layer: NULL argument");
        if (p->ctx != ctx) FD_THROW(FD_ERR_INVALID_ARGUMENT, "objects belong to different contexts");
        if (p->filter_kind != FD_LAYER_NONE) FD_THROW(FD_ERR_INVALID_ARGUMENT, "FhogFilter needs a gray pyramid (no layer filter)");
        fd_pyramid_require_single(p, "fd_pyramid_fhog_layer");
        if (layer < 0 || layer >= (int)p->kept.size()) FD_THROW(FD_ERR_INVALID_ARGUMENT, "no such pyramid layer: %d", layer);
        HIP_CHECK(hipSetDevice(ctx->device));
        const HostLayer& L = p->all[p->kept[layer]];
        FhogScratch& S = scratch(ctx);
        int rows, cols;
        run_fhog_single(ctx, S, p->arena.as<uint8_t>() + L.gray_off, L.w, L.h, L.w, *fp, rows, cols);
        if (rows && cols)
            HIP_CHECK(hipMemcpyAsync(out, S.desc.p, sizeof(float) * (size_t)rows * cols * (3 * fp->unsigned_bins + 4), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int fd_aggregated_create(fd_ctx* ctx, const fd_aggregated_params* prm, fd_aggregated** out) {
    return fd_guard(ctx, [&] {
        if (!ctx || !prm || !out || !prm->svm_weights) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_aggregated_create: NULL argument");
        if (prm->window_w < 1 || prm->window_h < 1 || prm->octave_layer_count < 1)
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "AggregatedFeaturesDetector: window size and octave layer count must be positive");
        if (prm->fhog.cell_size < 1 || prm->fhog.unsigned_bins < 1 || 2 * prm->fhog.unsigned_bins > FHOG_MAX_SBINS || !(prm->fhog.alpha > 0))
            FD_THROW(FD_ERR_INVALID_ARGUMENT, "AggregatedFeaturesDetector: invalid FhogFilter parameters");
        HIP_CHECK(hipSetDevice(ctx->device));
        std::unique_ptr<fd_aggregated> a(new fd_aggregated());
        a->ctx = ctx;
        a->prm = *prm;
        const size_t nw = (size_t)prm->window_w * prm->window_h * (3 * prm->fhog.unsigned_bins + 4);
        a->weights.assign(prm->svm_weights, prm->svm_weights + nw);
        a->prm.svm_weights = nullptr;
        a->dweights.reserve(sizeof(float) * nw);
        HIP_CHECK(hipMemcpy(a->dweights.p, a->weights.data(), sizeof(float) * nw, hipMemcpyHostToDevice));
        *out = a.release();
    });
}

void fd_aggregated_destroy(fd_aggregated* a) { delete a; }

int fd_aggregated_detect(fd_ctx* ctx, fd_aggregated* a, const uint8_t* image, int width, int height, int channels, int is_device, fd_box* out,
                         int cap, int* count, fd_box* candidates, int cand_cap, int* cand_count) {
    return fd_guard(ctx, [&] {
        if (!ctx || !a || !image || !count) FD_THROW(FD_ERR_INVALID_ARGUMENT, "fd_aggregated_detect: NULL argument");
        if (a->ctx != ctx) FD_THROW(FD_ERR_INVALID_ARGUMENT, "objects belong to different contexts");
        const fd_aggregated_params& P = a->prm;
        HIP_CHECK(hipSetDevice(ctx->device));
        // feature pyramid limits (AggregatedFeaturesExtractor.cpp:30-31,47-52,58-77), recomputed when the image size changes
        if (!a->pyr || a->pyrW != width || a->pyrH != height) {
            if (a->pyr) { fd_pyramid_destroy(a->pyr); a->pyr = nullptr; }
            a->layerTable.clear();
            const int patchWpx = P.window_w * P.fhog.cell_size, patchHpx = P.window_h * P.fhog.cell_size;
            const double inc = std::pow(0.5, 1. / P.octave_layer_count);
            double maxScale = 1.0;
            if (P.min_window_width > patchWpx) {
                const double m = (double)patchWpx / P.min_window_width;
                const int minLayerIndex = (int)std::ceil(std::log(m) / std::log(inc));
                maxScale = std::pow(inc, minLayerIndex);
            }
            const double aspectRatio = (double)patchHpx / (double)patchWpx, imageAspectRatio = (double)height / (double)width;
            const int maxWidth = aspectRatio > imageAspectRatio ? (int)(height / aspectRatio) : width;
            const double m = (double)patchWpx / maxWidth;
            const int maxLayerIndex = (int)(std::log(m) / std::log(inc));
            const double minScale = std::pow(inc, maxLayerIndex);
            const int rc = fd_pyramid_create(ctx, P.octave_layer_count, minScale, maxScale, &a->pyr);
            if (rc != FD_OK) throw FdError{rc, ctx->error};
            a->pyrW = width; a->pyrH = height;
        }
        {
            const int rc = fd_pyramid_update(a->pyr, image, width, height, channels, is_device);
            if (rc != FD_OK) throw FdError{rc, ctx->error};
        }
        fd_pyramid* p = a->pyr;
        if (p->kept.size() < 2)   // ImagePyramid::estimateLambdas (ImagePyramid.cpp:240-242) of the score pyramid
            FD_THROW(FD_ERR_RUNTIME, "ImagePyramid: at least two pyramid layers are needed to estimate the lambdas");
        FhogScratch& S = scratch(ctx);
        const int D = 3 * P.fhog.unsigned_bins + 4;
        // layer table of this pyramid geometry (rebuilt with the pyramid): descriptors and score maps of all layers run as
        // single launches over the table
        if (a->layerTable.empty() || a->arenaAt != p->arena.p) {
            a->layerTable.resize(p->kept.size());
            for (size_t li = 0; li < p->kept.size(); ++li) {
                const HostLayer& L = p->all[p->kept[li]];
                FhogLayerDev& T = a->layerTable[li];
                std::memset(&T, 0, sizeof(T));
                T.img = p->arena.as<uint8_t>() + L.gray_off; T.w = L.w; T.h = L.h; T.stride = L.w; T.channels = 1;
                T.vh = std::max(L.h / P.fhog.cell_size - P.window_h + 1, 0);
                T.vw = std::max(L.w / P.fhog.cell_size - P.window_w + 1, 0);
                if (T.vw == 0 || T.vh == 0) T.vw = T.vh = 0;
            }
            a->layout = layout_layers(a->layerTable, P.fhog);
            a->dlayers.reserve(sizeof(FhogLayerDev) * a->layerTable.size());
            HIP_CHECK(hipMemcpy(a->dlayers.p, a->layerTable.data(), sizeof(FhogLayerDev) * a->layerTable.size(), hipMemcpyHostToDevice));
            a->arenaAt = p->arena.p;
        }
        const int nLayers = (int)a->layerTable.size();
        std::vector<size_t> off(p->kept.size() + 1, 0);
        std::vector<int> vw(p->kept.size()), vh(p->kept.size());
        for (size_t li = 0; li < p->kept.size(); ++li) {
            vw[li] = a->layerTable[li].vw; vh[li] = a->layerTable[li].vh;
            off[li] = (size_t)a->layerTable[li].posBase;
        }
        off[p->kept.size()] = (size_t)a->layout.positions;
        a->scores.reserve(sizeof(float) * std::max<size_t>(off.back(), 1));
        run_fhog(ctx, S, a->dlayers.as<FhogLayerDev>(), nLayers, a->layout, P.fhog);
        if (a->layout.positions > 0) {
            if (D <= 32) {
                hipLaunchKernelGGL(k_fhog_score, dim3(a->layout.posBlocks), dim3(256), 0, ctx->stream, a->dlayers.as<FhogLayerDev>(), nLayers,
                                   S.desc.as<float>(), D, a->dweights.as<float>(), P.window_h, P.window_w, -P.svm_bias, a->scores.as<float>());
            } else {
                int maxPos = 0;
                for (const FhogLayerDev& T : a->layerTable) maxPos = std::max(maxPos, T.vw * T.vh);
                hipLaunchKernelGGL(k_fhog_score_wide, dim3((maxPos + 255) / 256, nLayers), dim3(256), 0, ctx->stream, a->dlayers.as<FhogLayerDev>(),
                                   S.desc.as<float>(), D, a->dweights.as<float>(), P.window_h, P.window_w, -P.svm_bias, a->scores.as<float>());
            }
            HIP_CHECK(hipGetLastError());
        }
        std::vector<float> hs(off.back());
        if (!hs.empty()) HIP_CHECK(hipMemcpyAsync(hs.data(), a->scores.p, sizeof(float) * hs.size(), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        // getPositiveWindows (AggregatedFeaturesDetector.cpp:87-106): layer, row, column order
        std::vector<fd_box> cand;
        for (size_t li = 0; li < p->kept.size(); ++li) {
            const HostLayer& L = p->all[p->kept[li]];
            const double scaleX = (double)L.w / (double)width, scaleY = (double)L.h / (double)height;   // ImagePyramid.cpp:178-179,187-188
            for (int y = 0; y < vh[li]; ++y)
                for (int x = 0; x < vw[li]; ++x) {
                    const float score = hs[off[li] + (size_t)y * vw[li] + x];
                    if (!(score > P.score_threshold)) continue;
                    const int cs = P.fhog.cell_size;
                    const int bx = (int)std::round((x * cs) / scaleX), by = (int)std::round((y * cs) / scaleY);
                    const int bw = (int)std::round((P.window_w * cs) / scaleX), bh = (int)std::round((P.window_h * cs) / scaleY);
                    const int cx = bx + bw / 2, cy = by + bh / 2;                                // Patch::computeCenter
                    const int rw = (int)(P.width_scale * bw), rh = (int)(P.height_scale * bh);  // rescaleWindow :108-112
                    cand.push_back(fd_box{score, cx - rw / 2, cy - rh / 2, rw, rh});
                }
        }
        if (cand_count) *cand_count = (int)cand.size();
        if (candidates)
            for (size_t i = 0; i < cand.size() && (int)i < cand_cap; ++i) candidates[i] = cand[i];
        std::vector<fd_box> fin(cand.size());
        int nfin = 0;
        const int rc = fd_nms_iou(cand.data(), (int)cand.size(), P.nms_overlap_threshold, P.nms_maximum_type, fin.data(), &nfin);
        if (rc != FD_OK) FD_THROW(rc, "NonMaximumSuppression failed (overlap threshold %g, maximum type %d)", P.nms_overlap_threshold, P.nms_maximum_type);
        *count = nfin;
        for (int i = 0; i < nfin && i < cap && out; ++i) out[i] = fin[i];
        if (out && nfin > cap) FD_THROW(FD_ERR_CAPACITY, "fd_aggregated_detect: %d detections, capacity %d", nfin, cap);
    });
}

}  // extern "C"
