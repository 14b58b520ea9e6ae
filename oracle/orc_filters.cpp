// oracle/orc_filters.cpp -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
// Patch / layer filters of libImageProcessing that sit on the hot path.
#include "orc_common.h"
#include "orc_internal.h"
#include "oracle.h"
#include <array>
#include <complex>
#include <cstring>

namespace orc {

// HistEq64Filter.cpp:32-125
void histeq64(const uchar* src, int w, int h, int stride, uchar* dst) {
    float stretchFactor = 255.0f / (float)(w * h);
    float pdf_bins[64];
    for (int i = 0; i < 64; i++) pdf_bins[i] = 0.0f;
    for (int z = 0; z < h; z++) {
        const uchar* o = src + (size_t)z * stride;
        for (int i = 0; i < w; i++) pdf_bins[o[i] >> 2] = pdf_bins[o[i] >> 2] + 1;
    }
    for (int i = 0; i < 64; i++)
        if (pdf_bins[i] != 0) pdf_bins[i] = pdf_bins[i] * stretchFactor;
    float cdf_BINS[64];
    cdf_BINS[0] = pdf_bins[0];
    for (unsigned int i = 1; i < 64; i++) cdf_BINS[i] = cdf_BINS[i - 1] + pdf_bins[i];
    float LUTeq[256];
    for (int i = 0; i < 256; i++) LUTeq[i] = cdf_BINS[i >> 2];
    for (int z = 0; z < h; z++) {
        const uchar* o = src + (size_t)z * stride;
        uchar* f = dst + (size_t)z * w;
        for (int i = 0; i < w; i++) f[i] = (uchar)std::floor(LUTeq[o[i]] + 0.5);
    }
}

// GradientBinningFilter.cpp:18-60.  lut index = gx | gy<<8 (little-endian union), entry 2 or 4 bytes.
static void gradient_binning_lut(int bins, bool signedGradients, bool interpolate, uchar* lut) {
    const double PI = 3.1415926535897932384626433832795;  // CV_PI
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
            size_t index = (size_t)x | ((size_t)y << 8);
            if (!interpolate) {
                lut[2 * index] = (uchar)((uchar)std::round(bin) % (unsigned)bins);
                lut[2 * index + 1] = sat_u8(255 * magnitude);
            } else {
                uchar b0 = (uchar)((uchar)std::floor(bin) % (unsigned)bins);
                uchar b1 = (uchar)((uchar)std::ceil(bin) % (unsigned)bins);
                uchar w1 = sat_u8(255 * magnitude * (bin - std::floor(bin)));
                uchar w0 = sat_u8(255 * magnitude - w1);
                lut[4 * index] = b0;
                lut[4 * index + 1] = w0;
                lut[4 * index + 2] = b1;
                lut[4 * index + 3] = w1;
            }
        }
    }
}

// GradientBinningFilter.cpp:66-93
void gradient_binning(const uchar* grad2, int n, int bins, int signedGradients, int interpolate, uchar* dst) {
    const int e = interpolate ? 4 : 2;
    std::vector<uchar> lut((size_t)65536 * e);
    gradient_binning_lut(bins, signedGradients != 0, interpolate != 0, lut.data());
    for (int i = 0; i < n; ++i) {
        size_t index = (size_t)grad2[2 * i] | ((size_t)grad2[2 * i + 1] << 8);
        std::memcpy(dst + (size_t)e * i, lut.data() + e * index, e);
    }
}

// LbpFilter.cpp:20-44 (uniform map), :56-85, LbpFilter.hpp:88-180 (3x3 kernels, BORDER_REPLICATE)
void lbp(const uchar* src, int w, int h, int type, uchar* dst) {
    std::array<uchar, 256> map{};
    if (type == 1) {
        int nonUniformIndex = 0, emptyIndex = 1;
        for (unsigned i = 0; i < 256; ++i) {
            uchar code = (uchar)i;
            int transitions = 0, previousBit = (code >> 7) & 1;
            for (int pos = 0; pos < 8; ++pos) {
                int currentBit = (code >> pos) & 1;
                if (previousBit != currentBit) { transitions++; previousBit = currentBit; }
            }
            map[i] = (uchar)(transitions <= 2 ? emptyIndex++ : nonUniformIndex);
        }
    }
    auto at = [&](int y, int x) {
        y = y < 0 ? 0 : (y >= h ? h - 1 : y);
        x = x < 0 ? 0 : (x >= w ? w - 1 : x);
        return src[(size_t)y * w + x];
    };
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uchar c = at(y, x), code = 0;
            if (type == 0 || type == 1) {
                code |= (at(y - 1, x - 1) > c) << 7;
                code |= (at(y - 1, x) > c) << 6;
                code |= (at(y - 1, x + 1) > c) << 5;
                code |= (at(y, x + 1) > c) << 4;
                code |= (at(y + 1, x + 1) > c) << 3;
                code |= (at(y + 1, x) > c) << 2;
                code |= (at(y + 1, x - 1) > c) << 1;
                code |= (at(y, x - 1) > c) << 0;
                if (type == 1) code = map[code];
            } else if (type == 2) {
                code |= (at(y - 1, x) > c) << 3;
                code |= (at(y, x + 1) > c) << 2;
                code |= (at(y + 1, x) > c) << 1;
                code |= (at(y, x - 1) > c) << 0;
            } else {
                code |= (at(y - 1, x - 1) > c) << 3;
                code |= (at(y - 1, x + 1) > c) << 2;
                code |= (at(y + 1, x + 1) > c) << 1;
                code |= (at(y + 1, x - 1) > c) << 0;
            }
            dst[(size_t)y * w + x] = code;
        }
}

// GreyWorldNormalizationFilter.cpp:20-71 (continuous input)
static void greyworld(const uchar* bgr, int w, int h, uchar* dst) {
    const int n = w * h;
    double sum[3] = {0, 0, 0};
    uchar mx[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) {
            sum[c] += bgr[3 * i + c];
            if (bgr[3 * i + c] > mx[c]) mx[c] = bgr[3 * i + c];
        }
    double mean[3], maxNew[3];
    for (int c = 0; c < 3; ++c) { mean[c] = sum[c] / n; maxNew[c] = mx[c] / mean[c]; }
    double max = maxNew[0];
    if (maxNew[1] > max) max = maxNew[1];
    if (maxNew[2] > max) max = maxNew[2];
    double scale[3];
    for (int c = 0; c < 3; ++c) scale[c] = 255.0 / (mean[c] * max);
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) dst[3 * i + c] = sat_u8(cvRound(scale[c] * bgr[3 * i + c]));
}

// "whi" chain, ffpDetectApp.cpp:449-454: WhiteningFilter (WhiteningFilter.cpp:20-81) ->
// HistogramEqualizationFilter -> ConversionFilter(CV_32F, 1/127.5, -1) -> UnitNormFilter(L2).
// cv::dft is not available here and its butterfly order is not part of the reference's
// source, so the transforms are restated as plain separable DFTs in double with explicit complex
// arithmetic (row pass, then column pass; twiddles from std::polar).  Against OpenCV this agrees to
// float rounding of the whitened image (a rounding flip of a u8 pixel is possible where the value
// lies within ~1e-5 of .5); tests/test_oracle_golden.py pins it against numpy's FFT.
void whi_tables(int w, int h, float alpha, float cutoff, std::vector<double>& twRow, std::vector<double>& twCol,
                std::vector<float>& filt) {
    const double PI2 = 6.283185307179586476925286766559;
    twRow.resize((size_t)w * w * 2);
    twCol.resize((size_t)h * h * 2);
    for (int x = 0; x < w; ++x)
        for (int k = 0; k < w; ++k) {
            std::complex<double> t = std::polar(1.0, -PI2 * x * k / w);
            twRow[((size_t)x * w + k) * 2] = t.real();
            twRow[((size_t)x * w + k) * 2 + 1] = t.imag();
        }
    for (int y = 0; y < h; ++y)
        for (int k = 0; k < h; ++k) {
            std::complex<double> t = std::polar(1.0, -PI2 * y * k / h);
            twCol[((size_t)y * h + k) * 2] = t.real();
            twCol[((size_t)y * h + k) * 2 + 1] = t.imag();
        }
    filt.resize((size_t)w * h);   // WhiteningFilter.cpp:62-81, float arithmetic
    for (int row = 0; row < h; ++row)
        for (int col = 0; col < w; ++col) {
            int shiftedRow = (row + h / 2) % h, shiftedCol = (col + w / 2) % w;
            float fx = -0.5f + shiftedCol * (2 * 0.5f) / (w - 1);
            float fy = -0.5f + shiftedRow * (2 * 0.5f) / (h - 1);
            float rho = std::sqrt(fx * fx + fy * fy);
            float f = std::pow(rho, alpha);
            if (cutoff > 0) f *= std::exp(-std::pow(rho / cutoff, 4));
            filt[(size_t)row * w + col] = f;
        }
}

static void whi(const uchar* src, int w, int h, int stride, float alpha, float cutoff, float* dst, uchar* whitened = nullptr) {
    const int n = w * h;
    std::vector<double> twRow, twCol;
    std::vector<float> filt;
    whi_tables(w, h, alpha, cutoff, twRow, twCol, filt);
    std::vector<double> Tre((size_t)n), Tim((size_t)n), Fre((size_t)n), Fim((size_t)n);
    // forward DFT with DFT_SCALE: rows (real input), then columns
    for (int v = 0; v < h; ++v)
        for (int x = 0; x < w; ++x) {
            double sr = 0, si = 0;
            for (int k = 0; k < w; ++k) {
                const double p = (double)src[(size_t)v * stride + k];
                sr = sr + p * twRow[((size_t)x * w + k) * 2];
                si = si + p * twRow[((size_t)x * w + k) * 2 + 1];
            }
            Tre[(size_t)v * w + x] = sr;
            Tim[(size_t)v * w + x] = si;
        }
    for (int y = 0; y < h; ++y)
        for (int u = 0; u < w; ++u) {
            double sr = 0, si = 0;
            for (int k = 0; k < h; ++k) {
                const double ar = Tre[(size_t)k * w + u], ai = Tim[(size_t)k * w + u];
                const double br = twCol[((size_t)y * h + k) * 2], bi = twCol[((size_t)y * h + k) * 2 + 1];
                sr = sr + (ar * br - ai * bi);
                si = si + (ar * bi + ai * br);
            }
            // the spectrum is a CV_32FC2 Mat; the whitening filter multiplies it in float (WhiteningFilter.cpp:38-45)
            const float f = filt[(size_t)y * w + u];
            Fre[(size_t)y * w + u] = (double)((float)(sr / n) * f);
            Fim[(size_t)y * w + u] = (double)((float)(si / n) * f);
        }
    // inverse DFT, DFT_REAL_OUTPUT: only the half-spectrum is consumed, the rest is implied by conjugate
    // symmetry (OpenCV packs the complex input into CCS form).  Rows, then columns (real part only).
    std::vector<double> Ure((size_t)n), Uim((size_t)n);
    for (int v = 0; v < h; ++v)
        for (int x = 0; x < w; ++x) {
            double sr = 0, si = 0;
            for (int u = 0; u < w; ++u) {
                const int rr = (h - v) % h, cc = (w - u) % w;
                const bool own = u < cc || (u == cc && v <= rr);
                const double gr = own ? Fre[(size_t)v * w + u] : Fre[(size_t)rr * w + cc];
                const double gi = own ? Fim[(size_t)v * w + u] : -Fim[(size_t)rr * w + cc];
                const double br = twRow[((size_t)x * w + u) * 2], bi = -twRow[((size_t)x * w + u) * 2 + 1];   // conjugate twiddle
                sr = sr + (gr * br - gi * bi);
                si = si + (gr * bi + gi * br);
            }
            Ure[(size_t)v * w + x] = sr;
            Uim[(size_t)v * w + x] = si;
        }
    std::vector<uchar> u8((size_t)n), eq((size_t)n);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            double sr = 0;
            for (int v = 0; v < h; ++v) {
                const double br = twCol[((size_t)y * h + v) * 2], bi = -twCol[((size_t)y * h + v) * 2 + 1];
                sr = sr + (Ure[(size_t)v * w + x] * br - Uim[(size_t)v * w + x] * bi);
            }
            const float val = (float)sr;
            u8[(size_t)y * w + x] = sat_u8((double)(val * 1.0f + 127.0f));
        }
    if (whitened) {   // WhiteningFilter::applyTo alone (WhiteningFilter.cpp:20-58)
        std::copy(u8.begin(), u8.end(), whitened);
        return;
    }
    equalize_hist(u8.data(), w, h, w, eq.data());
    const float a = (float)(1.0 / 127.5), b = -1.0f;
    double norm2 = 0;
    for (int i = 0; i < n; ++i) {
        dst[i] = eq[i] * a + b;
        norm2 += (double)dst[i] * dst[i];
    }
    double norm = std::sqrt(norm2);
    const float eps = 1e-4f;
    float inv = (float)(1.0 / (norm + eps));
    for (int i = 0; i < n; ++i) dst[i] = dst[i] * inv;
}

// ---------------------------------------------------------------------------------------
// HistogramFilter::createCellHistograms, HistogramFilter.cpp:23-197 (+createCache :199-220)
// ---------------------------------------------------------------------------------------
struct CacheEntry { int index1, index2; float weight1, weight2; };
static void createCache(std::vector<CacheEntry>& cache, unsigned size, int count) {
    cache.clear();
    for (unsigned m = 0; m < size; ++m) {
        CacheEntry e;
        double realIndex = (double)count * ((double)m + 0.5) / (double)size - 0.5;
        e.index1 = (int)std::floor(realIndex);
        e.index2 = e.index1 + 1;
        e.weight2 = (float)(realIndex - e.index1);
        e.weight1 = 1.f - e.weight2;
        if (e.index1 < 0) { e.index1 = e.index2; e.weight1 = 0; }
        else if (e.index2 >= count) { e.index2 = e.index1; e.weight2 = 0; }
        cache.push_back(e);
    }
}

static void createCellHistograms(const uchar* img, int w, int h, int ch, int strideBytes, std::vector<float>& hist,
                                 int binCount, int rowCount, int columnCount, bool interpolate) {
    if (ch != 1 && ch != 2 && ch != 4) throw std::runtime_error("HistogramFilter: image must have one, two or four channels");
    hist.assign((size_t)rowCount * columnCount * binCount, 0.f);
    const float factor = 1.f / 255.f;
    auto H = [&](int r, int c) { return hist.data() + ((size_t)r * columnCount + c) * binCount; };
    if (interpolate) {
        std::vector<CacheEntry> rowCache, colCache;
        createCache(rowCache, h, rowCount);
        createCache(colCache, w, columnCount);
        for (int y = 0; y < h; ++y) {
            const uchar* row = img + (size_t)y * strideBytes;
            int r0 = rowCache[y].index1, r1 = rowCache[y].index2;
            float rw1 = rowCache[y].weight2, rw0 = rowCache[y].weight1;
            for (int x = 0; x < w; ++x) {
                int c0 = colCache[x].index1, c1 = colCache[x].index2;
                float cw1 = colCache[x].weight2, cw0 = colCache[x].weight1;
                const uchar* px = row + (size_t)x * ch;
                int nb = ch == 4 ? 2 : 1;
                uchar b[2];
                float wt[2];
                b[0] = px[0];
                wt[0] = ch == 1 ? 1.f : factor * px[1];
                if (ch == 4) { b[1] = px[2]; wt[1] = factor * px[3]; }
                auto add = [&](int r, int c, float wr, float wc) {
                    float* hv = H(r, c);
                    for (int k = 0; k < nb; ++k) {
                        if (ch == 1) hv[b[k]] += wr * wc;
                        else hv[b[k]] += wt[k] * wr * wc;
                    }
                };
                if (r0 >= 0 && c0 >= 0) add(r0, c0, rw0, cw0);
                if (r0 >= 0 && c1 < columnCount) add(r0, c1, rw0, cw1);
                if (r1 < rowCount && c0 >= 0) add(r1, c0, rw1, cw0);
                if (r1 < rowCount && c1 < columnCount) add(r1, c1, rw1, cw1);
            }
        }
    } else {
        float* hv = hist.data();
        for (int cr = 0; cr < rowCount; ++cr)
            for (int cc = 0; cc < columnCount; ++cc) {
                int startRow = (cr * h) / rowCount, startCol = (cc * w) / columnCount;
                int endRow = ((cr + 1) * h) / rowCount, endCol = ((cc + 1) * w) / columnCount;
                for (int y = startRow; y < endRow; ++y) {
                    const uchar* row = img + (size_t)y * strideBytes;
                    for (int x = startCol; x < endCol; ++x) {
                        const uchar* px = row + (size_t)x * ch;
                        if (ch == 1) hv[px[0]]++;
                        else if (ch == 2) hv[px[0]] += factor * px[1];
                        else { hv[px[0]] += factor * px[1]; hv[px[2]] += factor * px[3]; }
                    }
                }
                hv += binCount;
            }
    }
}

// HogFilter.cpp:58-122
int hog_filter(const uchar* img, int w, int h, int ch, int strideBytes, int binCount, int cellW, int cellH,
               int blockW, int blockH, bool interpolate, bool signedAndUnsigned, std::vector<float>& out) {
    const float eps = 1e-4f;
    int cellRowCount = cvRound((double)h / (double)cellH);
    int cellColumnCount = cvRound((double)w / (double)cellW);
    std::vector<float> cells;
    createCellHistograms(img, w, h, ch, strideBytes, cells, binCount, cellRowCount, cellColumnCount, interpolate);
    int binHalfCount = binCount / 2;
    std::vector<float> energies((size_t)cellRowCount * cellColumnCount);
    for (int ci = 0; ci < cellRowCount * cellColumnCount; ++ci) {
        const float* hv = cells.data() + (size_t)ci * binCount;
        float energy = 0;
        if (signedAndUnsigned) {
            for (int b = 0; b < binHalfCount; ++b) {
                float u = hv[b] + hv[binHalfCount + b];
                energy += u * u;
            }
        } else {
            for (int b = 0; b < binCount; ++b) energy += hv[b] * hv[b];
        }
        energies[ci] = energy;
    }
    int blockHistogramSize = signedAndUnsigned ? blockW * blockH * (binCount + binHalfCount) : blockW * blockH * binCount;
    int blockRowCount = cellRowCount - blockH + 1, blockColumnCount = cellColumnCount - blockW + 1;
    if (blockRowCount < 0) blockRowCount = 0;
    if (blockColumnCount < 0) blockColumnCount = 0;
    out.assign((size_t)blockRowCount * blockColumnCount * blockHistogramSize, 0.f);
    float* o = out.data();
    for (int br = 0; br < blockRowCount; ++br)
        for (int bc = 0; bc < blockColumnCount; ++bc) {
            float energy = 0;
            for (int cr = br; cr < br + blockH; ++cr)
                for (int cc = bc; cc < bc + blockW; ++cc) energy += energies[(size_t)cr * cellColumnCount + cc];
            float normalizer = 1.f / std::sqrt(energy + eps);
            for (int cr = br; cr < br + blockH; ++cr)
                for (int cc = bc; cc < bc + blockW; ++cc) {
                    const float* hv = cells.data() + ((size_t)cr * cellColumnCount + cc) * binCount;
                    for (int b = 0; b < binCount; ++b) o[b] = normalizer * hv[b];
                    o += binCount;
                    if (signedAndUnsigned) {
                        for (int b = 0; b < binHalfCount; ++b) o[b] = normalizer * (hv[b] + hv[binHalfCount + b]);
                        o += binHalfCount;
                    }
                }
        }
    return (int)out.size();
}

// HistogramFilter.cpp:222-251
static void normalizeL2(float* v, int n) {
    const float eps = 1e-4f;
    double s = 0;
    for (int i = 0; i < n; ++i) s += (double)v[i] * v[i];
    float norm = (float)std::sqrt(s);
    float inv = (float)(1.0 / (double)(norm + eps));
    for (int i = 0; i < n; ++i) v[i] = v[i] * inv;
}
static void normalizeL1(float* v, int n) {
    const float eps = 1e-4f;
    double s = 0;
    for (int i = 0; i < n; ++i) s += std::fabs((double)v[i]);
    float norm = (float)s;
    float inv = (float)(1.0 / (double)(norm + eps));
    for (int i = 0; i < n; ++i) v[i] = v[i] * inv;
}
static void normalizeHist(float* v, int n, int normalization) {
    switch (normalization) {
        case 1: normalizeL2(v, n); break;
        case 2:
            normalizeL2(v, n);
            for (int i = 0; i < n; ++i) v[i] = std::min(v[i], (float)0.2);
            normalizeL2(v, n);
            break;
        case 3: normalizeL1(v, n); break;
        case 4:
            normalizeL1(v, n);
            for (int i = 0; i < n; ++i) v[i] = std::sqrt(v[i]);
            break;
        default: break;
    }
}

// SpatialHistogramFilter.cpp:56-94
static int spatial_histogram(const uchar* img, int w, int h, int ch, int strideBytes, int binCount, int cellW, int cellH,
                             int blockW, int blockH, bool interpolate, bool concatenate, int normalization,
                             std::vector<float>& out) {
    int cellRowCount = cvRound((double)h / (double)cellH);
    int cellColumnCount = cvRound((double)w / (double)cellW);
    if (blockW == 1 && blockH == 1) {
        createCellHistograms(img, w, h, ch, strideBytes, out, binCount, cellRowCount, cellColumnCount, interpolate);
        normalizeHist(out.data(), (int)out.size(), normalization);
        return (int)out.size();
    }
    std::vector<float> cells;
    createCellHistograms(img, w, h, ch, strideBytes, cells, binCount, cellRowCount, cellColumnCount, interpolate);
    int blockHistogramSize = concatenate ? blockW * blockH * binCount : binCount;
    int blockRowCount = cellRowCount - blockH + 1, blockColumnCount = cellColumnCount - blockW + 1;
    out.assign((size_t)blockRowCount * blockColumnCount * blockHistogramSize, 0.f);
    float* o = out.data();
    for (int br = 0; br < blockRowCount; ++br)
        for (int bc = 0; bc < blockColumnCount; ++bc) {
            float* blockStart = o;
            for (int cr = br; cr < br + blockH; ++cr)
                for (int cc = bc; cc < bc + blockW; ++cc) {
                    const float* hv = cells.data() + ((size_t)cr * cellColumnCount + cc) * binCount;
                    for (int b = 0; b < binCount; ++b) o[b] += hv[b];
                    if (concatenate) o += binCount;
                }
            if (!concatenate) o += binCount;
            normalizeHist(blockStart, blockHistogramSize, normalization);
        }
    return (int)out.size();
}

// PyramidHogFilter.cpp:33-113 / SpatialPyramidHistogramFilter.cpp:37-81: cell histograms on the finest
// 2^L x 2^L grid, every coarser level = sum of its 2x2 children (row-major child order, accumulated
// into zeros), all levels concatenated coarse -> fine, each histogram normalised on its own.
static void combine_levels(float* base, int histogramCount, int maxLevel, int binCount) {
    float* cellValues = base + (size_t)(histogramCount - (1 << (2 * maxLevel))) * binCount;
    for (int level = maxLevel - 1; level >= 0; --level) {
        float* blockValues = cellValues - (size_t)(1 << (2 * level)) * binCount;
        int blockCount = 1 << level, cellCount = blockCount << 1;
        float* bv = blockValues;
        for (int br = 0; br < blockCount; ++br)
            for (int bc = 0; bc < blockCount; ++bc) {
                for (int cr = 2 * br; cr < 2 * (br + 1); ++cr)
                    for (int cc = 2 * bc; cc < 2 * (bc + 1); ++cc) {
                        const float* cv = cellValues + (size_t)(cr * cellCount + cc) * binCount;
                        for (int b = 0; b < binCount; ++b) bv[b] += cv[b];
                    }
                bv += binCount;
            }
        cellValues = blockValues;
    }
}

static int pyramid_hog(const uchar* img, int w, int h, int ch, int strideBytes, int binCount, int levelCount, bool interpolate,
                       bool signedAndUnsigned, std::vector<float>& out) {
    if (binCount <= 0) throw std::invalid_argument("PyramidHogFilter: binCount must be greater than zero");
    if (levelCount <= 0) throw std::invalid_argument("PyramidHogFilter: levelCount must be greater than zero");
    if (signedAndUnsigned && binCount % 2 != 0)
        throw std::invalid_argument("PyramidHogFilter: the bin size must be even for signed and unsigned gradients to be combined");
    const float eps = 1e-4f;
    int maxLevel = levelCount - 1, histogramCount = 0;
    for (int level = 0; level < levelCount; ++level) histogramCount += 1 << (2 * level);
    int binHalfCount = binCount / 2;
    int realBinCount = signedAndUnsigned ? 3 * binCount / 2 : binCount;
    out.assign((size_t)histogramCount * realBinCount, 0.f);
    int gridCount = 1 << maxLevel;
    std::vector<float> cells;
    createCellHistograms(img, w, h, ch, strideBytes, cells, binCount, gridCount, gridCount, interpolate);
    {   // copyCellHistograms :57-74
        const float* sv = cells.data();
        float* dv = out.data() + (size_t)(histogramCount - gridCount * gridCount) * realBinCount;
        for (int i = 0; i < gridCount * gridCount; ++i) {
            for (int b = 0; b < binCount; ++b) dv[b] = sv[b];
            dv += binCount;
            if (signedAndUnsigned) {
                for (int b = 0; b < binHalfCount; ++b) dv[b] = sv[b] + sv[binHalfCount + b];
                dv += binHalfCount;
            }
            sv += binCount;
        }
    }
    combine_levels(out.data(), histogramCount, maxLevel, realBinCount);
    float* hv = out.data();   // normalizeHistograms :94-113
    for (int i = 0; i < histogramCount; ++i) {
        float energy = 0;
        if (signedAndUnsigned) {
            for (int b = binCount; b < realBinCount; ++b) energy += hv[b] * hv[b];
        } else {
            for (int b = 0; b < binCount; ++b) energy += hv[b] * hv[b];
        }
        float normalizer = 1.f / std::sqrt(energy + eps);
        for (int b = 0; b < realBinCount; ++b) hv[b] = normalizer * hv[b];
        hv += realBinCount;
    }
    return (int)out.size();
}

static int spatial_pyramid_histogram(const uchar* img, int w, int h, int ch, int strideBytes, int binCount, int levelCount,
                                     bool interpolate, int normalization, std::vector<float>& out) {
    if (binCount <= 0) throw std::invalid_argument("SpatialPyramidHistogramFilter: binCount must be greater than zero");
    if (levelCount <= 0) throw std::invalid_argument("SpatialPyramidHistogramFilter: levelCount must be greater than zero");
    int maxLevel = levelCount - 1, histogramCount = 0;
    for (int level = 0; level < levelCount; ++level) histogramCount += 1 << (2 * level);
    out.assign((size_t)histogramCount * binCount, 0.f);
    int gridCount = 1 << maxLevel;
    std::vector<float> cells;
    createCellHistograms(img, w, h, ch, strideBytes, cells, binCount, gridCount, gridCount, interpolate);
    std::copy(cells.begin(), cells.end(), out.begin() + (size_t)(histogramCount - gridCount * gridCount) * binCount);
    combine_levels(out.data(), histogramCount, maxLevel, binCount);
    for (int i = 0; i < histogramCount; ++i) normalizeHist(out.data() + (size_t)i * binCount, binCount, normalization);
    return (int)out.size();
}

// ---------------------------------------------------------------------------------------
// imageprocessing::filtering::FhogFilter (FhogFilter.cpp:20-132, FhogFilter.hpp:120-207) on a CV_8UC1 image +
// FhogAggregationFilter::computeDescriptors (FhogAggregationFilter.cpp:38-168).  SURVEY.md 8(f) row 2.
// Output: rows x cols cells (image size / cellSize, integer division) x (3 * unsignedBinCount + 4) floats.
// ---------------------------------------------------------------------------------------
struct FhogCoeff { int index1, index2; float weight1, weight2; };

static std::vector<FhogCoeff> fhog_interp_coefficients(int sizeInPixels, int sizeInCells, int cellSize, bool interpolateCells) {
    std::vector<FhogCoeff> c((size_t)sizeInPixels);
    if (interpolateCells) {
        for (int pixel = 0; pixel < sizeInPixels; ++pixel) {   // FhogFilter.cpp:76-91
            float realCellIndex = (pixel + 0.5f) / cellSize - 0.5f;
            int index1 = (int)std::floor(realCellIndex);
            int index2 = index1 + 1;
            float weight2 = realCellIndex - index1;
            float weight1 = index2 - realCellIndex;
            if (index1 < 0) { index1 = index2; weight1 = 0; }
            else if (index2 >= sizeInCells) { index2 = index1; weight2 = 0; }
            c[pixel] = FhogCoeff{index1, index2, weight1, weight2};
        }
    } else {
        for (int pixel = 0; pixel < sizeInPixels; ++pixel) c[pixel] = FhogCoeff{pixel / cellSize, -1, 1, 0};
    }
    return c;
}

// one entry of the 512 x 512 gradient look-up table (FhogFilter.cpp:35-57): dx, dy in [1, 511] are gradient + 256
static FhogCoeff fhog_lut_entry(int gradientCodeX, int gradientCodeY, int signedBinCount, bool interpolateBins, float& magnitude) {
    const float TWO_PI = (float)(2 * M_PI);                    // GradientOrientationFilter.hpp:57
    const float value2bin = signedBinCount / TWO_PI;           // FhogFilter.cpp:27
    float gradientX = (gradientCodeX - 256) / (255.0f * 2.0f);
    float gradientY = (gradientCodeY - 256) / (255.0f * 2.0f);
    magnitude = std::sqrt(gradientX * gradientX + gradientY * gradientY);   // GradientMagnitudeFilter.cpp:80-86
    float orientation = std::atan2(gradientY, gradientX);      // GradientOrientationFilter.cpp:137-144 (full range)
    if (orientation < 0) orientation += TWO_PI;
    FhogCoeff bins;
    if (interpolateBins) {                                     // computeInterpolatedBins :111-121
        const float bin = orientation * value2bin;
        bins.index1 = (int)bin;
        bins.index2 = bins.index1 + 1;
        if (bins.index2 == signedBinCount) bins.index2 = 0;
        bins.weight2 = magnitude * (bin - bins.index1);
        bins.weight1 = magnitude - bins.weight2;
    } else {                                                   // computeBin :103-108
        int bin = (int)(orientation * value2bin + 0.5f);
        if (bin == signedBinCount) bin = 0;
        bins.index1 = bin; bins.index2 = 0; bins.weight1 = magnitude; bins.weight2 = 0;
    }
    return bins;
}

int fhog_filter(const uchar* img, int w, int h, int stride, int cellSize, int unsignedBinCount, bool interpolateBins, bool interpolateCells,
                float alpha, std::vector<float>& out, int& rowsOut, int& colsOut, int channels) {
    if (channels != 1 && channels != 3) throw std::invalid_argument("FhogFilter: the image type must be CV_8UC1 or CV_8UC3");
    if (unsignedBinCount < 1) throw std::invalid_argument("FhogFilter: unsignedBinCount must be bigger than zero");
    if (alpha <= 0) throw std::invalid_argument("FhogAggregationFilter: alpha must be bigger than zero");
    const int signedBinCount = 2 * unsignedBinCount;
    const int rows = h / cellSize, cols = w / cellSize;
    const int D = signedBinCount + unsignedBinCount + 4;
    rowsOut = rows; colsOut = cols;
    out.assign((size_t)rows * cols * D, 0.f);
    if (rows == 0 || cols == 0) return 0;
    // computeSignedHistograms<true> (FhogFilter.hpp:120-131): pixels of the cell-covered region in row-major order
    std::vector<FhogCoeff> rowCoeff = fhog_interp_coefficients(rows * cellSize, rows, cellSize, interpolateCells);
    std::vector<FhogCoeff> colCoeff = fhog_interp_coefficients(cols * cellSize, cols, cellSize, interpolateCells);
    auto H = [&](int r, int c) { return out.data() + ((size_t)r * cols + c) * D; };
    for (int imageRow = 0; imageRow < (int)rowCoeff.size(); ++imageRow)
        for (int imageCol = 0; imageCol < (int)colCoeff.size(); ++imageCol) {
            int prevRow = std::max(imageRow - 1, 0), nextRow = std::min(imageRow + 1, h - 1);
            int prevCol = std::max(imageCol - 1, 0), nextCol = std::min(imageCol + 1, w - 1);
            FhogCoeff b;
            if (channels == 1) {   // getBinCoefficients<true> :133-142
                int dx = img[(size_t)imageRow * stride + nextCol] - img[(size_t)imageRow * stride + prevCol] + 256;
                int dy = img[(size_t)nextRow * stride + imageCol] - img[(size_t)prevRow * stride + imageCol] + 256;
                float magnitude;
                b = fhog_lut_entry(dx, dy, signedBinCount, interpolateBins, magnitude);
            } else {               // getBinCoefficients<false> :144-172: the channel with the largest gradient magnitude
                const uchar* up = img + (size_t)prevRow * stride + 3 * imageCol;
                const uchar* down = img + (size_t)nextRow * stride + 3 * imageCol;
                const uchar* left = img + (size_t)imageRow * stride + 3 * prevCol;
                const uchar* right = img + (size_t)imageRow * stride + 3 * nextCol;
                float m1, m2, m3;
                const FhogCoeff b1 = fhog_lut_entry(right[0] - left[0] + 256, down[0] - up[0] + 256, signedBinCount, interpolateBins, m1);
                const FhogCoeff b2 = fhog_lut_entry(right[1] - left[1] + 256, down[1] - up[1] + 256, signedBinCount, interpolateBins, m2);
                const FhogCoeff b3 = fhog_lut_entry(right[2] - left[2] + 256, down[2] - up[2] + 256, signedBinCount, interpolateBins, m3);
                if (m1 > m2) b = m1 > m3 ? b1 : b3;
                else b = m2 > m3 ? b2 : b3;
            }
            const FhogCoeff& rc = rowCoeff[imageRow];
            const FhogCoeff& cc = colCoeff[imageCol];
            if (interpolateCells) {   // addToSignedHistograms :173-205
                float* h11 = H(rc.index1, cc.index1); float* h12 = H(rc.index1, cc.index2);
                float* h21 = H(rc.index2, cc.index1); float* h22 = H(rc.index2, cc.index2);
                if (interpolateBins) {
                    h11[b.index1] += b.weight1 * rc.weight1 * cc.weight1;
                    h11[b.index2] += b.weight2 * rc.weight1 * cc.weight1;
                    h12[b.index1] += b.weight1 * rc.weight1 * cc.weight2;
                    h12[b.index2] += b.weight2 * rc.weight1 * cc.weight2;
                    h21[b.index1] += b.weight1 * rc.weight2 * cc.weight1;
                    h21[b.index2] += b.weight2 * rc.weight2 * cc.weight1;
                    h22[b.index1] += b.weight1 * rc.weight2 * cc.weight2;
                    h22[b.index2] += b.weight2 * rc.weight2 * cc.weight2;
                } else {
                    h11[b.index1] += b.weight1 * rc.weight1 * cc.weight1;
                    h12[b.index1] += b.weight1 * rc.weight1 * cc.weight2;
                    h21[b.index1] += b.weight1 * rc.weight2 * cc.weight1;
                    h22[b.index1] += b.weight1 * rc.weight2 * cc.weight2;
                }
            } else {
                float* hh = H(rc.index1, cc.index1);
                if (interpolateBins) { hh[b.index1] += b.weight1; hh[b.index2] += b.weight2; }
                else hh[b.index1] += b.weight1;
            }
        }
    // FhogAggregationFilter::computeDescriptors, in place on the same buffer (FhogFilter.cpp:70)
    const float eps = 1e-4f;
    std::vector<float> energies((size_t)rows * cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {   // computeGradientEnergy :53-61
            const float* sh = H(r, c);
            float energy = 0;
            for (int bin = 0; bin < unsignedBinCount; ++bin) {
                float u = sh[bin] + sh[bin + unsignedBinCount];
                energy += u * u;
            }
            energies[(size_t)r * cols + c] = energy;
        }
    auto E = [&](int r, int c) { return energies[(size_t)r * cols + c]; };
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            int pr = std::max(r - 1, 0), nr = std::min(r + 1, rows - 1), pc = std::max(c - 1, 0), nc = std::min(c + 1, cols - 1);
            float n[4] = {   // computeNormalizers :77-99
                1.f / std::sqrt(E(pr, pc) + E(pr, c) + E(r, pc) + E(r, c) + eps),
                1.f / std::sqrt(E(pr, c) + E(pr, nc) + E(r, c) + E(r, nc) + eps),
                1.f / std::sqrt(E(r, pc) + E(r, c) + E(nr, pc) + E(nr, c) + eps),
                1.f / std::sqrt(E(r, c) + E(r, nc) + E(nr, c) + E(nr, nc) + eps)};
            float* d = H(r, c);
            float energy[4] = {0, 0, 0, 0};
            auto normalized = [&](float value, float* v4) { for (int i = 0; i < 4; ++i) v4[i] = std::min(alpha, n[i] * value); };
            for (int bin = 0; bin < unsignedBinCount; ++bin) {   // computeDescriptor :101-148
                float v4[4];
                normalized(d[bin] + d[bin + unsignedBinCount], v4);
                d[signedBinCount + bin] = 0.5 * (v4[0] + v4[1] + v4[2] + v4[3]);
            }
            for (int bin = 0; bin < signedBinCount; ++bin) {
                float v4[4];
                normalized(d[bin], v4);
                d[bin] = 0.5 * (v4[0] + v4[1] + v4[2] + v4[3]);
                for (int i = 0; i < 4; ++i) energy[i] += v4[i];
            }
            for (int i = 0; i < 4; ++i) d[signedBinCount + unsignedBinCount + i] = 0.2357 * energy[i];
        }
    return (int)out.size();
}


}  // namespace orc

using namespace orc;
extern "C" {
void orc_histeq64(const uint8_t* s, int w, int h, int stride, uint8_t* d) { histeq64(s, w, h, stride, d); }
void orc_gradient_binning_lut(int bins, int sg, int interp, uint8_t* lut) { gradient_binning_lut(bins, sg != 0, interp != 0, lut); }
void orc_gradient_binning(const uint8_t* g, int n, int bins, int sg, int interp, uint8_t* d) { gradient_binning(g, n, bins, sg, interp, d); }
void orc_lbp(const uint8_t* s, int w, int h, int type, uint8_t* d) { lbp(s, w, h, type, d); }
void orc_greyworld(const uint8_t* s, int w, int h, uint8_t* d) { greyworld(s, w, h, d); }
void orc_whi(const uint8_t* s, int w, int h, int stride, float alpha, float cutoff, float* d) { whi(s, w, h, stride, alpha, cutoff, d); }
void orc_whitening(const uint8_t* s, int w, int h, int stride, float alpha, float cutoff, uint8_t* d) { whi(s, w, h, stride, alpha, cutoff, nullptr, d); }
int orc_hog_filter(const uint8_t* img, int w, int h, int ch, int stride, int bins, int cw, int chh, int bw, int bh,
                   int interp, int sau, float* out) {
    std::vector<float> v;
    int n = hog_filter(img, w, h, ch, stride, bins, cw, chh, bw, bh, interp != 0, sau != 0, v);
    if (out) std::memcpy(out, v.data(), sizeof(float) * v.size());
    return n;
}
int orc_spatial_histogram(const uint8_t* img, int w, int h, int ch, int stride, int bins, int cw, int chh, int bw, int bh,
                          int interp, int concat, int normalization, float* out) {
    std::vector<float> v;
    int n = spatial_histogram(img, w, h, ch, stride, bins, cw, chh, bw, bh, interp != 0, concat != 0, normalization, v);
    if (out) std::memcpy(out, v.data(), sizeof(float) * v.size());
    return n;
}
int orc_pyramid_hog(const uint8_t* img, int w, int h, int ch, int stride, int bins, int levels, int interpolate, int sau, float* out) {
    std::vector<float> v;
    int n = pyramid_hog(img, w, h, ch, stride, bins, levels, interpolate != 0, sau != 0, v);
    if (out) std::memcpy(out, v.data(), sizeof(float) * v.size());
    return n;
}
int orc_spatial_pyramid_histogram(const uint8_t* img, int w, int h, int ch, int stride, int bins, int levels, int interpolate,
                                  int normalization, float* out) {
    std::vector<float> v;
    int n = spatial_pyramid_histogram(img, w, h, ch, stride, bins, levels, interpolate != 0, normalization, v);
    if (out) std::memcpy(out, v.data(), sizeof(float) * v.size());
    return n;
}
int orc_fhog(const uint8_t* img, int w, int h, int stride, int cellSize, int unsignedBinCount, int interpolateBins, int interpolateCells,
             float alpha, float* out, int* rows, int* cols) {
    return orc_fhog_channels(img, w, h, 1, stride, cellSize, unsignedBinCount, interpolateBins, interpolateCells, alpha, out, rows, cols);
}
int orc_fhog_channels(const uint8_t* img, int w, int h, int channels, int stride, int cellSize, int unsignedBinCount, int interpolateBins,
                      int interpolateCells, float alpha, float* out, int* rows, int* cols) {
    std::vector<float> v;
    int r, c;
    int n = fhog_filter(img, w, h, stride, cellSize, unsignedBinCount, interpolateBins != 0, interpolateCells != 0, alpha, v, r, c, channels);
    if (rows) *rows = r;
    if (cols) *cols = c;
    if (out) std::memcpy(out, v.data(), sizeof(float) * v.size());
    return n;
}
}
