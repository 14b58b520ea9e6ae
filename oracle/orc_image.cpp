// oracle/orc_image.cpp -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
// OpenCV 2.4 image primitives the hot path calls, the image pyramid and the window enumeration.
#include "orc_common.h"
#include "orc_internal.h"
#include "oracle.h"
#include <cstring>

namespace orc {

// cv::cvtColor(CV_BGR2GRAY) for 8U: fixed point, yuv_shift = 14, B2Y=1868 G2Y=9617 R2Y=4899
// (call site: GrayscaleFilter.cpp:20)
void bgr2gray(const uchar* bgr, int w, int h, uchar* gray) {
    for (size_t i = 0, n = (size_t)w * h; i < n; ++i) {
        int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
        gray[i] = (uchar)((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14);
    }
}

// cv::resize(..., INTER_LINEAR) for 8UC1 (call site ImagePyramid.cpp:177): 11-bit fixed-point
// coefficients, horizontal pass into int32, vertical pass with the (>>4, >>16, +2, >>2) cast.
void resize_linear_u8(const uchar* src, int sw, int sh, uchar* dst, int dw, int dh) {
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(2 * (size_t)dw), ibeta(2 * (size_t)dh);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = (short)cvRound((1.f - fx) * 2048);
        ialpha[2 * dx + 1] = (short)cvRound(fx * 2048);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[2 * dy] = (short)cvRound((1.f - fy) * 2048);
        ibeta[2 * dy + 1] = (short)cvRound(fy * 2048);
    }
    std::vector<int> r0(dw), r1(dw);
    auto hrow = [&](int sy, std::vector<int>& out) {
        const uchar* S = src + (size_t)sy * sw;
        for (int dx = 0; dx < dw; ++dx) {
            int sx = xofs[dx];
            int s1 = sx + 1 < sw ? S[sx + 1] : S[sx];  // weight is 0 there
            out[dx] = S[sx] * ialpha[2 * dx] + s1 * ialpha[2 * dx + 1];
        }
    };
    auto clip = [&](int y) { return y < 0 ? 0 : (y >= sh ? sh - 1 : y); };
    for (int dy = 0; dy < dh; ++dy) {
        hrow(clip(yofs[dy]), r0);
        hrow(clip(yofs[dy] + 1), r1);
        int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        uchar* D = dst + (size_t)dy * dw;
        for (int x = 0; x < dw; ++x)
            D[x] = (uchar)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
    }
}

// cv::resize(..., INTER_LINEAR) for 32FC1 (call site DescriptorExtractor.hpp:183): same
// coordinates, float coefficients, horizontal then vertical pass in float.
void resize_linear_f32(const float* src, int sw, int sh, float* dst, int dw, int dh) {
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<float> alpha(2 * (size_t)dw), beta(2 * (size_t)dh);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        alpha[2 * dx] = 1.f - fx;
        alpha[2 * dx + 1] = fx;
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloor(fy);
        fy -= sy;
        yofs[dy] = sy;
        beta[2 * dy] = 1.f - fy;
        beta[2 * dy + 1] = fy;
    }
    std::vector<float> r0(dw), r1(dw);
    auto hrow = [&](int sy, std::vector<float>& out) {
        const float* S = src + (size_t)sy * sw;
        for (int dx = 0; dx < dw; ++dx) {
            int sx = xofs[dx];
            float s1 = sx + 1 < sw ? S[sx + 1] : S[sx];
            out[dx] = S[sx] * alpha[2 * dx] + s1 * alpha[2 * dx + 1];
        }
    };
    auto clip = [&](int y) { return y < 0 ? 0 : (y >= sh ? sh - 1 : y); };
    for (int dy = 0; dy < dh; ++dy) {
        hrow(clip(yofs[dy]), r0);
        hrow(clip(yofs[dy] + 1), r1);
        float b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        float* D = dst + (size_t)dy * dw;
        for (int x = 0; x < dw; ++x) D[x] = r0[x] * b0 + r1[x] * b1;
    }
}

// cv::pyrDown for 8UC1 (call site ImagePyramid.cpp:186): [1 4 6 4 1]x[1 4 6 4 1], (sum+128)>>8,
// BORDER_REFLECT_101, dst = ((w+1)/2, (h+1)/2)
void pyrdown_u8(const uchar* src, int sw, int sh, uchar* dst) {
    int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    std::vector<int> rows(5 * (size_t)dw);
    for (int y = 0; y < dh; ++y) {
        for (int k = 0; k < 5; ++k) {
            int sy = reflect101(2 * y - 2 + k, sh);
            const uchar* S = src + (size_t)sy * sw;
            int* R = rows.data() + (size_t)k * dw;
            for (int x = 0; x < dw; ++x) {
                int x0 = reflect101(2 * x - 2, sw), x1 = reflect101(2 * x - 1, sw), x2 = 2 * x;
                int x3 = reflect101(2 * x + 1, sw), x4 = reflect101(2 * x + 2, sw);
                R[x] = S[x2] * 6 + (S[x1] + S[x3]) * 4 + S[x0] + S[x4];
            }
        }
        uchar* D = dst + (size_t)y * dw;
        for (int x = 0; x < dw; ++x) {
            int v = rows[2 * dw + x] * 6 + (rows[dw + x] + rows[3 * dw + x]) * 4 + rows[x] + rows[4 * dw + x];
            D[x] = (uchar)((v + 128) >> 8);
        }
    }
}

// cv::blur(img, dst, Size(k,k)) for 8UC1: normalised box filter, anchor k/2, BORDER_REFLECT_101,
// saturate_cast<uchar>(sum * (1.0/(k*k))).   (call site GradientFilter.cpp:45)
static void box_blur_u8(const uchar* src, int w, int h, int k, uchar* dst) {
    const int a = k / 2;
    const double scale = 1.0 / ((double)k * k);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int s = 0;
            for (int j = 0; j < k; ++j) {
                const uchar* S = src + (size_t)reflect101(y - a + j, h) * w;
                for (int i = 0; i < k; ++i) s += S[reflect101(x - a + i, w)];
            }
            dst[(size_t)y * w + x] = sat_u8(s * scale);
        }
}

// GradientFilter.cpp:16-59: optional blur, cv::Sobel(dx) and cv::Sobel(dy) with ddepth = 8U, delta = 127, merged into 2 channels
// (x first).  cv::Sobel = sepFilter2D with the kernels of getDerivKernels (OpenCV imgproc/deriv.cpp, getSobelKernels /
// getScharrKernels): derivative taps d and smoothing taps s,
//   ksize 1: d = [-1 0 1], s = [1]               scale 1/2      (GradientFilter::getScale: 1 / 2^(2 ksize - 3); 1/2 for ksize 1)
//   ksize 3: d = [-1 0 1], s = [1 2 1]           scale 1/8
//   ksize 5: d = [-1 -2 0 2 1], s = [1 4 6 4 1]  scale 1/128
//   ksize 7: d = [-1 -4 -5 0 5 4 1], s = [1 6 15 20 15 6 1]   scale 1/2048
//   CV_SCHARR (-1): d = [-1 0 1], s = [3 10 3]   scale 1/32
// (the scale is folded into one of the float kernels, cv::Sobel).  BORDER_REFLECT_101.  Every scale is a power of two and every
// partial sum fits 24 bits, so sepFilter2D's float arithmetic is exact whatever its order: out = saturate(cvRound(127 + scale * D))
// with the integer 2-D derivative D.
void gradient_filter(const uchar* src, int w, int h, int ksize, int blur, uchar* dst2) {
    static const int D1[3] = {-1, 0, 1}, S1[1] = {1}, S3[3] = {1, 2, 1}, D5[5] = {-1, -2, 0, 2, 1}, S5[5] = {1, 4, 6, 4, 1},
                     D7[7] = {-1, -4, -5, 0, 5, 4, 1}, S7[7] = {1, 6, 15, 20, 15, 6, 1}, SS[3] = {3, 10, 3};
    const int *dk, *sk;
    int nd, ns;
    double scale;
    switch (ksize) {
        case 1: dk = D1; nd = 3; sk = S1; ns = 1; scale = 1.0 / 2; break;
        case 3: dk = D1; nd = 3; sk = S3; ns = 3; scale = 1.0 / 8; break;
        case 5: dk = D5; nd = 5; sk = S5; ns = 5; scale = 1.0 / 128; break;
        case 7: dk = D7; nd = 7; sk = S7; ns = 7; scale = 1.0 / 2048; break;
        case -1: dk = D1; nd = 3; sk = SS; ns = 3; scale = 1.0 / 32; break;   // CV_SCHARR
        default: throw std::invalid_argument("GradientFilter: the kernel size must be 1, 3, 5, 7 or CV_SCHARR");
    }
    std::vector<uchar> tmp;
    if (blur > 0) {
        tmp.resize((size_t)w * h);
        box_blur_u8(src, w, h, blur, tmp.data());
        src = tmp.data();
    }
    const int ad = nd / 2, as = ns / 2;
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            int gx = 0, gy = 0;
            for (int j = 0; j < ns; ++j) {       // gx: derivative along x, smoothing along y
                const uchar* S = src + (size_t)reflect101(y + j - as, h) * w;
                int r = 0;
                for (int i = 0; i < nd; ++i) r += dk[i] * S[reflect101(x + i - ad, w)];
                gx += sk[j] * r;
            }
            for (int j = 0; j < nd; ++j) {       // gy: derivative along y, smoothing along x
                const uchar* S = src + (size_t)reflect101(y + j - ad, h) * w;
                int r = 0;
                for (int i = 0; i < ns; ++i) r += sk[i] * S[reflect101(x + i - as, w)];
                gy += dk[j] * r;
            }
            dst2[2 * ((size_t)y * w + x)] = sat_u8(127.0 + scale * gx);
            dst2[2 * ((size_t)y * w + x) + 1] = sat_u8(127.0 + scale * gy);
        }
    }
}

// cv::equalizeHist (OpenCV >= 2.4.4 implementation; call site HistogramEqualizationFilter.cpp)
void equalize_hist(const uchar* src, int w, int h, int stride, uchar* dst) {
    int hist[256] = {0};
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) hist[src[(size_t)y * stride + x]]++;
    const int total = w * h;
    int i = 0;
    while (!hist[i]) ++i;
    if (hist[i] == total) {
        for (int k = 0; k < total; ++k) dst[k] = (uchar)i;
        return;
    }
    float scale = (256 - 1.f) / (total - hist[i]);
    int lut[256] = {0};
    int sum = 0;
    for (lut[i++] = 0; i < 256; ++i) {
        sum += hist[i];
        lut[i] = sat_u8((double)(sum * scale));
    }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) dst[(size_t)y * w + x] = (uchar)lut[src[(size_t)y * stride + x]];
}

// ---------------------------------------------------------------------------------------
// ImagePyramid (ImagePyramid.cpp:67-92 ctors, :170-198 createLayers)
// ---------------------------------------------------------------------------------------
Pyramid::Pyramid(size_t octl, double minS, double maxS) : octaveLayerCount(octl), minScale(minS), maxScale(maxS) {
    if (octl == 0) throw std::invalid_argument("the number of layers per octave must be greater than zero");
    if (minS <= 0) throw std::invalid_argument("the minimum scale factor must be greater than zero");
    if (maxS > 1) throw std::invalid_argument("the maximum scale factor must not exceed one");
    incScale = std::pow(0.5, 1. / octl);
}

Pyramid* Pyramid::fromInc(double inc, double minS, double maxS) {
    if (inc <= 0 || inc >= 1)
        throw std::invalid_argument("the incremental scale factor must be greater than zero and smaller than one");
    size_t octl = (size_t)std::round(std::log(0.5) / std::log(inc));
    return new Pyramid(octl, minS, maxS);
}

ImgU8 Pyramid::applyLayerFilter(const ImgU8& gray) const {
    if (filterKind == 0) return gray;
    if (filterKind == 1) {
        ImgU8 g(gray.w, gray.h, 2);
        gradient_filter(gray.d.data(), gray.w, gray.h, gradKernel, blurKernel, g.d.data());
        ImgU8 b(gray.w, gray.h, interpolate ? 4 : 2);
        gradient_binning(g.d.data(), gray.w * gray.h, bins, signedGradients, interpolate, b.d.data());
        return b;
    }
    if (filterKind == 2) {
        ImgU8 l(gray.w, gray.h, 1);
        lbp(gray.d.data(), gray.w, gray.h, lbpType, l.d.data());
        return l;
    }
    throw std::invalid_argument("unknown layer filter kind");
}

void Pyramid::update(const uchar* img, int w, int h, int ch) {
    layers.clear();
    imgW = w;
    imgH = h;
    ImgU8 filtered(w, h, 1);
    if (ch == 3) bgr2gray(img, w, h, filtered.d.data());
    else if (ch == 1) std::memcpy(filtered.d.data(), img, (size_t)w * h);
    else throw std::invalid_argument("image must have 1 or 3 channels");
    for (size_t i = 0; i < octaveLayerCount; ++i) {
        double scaleFactor = std::pow(incScale, (double)i);
        ImgU8 scaled(cvRound(filtered.w * scaleFactor), cvRound(filtered.h * scaleFactor), 1);
        resize_linear_u8(filtered.d.data(), filtered.w, filtered.h, scaled.d.data(), scaled.w, scaled.h);
        if (scaleFactor <= maxScale && scaleFactor >= minScale) {
            Layer L{(int)i, scaleFactor, (double)scaled.w / filtered.w, (double)scaled.h / filtered.h,
                    applyLayerFilter(scaled)};
            layers.push_back(std::move(L));
        }
        ImgU8 prev = scaled;
        scaleFactor *= 0.5;
        for (size_t j = 1; scaleFactor >= minScale && prev.w > 1; ++j, scaleFactor *= 0.5) {
            ImgU8 down((prev.w + 1) / 2, (prev.h + 1) / 2, 1);
            pyrdown_u8(prev.d.data(), prev.w, prev.h, down.d.data());
            if (scaleFactor <= maxScale) {
                Layer L{(int)(i + j * octaveLayerCount), scaleFactor, (double)down.w / filtered.w,
                        (double)down.h / filtered.h, applyLayerFilter(down)};
                layers.push_back(std::move(L));
            }
            prev = std::move(down);
        }
    }
    std::sort(layers.begin(), layers.end(), [](const Layer& a, const Layer& b) { return a.index < b.index; });
}

// DirectPyramidFeatureExtractor.cpp:75-123 (window enumeration only; strict '<' bounds)
void enumerate_windows(const Pyramid& p, int pw, int ph, int stepX, int stepY, const int* roiIn,
                       std::vector<Window>& out) {
    if (stepX < 1) throw std::invalid_argument("DirectPyramidFeatureExtractor: stepX has to be greater than zero");
    if (stepY < 1) throw std::invalid_argument("DirectPyramidFeatureExtractor: stepY has to be greater than zero");
    int rx = 0, ry = 0, rw = 0, rh = 0;
    if (roiIn) { rx = roiIn[0]; ry = roiIn[1]; rw = roiIn[2]; rh = roiIn[3]; }
    if (rx == 0 && ry == 0 && rw == 0 && rh == 0) {
        rw = p.imgW;
        rh = p.imgH;
    } else {
        int nx = std::max(0, rx), ny = std::max(0, ry);
        // note: the reference clamps x/y first and then uses the *clamped* x/y in the width formula
        rw = std::min(p.imgW, rw + nx) - nx;
        rh = std::min(p.imgH, rh + ny) - ny;
        rx = nx;
        ry = ny;
    }
    for (size_t li = 0; li < p.layers.size(); ++li) {
        const Layer& L = p.layers[li];
        int ow = L.getOriginal(pw), oh = L.getOriginal(ph);
        int bx = L.getScaled(rx), by = L.getScaled(ry);
        int ex = L.getScaled(rx + rw), ey = L.getScaled(ry + rh);
        for (int y = by; y + ph < ey; y += stepY)
            for (int x = bx; x + pw < ex; x += stepX) {
                Window wdw{(int)li, x, y, L.getOriginal(x) + ow / 2, L.getOriginal(y) + oh / 2, ow, oh};
                out.push_back(wdw);
            }
    }
}

}  // namespace orc

using namespace orc;
extern "C" {
void orc_bgr2gray(const uint8_t* bgr, int w, int h, uint8_t* gray) { bgr2gray(bgr, w, h, gray); }
void orc_resize_linear_u8(const uint8_t* s, int sw, int sh, uint8_t* d, int dw, int dh) { resize_linear_u8(s, sw, sh, d, dw, dh); }
void orc_resize_linear_f32(const float* s, int sw, int sh, float* d, int dw, int dh) { resize_linear_f32(s, sw, sh, d, dw, dh); }
void orc_pyrdown_u8(const uint8_t* s, int sw, int sh, uint8_t* d) { pyrdown_u8(s, sw, sh, d); }
void orc_gradient_filter(const uint8_t* s, int w, int h, int k, int blur, uint8_t* d) { gradient_filter(s, w, h, k, blur, d); }
void orc_equalize_hist(const uint8_t* s, int w, int h, int stride, uint8_t* d) { equalize_hist(s, w, h, stride, d); }

orc_pyramid* orc_pyramid_create(int octl, double minS, double maxS) {
    try { return (orc_pyramid*)new Pyramid((size_t)octl, minS, maxS); } catch (...) { return nullptr; }
}
orc_pyramid* orc_pyramid_create_inc(double inc, double minS, double maxS) {
    try { return (orc_pyramid*)Pyramid::fromInc(inc, minS, maxS); } catch (...) { return nullptr; }
}
void orc_pyramid_destroy(orc_pyramid* p) { delete (Pyramid*)p; }
void orc_pyramid_set_layer_filter(orc_pyramid* p_, int kind, int bins, int sg, int interp, int gk, int bk, int lbpType) {
    Pyramid* p = (Pyramid*)p_;
    p->filterKind = kind; p->bins = bins; p->signedGradients = sg; p->interpolate = interp;
    p->gradKernel = gk; p->blurKernel = bk; p->lbpType = lbpType;
}
void orc_pyramid_update(orc_pyramid* p, const uint8_t* img, int w, int h, int ch) { ((Pyramid*)p)->update(img, w, h, ch); }
int orc_pyramid_octave_layers(const orc_pyramid* p) { return (int)((const Pyramid*)p)->octaveLayerCount; }
double orc_pyramid_inc_scale(const orc_pyramid* p) { return ((const Pyramid*)p)->incScale; }
int orc_pyramid_num_layers(const orc_pyramid* p) { return (int)((const Pyramid*)p)->layers.size(); }
void orc_pyramid_layer_info(const orc_pyramid* p_, int i, int* index, double* scale, int* w, int* h, int* ch) {
    const Layer& L = ((const Pyramid*)p_)->layers[i];
    *index = L.index; *scale = L.scale; *w = L.img.w; *h = L.img.h; *ch = L.img.ch;
}
const uint8_t* orc_pyramid_layer_data(const orc_pyramid* p, int i) { return ((const Pyramid*)p)->layers[i].img.d.data(); }
int64_t orc_extract_windows(const orc_pyramid* p, int pw, int ph, int sx, int sy, const int* roi, int32_t* out, int64_t cap) {
    std::vector<Window> w;
    enumerate_windows(*(const Pyramid*)p, pw, ph, sx, sy, roi, w);
    for (int64_t i = 0; i < (int64_t)w.size() && i < cap; ++i) {
        int32_t* o = out + 7 * i;
        o[0] = w[i].layer; o[1] = w[i].lx; o[2] = w[i].ly; o[3] = w[i].cx; o[4] = w[i].cy; o[5] = w[i].ow; o[6] = w[i].oh;
    }
    return (int64_t)w.size();
}
}
