// oracle/orc_sdm.cpp -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
// libSupervisedDescent: VLFeat HOG restatement, VlHogDescriptorExtractor, SdmLandmarkModelFitting.
// The HOG restatement is validated bit-for-bit against the reference's own hog.c (oracle/_ref).
#include "orc_common.h"
#include "orc_internal.h"
#include "oracle.h"
#include <cstring>

namespace orc {

// hog.c:174-218 (orientation table, dimension), :539-573 (buffer sizes), :596-721 (put_image,
// hard orientation assignment since useBilinearOrientationAssigment = FALSE, :185),
// :858-1063 (extract).  Single channel, transposed = false.
static int vlhog(const float* image, int width, int height, int cellSize, int numOri, int variant,
                 std::vector<float>& features, int& hogW, int& hogH) {
    const double VL_PI = 3.141592653589793;
    hogW = (width + cellSize / 2) / cellSize;
    hogH = (height + cellSize / 2) / cellSize;
    const int dimension = variant == 1 ? 3 * numOri + 4 : 4 * numOri;
    const size_t hogStride = (size_t)hogW * hogH;
    std::vector<float> oX(numOri), oY(numOri);
    for (int o = 0; o < numOri; ++o) {
        double angle = o * VL_PI / numOri;
        oX[o] = (float)std::cos(angle);
        oY[o] = (float)std::sin(angle);
    }
    std::vector<float> hog(hogStride * numOri * 2, 0.f), hogNorm(hogStride, 0.f);
    auto at = [&](long x, long y, long k) -> float& { return hog[x + y * hogW + k * hogStride]; };

    for (int y = 1; y < height - 1; ++y)
        for (int x = 1; x < width - 1; ++x) {
            const float* it = image + (size_t)y * width + x;
            float gradx = 0, grady = 0, grad2 = 0;
            {
                float gx_ = *(it + 1) - *(it - 1);
                float gy_ = *(it + width) - *(it - width);
                float g2_ = gx_ * gx_ + gy_ * gy_;
                if (g2_ > grad2) { gradx = gx_; grady = gy_; grad2 = g2_; }
            }
            float grad = sqrtf(grad2);
            gradx /= (grad > 1e-10 ? grad : 1e-10);  // VL_MAX(grad, 1e-10): double division, stored to float
            grady /= (grad > 1e-10 ? grad : 1e-10);
            float w0 = 0, w1 = 0;
            long b0 = -1, b1 = -1;
            for (int k = 0; k < numOri; ++k) {
                float score = gradx * oX[k] + grady * oY[k];
                long bin = k;
                if (score < 0) { score = -score; bin += numOri; }
                if (score > w0) { b1 = b0; w1 = w0; b0 = bin; w0 = score; }
                else if (score > w1) { b1 = bin; w1 = score; }
            }
            (void)b1; (void)w1;
            w0 = 1;  // hard assignment
            const long orientation = b0;
            if (orientation < 0) continue;
            float hx = (x + 0.5) / cellSize - 0.5;
            float hy = (y + 0.5) / cellSize - 0.5;
            long binx = (long)hx; if (!(hx >= 0 || (float)binx == hx)) binx -= 1;  // vl_floor_f
            long biny = (long)hy; if (!(hy >= 0 || (float)biny == hy)) biny -= 1;
            float wx2 = hx - binx, wy2 = hy - biny;
            float wx1 = 1.0 - wx2, wy1 = 1.0 - wy2;
            wx1 *= w0; wx2 *= w0; wy1 *= w0; wy2 *= w0;
            if (binx >= 0 && biny >= 0) at(binx, biny, orientation) += grad * wx1 * wy1;
            if (binx < hogW - 1 && biny >= 0) at(binx + 1, biny, orientation) += grad * wx2 * wy1;
            if (binx < hogW - 1 && biny < hogH - 1) at(binx + 1, biny + 1, orientation) += grad * wx2 * wy2;
            if (binx >= 0 && biny < hogH - 1) at(binx, biny + 1, orientation) += grad * wx1 * wy2;
        }

    // squared L2 norm of the undirected histogram per cell
    for (int k = 0; k < numOri; ++k)
        for (size_t c = 0; c < hogStride; ++c) {
            float h = hog[c + k * hogStride] + hog[c + (size_t)(k + numOri) * hogStride];
            hogNorm[c] += h * h;
        }
    features.assign(hogStride * dimension, 0.f);
    auto N = [&](long x, long y) -> double { return hogNorm[x + y * hogW]; };
    for (long y = 0; y < hogH; ++y)
        for (long x = 0; x < hogW; ++x) {
            long xm = std::max(x - 1, 0L), xp = std::min(x + 1, (long)hogW - 1);
            long ym = std::max(y - 1, 0L), yp = std::min(y + 1, (long)hogH - 1);
            double n1 = N(xm, ym), n2 = N(x, ym), n3 = N(xp, ym), n4 = N(xm, y), n5 = N(x, y), n6 = N(xp, y),
                   n7 = N(xm, yp), n8 = N(x, yp), n9 = N(xp, yp);
            double f1 = 1.0 / std::sqrt(n1 + n2 + n4 + n5 + 1e-4);
            double f2 = 1.0 / std::sqrt(n2 + n3 + n5 + n6 + 1e-4);
            double f3 = 1.0 / std::sqrt(n4 + n5 + n7 + n8 + 1e-4);
            double f4 = 1.0 / std::sqrt(n5 + n6 + n8 + n9 + 1e-4);
            double t1 = 0, t2 = 0, t3 = 0, t4 = 0;
            float* o = features.data() + x + hogW * y;
            const float* it = hog.data() + x + hogW * y;
            for (int k = 0; k < numOri; ++k) {
                double ha = it[hogStride * k], hb = it[hogStride * (k + numOri)];
                double ha1 = f1 * ha, ha2 = f2 * ha, ha3 = f3 * ha, ha4 = f4 * ha;
                double hb1 = f1 * hb, hb2 = f2 * hb, hb3 = f3 * hb, hb4 = f4 * hb;
                double hc1 = ha1 + hb1, hc2 = ha2 + hb2, hc3 = ha3 + hb3, hc4 = ha4 + hb4;
                ha1 = std::min(0.2, ha1); ha2 = std::min(0.2, ha2); ha3 = std::min(0.2, ha3); ha4 = std::min(0.2, ha4);
                hb1 = std::min(0.2, hb1); hb2 = std::min(0.2, hb2); hb3 = std::min(0.2, hb3); hb4 = std::min(0.2, hb4);
                hc1 = std::min(0.2, hc1); hc2 = std::min(0.2, hc2); hc3 = std::min(0.2, hc3); hc4 = std::min(0.2, hc4);
                t1 += hc1; t2 += hc2; t3 += hc3; t4 += hc4;
                if (variant == 1) {
                    *o = 0.5 * (ha1 + ha2 + ha3 + ha4);
                    *(o + hogStride * numOri) = 0.5 * (hb1 + hb2 + hb3 + hb4);
                    *(o + 2 * hogStride * numOri) = 0.5 * (hc1 + hc2 + hc3 + hc4);
                } else {
                    *o = hc1;
                    *(o + hogStride * numOri) = hc2;
                    *(o + 2 * hogStride * numOri) = hc3;
                    *(o + 3 * hogStride * numOri) = hc4;
                }
                o += hogStride;
            }
            if (variant == 1) {
                o += 2 * hogStride * numOri;
                *o = (1.0f / sqrtf(18.0f)) * t1; o += hogStride;
                *o = (1.0f / sqrtf(18.0f)) * t2; o += hogStride;
                *o = (1.0f / sqrtf(18.0f)) * t3; o += hogStride;
                *o = (1.0f / sqrtf(18.0f)) * t4; o += hogStride;
            }
        }
    return dimension;
}

// DescriptorExtractor.hpp:106-219
static int sdm_descriptors(const uchar* gray, int W, int H, const float* px, const float* py, int n, int windowSizeHalf,
                           int variant, int numCells, int cellSize, int numBins, std::vector<float>& out) {
    int patchWidthHalf;
    const bool adaptive = windowSizeHalf > 0;
    if (adaptive) { patchWidthHalf = windowSizeHalf; cellSize = 10; numCells = 3; numBins = 9; }
    else patchWidthHalf = numCells * (cellSize / 2);
    (void)numCells;
    int len = -1;
    const int side = 2 * patchWidthHalf;
    std::vector<float> roi((size_t)side * side), resized, feat, hogArray;
    for (int i = 0; i < n; ++i) {
        int x = cvRound(px[i]), y = cvRound(py[i]);
        int ox = x - patchWidthHalf, oy = y - patchWidthHalf;  // roi origin in image coordinates
        if (x - patchWidthHalf < 0 || y - patchWidthHalf < 0 || x + patchWidthHalf >= W || y + patchWidthHalf >= H) {
            int borderLeft = (x - patchWidthHalf) < 0 ? std::abs(x - patchWidthHalf) : 0;
            int borderTop = (y - patchWidthHalf) < 0 ? std::abs(y - patchWidthHalf) : 0;
            int borderRight = (x + patchWidthHalf) >= W ? std::abs(W - (x + patchWidthHalf)) : 0;
            int borderBottom = (y + patchWidthHalf) >= H ? std::abs(H - (y + patchWidthHalf)) : 0;
            // roi in the extended image; note the reference's quirk: y uses borderRight (:171)
            int rx = (x - patchWidthHalf) + borderLeft, ry = (y - patchWidthHalf) + borderRight;
            int EW = W + borderLeft + borderRight, EH = H + borderTop + borderBottom;
            if (rx < 0 || ry < 0 || rx + side > EW || ry + side > EH) return -1;  // cv::Mat(roi) assertion -> exception
            ox = rx - borderLeft;  // back to source-image coordinates (may be negative)
            oy = ry - borderTop;
        }
        for (int r = 0; r < side; ++r)
            for (int c = 0; c < side; ++c) {
                int sx = ox + c, sy = oy + r;
                roi[(size_t)r * side + c] = (sx >= 0 && sy >= 0 && sx < W && sy < H) ? (float)gray[(size_t)sy * W + sx] : 0.f;
            }
        const float* img = roi.data();
        int iw = side, ih = side;
        if (adaptive) {
            resized.resize(30 * 30);
            if (side == 30) std::memcpy(resized.data(), roi.data(), sizeof(float) * 900);
            else resize_linear_f32(roi.data(), side, side, resized.data(), 30, 30);
            img = resized.data(); iw = ih = 30;
        }
        int ww, hh;
        int dd = vlhog(img, iw, ih, cellSize, numBins, variant, hogArray, ww, hh);
        // transpose each of the dd planes (hh x ww) and stack them (DescriptorExtractor.hpp:198-205)
        if (len < 0) { len = ww * hh * dd; out.assign((size_t)n * len, 0.f); }
        float* o = out.data() + (size_t)i * len;
        for (int j = 0; j < dd; ++j)
            for (int c = 0; c < ww; ++c)
                for (int r = 0; r < hh; ++r) o[(size_t)j * ww * hh + (size_t)c * hh + r] = hogArray[(size_t)j * ww * hh + (size_t)r * ww + c];
    }
    return len;
}

// SdmLandmarkModel.hpp:199-256 (adaptive branch).  cv::Mat float gemm accumulates in double? No:
// OpenCV's GEMM for CV_32F (gemmImpl<float,double>) accumulates each dot product in double and
// rounds once to float; the bias row is added afterwards in float (MatExpr a*b + c -> gemm with
// beta=1 adds inside the double accumulator).  We restate it as double accumulate + one rounding.
// descParams == NULL: the branch the reference compiles in (`if (true) { // adaptive`, :209,243).  Otherwise the `else` branches of
// :236-238 and :246-248 ("non-adaptive, the descriptorExtractor has all necessary params"): getDescriptors(image, points) with
// windowSizeHalf = 0, i.e. the extractor's own {numCells, cellSize, numBins} (descParams[3*step ..]), and modelShape + deltaShape.t()
// without the face-size factor.
static int sdm_optimize(const uchar* gray, int W, int H, float* shape, int L, int S, const float* const* R,
                        const int* Rrows, int variant, const int* descParams = nullptr) {
    std::vector<float> px(L), py(L), feats, delta(2 * (size_t)L);
    for (int step = 0; step < S; ++step) {
        for (int i = 0; i < L; ++i) { px[i] = shape[i]; py[i] = shape[i + L]; }
        if (descParams) {
            const int* dp = descParams + 3 * step;
            int len = sdm_descriptors(gray, W, H, px.data(), py.data(), L, 0, variant, dp[0], dp[1], dp[2], feats);
            if (len < 0) return -1;
            const int F = len * L;
            if (Rrows[step] != F + 1) return -2;
            const float* Rm = R[step];
            for (int j = 0; j < 2 * L; ++j) {
                double acc = 0;
                for (int k = 0; k < F; ++k) acc += (double)feats[k] * (double)Rm[(size_t)k * 2 * L + j];
                acc += (double)Rm[(size_t)F * 2 * L + j];
                delta[j] = (float)acc;
            }
            for (int j = 0; j < 2 * L; ++j) shape[j] = shape[j] + delta[j];
            continue;
        }
        float a1x = (shape[8] + shape[9]) / 2.0f, a1y = (shape[8 + L] + shape[9 + L]) / 2.0f;
        float a2x = (shape[11] + shape[12]) / 2.0f, a2y = (shape[11 + L] + shape[12 + L]) / 2.0f;
        // cv::norm(Vec2f) = sqrt of the double-accumulated squares
        double dx = (double)(a1x - a2x), dy = (double)(a1y - a2y);
        float dist = (float)std::sqrt(dx * dx + dy * dy);
        float windowSize = dist / 2.0f;
        float windowSizeHalf = windowSize / 2;
        windowSizeHalf = std::round(windowSizeHalf * (1 / (1 + std::exp((step + 1) - S))));
        const int NUM_CELL = 3;
        int wshi = (int)windowSizeHalf + NUM_CELL - ((int)windowSizeHalf % NUM_CELL);
        int len = sdm_descriptors(gray, W, H, px.data(), py.data(), L, wshi, variant, 3, 10, 9, feats);
        if (len < 0) return -1;
        const int F = len * L;
        if (Rrows[step] != F + 1) return -2;
        const float* Rm = R[step];
        for (int j = 0; j < 2 * L; ++j) {
            double acc = 0;
            for (int k = 0; k < F; ++k) acc += (double)feats[k] * (double)Rm[(size_t)k * 2 * L + j];
            acc += (double)Rm[(size_t)F * 2 * L + j];
            delta[j] = (float)acc;
        }
        for (int j = 0; j < 2 * L; ++j) shape[j] = shape[j] + delta[j] * dist;
    }
    return 0;
}

}  // namespace orc

using namespace orc;
extern "C" {
int orc_vlhog(const float* img, int w, int h, int cellSize, int numOrient, int variant, float* out, int* hogW, int* hogH) {
    std::vector<float> f;
    int d = vlhog(img, w, h, cellSize, numOrient, variant, f, *hogW, *hogH);
    if (out) std::memcpy(out, f.data(), sizeof(float) * f.size());
    return d;
}
int orc_sdm_descriptors(const uint8_t* gray, int w, int h, const float* px, const float* py, int n, int wsh, int variant,
                        int numCells, int cellSize, int numBins, float* out) {
    std::vector<float> f;
    int len = sdm_descriptors(gray, w, h, px, py, n, wsh, variant, numCells, cellSize, numBins, f);
    if (len > 0 && out) std::memcpy(out, f.data(), sizeof(float) * f.size());
    return len;
}
void orc_sdm_align_rigid(float* shape, int L, const int* fb) {
    // SdmLandmarkModel.hpp:168-169: "(x + 0.5f) * w + bx" is a cv::MatExpr, which OpenCV folds into one
    // convertTo(alpha = w, beta = 0.5*w + bx) evaluated as float(x)*float(alpha) + float(beta).
    const float ax = (float)(double)fb[2], bx = (float)(0.5 * fb[2] + fb[0]);
    const float ay = (float)(double)fb[3], by = (float)(0.5 * fb[3] + fb[1]);
    for (int i = 0; i < L; ++i) {
        shape[i] = shape[i] * ax + bx;
        shape[i + L] = shape[i + L] * ay + by;
    }
}
int orc_sdm_optimize(const uint8_t* gray, int w, int h, float* shape, int L, int S, const float* const* R,
                     const int* Rrows, int variant) {
    return sdm_optimize(gray, w, h, shape, L, S, R, Rrows, variant);
}
int orc_sdm_optimize_fixed(const uint8_t* gray, int w, int h, float* shape, int L, int S, const float* const* R,
                           const int* Rrows, int variant, const int* descParams) {
    return sdm_optimize(gray, w, h, shape, L, S, R, Rrows, variant, descParams);
}
}
