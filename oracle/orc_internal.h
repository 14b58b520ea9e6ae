// oracle/orc_internal.h -- TEST INFRASTRUCTURE ONLY (see orc_common.h). Internal declarations.
#pragma once
#include "orc_common.h"

namespace orc {

void bgr2gray(const uchar* bgr, int w, int h, uchar* gray);
void resize_linear_u8(const uchar* src, int sw, int sh, uchar* dst, int dw, int dh);
void resize_linear_f32(const float* src, int sw, int sh, float* dst, int dw, int dh);
void pyrdown_u8(const uchar* src, int sw, int sh, uchar* dst);
void gradient_filter(const uchar* src, int w, int h, int ksize, int blur, uchar* dst2);
void equalize_hist(const uchar* src, int w, int h, int stride, uchar* dst);
void gradient_binning(const uchar* grad2, int n, int bins, int signedGradients, int interpolate, uchar* dst);
void lbp(const uchar* src, int w, int h, int type, uchar* dst);
void histeq64(const uchar* src, int w, int h, int stride, uchar* dst);
int hog_filter(const uchar* img, int w, int h, int ch, int strideBytes, int bins, int cellW, int cellH,
               int blockW, int blockH, bool interpolate, bool signedAndUnsigned, std::vector<float>& out);

struct Pyramid {
    size_t octaveLayerCount;
    double incScale, minScale, maxScale;
    int filterKind = 0, bins = 9, signedGradients = 0, interpolate = 0, gradKernel = 1, blurKernel = 0, lbpType = 0;
    int imgW = 0, imgH = 0;
    std::vector<Layer> layers;
    Pyramid(size_t octl, double minS, double maxS);
    static Pyramid* fromInc(double inc, double minS, double maxS);
    void update(const uchar* img, int w, int h, int ch);
    ImgU8 applyLayerFilter(const ImgU8& gray) const;
};

struct Window {
    int layer, lx, ly, cx, cy, ow, oh;
};
void enumerate_windows(const Pyramid& p, int pw, int ph, int stepX, int stepY, const int* roi,
                       std::vector<Window>& out);

struct Wvm {
    int fw, fh, numFilters, numUsed, numPerLevel;
    float basisParam, bias;
    std::vector<float> thresholds, hkWeights;
    std::vector<double> pp, val;
    std::vector<int> valOff, recOff;
    std::vector<uchar> rects;
    double logisticA, logisticB;
    void eval(const uchar* patch, int& lastLevel, float& fout) const;
    bool classify(int lastLevel, double fout) const {
        return lastLevel + 1 == numFilters && fout >= thresholds[lastLevel];
    }
    double probability(double fout) const { return 1.0f / (1.0f + std::exp(logisticA + logisticB * fout)); }
};

struct Svm {
    int kernel;  // 0 linear 1 poly 2 rbf 3 hik
    double p0, p1, p2;
    int nsv, dim, dtype;  // dtype 0 u8 1 f32
    std::vector<uchar> svU8;
    std::vector<float> svF32;
    std::vector<float> coeff;
    float bias, threshold;
    double logisticA, logisticB;
    double kernelValue(const void* x, int i) const;
    double distance(const void* x) const;
    bool classify(double d) const { return d >= threshold; }
    double probability(double d) const {
        double fABp = logisticA + logisticB * d;
        return fABp >= 0 ? std::exp(-fABp) / (1.0 + std::exp(-fABp)) : 1.0 / (1.0 + std::exp(fABp));
    }
};

// filtering::FhogFilter on a gray image (orc_filters.cpp)
int fhog_filter(const uchar* img, int w, int h, int stride, int cellSize, int unsignedBinCount, bool interpolateBins, bool interpolateCells,
                float alpha, std::vector<float>& out, int& rowsOut, int& colsOut, int channels = 1);

// RvmClassifier (RvmClassifier.cpp:75-110) + ProbabilisticRvmClassifier (ProbabilisticRvmClassifier.cpp:52-64)
struct Rvm {
    Svm store;                    // kernel + reduced set vectors (f32), reuses Svm::kernelValue
    std::vector<float> coeff;     // packed lower triangle: coefficients[k][i] at k(k+1)/2 + i
    std::vector<float> thresholds;
    int numFilters, numUse;
    float bias;
    double logisticA, logisticB;
    void eval(const float* x, int& lastLevel, double& distance) const;
    bool classify(int lastLevel, double d) const { return lastLevel + 1 == numUse && d >= thresholds[lastLevel]; }
    double probability(double d) const { return 1.0f / (1.0f + std::exp(logisticA + logisticB * d)); }
};

}  // namespace orc
