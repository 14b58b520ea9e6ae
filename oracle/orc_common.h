// oracle/orc_common.h -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement ("oracle") of the elador/FeatureDetection hot path.  Nothing in the
// product (featuredetection_amd/, include/) may include, link or call this code; only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
//
// Parity status: the reference ships NO tests, golden vectors or classifier models for this
// path and its libraries need OpenCV 2.4 + Boost which are absent here, so
// the whole-path parity is "UNPINNED" by the reference.  What IS pinned: the three reference
// translation units that compile standalone (oracle/_ref: hog.c, IImg.cpp, svm.cpp) are used
// by tests/ to validate the corresponding restatements bit-for-bit (VLFeat HOG, integral
// image) or to fp64 round-off (libsvm RBF/HIK/poly/linear decision values).  OpenCV primitives
// (cvtColor, resize, pyrDown, Sobel, equalizeHist, minMaxLoc) are restated from the OpenCV
// 2.4 algorithms and are *defined* as the spec here.  Guard against slips in
// this restatement: tests/test_oracle_golden.py holds SECOND restatements, written separately in
// numpy / Python scalars from the reference's source, of cvtColor / resize / pyrDown, HistEq64,
// the WVM cascade, the gradient / binning / HOG filter chain, OverlapElimination,
// nonMaximaSuppression and the grey-world filter; the C++ code here must agree with them bit for bit.  They are the same author's
// second reading of the same sources: they catch transcription slips, they are NOT a pin by the reference.
// Round 5 adds the reference's one trained model (detect-landmarks/share/models/SDM_Model_HOG_Zhenhua_11012014.txt) as committed
// data with hog.c's outputs for the patches its regressors visit (tests/golden/make_sdm_real.py).
#pragma once
#include <cstdint>
#include <cmath>
#include <vector>
#include <algorithm>
#include <stdexcept>
#include <string>

namespace orc {

typedef unsigned char uchar;

// cvRound: round-half-to-even (OpenCV uses lrint / SSE cvtsd2si in default rounding mode).
static inline int cvRound(double v) { return (int)std::lrint(v); }
static inline int cvFloor(double v) { int i = (int)v; return i - (v < i); }
static inline uchar sat_u8(int v) { return (uchar)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
static inline uchar sat_u8(double v) { return sat_u8(cvRound(v)); }

// cv::borderInterpolate(p, len, BORDER_REFLECT_101)
static inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

struct ImgU8 {
    int w = 0, h = 0, ch = 1;
    std::vector<uchar> d;
    ImgU8() {}
    ImgU8(int w_, int h_, int ch_ = 1) : w(w_), h(h_), ch(ch_), d((size_t)w_ * h_ * ch_) {}
    uchar* row(int y) { return d.data() + (size_t)y * w * ch; }
    const uchar* row(int y) const { return d.data() + (size_t)y * w * ch; }
};

struct Layer {
    int index;
    double scale, scaleX, scaleY;
    ImgU8 img;  // after layer filters (1, 2 or 4 channels)
    // ImagePyramidLayer.hpp:65-67,98-100
    int getScaled(int v) const { return cvRound(v * scale); }
    int getOriginal(int v) const { return cvRound(v / scale); }
};

}  // namespace orc
