// oracle/orc_detect.cpp -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
// libDetection: SlidingWindowDetector, OverlapElimination, FiveStageSlidingWindowDetector.
#include "orc_common.h"
#include "orc_internal.h"
#include "oracle.h"
#include <cstring>
#include <memory>
#include <chrono>

namespace orc {

// Phase timers of bench.py's cpu_baseline leg (update / extract / classify).  Thread-local, off by
// default; the two clock reads per window cost ~50 ns against >= 2 us of work per window.
struct PhaseTimes { bool on = false; double extract = 0, classify = 0; };
static thread_local PhaseTimes g_phase;
struct PhaseScope {
    double& acc;
    std::chrono::steady_clock::time_point t0;
    bool on;
    explicit PhaseScope(double& a) : acc(a), on(g_phase.on) { if (on) t0 = std::chrono::steady_clock::now(); }
    ~PhaseScope() { if (on) acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

// OverlapElimination.cpp:44-105.  The reference sorts shared_ptrs through boost::indirect_iterator
// with std::greater<ClassifiedPatch> (probability only), i.e. a plain std::sort on the same sequence.
static std::vector<int> overlap_elimination(const std::vector<orc_det>& in, float distIn, float ratioIn) {
    std::vector<int> cand(in.size());
    for (size_t i = 0; i < in.size(); ++i) cand[i] = (int)i;
    if (cand.empty()) return cand;
    float dist = distIn;
    float ratio = ((ratioIn > 0.0f) && (ratioIn <= 1.0f)) ? ratioIn : 0.0f;
    float d;
    std::sort(cand.begin(), cand.end(), [&](int a, int b) { return in[a].prob > in[b].prob; });
    for (auto accepted = cand.begin(); accepted != cand.end(); accepted++) {
        for (auto proband = accepted + 1; proband != cand.end();) {
            const orc_det &A = in[*accepted], &P = in[*proband];
            if (dist <= 1.0) d = dist * std::max(A.w, P.w);
            else d = dist;
            if ((std::abs(A.cx - P.cx) < d) && (std::abs(A.cy - P.cy) < d) &&
                (((float)std::min(A.w, P.w) / (float)std::max(A.w, P.w)) > ratio))
                proband = cand.erase(proband);
            else
                proband++;
        }
    }
    return cand;
}

// cv::minMaxLoc(src(rows,cols), NULL, &maxVal, NULL, &maxLoc, mask) on a CV_32F sub-matrix
// (OpenCV 2.4 minMaxIdx: strict '>' so the first row-major maximum wins; an all-zero mask yields
// maxVal = 0 and location (-1,-1)).
static void maxloc(const float* src, int W, int r0, int r1, int c0, int c1, const uchar* mask, int maskStride,
                   double& maxVal, int& mx, int& my) {
    float best = -3.402823466e+38F;
    size_t idx = 0, k = 1;
    for (int r = r0; r < r1; ++r)
        for (int c = c0; c < c1; ++c, ++k) {
            if (mask && !mask[(size_t)(r - r0) * maskStride + (c - c0)]) continue;
            float v = src[(size_t)r * W + c];
            if (v > best) { best = v; idx = k; }
        }
    if (idx == 0) { maxVal = 0; mx = -1; my = -1; }
    else { maxVal = best; size_t o = idx - 1; int cols = c1 - c0; my = (int)(o / cols); mx = (int)(o % cols); }
}

// nonMaximaSuppression, FiveStageSlidingWindowDetector.cpp:143-184
static void block_nms(const float* src, int M, int N, int sz, const uchar* mask, uchar* dst) {
    const bool masked = mask != nullptr;
    std::memset(dst, 0, (size_t)M * N);
    std::vector<uchar> bm;
    for (int m = 0; m < M; m += sz + 1)
        for (int n = 0; n < N; n += sz + 1) {
            int ic0 = m, ic1 = std::min(m + sz + 1, M), jc0 = n, jc1 = std::min(n + sz + 1, N);
            double vcmax, vnmax;
            int ix, iy;
            // candidate: maximum inside the block (mask restricted to the block)
            std::vector<uchar> cm;
            if (masked) {
                cm.resize((size_t)(ic1 - ic0) * (jc1 - jc0));
                for (int r = ic0; r < ic1; ++r)
                    for (int c = jc0; c < jc1; ++c) cm[(size_t)(r - ic0) * (jc1 - jc0) + (c - jc0)] = mask[(size_t)r * N + c];
            }
            maxloc(src, N, ic0, ic1, jc0, jc1, masked ? cm.data() : nullptr, jc1 - jc0, vcmax, ix, iy);
            int ccx = ix + jc0, ccy = iy + ic0;
            int in0 = std::max(ccy - sz, 0), in1 = std::min(ccy + sz + 1, M);
            int jn0 = std::max(ccx - sz, 0), jn1 = std::min(ccx + sz + 1, N);
            int ih = in1 - in0, jw = jn1 - jn0;
            if (ih <= 0 || jw <= 0) continue;  // cannot happen for sz >= 0 inside the image
            bm.assign((size_t)ih * jw, 255);
            int iis0 = ic0 - in0, iis1 = std::min(ic0 - in0 + sz + 1, ih);
            int jis0 = jc0 - jn0, jis1 = std::min(jc0 - jn0 + sz + 1, jw);
            for (int r = iis0; r < iis1; ++r)
                for (int c = jis0; c < jis1; ++c) bm[(size_t)r * jw + c] = 0;
            if (masked)  // mask(in,jn).mul(blockmask): non-zero iff both non-zero
                for (int r = 0; r < ih; ++r)
                    for (int c = 0; c < jw; ++c)
                        if (!mask[(size_t)(r + in0) * N + (c + jn0)]) bm[(size_t)r * jw + c] = 0;
            maxloc(src, N, in0, in1, jn0, jn1, bm.data(), jw, vnmax, ix, iy);
            if (vcmax > vnmax) dst[(size_t)ccy * N + ccx] = 255;
        }
}

static orc_det make_det(const Window& w) {
    orc_det d;
    std::memset(&d, 0, sizeof(d));
    d.cx = w.cx; d.cy = w.cy; d.w = w.ow; d.h = w.oh; d.layer = w.layer; d.lx = w.lx; d.ly = w.ly;
    d.level = -1; d.positive = 0; d.fout = 0; d.prob = 0.5;
    return d;
}

struct Scored {
    orc_det det;
    std::vector<uchar> data;  // HistEq64'd patch (Patch::data)
};

// SlidingWindowDetector.cpp:53-78 / :87-98 with DirectPyramidFeatureExtractor + HistEq64Filter + PWVM
static void sliding_wvm(const Pyramid& p, const Wvm& m, int stepX, int stepY, const int* roi,
                        std::vector<Scored>& positives, int32_t* all_level, float* all_fout) {
    std::vector<Window> wins;
    enumerate_windows(p, m.fw, m.fh, stepX, stepY, roi, wins);
    std::vector<uchar> eq((size_t)m.fw * m.fh);
    for (size_t i = 0; i < wins.size(); ++i) {
        const Window& w = wins[i];
        const ImgU8& img = p.layers[w.layer].img;
        { PhaseScope ps(g_phase.extract); histeq64(img.d.data() + (size_t)w.ly * img.w + w.lx, m.fw, m.fh, img.w, eq.data()); }
        int level; float fout;
        { PhaseScope ps(g_phase.classify); m.eval(eq.data(), level, fout); }
        if (all_level) all_level[i] = level;
        if (all_fout) all_fout[i] = fout;
        if (m.classify(level, fout)) {
            Scored s;
            s.det = make_det(w);
            s.det.level = level; s.det.fout = fout; s.det.positive = 1; s.det.prob = m.probability(fout);
            s.data = eq;
            positives.push_back(std::move(s));
        }
    }
}

// DirectPyramidFeatureExtractor::extract(x, y, width, height) (DirectPyramidFeatureExtractor.cpp:67-73,134-153) with
// ImagePyramid::getLayer(double scaleFactor) / getLayer(int index) (ImagePyramid.cpp:300-310): the window of the
// sample, or false when there is no such layer / the patch leaves the layer.
static bool extract_single(const Pyramid& p, int pw, int ph, int x, int y, int width, int height, Window& w) {
    if (p.layers.empty()) return false;
    const double scaleFactor = (double)pw / (double)width;
    const double power = std::log(scaleFactor) / std::log(p.incScale);
    const int index = (int)std::round(power);
    const int realIndex = index - p.layers.front().index;   // layers are sorted by index, consecutive
    if (realIndex < 0 || realIndex >= (int)p.layers.size()) return false;
    const Layer& L = p.layers[realIndex];
    const int bx = cvRound((x - width / 2) * L.scale), by = cvRound((y - height / 2) * L.scale);
    if (bx < 0 || by < 0 || bx + pw > L.img.w || by + ph > L.img.h) return false;
    w.layer = realIndex; w.lx = bx; w.ly = by;
    w.ow = cvRound(pw / L.scale); w.oh = cvRound(ph / L.scale);
    w.cx = cvRound(bx / L.scale) + w.ow / 2;
    w.cy = cvRound(by / L.scale) + w.oh / 2;
    return true;
}

// condensation::WvmSvmModel::evaluate(image, samples) (WvmSvmModel.cpp:69-118) with a DirectPyramidFeatureExtractor +
// HistEq64Filter.  The cache map is keyed by shared_ptr<Patch> with the default (pointer) hash and equality
// (WvmSvmModel.hpp:61) and extract() makes a new Patch per call, so the cache never hits: every sample is scored.
static void wvm_svm_evaluate(const Pyramid& p, const Wvm& wvm, const Svm& svm, int n, const int32_t* xywh, uint8_t* target, double* weight) {
    struct Remaining { int sample; double prob; std::vector<uchar> data; };
    std::vector<Remaining> remaining;
    std::vector<uchar> eq((size_t)wvm.fw * wvm.fh);
    for (int i = 0; i < n; ++i) {
        target[i] = 0;
        Window w;
        if (!extract_single(p, wvm.fw, wvm.fh, xywh[4 * i], xywh[4 * i + 1], xywh[4 * i + 2], xywh[4 * i + 3], w)) {
            weight[i] = 0;
            continue;
        }
        const ImgU8& img = p.layers[w.layer].img;
        histeq64(img.d.data() + (size_t)w.ly * img.w + w.lx, wvm.fw, wvm.fh, img.w, eq.data());
        int level; float fout;
        wvm.eval(eq.data(), level, fout);
        const double prob = wvm.probability(fout);
        if (wvm.classify(level, fout)) remaining.push_back(Remaining{i, prob, eq});
        weight[i] = 0.5 * prob;
    }
    if (!remaining.empty()) {
        if (remaining.size() > 8) {
            // sort(make_indirect_iterator(...), greater<ClassifiedPatch>()): probability descending, unstable;
            // restated with the same std::sort on the same sequence of keys
            std::sort(remaining.begin(), remaining.end(), [](const Remaining& a, const Remaining& b) { return a.prob > b.prob; });
            remaining.resize(8);
        }
        for (const Remaining& r : remaining) {
            const double dist = svm.distance(r.data.data());
            target[r.sample] = svm.classify(dist) ? 1 : 0;
            weight[r.sample] = 2 * weight[r.sample] * svm.probability(dist);
        }
    }
}

// FiveStageSlidingWindowDetector.cpp:187-320 (roi == nullptr) and :331-380 (roi != nullptr)
static std::vector<orc_det> five_stage(const Pyramid& p, int imgW, int imgH, const Wvm& wvm, const Svm& svm,
                                       float oeDist, float oeRatio, int stepX, int stepY, const int* roi,
                                       int32_t* counts) {
    std::vector<Scored> cls;
    sliding_wvm(p, wvm, stepX, stepY, roi, cls, nullptr, nullptr);
    if (counts) counts[0] = (int)cls.size();
    PhaseScope psTail(g_phase.classify);   // OE + SVM + NMS belong to the classify phase
    // OE
    std::vector<orc_det> dets(cls.size());
    for (size_t i = 0; i < cls.size(); ++i) dets[i] = cls[i].det;
    std::vector<int> keep = overlap_elimination(dets, oeDist, oeRatio);
    if (counts) counts[1] = (int)keep.size();
    // SVM stage: classify() only -> ClassifiedPatch(patch, bool) => probability 0.5 (ClassifiedPatch.hpp:29-30)
    std::vector<orc_det> svmPos;
    for (int k : keep) {
        double dist = svm.distance(cls[k].data.data());
        if (svm.classify(dist)) {
            orc_det d = cls[k].det;
            d.fout = (float)dist; d.positive = 1; d.prob = 0.5;
            svmPos.push_back(d);
        }
    }
    if (counts) counts[2] = (int)svmPos.size();
    if (!roi) {
        std::vector<float> map((size_t)imgW * imgH, 0.f);
        for (const auto& d : svmPos)
            if (map[(size_t)d.cy * imgW + d.cx] < d.prob) map[(size_t)d.cy * imgW + d.cx] = (float)d.prob;
        std::vector<uchar> mask((size_t)imgW * imgH), maxima((size_t)imgW * imgH);
        for (size_t i = 0; i < map.size(); ++i) mask[i] = map[i] > 0.3f ? 255 : 0;
        block_nms(map.data(), imgH, imgW, 35, mask.data(), maxima.data());
        size_t nz = 0;
        for (uchar v : maxima) nz += v != 0;
        if (nz == 0) {
            block_nms(map.data(), imgH, imgW, 35, nullptr, maxima.data());
            for (uchar v : maxima) nz += v != 0;
            if (nz == 0) { if (counts) counts[3] = (int)svmPos.size(); return svmPos; }
        }
        std::sort(svmPos.begin(), svmPos.end(), [](const orc_det& a, const orc_det& b) { return a.prob > b.prob; });
        std::vector<orc_det> res;
        for (int y = 0; y < imgH; ++y)  // cv::findNonZero: row-major order
            for (int x = 0; x < imgW; ++x) {
                if (!maxima[(size_t)y * imgW + x]) continue;
                auto it = std::find_if(svmPos.begin(), svmPos.end(), [&](const orc_det& a) { return a.cx == x && a.cy == y; });
                if (it != svmPos.end()) res.push_back(*it);  // (the reference dereferences end() here: UB)
            }
        svmPos = res;
    }
    std::sort(svmPos.begin(), svmPos.end(), [](const orc_det& a, const orc_det& b) { return a.prob > b.prob; });
    if (counts) counts[3] = (int)svmPos.size();
    return svmPos;
}

}  // namespace orc

using namespace orc;
extern "C" {
int orc_overlap_elimination(int n, const orc_det* in, float dist, float ratio, int32_t* idx_out) {
    std::vector<orc_det> v(in, in + n);
    std::vector<int> k = overlap_elimination(v, dist, ratio);
    for (size_t i = 0; i < k.size(); ++i) idx_out[i] = k[i];
    return (int)k.size();
}
void orc_block_nms(const float* map, int H, int W, int sz, const uint8_t* mask, uint8_t* dst) { block_nms(map, H, W, sz, mask, dst); }

int64_t orc_sliding_wvm(const orc_pyramid* p, const orc_wvm* m, int stepX, int stepY, const int* roi, orc_det* out,
                        int64_t cap, int32_t* all_level, float* all_fout) {
    std::vector<Scored> pos;
    sliding_wvm(*(const Pyramid*)p, *(const Wvm*)m, stepX, stepY, roi, pos, all_level, all_fout);
    for (int64_t i = 0; i < (int64_t)pos.size() && i < cap; ++i) out[i] = pos[i].det;
    return (int64_t)pos.size();
}

// detection::AggregatedFeaturesDetector (AggregatedFeaturesDetector.cpp:37-128) with a GrayscaleFilter image filter and a
// FhogFilter layer filter on an AggregatedFeaturesExtractor (AggregatedFeaturesExtractor.cpp:34-86): feature pyramid
// ImagePyramid(octaveLayerCount, 0.5, 1) whose minimum / maximum scale factors follow the image and the minimum window
// width, score pyramid = ConvolutionFilter(CV_32F) of the linear SVM's support vector with anchor (0, 0) and delta = -bias
// (ConvolutionFilter.cpp:27-43: per channel cv::filter2D, summed channel by channel onto delta), windows with
// score > threshold in layer / row / column order, bounds through the layer's actual x / y scale, rescaleWindow.
// Returns the candidates (before NonMaximumSuppression).
int orc_aggregated_candidates(const uint8_t* img, int w, int h, int ch, int cellSize, int unsignedBinCount, int interpolateBins,
                              int interpolateCells, float alpha, int windowW, int windowH, int octaveLayerCount, int minWindowWidth,
                              float widthScale, float heightScale, const float* svmWeights, float svmBias, float scoreThreshold,
                              float* outScore, int32_t* outXywh, int cap) {
    const int D = 3 * unsignedBinCount + 4;
    const int patchWpx = windowW * cellSize, patchHpx = windowH * cellSize;
    const double inc = std::pow(0.5, 1. / octaveLayerCount);   // ImagePyramid.cpp:67-78
    double maxScale = 1.0;
    if (minWindowWidth > patchWpx) {   // AggregatedFeaturesExtractor.cpp:30-31,47-52
        double m = (double)patchWpx / minWindowWidth;
        int minLayerIndex = (int)std::ceil(std::log(m) / std::log(inc));
        maxScale = std::pow(inc, minLayerIndex);
    }
    double minScale;
    {   // getMinScaleFactor / getMaxWidth :64-77
        double aspectRatio = (double)patchHpx / (double)patchWpx;
        double imageAspectRatio = (double)h / (double)w;
        int maxWidth = aspectRatio > imageAspectRatio ? (int)(h / aspectRatio) : w;
        double m = (double)patchWpx / maxWidth;
        int maxLayerIndex = (int)(std::log(m) / std::log(inc));
        minScale = std::pow(inc, maxLayerIndex);
    }
    Pyramid pyr((size_t)octaveLayerCount, minScale, maxScale);
    pyr.update(img, w, h, ch);
    if (pyr.layers.size() < 2) return -1;   // ImagePyramid::estimateLambdas (ImagePyramid.cpp:240-242) throws
    int n = 0;
    for (const Layer& L : pyr.layers) {
        std::vector<float> F;
        int rows, cols;
        fhog_filter(L.img.d.data(), L.img.w, L.img.h, L.img.w, cellSize, unsignedBinCount, interpolateBins != 0, interpolateCells != 0, alpha, F,
                    rows, cols);
        int validHeight = rows - windowH + 1, validWidth = cols - windowW + 1;   // AggregatedFeaturesDetector.cpp:91-92
        for (int y = 0; y < validHeight; ++y)
            for (int x = 0; x < validWidth; ++x) {
                float score = -svmBias;   // filtered = delta; filtered += filter2D(channel_i, kernel_i)
                for (int c = 0; c < D; ++c) {
                    float s = 0;
                    for (int ky = 0; ky < windowH; ++ky)
                        for (int kx = 0; kx < windowW; ++kx)
                            s += svmWeights[((size_t)ky * windowW + kx) * D + c] * F[((size_t)(y + ky) * cols + (x + kx)) * D + c];
                    score += s;
                }
                if (score > scoreThreshold) {
                    // computeBoundsInImagePixels (AggregatedFeaturesExtractor.cpp:123-130), rescaleWindow (:108-112)
                    int bx = (int)std::round((x * cellSize) / L.scaleX), by = (int)std::round((y * cellSize) / L.scaleY);
                    int bw = (int)std::round((windowW * cellSize) / L.scaleX), bh = (int)std::round((windowH * cellSize) / L.scaleY);
                    int cx = bx + bw / 2, cy = by + bh / 2;
                    int rw = (int)(widthScale * bw), rh = (int)(heightScale * bh);
                    if (n < cap) {
                        outScore[n] = score;
                        outXywh[4 * n] = cx - rw / 2; outXywh[4 * n + 1] = cy - rh / 2; outXywh[4 * n + 2] = rw; outXywh[4 * n + 3] = rh;
                    }
                    ++n;
                }
            }
    }
    return n;
}

// NonMaximumSuppression.cpp:27-118
int orc_nms_iou(int n, const float* score, const int32_t* xywh, double overlapThreshold, int maximumType, float* outScore, int32_t* outXywh) {
    struct Det { float score; int x, y, w, h; };
    std::vector<Det> candidates;
    for (int i = 0; i < n; ++i) candidates.push_back(Det{score[i], xywh[4 * i], xywh[4 * i + 1], xywh[4 * i + 2], xywh[4 * i + 3]});
    std::vector<Det> finalDetections;
    if (overlapThreshold == 1.0) {
        finalDetections = candidates;
    } else {
        std::sort(candidates.begin(), candidates.end(), [](const Det& a, const Det& b) { return a.score < b.score; });   // sortByScore :35-39
        auto computeOverlap = [](const Det& a, const Det& b) {   // :58-62
            int x1 = std::max(a.x, b.x), y1 = std::max(a.y, b.y);
            int x2 = std::min(a.x + a.w, b.x + b.w), y2 = std::min(a.y + a.h, b.y + b.h);
            double intersectionArea = (x2 - x1 <= 0 || y2 - y1 <= 0) ? 0 : (x2 - x1) * (y2 - y1);
            double unionArea = a.w * a.h + b.w * b.h - intersectionArea;
            return intersectionArea / unionArea;
        };
        std::vector<std::vector<Det>> clusters;
        while (!candidates.empty()) {   // cluster :41-46, extractOverlappingDetections :48-57
            Det detection = candidates.back();
            std::vector<Det> overlappingDetections;
            auto firstOverlapping = std::stable_partition(candidates.begin(), candidates.end(), [&](const Det& candidate) {
                return computeOverlap(detection, candidate) <= overlapThreshold;
            });
            std::move(firstOverlapping, candidates.end(), std::back_inserter(overlappingDetections));
            std::reverse(overlappingDetections.begin(), overlappingDetections.end());
            candidates.erase(firstOverlapping, candidates.end());
            if (overlappingDetections.empty()) return -1;   // overlap threshold > 1: endless loop in the reference
            clusters.push_back(overlappingDetections);
        }
        for (const std::vector<Det>& cluster : clusters) {   // getMaximum :72-118
            Det r = cluster.front();
            if (maximumType == 1) {
                double xSum = 0, ySum = 0, wSum = 0, hSum = 0;
                for (const Det& e : cluster) { xSum += e.x; ySum += e.y; wSum += e.w; hSum += e.h; }
                r.x = (int)std::round(xSum / cluster.size()); r.y = (int)std::round(ySum / cluster.size());
                r.w = (int)std::round(wSum / cluster.size()); r.h = (int)std::round(hSum / cluster.size());
            } else if (maximumType == 2) {
                double weightSum = 0, xSum = 0, ySum = 0, wSum = 0, hSum = 0;
                for (const Det& e : cluster) {
                    double weight = e.score;
                    weightSum += weight;
                    xSum += weight * e.x; ySum += weight * e.y; wSum += weight * e.w; hSum += weight * e.h;
                }
                r.x = (int)std::round(xSum / weightSum); r.y = (int)std::round(ySum / weightSum);
                r.w = (int)std::round(wSum / weightSum); r.h = (int)std::round(hSum / weightSum);
            }
            finalDetections.push_back(r);
        }
    }
    for (size_t i = 0; i < finalDetections.size(); ++i) {
        outScore[i] = finalDetections[i].score;
        outXywh[4 * i] = finalDetections[i].x; outXywh[4 * i + 1] = finalDetections[i].y;
        outXywh[4 * i + 2] = finalDetections[i].w; outXywh[4 * i + 3] = finalDetections[i].h;
    }
    return (int)finalDetections.size();
}

int orc_extract_single(const orc_pyramid* p, int pw, int ph, int x, int y, int width, int height, int32_t* out7) {
    Window w;
    if (!extract_single(*(const Pyramid*)p, pw, ph, x, y, width, height, w)) return 0;
    out7[0] = w.layer; out7[1] = w.lx; out7[2] = w.ly; out7[3] = w.cx; out7[4] = w.cy; out7[5] = w.ow; out7[6] = w.oh;
    return 1;
}
void orc_wvm_svm_evaluate(const orc_pyramid* p, const orc_wvm* wvm, const orc_svm* svm, int n, const int32_t* xywh, uint8_t* target,
                          double* weight) {
    wvm_svm_evaluate(*(const Pyramid*)p, *(const Wvm*)wvm, *(const Svm*)svm, n, xywh, target, weight);
}

int orc_five_stage(const orc_pyramid* p, int imgW, int imgH, const orc_wvm* wvm, const orc_svm* svm, float oeDist,
                   float oeRatio, int stepX, int stepY, const int* roi, orc_det* out, int cap, int32_t* counts) {
    std::vector<orc_det> r = five_stage(*(const Pyramid*)p, imgW, imgH, *(const Wvm*)wvm, *(const Svm*)svm, oeDist,
                                        oeRatio, stepX, stepY, roi, counts);
    for (int i = 0; i < (int)r.size() && i < cap; ++i) out[i] = r[i];
    return (int)r.size();
}

void orc_phase_timing(int enable) { g_phase.on = enable != 0; g_phase.extract = g_phase.classify = 0; }
void orc_phase_get(double* extract_s, double* classify_s) { *extract_s = g_phase.extract; *classify_s = g_phase.classify; }

// bench.py cpu_baseline only: the same loop over windows first, first + step, ... (a bounded sample of one full-size frame, or one
// thread's share of it); *visited = windows evaluated.  all_dist / feat_out are indexed by window like the full run.
int64_t orc_sliding_hog_svm_sample(const orc_pyramid* p_, const orc_svm* svm_, int pw, int ph, int stepX, int stepY, int bins,
                                   int cell, int block, int interpolate, int signedAndUnsigned, orc_det* out, int64_t cap,
                                   double* all_dist, float* feat_out, int64_t feat_cap_windows, int64_t first, int64_t step,
                                   int64_t* visited) {
    const Pyramid& p = *(const Pyramid*)p_;
    const Svm* svm = (const Svm*)svm_;
    std::vector<Window> wins;
    enumerate_windows(p, pw, ph, stepX, stepY, nullptr, wins);
    std::vector<float> feat;
    int64_t npos = 0, nvis = 0;
    if (step < 1) step = 1;
    if (first < 0) first = 0;
    for (size_t i = (size_t)first; i < wins.size(); i += (size_t)step) {
        ++nvis;
        const Window& w = wins[i];
        const ImgU8& img = p.layers[w.layer].img;
        const uchar* src = img.d.data() + ((size_t)w.ly * img.w + w.lx) * img.ch;
        { PhaseScope ps(g_phase.extract);
          hog_filter(src, pw, ph, img.ch, img.w * img.ch, bins, cell, cell, block, block, interpolate != 0,
                     signedAndUnsigned != 0, feat); }
        if (feat_out && (int64_t)i < feat_cap_windows)
            std::memcpy(feat_out + i * feat.size(), feat.data(), sizeof(float) * feat.size());
        if (!svm) continue;
        double dist;
        { PhaseScope ps(g_phase.classify); dist = svm->distance(feat.data()); }
        if (all_dist) all_dist[i] = dist;
        if (svm->classify(dist)) {
            if (npos < cap && out) {
                orc_det d = make_det(w);
                d.fout = (float)dist; d.positive = 1; d.prob = svm->probability(dist);
                out[npos] = d;
            }
            ++npos;
        }
    }
    if (visited) *visited = nvis;
    return svm ? npos : (int64_t)wins.size();
}

int64_t orc_sliding_hog_svm(const orc_pyramid* p_, const orc_svm* svm_, int pw, int ph, int stepX, int stepY, int bins,
                            int cell, int block, int interpolate, int signedAndUnsigned, orc_det* out, int64_t cap,
                            double* all_dist, float* feat_out, int64_t feat_cap_windows) {
    return orc_sliding_hog_svm_sample(p_, svm_, pw, ph, stepX, stepY, bins, cell, block, interpolate, signedAndUnsigned, out, cap, all_dist,
                                      feat_out, feat_cap_windows, 0, 1, nullptr);
}
}
