// host_selftest_app -- exercises the parts of the reference-shaped host layer that have no app of their own, and writes raw
// results for tests/test_gpu_host_apps.py to compare with the oracle:
//   * 15 extractors on ImagePyramids built on ONE source pyramid, all updated with the same VersionedImage: the source is
//     built once (Version mechanism, ImagePyramid.cpp:100-104,146-168), every view exposes the layers of its scale range
//   * DirectPyramidFeatureExtractor::extract(stepX, stepY, roi, firstLayer, lastLayer, stepLayer) on a HOG chain
//   * FilteringPyramidFeatureExtractor: fused chain vs per-Mat composition of the same filters
//   * the stand-alone ImageFilter::applyTo(const Mat&) of every filter
// usage: host_selftest_app <image.pgm> <out-dir>
#include <cstdio>
#include <fstream>
#include <iostream>
#include "detection/detection_all.hpp"
#include "classification/RvmClassifier.hpp"
#include "classification/ProbabilisticRvmClassifier.hpp"
#include "imageprocessing/FilteringFeatureExtractor.hpp"
#include "imageprocessing/ZeroMeanUnitVarianceFilter.hpp"

using namespace imageprocessing;
using std::make_shared;
using std::shared_ptr;
using std::string;

static cv::Mat read_pgm(const string& path) {
    std::ifstream f(path.c_str(), std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("cannot open image " + path);
    string magic;
    int w, h, maxv;
    f >> magic >> w >> h >> maxv;
    f.get();
    if (magic != "P5" || maxv != 255) throw std::runtime_error("binary PGM expected");
    cv::Mat img(h, w, CV_8UC1);
    f.read((char*)img.data, (size_t)w * h);
    return img;
}
static void dump(const string& path, const void* p, size_t bytes) {
    std::ofstream f(path.c_str(), std::ios::binary);
    f.write((const char*)p, (std::streamsize)bytes);
}
static void dumpMat(const string& path, const cv::Mat& m) {
    cv::Mat c = m.isContinuous() ? m : m.clone();
    dump(path, c.data, (size_t)c.rows * c.cols * c.elemSize());
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s <image.pgm> <out-dir>\n", argv[0]); return 2; }
    try {
        const string out = string(argv[2]) + "/";
        cv::Mat img = read_pgm(argv[1]);
        auto image = make_shared<VersionedImage>(img);

        // ---- 1. one source pyramid, 15 views with the scale ranges / patch sizes of ffpDetectApp's detectors
        auto source = make_shared<ImagePyramid>(0.9, 0.09, 0.7);
        source->addImageFilter(make_shared<GrayscaleFilter>());
        const double ranges[3][2] = {{0.09, 0.25}, {0.5, 0.7}, {0.3, 0.45}};
        std::vector<shared_ptr<DirectPyramidFeatureExtractor>> extractors;
        for (int i = 0; i < 15; ++i) {
            auto view = make_shared<ImagePyramid>(source, ranges[i % 3][0], ranges[i % 3][1]);
            extractors.push_back(make_shared<DirectPyramidFeatureExtractor>(view, i % 2 ? 24 : 20, i % 2 ? 24 : 20));
        }
        const long before = ImagePyramid::buildCount();
        for (auto& e : extractors) e->update(image);
        for (auto& e : extractors) e->update(image);   // same version: nothing happens
        std::printf("builds %ld\n", ImagePyramid::buildCount() - before);
        std::printf("source_layers %zu\n", source->getLayers().size());
        for (int i = 0; i < 3; ++i) {
            auto scales = extractors[i]->getLayerScales();
            std::printf("view %d layers %zu first %d last %d\n", i, scales.size(), scales.empty() ? -1 : scales.front().first, scales.empty() ? -1 : scales.back().first);
            auto patches = extractors[i]->extract(4, 4);
            std::printf("view %d patches %zu\n", i, patches.size());
            std::vector<int32_t> geo;
            for (auto& p : patches) { geo.push_back(p->getX()); geo.push_back(p->getY()); geo.push_back(p->getWidth()); geo.push_back(p->getHeight()); }
            dump(out + "view" + std::to_string(i) + "_geo.bin", geo.data(), geo.size() * 4);
        }
        // the layer step of extract() counts from the FIRST LAYER OF THE VIEW (pyramid->getLayers().begin(),
        // DirectPyramidFeatureExtractor.cpp:94-99), wherever that layer sits in the source pyramid
        for (int i = 0; i < 3; ++i) {
            auto patches = extractors[i]->extract(4, 4, cv::Rect(), -1, -1, 2);
            std::vector<int32_t> geo;
            for (auto& p : patches) { geo.push_back(p->getX()); geo.push_back(p->getY()); geo.push_back(p->getWidth()); geo.push_back(p->getHeight()); }
            dump(out + "view" + std::to_string(i) + "_step2_geo.bin", geo.data(), geo.size() * 4);
        }
        image->setData(img);   // new version: exactly one rebuild, whoever asks first
        for (auto& e : extractors) e->update(image);
        std::printf("builds_after_new_version %ld\n", ImagePyramid::buildCount() - before);

        // ---- 2. layer sub-range + ROI on a HOG chain (DirectPyramidFeatureExtractor.cpp:75-123)
        auto hogPyr = make_shared<ImagePyramid>((size_t)3, 0.2, 0.8);
        hogPyr->addLayerFilter(make_shared<GradientFilter>(1));
        hogPyr->addLayerFilter(make_shared<GradientBinningFilter>(9));
        auto hogEx = make_shared<DirectPyramidFeatureExtractor>(hogPyr, 20, 20);
        hogEx->addPatchFilter(make_shared<HogFilter>(9, 5, 2));
        hogEx->update(image);
        auto scales = hogEx->getLayerScales();
        const int first = scales[1].first, last = scales[scales.size() - 2].first;
        auto sub = hogEx->extract(3, 3, cv::Rect(40, 30, 200, 160), first, last, 2);
        std::printf("hog_sub first %d last %d patches %zu\n", first, last, sub.size());
        {
            std::vector<int32_t> geo;
            std::vector<float> feat;
            for (auto& p : sub) {
                geo.push_back(p->getX()); geo.push_back(p->getY()); geo.push_back(p->getWidth()); geo.push_back(p->getHeight());
                const float* f = p->getData().ptr<float>(0);
                feat.insert(feat.end(), f, f + p->getData().cols);
            }
            dump(out + "hog_sub_geo.bin", geo.data(), geo.size() * 4);
            dump(out + "hog_sub_feat.bin", feat.data(), feat.size() * 4);
        }

        // ---- 3. FilteringPyramidFeatureExtractor: fused whi chain vs the same filters applied per Mat
        auto grayPyr = make_shared<ImagePyramid>((size_t)2, 0.3, 0.5);
        auto direct = make_shared<DirectPyramidFeatureExtractor>(grayPyr, 20, 20);
        auto filtering = make_shared<FilteringPyramidFeatureExtractor>(direct);
        filtering->addPatchFilter(make_shared<WhiteningFilter>());
        filtering->addPatchFilter(make_shared<HistogramEqualizationFilter>());
        filtering->addPatchFilter(make_shared<ConversionFilter>(CV_32F, 1.0 / 127.5, -1.0));
        filtering->addPatchFilter(make_shared<UnitNormFilter>(cv::NORM_L2));
        filtering->update(image);
        auto fusedPatches = filtering->extract(7, 7);
        auto rawPatches = direct->extract(7, 7);
        ChainedFilter chain;
        chain.add(make_shared<WhiteningFilter>());
        chain.add(make_shared<HistogramEqualizationFilter>());
        chain.add(make_shared<ConversionFilter>(CV_32F, 1.0 / 127.5, -1.0));
        chain.add(make_shared<UnitNormFilter>(cv::NORM_L2));
        size_t mismatches = 0;
        const size_t ncheck = std::min<size_t>(rawPatches.size(), 40);
        for (size_t i = 0; i < ncheck; ++i) {
            cv::Mat m = chain.applyTo(rawPatches[i]->getData());
            if (std::memcmp(m.data, fusedPatches[i]->getData().data, sizeof(float) * 400) != 0) ++mismatches;
        }
        std::printf("filtering fused %d patches %zu per_mat_checked %zu mismatches %zu\n", filtering->getFusedExtractor() ? 1 : 0, fusedPatches.size(), ncheck, mismatches);
        // a chain the kernels do not fuse falls back to per-Mat composition (LbpFilter as a patch filter)
        auto generic = make_shared<FilteringPyramidFeatureExtractor>(direct);
        generic->addPatchFilter(make_shared<LbpFilter>(LbpFilter::Type::LBP8));
        auto lbpPatches = generic->extract(9, 9);
        std::printf("generic fused %d patches %zu\n", generic->getFusedExtractor() ? 1 : 0, lbpPatches.size());
        if (!lbpPatches.empty()) dumpMat(out + "generic_lbp_patch0.bin", lbpPatches[0]->getData());
        if (!rawPatches.empty()) dumpMat(out + "raw_patch0.bin", direct->extract(9, 9)[0]->getData());

        // ---- 4. stand-alone applyTo of every filter on a crop
        cv::Mat crop = cv::Mat(img, cv::Rect(16, 24, 64, 48)).clone();
        dumpMat(out + "crop.bin", crop);
        cv::Mat grad = GradientFilter(3).applyTo(crop);
        dumpMat(out + "grad.bin", grad);
        cv::Mat bins = GradientBinningFilter(9, false, true).applyTo(grad);
        dumpMat(out + "bins.bin", bins);
        dumpMat(out + "lbp.bin", LbpFilter(LbpFilter::Type::LBP8_UNIFORM).applyTo(crop));
        cv::Mat p20 = cv::Mat(bins, cv::Rect(4, 4, 20, 20)).clone();
        dumpMat(out + "hog.bin", HogFilter(9, 5, 2, true).applyTo(p20));
        dumpMat(out + "sphist.bin", SpatialHistogramFilter(9, 5, 2, true, false, HistogramFilter::Normalization::L2HYS).applyTo(p20));
        dumpMat(out + "phog.bin", PyramidHogFilter(9, 2, true).applyTo(p20));
        dumpMat(out + "sppyr.bin", SpatialPyramidHistogramFilter(9, 2, true, HistogramFilter::Normalization::L1NORM).applyTo(p20));
        cv::Mat g20 = cv::Mat(crop, cv::Rect(10, 10, 20, 20)).clone();
        dumpMat(out + "whitened.bin", WhiteningFilter().applyTo(g20));
        dumpMat(out + "converted.bin", ConversionFilter(CV_32F, 1.0 / 255.0, 0.0).applyTo(g20));
        dumpMat(out + "unitnorm.bin", UnitNormFilter(cv::NORM_L2).applyTo(g20));
        dumpMat(out + "zmuv.bin", ZeroMeanUnitVarianceFilter().applyTo(g20));
        {   // FilteringFeatureExtractor (FilteringFeatureExtractor.hpp:20-62): single-patch extraction + a patch filter applied per Mat
            auto ffe = make_shared<FilteringFeatureExtractor>(direct);
            ffe->addPatchFilter(make_shared<ZeroMeanUnitVarianceFilter>());
            ffe->update(image);
            auto pf = ffe->extract(200, 150, 60, 60), pr = direct->extract(200, 150, 60, 60);
            if (pf && pr) {
                dumpMat(out + "ffe_patch.bin", pf->getData());
                dumpMat(out + "ffe_raw.bin", pr->getData());
            }
        }
        cv::Mat row = ReshapingFilter(1).applyTo(g20);
        std::printf("reshaped %d x %d\n", row.rows, row.cols);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
