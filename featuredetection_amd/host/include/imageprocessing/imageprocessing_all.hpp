// imageprocessing/*.hpp of the reference, re-implemented on top of the C ABI (include/fd_hip.h).
// Class names, namespaces, constructor and method signatures follow the reference headers cited per
// class; the work is done by the HIP kernels behind fd_hip.h.  Filters that the GPU path fuses into a
// kernel (GradientFilter, GradientBinningFilter, HogFilter, LbpFilter as layer filter, HistEq64Filter as
// patch filter) are recognised by type when they are added to a pyramid / extractor; every filter also has
// its stand-alone ImageFilter::applyTo(const Mat&) form (one kernel launch per Mat: for composition and
// tests, not for the sliding-window hot loop).
#pragma once
#include <memory>
#include <string>
#include <utility>
#include <vector>
#include "fdcompat/cv.hpp"
#include "fdcompat/runtime.hpp"

namespace imageprocessing {

// Version.hpp:18-46 -- (instance id, counter) pair used to avoid rebuilding pyramids
class Version {
public:
    Version() : instance(-1), counter(0) {}
    static Version fresh() { Version v; v.instance = nextInstance()++; return v; }
    bool operator==(const Version& o) const { return instance == o.instance && counter == o.counter; }
    bool operator!=(const Version& o) const { return !(*this == o); }
    Version& operator++() { ++counter; return *this; }
private:
    static int& nextInstance() { static int n = 0; return n; }
    int instance, counter;
};

// VersionedImage.hpp:19-67
class VersionedImage {
public:
    VersionedImage() : data(), version(Version::fresh()) {}
    explicit VersionedImage(const cv::Mat& d) : data(d), version(Version::fresh()) {}
    cv::Mat& getData() { return data; }
    const cv::Mat& getData() const { return data; }
    void setData(const cv::Mat& d) { data = d; ++version; }
    Version getVersion() const { return version; }
private:
    cv::Mat data;
    Version version;
};

// ImageFilter.hpp:18-57
class ImageFilter {
public:
    virtual ~ImageFilter() {}
    cv::Mat applyTo(const cv::Mat& image) const { cv::Mat filtered; return applyTo(image, filtered); }
    virtual cv::Mat applyTo(const cv::Mat& image, cv::Mat& filtered) const = 0;
    virtual void applyInPlace(cv::Mat& image) const { image = applyTo(image); }
};

// ChainedFilter.cpp:36-50
class ChainedFilter : public ImageFilter {
public:
    using ImageFilter::applyTo;
    void add(std::shared_ptr<ImageFilter> filter) { filters.push_back(filter); }
    const std::vector<std::shared_ptr<ImageFilter>>& getFilters() const { return filters; }
    cv::Mat applyTo(const cv::Mat& image, cv::Mat& filtered) const override {
        if (filters.empty()) { image.copyTo(filtered); return filtered; }
        filtered = image;
        for (const auto& f : filters) { cv::Mat tmp; f->applyTo(filtered, tmp); filtered = tmp; }
        return filtered;
    }
private:
    std::vector<std::shared_ptr<ImageFilter>> filters;
};

// GrayscaleFilter.cpp:18-24 (image filter of the pyramid: fused into fd_pyramid_update)
class GrayscaleFilter : public ImageFilter {
public:
    using ImageFilter::applyTo;
    cv::Mat applyTo(const cv::Mat& image, cv::Mat& filtered) const override;
};

// HistEq64Filter.cpp:32-125 (patch filter: fused into the WVM kernel; stand-alone via fd_histeq64_batch)
class HistEq64Filter : public ImageFilter {
public:
    using ImageFilter::applyTo;
    cv::Mat applyTo(const cv::Mat& image, cv::Mat& filtered) const override;
};

// The "whi" patch filter chain of ffpDetectApp.cpp:449-454.  Added in this order to a (Filtering / Direct) pyramid feature
// extractor the four filters run as one fused kernel (fd_extract_whi / fd_detect_whi_svm); each also has its per-Mat form
// (fd_whitening_batch, fd_equalize_hist_batch, fd_convert_batch, fd_unit_norm_batch).
// WhiteningFilter.hpp:31 / WhiteningFilter.cpp:18-81
class WhiteningFilter : public ImageFilter {
public:
    using ImageFilter::applyTo;
    explicit WhiteningFilter(float alpha = 1, float cutoffFrequency = 0.390625f) : alpha(alpha), cutoffFrequency(cutoffFrequency) {}
    cv::Mat applyTo(const cv::Mat& image, cv::Mat& filtered) const override;
    float alpha, cutoffFrequency;
};
// HistogramEqualizationFilter.cpp (cv::equalizeHist)
class HistogramEqualizationFilter : public ImageFilter {
public:
    using ImageFilter::applyTo;
    cv::Mat applyTo(const cv::Mat& image, cv::Mat& filtered) const override;
};
// ConversionFilter.hpp: convertTo(type, alpha, beta)
class ConversionFilter : public ImageFilter {
public:
    using ImageFilter::applyTo;
    explicit ConversionFilter(int type, double alpha = 1, double beta = 0) : type(type), alpha(alpha), beta(beta) {}
    cv::Mat applyTo(const cv::Mat& image, cv::Mat& filtered) const override;
    int type;
    double alpha, beta;
};
// UnitNormFilter.hpp / UnitNormFilter.cpp (eps 1e-4)
class UnitNormFilter : public ImageFilter {
public:
    using ImageFilter::applyTo;
    explicit UnitNormFilter(int normType = cv::NORM_L2) : normType(normType) {}
    cv::Mat applyTo(const cv::Mat& image, cv::Mat& filtered) const override;
    int normType;
};
// ZeroMeanUnitVarianceFilter.cpp:21-34 (ffpDetectApp.cpp:77 includes it; host only: cv::meanStdDev in double, then (x - mean) / deviation
// on the CV_32F copy, all zeros for a constant image)
class ZeroMeanUnitVarianceFilter : public ImageFilter {
public:
    using ImageFilter::applyTo;
    ZeroMeanUnitVarianceFilter() {}
    cv::Mat applyTo(const cv::Mat& image, cv::Mat& filtered) const override;
};
// ReshapingFilter.hpp: Mat::reshape(channels, rows) (ffpDetectApp.cpp:465-467 turns patches into row vectors); the fused
// kernels work on flat vectors, so inside a fused chain it is a no-op
class ReshapingFilter : public ImageFilter {
public:
    using ImageFilter::applyTo;
    explicit ReshapingFilter(int rows, int channels = 0) : rows(rows), channels(channels) {}
    cv::Mat applyTo(const cv::Mat& image, cv::Mat& filtered) const override;
    int rows, channels;
};

// GreyWorldNormalizationFilter.cpp:20-71
class GreyWorldNormalizationFilter : public ImageFilter {
public:
    using ImageFilter::applyTo;
    cv::Mat applyTo(const cv::Mat& image, cv::Mat& filtered) const override;
};

// GradientFilter.cpp:16-59 (fused as a layer filter; applyTo: CV_8UC1 -> CV_8UC2 via fd_gradient_image)
class GradientFilter : public ImageFilter {
public:
    using ImageFilter::applyTo;
    explicit GradientFilter(int kernelSize, int blurKernelSize = 0);
    cv::Mat applyTo(const cv::Mat& image, cv::Mat& filtered) const override;
    int kernelSize, blurKernelSize;
};

// GradientBinningFilter.cpp:18-93 (fused as a layer filter; applyTo: CV_8UC2 -> CV_8UC2 / CV_8UC4 via fd_gradient_binning_image)
class GradientBinningFilter : public ImageFilter {
public:
    using ImageFilter::applyTo;
    explicit GradientBinningFilter(unsigned int bins, bool signedGradients = false, bool interpolate = false);
    cv::Mat applyTo(const cv::Mat& image, cv::Mat& filtered) const override;
    unsigned int getBinCount() const { return bins; }
    unsigned int bins;
    bool signedGradients, interpolate;
};

// LbpFilter.hpp (fused as a layer filter; applyTo via fd_lbp_image)
class LbpFilter : public ImageFilter {
public:
    using ImageFilter::applyTo;
    enum class Type { LBP8, LBP8_UNIFORM, LBP4, LBP4_ROTATED };
    explicit LbpFilter(Type type = Type::LBP8) : type(type) {}
    cv::Mat applyTo(const cv::Mat& image, cv::Mat& filtered) const override;
    unsigned int getBinCount() const;
    Type type;
};

// HistogramFilter.hpp:24-33 -- base of the histogram patch filters.  On this backend they run fused on the
// pyramid's bin-image layers (k_hog_tile / k_hist_features); applyTo(Mat) runs the same kernel on one bin-image
// patch (fd_hist_patch_batch) and returns the 1 x F CV_32F feature vector.
class HistogramFilter : public ImageFilter {
public:
    enum class Normalization { NONE, L2NORM, L2HYS, L1NORM, L1SQRT };
    explicit HistogramFilter(Normalization normalization) : normalization(normalization) {}
    Normalization normalization;
};

// HogFilter.hpp / HogFilter.cpp:16-57
class HogFilter : public HistogramFilter {
public:
    using ImageFilter::applyTo;
    explicit HogFilter(int binCount, int cellSize = 5, int blockSize = 2, bool interpolate = false, bool signedAndUnsigned = false);
    HogFilter(int binCount, int cellWidth, int cellHeight, int blockWidth, int blockHeight, bool interpolate = false,
              bool signedAndUnsigned = false);
    cv::Mat applyTo(const cv::Mat& image, cv::Mat& filtered) const override;
    int binCount, cellWidth, cellHeight, blockWidth, blockHeight;
    bool interpolate, signedAndUnsigned;
};

// SpatialHistogramFilter.hpp:38-54 / SpatialHistogramFilter.cpp:16-54
class SpatialHistogramFilter : public HistogramFilter {
public:
    using ImageFilter::applyTo;
    SpatialHistogramFilter(int binCount, int cellSize, int blockSize, bool interpolate, bool concatenate = false,
                           Normalization normalization = Normalization::NONE);
    SpatialHistogramFilter(int binCount, int cellWidth, int cellHeight, int blockWidth, int blockHeight, bool interpolate,
                           bool concatenate = false, Normalization normalization = Normalization::NONE);
    cv::Mat applyTo(const cv::Mat& image, cv::Mat& filtered) const override;
    int binCount, cellWidth, cellHeight, blockWidth, blockHeight;
    bool interpolate, concatenate;
};

// PyramidHogFilter.hpp:35 / PyramidHogFilter.cpp:15-31
class PyramidHogFilter : public HistogramFilter {
public:
    using ImageFilter::applyTo;
    PyramidHogFilter(int binCount, int levelCount, bool interpolate = false, bool signedAndUnsigned = false);
    cv::Mat applyTo(const cv::Mat& image, cv::Mat& filtered) const override;
    int binCount, levelCount;
    bool interpolate, signedAndUnsigned;
};

// SpatialPyramidHistogramFilter.hpp:36 / SpatialPyramidHistogramFilter.cpp:21-35
class SpatialPyramidHistogramFilter : public HistogramFilter {
public:
    using ImageFilter::applyTo;
    SpatialPyramidHistogramFilter(int binCount, int levelCount, bool interpolate = false, Normalization normalization = Normalization::NONE);
    cv::Mat applyTo(const cv::Mat& image, cv::Mat& filtered) const override;
    int binCount, levelCount;
    bool interpolate;
};

// ImagePyramidLayer.hpp:34-163
class ImagePyramidLayer {
public:
    ImagePyramidLayer(int index, double scale, double scaleX, double scaleY, const cv::Mat& image)
        : index(index), scale(scale), scaleX(scaleX), scaleY(scaleY), image(image) {}
    int getScaled(int value) const { return cv::cvRound(value * scale); }
    int getOriginal(int value) const { return cv::cvRound(value / scale); }
    cv::Size getSize() const { return cv::Size(image.cols, image.rows); }
    int getIndex() const { return index; }
    double getScaleFactor() const { return scale; }
    const cv::Mat& getScaledImage() const { return image; }
private:
    int index;
    double scale, scaleX, scaleY;
    cv::Mat image;
};

// ImagePyramid.hpp / ImagePyramid.cpp:67-92,116-128,148-198,300-328 -- backed by fd_pyramid
class ImagePyramid {
public:
    ImagePyramid(size_t octaveLayerCount, double minScaleFactor, double maxScaleFactor = 1);
    ImagePyramid(double incrementalScaleFactor, double minScaleFactor, double maxScaleFactor = 1);
    // ImagePyramid.cpp:100-104: same layers as `pyramid`, restricted to scale factors within [min, max]; nothing is rebuilt,
    // the layers of the source (and its device arena) are shared (createLayers(const ImagePyramid&), :200-235, without the
    // approximated in-between layers of createApproximated)
    ImagePyramid(std::shared_ptr<ImagePyramid> pyramid, double minScaleFactor, double maxScaleFactor = 1);
    ~ImagePyramid();
    ImagePyramid(const ImagePyramid&) = delete;
    ImagePyramid& operator=(const ImagePyramid&) = delete;
    void addImageFilter(const std::shared_ptr<ImageFilter>& filter);   // GrayscaleFilter
    void addLayerFilter(const std::shared_ptr<ImageFilter>& filter);   // GradientFilter, GradientBinningFilter, LbpFilter
    void update(const cv::Mat& image);
    void update(const std::shared_ptr<VersionedImage>& image);
    void setSource(const cv::Mat& image) { setSource(std::make_shared<VersionedImage>(image)); }
    void setSource(const std::shared_ptr<VersionedImage>& image);     // ImagePyramid.cpp:132-139
    void setSource(const std::shared_ptr<ImagePyramid>& pyramid);     // ImagePyramid.cpp:141-144
    void update();                                                     // ImagePyramid.cpp:146-168
    const std::vector<std::shared_ptr<ImagePyramidLayer>>& getLayers() const;  // downloads the layers on first use
    const std::shared_ptr<ImagePyramidLayer> getLayer(int index) const;
    double getMinScaleFactor() const { return minScaleFactor; }
    double getMaxScaleFactor() const { return maxScaleFactor; }
    double getIncrementalScaleFactor() const;
    cv::Size getImageSize() const { return imageSize; }
    std::vector<std::pair<int, double>> getLayerScales() const;
    std::vector<cv::Size> getLayerSizes() const;
    fd_pyramid* native() const { return sourcePyramid ? sourcePyramid->native() : handle; }
    std::shared_ptr<ImagePyramid> getSourcePyramid() const { return sourcePyramid; }
    // a second native pyramid with this pyramid's parameters that holds `frames` equally sized frames at once (backend extension
    // behind FiveStageSlidingWindowDetector::detectFrames); NULL for pyramids on a source pyramid or with layer filters
    fd_pyramid* createFramesPyramid(int frames) const;
    // Layer sub-range / region of interest of the extraction or detection call that follows (fd_pyramid_select), intersected
    // with the scale range of a pyramid that views another one; reset when the guard goes out of scope.
    class Selection {
    public:
        // viewFirst / viewLast: layer index range of a pyramid built on another one (-1: the pyramid owns its layers)
        Selection(fd_pyramid* h, int first, int last, int step, const cv::Rect* roi, int viewFirst = -1, int viewLast = -1);
        ~Selection();
        Selection(Selection&& o) : handle(o.handle) { o.handle = nullptr; }
        Selection(const Selection&) = delete;
    private:
        fd_pyramid* handle;
    };
    Selection select(int firstLayer = -1, int lastLayer = -1, int stepLayer = 1, const cv::Rect* roi = nullptr) const;
    static long buildCount();   // pyramids actually (re)built so far: the VersionedImage mechanism at work (tests)
private:
    void applyLayerFilterConfig();
    void viewRange(int& first, int& last) const;   // layer indices of the source inside [minScaleFactor, maxScaleFactor]
    fd_pyramid* handle;
    std::shared_ptr<ImagePyramid> sourcePyramid;
    std::shared_ptr<VersionedImage> sourceImage;
    double minScaleFactor, maxScaleFactor;
    size_t ctorOctaveLayers = 0;          // constructor arguments (one of the two forms)
    double ctorIncremental = 0;
    cv::Size imageSize;
    Version version;
    std::shared_ptr<GradientFilter> gradient;
    std::shared_ptr<GradientBinningFilter> binning;
    std::shared_ptr<LbpFilter> lbp;
    mutable std::vector<std::shared_ptr<ImagePyramidLayer>> layers;
    mutable bool layersValid;
};

// Patch.hpp:28-243
class Patch {
public:
    Patch() : center(0, 0), size(0, 0), data() {}
    Patch(int x, int y, int width, int height, const cv::Mat& data) : center(x, y), size(width, height), data(data) {}
    Patch(const Patch& other) : center(other.center), size(other.size), data(other.data.clone()) {}
    bool operator==(const Patch& o) const {
        return center.x == o.center.x && center.y == o.center.y && size.width == o.size.width && size.height == o.size.height;
    }
    cv::Rect getBounds() const { return cv::Rect(center.x - size.width / 2, center.y - size.height / 2, size.width, size.height); }
    int getX() const { return center.x; }
    int getY() const { return center.y; }
    int getWidth() const { return size.width; }
    int getHeight() const { return size.height; }
    cv::Mat& getData() { return data; }
    const cv::Mat& getData() const { return data; }
private:
    cv::Point center;
    cv::Size size;
    cv::Mat data;
};

// FeatureExtractor.hpp:22-55
class FeatureExtractor {
public:
    virtual ~FeatureExtractor() {}
    void update(const cv::Mat& image) { update(std::make_shared<VersionedImage>(image)); }
    virtual void update(std::shared_ptr<VersionedImage> image) = 0;
    virtual std::shared_ptr<Patch> extract(int x, int y, int width, int height) const = 0;
};

// PyramidFeatureExtractor.hpp:21-119
class PyramidFeatureExtractor : public FeatureExtractor {
public:
    using FeatureExtractor::update;
    using FeatureExtractor::extract;
    virtual std::vector<std::shared_ptr<Patch>> extract(int stepX, int stepY, cv::Rect roi = cv::Rect(), int firstLayer = -1,
                                                        int lastLayer = -1, int stepLayer = 1) const = 0;
    virtual std::shared_ptr<Patch> extract(int layer, int x, int y) const = 0;
    virtual int getLayerIndex(int width, int height) const = 0;
    virtual double getMinScaleFactor() const = 0;
    virtual double getMaxScaleFactor() const = 0;
    virtual double getIncrementalScaleFactor() const = 0;
    virtual cv::Size getPatchSize() const = 0;
    virtual cv::Size getImageSize() const = 0;
    virtual std::vector<std::pair<int, double>> getLayerScales() const = 0;
    virtual std::vector<cv::Size> getLayerSizes() const = 0;
    virtual std::vector<cv::Size> getPatchSizes() const = 0;
};

// DirectPyramidFeatureExtractor.hpp / .cpp:27-229
class DirectPyramidFeatureExtractor : public PyramidFeatureExtractor {
public:
    using PyramidFeatureExtractor::update;
    using PyramidFeatureExtractor::extract;
    DirectPyramidFeatureExtractor(std::shared_ptr<ImagePyramid> pyramid, int width, int height);
    void addImageFilter(std::shared_ptr<ImageFilter> filter) { pyramid->addImageFilter(filter); }
    void addLayerFilter(std::shared_ptr<ImageFilter> filter) { pyramid->addLayerFilter(filter); }
    void addPatchFilter(std::shared_ptr<ImageFilter> filter);          // HistEq64Filter or one HistogramFilter
    void update(std::shared_ptr<VersionedImage> image) override { pyramid->update(image); }
    std::shared_ptr<Patch> extract(int x, int y, int width, int height) const override;
    std::vector<std::shared_ptr<Patch>> extract(int stepX, int stepY, cv::Rect roi = cv::Rect(), int firstLayer = -1, int lastLayer = -1,
                                                int stepLayer = 1) const override;
    std::shared_ptr<Patch> extract(int layer, int x, int y) const override;
    int getLayerIndex(int width, int height) const override;
    double getMinScaleFactor() const override { return pyramid->getMinScaleFactor(); }
    double getMaxScaleFactor() const override { return pyramid->getMaxScaleFactor(); }
    double getIncrementalScaleFactor() const override { return pyramid->getIncrementalScaleFactor(); }
    cv::Size getPatchSize() const override { return cv::Size(patchWidth, patchHeight); }
    cv::Size getImageSize() const override { return pyramid->getImageSize(); }
    std::vector<std::pair<int, double>> getLayerScales() const override { return pyramid->getLayerScales(); }
    std::vector<cv::Size> getLayerSizes() const override { return pyramid->getLayerSizes(); }
    std::vector<cv::Size> getPatchSizes() const override;
    std::shared_ptr<ImagePyramid> getPyramid() { return pyramid; }
    const std::shared_ptr<ImagePyramid> getPyramid() const { return pyramid; }
    int getPatchWidth() const { return patchWidth; }
    int getPatchHeight() const { return patchHeight; }
    // which fused GPU path the patch filter chain maps to
    bool hasHistEq64() const { return (bool)histeq; }
    std::shared_ptr<HogFilter> getHogFilter() const { return hog; }   // non-interpolating square HogFilter: k_hog_tile path
    std::shared_ptr<HistogramFilter> getHistogramFilter() const { return hist; }
    // complete whi chain (WhiteningFilter, HistogramEqualizationFilter, ConversionFilter(CV_32F, 1/127.5, -1), UnitNormFilter(L2))
    std::shared_ptr<WhiteningFilter> getWhiChain() const { return whiStage == 4 ? whitening : nullptr; }
    // u8 feature space (0 gray, 1 hq64 = HistEq64Filter, 2 histeq = HistogramEqualizationFilter) followed by a
    // ConversionFilter(CV_32F, scale, shift): the input of a ProbabilisticRvmClassifier (fd_detect_rvm)
    std::shared_ptr<ConversionFilter> getConversion() const { return whiStage == 0 ? conversion : nullptr; }
    int getU8FeatureSpace() const { return histeq ? 1 : (equalization ? 2 : 0); }
    bool hasPatchFilters() const { return histeq || hist || whitening || equalization || conversion || reshaping; }
private:
    std::shared_ptr<Patch> extractFromLayer(const ImagePyramidLayer& layer, cv::Rect bounds) const;
    std::shared_ptr<ImagePyramid> pyramid;
    int patchWidth, patchHeight;
    std::shared_ptr<HistEq64Filter> histeq;
    std::shared_ptr<HogFilter> hog;
    std::shared_ptr<HistogramFilter> hist;
    std::shared_ptr<WhiteningFilter> whitening;
    std::shared_ptr<HistogramEqualizationFilter> equalization;
    std::shared_ptr<ConversionFilter> conversion;
    std::shared_ptr<ReshapingFilter> reshaping;
    int whiStage = 0;   // number of whi chain filters added so far (in order)
    std::shared_ptr<ChainedFilter> chain = std::make_shared<ChainedFilter>();   // every patch filter in the order added: the per-Mat form of extract(x, y, w, h) / extract(layer, x, y)
};

// FilteringPyramidFeatureExtractor.hpp:20-90 -- applies additional patch filters to the patches of another pyramid feature
// extractor (ffpDetectApp.cpp:445).  When the underlying extractor is a DirectPyramidFeatureExtractor without patch filters of
// its own and the added chain is one the kernels fuse (hq64, histeq, whi, histogram filters, + ConversionFilter /
// ReshapingFilter), the detectors run the fused GPU path (getFusedExtractor()); otherwise every patch goes through the filters'
// per-Mat applyTo, exactly like the reference.
class FilteringPyramidFeatureExtractor : public PyramidFeatureExtractor {
public:
    using PyramidFeatureExtractor::update;
    using PyramidFeatureExtractor::extract;
    explicit FilteringPyramidFeatureExtractor(std::shared_ptr<PyramidFeatureExtractor> extractor);
    void addPatchFilter(std::shared_ptr<ImageFilter> filter);
    void update(std::shared_ptr<VersionedImage> image) override { extractor->update(image); }
    std::shared_ptr<Patch> extract(int x, int y, int width, int height) const override;
    std::vector<std::shared_ptr<Patch>> extract(int stepX, int stepY, cv::Rect roi = cv::Rect(), int firstLayer = -1, int lastLayer = -1,
                                                int stepLayer = 1) const override;
    std::shared_ptr<Patch> extract(int layer, int x, int y) const override;
    int getLayerIndex(int width, int height) const override { return extractor->getLayerIndex(width, height); }
    double getMinScaleFactor() const override { return extractor->getMinScaleFactor(); }
    double getMaxScaleFactor() const override { return extractor->getMaxScaleFactor(); }
    double getIncrementalScaleFactor() const override { return extractor->getIncrementalScaleFactor(); }
    cv::Size getPatchSize() const override { return extractor->getPatchSize(); }
    cv::Size getImageSize() const override { return extractor->getImageSize(); }
    std::vector<std::pair<int, double>> getLayerScales() const override { return extractor->getLayerScales(); }
    std::vector<cv::Size> getLayerSizes() const override { return extractor->getLayerSizes(); }
    std::vector<cv::Size> getPatchSizes() const override { return extractor->getPatchSizes(); }
    std::shared_ptr<PyramidFeatureExtractor> getExtractor() const { return extractor; }
    // the equivalent DirectPyramidFeatureExtractor (same pyramid, same patch size, the whole chain as its patch filters) when the
    // chain maps to a fused kernel, else null
    std::shared_ptr<DirectPyramidFeatureExtractor> getFusedExtractor() const { return fused; }
private:
    std::shared_ptr<PyramidFeatureExtractor> extractor;
    std::shared_ptr<ChainedFilter> patchFilter;
    std::shared_ptr<DirectPyramidFeatureExtractor> fused;
};

// FilteringFeatureExtractor.hpp:20-62 (ffpDetectApp.cpp:74 includes it): any FeatureExtractor plus patch filters applied per Mat
class FilteringFeatureExtractor : public FeatureExtractor {
public:
    using FeatureExtractor::update;
    explicit FilteringFeatureExtractor(std::shared_ptr<FeatureExtractor> extractor) : extractor(extractor), patchFilter(std::make_shared<ChainedFilter>()) {}
    void addPatchFilter(std::shared_ptr<ImageFilter> filter) { patchFilter->add(filter); }
    void update(std::shared_ptr<VersionedImage> image) override { extractor->update(image); }
    std::shared_ptr<Patch> extract(int x, int y, int width, int height) const override {
        std::shared_ptr<Patch> patch = extractor->extract(x, y, width, height);
        if (patch) patchFilter->applyInPlace(patch->getData());
        return patch;
    }
private:
    std::shared_ptr<FeatureExtractor> extractor;
    std::shared_ptr<ChainedFilter> patchFilter;
};

// filtering/FhogFilter.hpp:55-56 / FhogFilter.cpp:20-72 (the cell descriptors of the AggregatedFeaturesDetector family)
namespace filtering {
class FhogFilter : public ImageFilter {
public:
    using ImageFilter::applyTo;
    explicit FhogFilter(int cellSize = 8, int unsignedBinCount = 9, bool interpolateBins = false, bool interpolateCells = true, float alpha = 0.2f);
    cv::Mat applyTo(const cv::Mat& image, cv::Mat& filtered) const override;   // CV_8UC1 -> rows x cols x (3B+4) CV_32F (stored as rows x cols*(3B+4))
    int cellSize, unsignedBinCount;
    bool interpolateBins, interpolateCells;
    float alpha;
};
}  // namespace filtering

}  // namespace imageprocessing
