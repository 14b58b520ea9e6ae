// detection/*.hpp of the reference on top of the C ABI (include/fd_hip.h).
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "classification/classification_all.hpp"
#include "imageprocessing/imageprocessing_all.hpp"

namespace detection {

// ClassifiedPatch.hpp:19-97
class ClassifiedPatch {
public:
    ClassifiedPatch(std::shared_ptr<imageprocessing::Patch> patch, bool positive, double probability = 0.5)
        : patch(patch), positive(positive), probability(probability) {}
    ClassifiedPatch(std::shared_ptr<imageprocessing::Patch> patch, std::pair<bool, double> result)
        : patch(patch), positive(result.first), probability(result.second) {}
    std::shared_ptr<imageprocessing::Patch> getPatch() { return patch; }
    const std::shared_ptr<imageprocessing::Patch> getPatch() const { return patch; }
    bool isPositive() const { return positive; }
    double getProbability() const { return probability; }
    bool operator<(const ClassifiedPatch& other) const { return probability < other.probability; }
    bool operator>(const ClassifiedPatch& other) const { return probability > other.probability; }
private:
    std::shared_ptr<imageprocessing::Patch> patch;
    bool positive;
    double probability;
};

// Detector.hpp:43-81
class Detector {
public:
    virtual ~Detector() {}
    virtual std::vector<std::shared_ptr<ClassifiedPatch>> detect(const cv::Mat& image) = 0;
    virtual std::vector<std::shared_ptr<ClassifiedPatch>> detect(const cv::Mat& image, const cv::Rect& roi) = 0;
    virtual std::vector<std::shared_ptr<ClassifiedPatch>> detect(std::shared_ptr<imageprocessing::VersionedImage> image) = 0;
    std::string landmark;
    // Backend extension.  On the fused paths the patches' pixels never leave the GPU: a returned patch carries its geometry and
    // probability, getPatch()->getData() is empty.  keepPatchData(true): every patch a detect() call RETURNS is extracted once more
    // through the feature extractor's single-patch path (PyramidFeatureExtractor::extract(x, y, width, height): the layer comes to
    // the host once per image, the patch filters run per Mat), so getData() holds what the reference's patch holds (Patch.hpp:28-243).
    void keepPatchData(bool keep) { patchData = keep; }
    bool keepsPatchData() const { return patchData; }
protected:
    void fillPatchData(const imageprocessing::PyramidFeatureExtractor& extractor, std::vector<std::shared_ptr<ClassifiedPatch>>& patches) const;
    bool patchData = false;
};

// OverlapElimination.hpp:45-55 / OverlapElimination.cpp:44-105
class OverlapElimination {
public:
    explicit OverlapElimination(float dist = 5.0f, float ratio = 0.0f) : dist(dist), ratio(ratio) {}
    std::vector<std::shared_ptr<ClassifiedPatch>> eliminate(std::vector<std::shared_ptr<ClassifiedPatch>>& classifiedPatches);
    float getDist() const { return dist; }
    float getRatio() const { return ratio; }
private:
    float dist, ratio;
};

// NonMaximumSuppression.hpp:20-117 / NonMaximumSuppression.cpp:27-118 (IoU clustering of the AggregatedFeaturesDetector family)
struct Detection {
    float score;
    cv::Rect bounds;
};
class NonMaximumSuppression {
public:
    enum class MaximumType { MAX_SCORE, AVERAGE, WEIGHTED_AVERAGE };
    explicit NonMaximumSuppression(double overlapThreshold, MaximumType maximumType = MaximumType::MAX_SCORE)
        : overlapThreshold(overlapThreshold), maximumType(maximumType) {}
    std::vector<Detection> eliminateRedundantDetections(std::vector<Detection> candidates) const;
    double getOverlapThreshold() const { return overlapThreshold; }
    MaximumType getMaximumType() const { return maximumType; }
private:
    double overlapThreshold;
    MaximumType maximumType;
};

// AggregatedFeaturesDetector.hpp:40-110 / AggregatedFeaturesDetector.cpp:37-128 (SimpleDetector interface: bounding boxes).
// On this backend the image filter must be a GrayscaleFilter and the layer filter a filtering::FhogFilter; the SVM must
// use a LinearKernel and hold one support vector of windowSize.height x (windowSize.width * channels) floats.
class AggregatedFeaturesDetector {
public:
    AggregatedFeaturesDetector(std::shared_ptr<imageprocessing::ImageFilter> imageFilter, std::shared_ptr<imageprocessing::ImageFilter> layerFilter,
                               int cellSize, cv::Size windowSize, int octaveLayerCount, std::shared_ptr<classification::SvmClassifier> svm,
                               std::shared_ptr<NonMaximumSuppression> nonMaximumSuppression, float widthScale = 1.0f, float heightScale = 1.0f,
                               int minWindowWidth = 0);
    ~AggregatedFeaturesDetector();
    std::vector<cv::Rect> detect(std::shared_ptr<imageprocessing::VersionedImage> image);
    std::vector<std::pair<cv::Rect, float>> detectWithScores(std::shared_ptr<imageprocessing::VersionedImage> image);
    std::vector<cv::Rect> detect(const cv::Mat& image) { return detect(std::make_shared<imageprocessing::VersionedImage>(image)); }
    std::vector<std::pair<cv::Rect, float>> detectWithScores(const cv::Mat& image) {
        return detectWithScores(std::make_shared<imageprocessing::VersionedImage>(image));
    }
    float getScoreThreshold() const { return scoreThreshold; }
private:
    fd_aggregated* handle = nullptr;
    float scoreThreshold;
};

// SlidingWindowDetector.hpp:41-93 / SlidingWindowDetector.cpp:40-98
class SlidingWindowDetector : public Detector {
public:
    explicit SlidingWindowDetector(std::shared_ptr<classification::ProbabilisticClassifier> classifier,
                                   std::shared_ptr<imageprocessing::PyramidFeatureExtractor> featureExtractor, int stepSizeX = 1,
                                   int stepSizeY = 1);
    std::vector<std::shared_ptr<ClassifiedPatch>> detect(const cv::Mat& image) override;
    std::vector<std::shared_ptr<ClassifiedPatch>> detect(const cv::Mat& image, const cv::Rect& roi) override;
    std::vector<std::shared_ptr<ClassifiedPatch>> detect(std::shared_ptr<imageprocessing::VersionedImage> image) override;
    const std::shared_ptr<imageprocessing::PyramidFeatureExtractor> getPyramidFeatureExtractor() const { return featureExtractor; }
    std::shared_ptr<classification::ProbabilisticClassifier> getClassifier() const { return classifier; }
    int getStepSizeX() const { return stepSizeX; }
    int getStepSizeY() const { return stepSizeY; }
private:
    std::vector<std::shared_ptr<ClassifiedPatch>> detect(const cv::Rect* roi) const;
    std::vector<std::shared_ptr<ClassifiedPatch>> detectWindows(const cv::Rect* roi) const;
    std::shared_ptr<classification::ProbabilisticClassifier> classifier;
    std::shared_ptr<imageprocessing::PyramidFeatureExtractor> featureExtractor;
    int stepSizeX, stepSizeY;
};

// FiveStageSlidingWindowDetector.hpp:36 / FiveStageSlidingWindowDetector.cpp:187-380
class FiveStageSlidingWindowDetector : public Detector {
public:
    FiveStageSlidingWindowDetector(std::shared_ptr<SlidingWindowDetector> slidingWindowDetector,
                                   std::shared_ptr<OverlapElimination> overlapElimination,
                                   std::shared_ptr<classification::ProbabilisticClassifier> strongClassifier);
    std::vector<std::shared_ptr<ClassifiedPatch>> detect(const cv::Mat& image) override;
    std::vector<std::shared_ptr<ClassifiedPatch>> detect(const cv::Mat& image, const cv::Rect& roi) override;
    std::vector<std::shared_ptr<ClassifiedPatch>> detect(std::shared_ptr<imageprocessing::VersionedImage> image) override;
    const std::shared_ptr<imageprocessing::PyramidFeatureExtractor> getPyramidFeatureExtractor() const {
        return slidingWindowDetector->getPyramidFeatureExtractor();
    }
    // Backend extension (not in the reference): detect(image) for every image of a sequence in one call.  Up to 64 equally sized
    // images share one multi-frame pyramid, one cascade run and one SVM launch (fd_pyramid_update_frames +
    // fd_detect_five_stage_frames); result i equals detect(images[i]).  Images of different sizes, pyramids on a source pyramid
    // and more than 64 images are handled by splitting / falling back to detect().
    std::vector<std::vector<std::shared_ptr<ClassifiedPatch>>> detectFrames(const std::vector<cv::Mat>& images);
    ~FiveStageSlidingWindowDetector();
private:
    std::vector<std::shared_ptr<ClassifiedPatch>> run(const cv::Mat& image, const cv::Rect* roi);
    fd_pyramid* framesPyramid = nullptr;   // detectFrames: the multi-frame twin of the extractor's pyramid
    int framesCount = 0;
    std::shared_ptr<SlidingWindowDetector> slidingWindowDetector;
    std::shared_ptr<OverlapElimination> overlapElimination;
    std::shared_ptr<classification::ProbabilisticClassifier> strongClassifier;
};

}  // namespace detection
