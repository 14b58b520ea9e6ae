// condensation/*.hpp of the reference (the measurement-model side of the particle filter) on top of the C ABI.
// the tracking-side caller of the WVM -> SVM path.
#pragma once
#include <memory>
#include <vector>
#include "classification/classification_all.hpp"
#include "imageprocessing/imageprocessing_all.hpp"

namespace condensation {

// Sample.hpp:30-330 (position, size, velocity, weight, target flag; the fields the measurement model touches)
class Sample {
public:
    Sample() : x(0), y(0), size(0), vx(0), vy(0), vsize(1), weight(1), target(false) {}
    Sample(int x, int y, int size) : x(x), y(y), size(size), vx(0), vy(0), vsize(1), weight(1), target(false) {}
    Sample(int x, int y, int size, int vx, int vy, float vsize) : x(x), y(y), size(size), vx(vx), vy(vy), vsize(vsize), weight(1), target(false) {}
    cv::Rect getBounds() const { return cv::Rect(x - getWidth() / 2, y - getHeight() / 2, getWidth(), getHeight()); }
    int getX() const { return x; }
    void setX(int v) { x = v; }
    int getY() const { return y; }
    void setY(int v) { y = v; }
    int getSize() const { return size; }
    void setSize(int v) { size = v; }
    int getWidth() const { return size; }
    int getHeight() const { return cv::cvRound(Sample::aspectRatio * size); }
    int getVx() const { return vx; }
    int getVy() const { return vy; }
    float getVSize() const { return vsize; }
    double getWeight() const { return weight; }
    void setWeight(double w) { weight = w; }
    bool isTarget() const { return target; }
    void setTarget(bool t) { target = t; }
    static void setAspectRatio(double ratio) { Sample::aspectRatio = ratio; }
    static double getAspectRatio() { return Sample::aspectRatio; }
    static double aspectRatio;   // Sample.cpp:12
private:
    int x, y, size, vx, vy;
    float vsize;
    double weight;
    bool target;
};

// MeasurementModel.hpp:25-56
class MeasurementModel {
public:
    virtual ~MeasurementModel() {}
    virtual void update(std::shared_ptr<imageprocessing::VersionedImage> image) = 0;
    virtual void evaluate(Sample& sample) const = 0;
    virtual void evaluate(std::shared_ptr<imageprocessing::VersionedImage> image, std::vector<std::shared_ptr<Sample>>& samples) {
        update(image);
        for (std::shared_ptr<Sample> sample : samples) evaluate(*sample);
    }
};

// WvmSvmModel.hpp / WvmSvmModel.cpp:36-118.  With a DirectPyramidFeatureExtractor + HistEq64Filter all samples are
// scored in one call of fd_wvm_svm_evaluate_samples.
class WvmSvmModel : public MeasurementModel {
public:
    WvmSvmModel(std::shared_ptr<imageprocessing::FeatureExtractor> featureExtractor, std::shared_ptr<classification::ProbabilisticWvmClassifier> wvm,
                std::shared_ptr<classification::ProbabilisticSvmClassifier> svm);
    void update(std::shared_ptr<imageprocessing::VersionedImage> image) override;
    void evaluate(Sample& sample) const override;
    void evaluate(std::shared_ptr<imageprocessing::VersionedImage> image, std::vector<std::shared_ptr<Sample>>& samples) override;
private:
    std::shared_ptr<imageprocessing::FeatureExtractor> featureExtractor;
    std::shared_ptr<classification::ProbabilisticWvmClassifier> wvm;
    std::shared_ptr<classification::ProbabilisticSvmClassifier> svm;
};

}  // namespace condensation
