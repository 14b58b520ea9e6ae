// superviseddescent/*.hpp of the reference on top of the C ABI (include/fd_hip.h).
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "fdcompat/cv.hpp"
#include "fdcompat/runtime.hpp"

namespace superviseddescent {

// DescriptorExtractor.hpp:40-51
class DescriptorExtractor {
public:
    virtual ~DescriptorExtractor() {}
    virtual cv::Mat getDescriptors(const cv::Mat image, std::vector<cv::Point2f> locations, int windowSizeHalf = 0) = 0;
    virtual std::string getParameterString() const = 0;
};

// DescriptorExtractor.hpp:83-231 -- fd_sdm_descriptors
class VlHogDescriptorExtractor : public DescriptorExtractor {
public:
    enum class VlHogType { DalalTriggs, Uoctti };
    explicit VlHogDescriptorExtractor(VlHogType vlhogType) : hogType(vlhogType), numCells(0), cellSize(0), numBins(0) {}
    VlHogDescriptorExtractor(VlHogType vlhogType, int numCells, int cellSize, int numBins)
        : hogType(vlhogType), numCells(numCells), cellSize(cellSize), numBins(numBins) {}
    cv::Mat getDescriptors(const cv::Mat image, std::vector<cv::Point2f> locations, int windowSizeHalf) override;
    std::string getParameterString() const override;
    VlHogType getType() const { return hogType; }
    int getNumCells() const { return numCells; }
    int getCellSize() const { return cellSize; }
    int getNumBins() const { return numBins; }
private:
    VlHogType hogType;
    int numCells, cellSize, numBins;
};

// SdmLandmarkModel.hpp:45-124 / SdmLandmarkModel.cpp:53-233
class SdmLandmarkModel {
public:
    SdmLandmarkModel() {}
    SdmLandmarkModel(cv::Mat meanLandmarks, std::vector<std::string> landmarkIdentifier, std::vector<cv::Mat> regressorData,
                     std::vector<std::shared_ptr<DescriptorExtractor>> descriptorExtractors, std::vector<std::string> descriptorTypes);
    int getNumLandmarks() const { return meanLandmarks.cols / 2; }
    int getNumCascadeSteps() const { return (int)regressorData.size(); }
    cv::Mat getMeanShape() const;                       // 2L x 1 column vector (copy)
    cv::Mat getRegressorData(int cascadeLevel) { return regressorData[cascadeLevel]; }
    std::shared_ptr<DescriptorExtractor> getDescriptorExtractor(int cascadeLevel) { return descriptorExtractors[cascadeLevel]; }
    std::string getDescriptorType(int cascadeLevel) { return descriptorTypes[cascadeLevel]; }
    std::vector<cv::Point2f> getMeanAsPoints() const;
    cv::Point2f getLandmarkAsPoint(std::string landmarkIdentifier, cv::Mat modelInstance = cv::Mat()) const;
    void save(std::string filename, std::string comment = "");
    static SdmLandmarkModel load(std::string filename);  // text format of SdmLandmarkModel.cpp:130-233
    const std::vector<std::string>& getLandmarkIdentifiers() const { return landmarkIdentifier; }
private:
    cv::Mat meanLandmarks;  // 1 x 2L
    std::vector<std::string> landmarkIdentifier;
    std::vector<cv::Mat> regressorData;
    std::vector<std::shared_ptr<DescriptorExtractor>> descriptorExtractors;
    std::vector<std::string> descriptorTypes;
};

// SdmLandmarkModel.hpp:143-259
class SdmLandmarkModelFitting {
public:
    // adaptive = true is the reference as compiled (`if (true) { // adaptive`, SdmLandmarkModel.hpp:209,243).  adaptive = false is the
    // `else` branch of the same lines (the extractors' own numCells / cellSize / numBins, no face-size factor), which upstream cannot
    // reach at run time; it exists here because the one model the reference ships (SDM_Model_HOG_Zhenhua_11012014.txt) needs it.
    explicit SdmLandmarkModelFitting(SdmLandmarkModel model, bool adaptive = true);
    ~SdmLandmarkModelFitting();
    cv::Mat alignRigid(cv::Mat modelShape, cv::Rect faceBox) const;    // :156-192
    cv::Mat optimize(cv::Mat modelShape, cv::Mat image);               // :199-256 -> fd_sdm_optimize_batch
    // batched form used by throughput callers: B images of equal size, B shapes (2L x 1 each)
    std::vector<cv::Mat> optimize(const std::vector<cv::Mat>& modelShapes, const std::vector<cv::Mat>& images);
private:
    SdmLandmarkModel model;
    fd_sdm* handle;
};

}  // namespace superviseddescent
