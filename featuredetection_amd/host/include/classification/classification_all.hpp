// classification/*.hpp of the reference on top of the C ABI (include/fd_hip.h).
#pragma once
#include <fstream>
#include <memory>
#include <string>
#include <utility>
#include <vector>
#include "fdcompat/cv.hpp"
#include "fdcompat/ptree.hpp"
#include "fdcompat/runtime.hpp"

namespace classification {

// BinaryClassifier.hpp:21-43
class BinaryClassifier {
public:
    virtual ~BinaryClassifier() {}
    virtual bool classify(const cv::Mat& featureVector) const = 0;
    virtual std::pair<bool, double> getConfidence(const cv::Mat& featureVector) const = 0;
};

// ProbabilisticClassifier.hpp:22-34
class ProbabilisticClassifier : public BinaryClassifier {
public:
    virtual ~ProbabilisticClassifier() {}
    virtual std::pair<bool, double> getProbability(const cv::Mat& featureVector) const = 0;
};

class KernelVisitor;
// Kernel.hpp:22-42
class Kernel {
public:
    virtual ~Kernel() {}
    virtual double compute(const cv::Mat& lhs, const cv::Mat& rhs) const;   // one-SV SVM on the GPU
    virtual int abiKernel() const = 0;                                       // FD_KERNEL_*
    virtual void abiParams(double& p0, double& p1, double& p2) const { p0 = p1 = p2 = 0; }
};
class LinearKernel : public Kernel {        // LinearKernel.hpp:27-29
public:
    int abiKernel() const override { return FD_KERNEL_LINEAR; }
};
class PolynomialKernel : public Kernel {    // PolynomialKernel.hpp:35-37
public:
    explicit PolynomialKernel(double alpha, double constant = 0, int degree = 2) : alpha(alpha), constant(constant), degree(degree) {}
    int abiKernel() const override { return FD_KERNEL_POLY; }
    void abiParams(double& p0, double& p1, double& p2) const override { p0 = alpha; p1 = constant; p2 = degree; }
    double getAlpha() const { return alpha; }
    double getConstant() const { return constant; }
    int getDegree() const { return degree; }
private:
    double alpha, constant;
    int degree;
};
class RbfKernel : public Kernel {           // RbfKernel.hpp:32-40
public:
    explicit RbfKernel(double gamma) : gamma(gamma) {}
    int abiKernel() const override { return FD_KERNEL_RBF; }
    void abiParams(double& p0, double& p1, double& p2) const override { p0 = gamma; p1 = p2 = 0; }
    double getGamma() const { return gamma; }
private:
    double gamma;
};
class HistogramIntersectionKernel : public Kernel {   // HistogramIntersectionKernel.hpp:28-36
public:
    int abiKernel() const override { return FD_KERNEL_HIK; }
};

// VectorMachineClassifier.hpp:23-78
class VectorMachineClassifier : public BinaryClassifier {
public:
    explicit VectorMachineClassifier(std::shared_ptr<Kernel> kernel) : kernel(kernel), bias(0), threshold(0) {}
    float getThreshold() const { return threshold; }
    virtual void setThreshold(float t) { threshold = t; }
    std::shared_ptr<Kernel> getKernel() { return kernel; }
    const std::shared_ptr<Kernel> getKernel() const { return kernel; }
    float getBias() const { return bias; }
protected:
    std::shared_ptr<Kernel> kernel;
    float bias, threshold;
};

// SvmClassifier.hpp / SvmClassifier.cpp:32-159 -- scoring by fd_svm_distance_batch
class SvmClassifier : public VectorMachineClassifier {
public:
    explicit SvmClassifier(std::shared_ptr<Kernel> kernel);
    ~SvmClassifier();
    bool classify(const cv::Mat& featureVector) const override;
    std::pair<bool, double> getConfidence(const cv::Mat& featureVector) const override;
    bool classify(double hyperplaneDistance) const { return hyperplaneDistance >= threshold; }
    double computeHyperplaneDistance(const cv::Mat& featureVector) const;
    void setSvmParameters(std::vector<cv::Mat> supportVectors, std::vector<float> coefficients, double bias);
    void setThreshold(float t) override { threshold = t; dirty = true; }
    void store(std::ofstream& file);                       // SvmClassifier.cpp:68-107 text format
    static std::shared_ptr<SvmClassifier> load(std::ifstream& file);   // :109-159
    static std::shared_ptr<SvmClassifier> loadFromText(const std::string& classifierFilename);   // :161-239 (FullPolynomial text models)
    const std::vector<cv::Mat>& getSupportVectors() const { return supportVectors; }
    const std::vector<float>& getCoefficients() const { return coefficients; }
    const fd_svm* native(double logisticA = 0.00556, double logisticB = -2.95) const;   // (re)builds the device model lazily
private:
    std::vector<cv::Mat> supportVectors;
    std::vector<float> coefficients;
    mutable fd_svm* handle;
    mutable bool dirty;
};

// ProbabilisticSvmClassifier.hpp / .cpp:36-164
class ProbabilisticSvmClassifier : public ProbabilisticClassifier {
public:
    explicit ProbabilisticSvmClassifier(std::shared_ptr<Kernel> kernel, double logisticA = 0.00556, double logisticB = -2.95)
        : svm(std::make_shared<SvmClassifier>(kernel)), logisticA(logisticA), logisticB(logisticB) {}
    explicit ProbabilisticSvmClassifier(std::shared_ptr<SvmClassifier> svm, double logisticA = 0.00556, double logisticB = -2.95)
        : svm(svm), logisticA(logisticA), logisticB(logisticB) {}
    bool classify(const cv::Mat& featureVector) const override { return svm->classify(featureVector); }
    std::pair<bool, double> getConfidence(const cv::Mat& featureVector) const override { return svm->getConfidence(featureVector); }
    std::pair<bool, double> getProbability(const cv::Mat& featureVector) const override;
    std::pair<bool, double> getProbability(double hyperplaneDistance) const;
    void setLogisticParameters(double a, double b) { logisticA = a; logisticB = b; }
    void store(std::ofstream& file);
    static std::shared_ptr<ProbabilisticSvmClassifier> load(std::ifstream& file);
    // ptree: classifierFile (SvmClassifier::loadFromText format, or the SvmClassifier::store stream format as the stand-in for
    // .mat models), optional logisticA / logisticB / threshold (ProbabilisticSvmClassifier.cpp:80-104)
    static std::shared_ptr<ProbabilisticSvmClassifier> load(const boost::property_tree::ptree& subtree);
    std::shared_ptr<SvmClassifier> getSvm() { return svm; }
    const std::shared_ptr<SvmClassifier> getSvm() const { return svm; }
    double getLogisticA() const { return logisticA; }
    double getLogisticB() const { return logisticB; }
private:
    std::shared_ptr<SvmClassifier> svm;
    double logisticA, logisticB;
};

// WvmClassifier.hpp / WvmClassifier.cpp:31-181 -- model arrays in the unit conventions of the Matlab
// loader (:352-769); scoring by fd_wvm_eval_batch / the fused fd_detect_* entry points.
class WvmClassifier : public VectorMachineClassifier {
public:
    struct Model {   // flat model, same fields as fd_wvm_model
        int filter_w = 0, filter_h = 0, num_filters = 0, num_used = 0, num_per_level = 0;
        float basis_param = 0, bias = 0;
        std::vector<float> thresholdsFromFile, hk_weights;
        std::vector<double> pp, val;
        std::vector<int32_t> val_off, rec_off;
        std::vector<uint8_t> rects;
    };
    WvmClassifier();
    ~WvmClassifier();
    bool classify(const cv::Mat& featureVector) const override;
    std::pair<bool, double> getConfidence(const cv::Mat& featureVector) const override;
    std::pair<int, double> computeHyperplaneDistance(const cv::Mat& featureVector) const;
    bool classify(std::pair<int, double> levelAndDistance) const;
    void setNumUsedFilters(int var);
    int getNumUsedFilters() { return model.num_used; }
    void setLimitReliabilityFilter(float var);
    float getLimitReliabilityFilter() { return limitReliabilityFilter; }
    // binary model file written by featuredetection_amd.synth / tools (the reference's .mat files are absent)
    static std::shared_ptr<WvmClassifier> loadFromFile(const std::string& classifierFilename);
    void setModel(const Model& m);
    const fd_wvm* native(double logisticA = 0.00556, double logisticB = -2.95) const;
    const Model& getModel() const { return model; }
private:
    Model model;
    std::vector<float> hierarchicalThresholds;
    float limitReliabilityFilter;
    mutable fd_wvm* handle;
    mutable bool dirty;
};

// RvmClassifier.hpp / RvmClassifier.cpp:42-126 -- scoring by fd_rvm_eval_batch / fd_detect_rvm
class RvmClassifier : public VectorMachineClassifier {
public:
    struct Model {   // flat model, same fields as fd_rvm_model
        int kernel = 2;
        double p0 = 0, p1 = 0, p2 = 0;
        int filter_w = 0, filter_h = 0, num_filters = 0;
        float bias = 0;
        std::vector<float> support_vectors, coefficients /* packed lower triangle */, thresholds;
    };
    explicit RvmClassifier(std::shared_ptr<Kernel> kernel, bool cascadedCoefficients = true);
    ~RvmClassifier();
    bool classify(const cv::Mat& featureVector) const override;
    std::pair<bool, double> getConfidence(const cv::Mat& featureVector) const override;
    std::pair<bool, double> getConfidence(std::pair<int, double> levelAndDistance) const;
    bool classify(std::pair<int, double> levelAndDistance) const;
    std::pair<int, double> computeHyperplaneDistance(const cv::Mat& featureVector) const;   // feature vector: CV_32F
    unsigned int getNumFiltersToUse() const { return numFiltersToUse; }
    void setNumFiltersToUse(unsigned int numFilters);
    // ptree: classifierFile (binary FDRVM1 written by featuredetection_amd.synth.save_rvm; the reference reads Matlab .mat
    // files through libmat, RvmClassifier.cpp:141-330)
    static std::shared_ptr<RvmClassifier> load(const boost::property_tree::ptree& subtree);
    static std::shared_ptr<RvmClassifier> loadFromFile(const std::string& classifierFilename);
    const fd_rvm* native(double logisticA = 0.0, double logisticB = -1.0) const;
    const Model& getModel() const { return model; }
private:
    Model model;
    unsigned int numFiltersToUse = 0;
    mutable fd_rvm* handle = nullptr;
    mutable bool dirty = true;
    mutable double builtA = 0, builtB = 0;
};

// ProbabilisticRvmClassifier.hpp / .cpp:32-115
class ProbabilisticRvmClassifier : public ProbabilisticClassifier {
public:
    explicit ProbabilisticRvmClassifier(std::shared_ptr<RvmClassifier> rvm, double logisticA = 0.00556, double logisticB = -2.95)
        : rvm(rvm), logisticA(logisticA), logisticB(logisticB) {}
    bool classify(const cv::Mat& featureVector) const override { return rvm->classify(featureVector); }
    std::pair<bool, double> getConfidence(const cv::Mat& featureVector) const override { return rvm->getConfidence(featureVector); }
    std::pair<bool, double> getProbability(const cv::Mat& featureVector) const override;
    std::pair<bool, double> getProbability(std::pair<int, double> levelAndDistance) const;
    void setLogisticParameters(double a, double b) { logisticA = a; logisticB = b; }
    // ptree: classifierFile (FDRVM1), optional logisticA / logisticB, numFiltersToUse
    static std::shared_ptr<ProbabilisticRvmClassifier> load(const boost::property_tree::ptree& subtree);
    std::shared_ptr<RvmClassifier> getRvm() { return rvm; }
    const std::shared_ptr<RvmClassifier> getRvm() const { return rvm; }
    double getLogisticA() const { return logisticA; }
    double getLogisticB() const { return logisticB; }
private:
    std::shared_ptr<RvmClassifier> rvm;
    double logisticA, logisticB;
};

// ProbabilisticWvmClassifier.hpp / .cpp:32-139
class ProbabilisticWvmClassifier : public ProbabilisticClassifier {
public:
    explicit ProbabilisticWvmClassifier(std::shared_ptr<WvmClassifier> wvm, double logisticA = 0.00556, double logisticB = -2.95)
        : wvm(wvm), logisticA(logisticA), logisticB(logisticB) {}
    bool classify(const cv::Mat& featureVector) const override { return wvm->classify(featureVector); }
    std::pair<bool, double> getConfidence(const cv::Mat& featureVector) const override { return wvm->getConfidence(featureVector); }
    std::pair<bool, double> getProbability(const cv::Mat& featureVector) const override;
    std::pair<bool, double> getProbability(std::pair<int, double> levelAndDistance) const;
    // ptree: classifierFile (binary .fdwvm), optional logisticA/logisticB, threshold (limitReliabilityFilter)
    static std::shared_ptr<ProbabilisticWvmClassifier> load(const boost::property_tree::ptree& subtree);
    std::shared_ptr<WvmClassifier> getWvm() { return wvm; }
    const std::shared_ptr<WvmClassifier> getWvm() const { return wvm; }
    double getLogisticA() const { return logisticA; }
    double getLogisticB() const { return logisticB; }
private:
    std::shared_ptr<WvmClassifier> wvm;
    double logisticA, logisticB;
};

}  // namespace classification
