// featuredetection_amd/host/src/fd_host.cpp -- implementation of the reference-shaped C++ classes
// (host/include/{imageprocessing,classification,detection,superviseddescent}) on top of the C ABI.
// Everything numerical is delegated to libfd_hip.so; this file is object plumbing only.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <sstream>
#include "detection/detection_all.hpp"
#include "superviseddescent/superviseddescent_all.hpp"

using cv::Mat;
using cv::uchar;
using std::make_shared;
using std::shared_ptr;
using std::string;
using std::vector;

namespace fdhost {
fd_ctx* context() {
    static fd_ctx* ctx = nullptr;
    if (!ctx) {
        const char* dev = std::getenv("FD_DEVICE");
        int rc = fd_ctx_create(dev ? std::atoi(dev) : 0, nullptr, &ctx);
        if (rc != FD_OK || !ctx) throw std::runtime_error("fd_ctx_create failed: no usable gfx950 device (this backend has no CPU fallback)");
    }
    return ctx;
}
}  // namespace fdhost
using fdhost::check;
using fdhost::context;

static Mat contiguous(const Mat& m) { return m.isContinuous() ? m : m.clone(); }

// =================================================================================================
namespace imageprocessing {

Mat GrayscaleFilter::applyTo(const Mat& image, Mat& filtered) const {
    if (image.channels() == 1) { image.copyTo(filtered); return filtered; }
    // a one-layer pyramid at scale 1 is exactly the grayscale image (resize to the same size is a copy)
    fd_pyramid* p = nullptr;
    check(fd_pyramid_create(context(), 1, 1.0, 1.0, &p));
    Mat src = contiguous(image);
    int rc = fd_pyramid_update(p, src.data, src.cols, src.rows, src.channels(), 0);
    if (rc == FD_OK) {
        filtered.create(src.rows, src.cols, CV_8UC1);
        rc = fd_pyramid_layer_download(p, 0, filtered.data);
    }
    fd_pyramid_destroy(p);
    check(rc);
    return filtered;
}

Mat HistEq64Filter::applyTo(const Mat& image, Mat& filtered) const {
    if (image.type() != CV_8UC1) throw std::invalid_argument("HistEq64Filter: the image must be of type CV_8UC1");
    Mat src = contiguous(image);
    Mat dst(src.rows, src.cols, CV_8UC1);
    check(fd_histeq64_batch(context(), src.data, 1, src.cols, src.rows, dst.data));
    filtered = dst;
    return filtered;
}

Mat GreyWorldNormalizationFilter::applyTo(const Mat& image, Mat& filtered) const {
    if (image.type() != CV_8UC3) throw std::invalid_argument("GreyWorldNormalizationFilter: The image type must be CV_8UC3");
    Mat src = contiguous(image);
    Mat dst(src.rows, src.cols, CV_8UC3);
    check(fd_greyworld(context(), src.data, src.cols, src.rows, dst.data, 0));
    filtered = dst;
    return filtered;
}

GradientFilter::GradientFilter(int kernelSize, int blurKernelSize) : kernelSize(kernelSize), blurKernelSize(blurKernelSize) {
    if (kernelSize != 1 && kernelSize != 3 && kernelSize != 5 && kernelSize != 7 && kernelSize != CV_SCHARR)
        throw std::invalid_argument("GradientFilter: the kernel size must be 1, 3, 5, 7 or CV_SCHARR");   // GradientFilter.cpp:16-19
}
// ---- stand-alone ImageFilter::applyTo(const Mat&) forms (ImageFilter.hpp:18-57): one kernel launch per Mat through the C ABI
Mat GradientFilter::applyTo(const Mat& image, Mat& filtered) const {
    if (image.type() != CV_8UC1) throw std::invalid_argument("GradientFilter: the image must be of type CV_8UC1");
    Mat src = contiguous(image);
    Mat dst(src.rows, src.cols, CV_8UC2);
    check(fd_gradient_filter_image(context(), src.data, src.cols, src.rows, kernelSize, blurKernelSize > 0 ? blurKernelSize : 0, dst.data));
    filtered = dst;
    return filtered;
}
GradientBinningFilter::GradientBinningFilter(unsigned int bins, bool signedGradients, bool interpolate)
    : bins(bins), signedGradients(signedGradients), interpolate(interpolate) {}
Mat GradientBinningFilter::applyTo(const Mat& image, Mat& filtered) const {
    if (image.type() != CV_8UC2) throw std::invalid_argument("GradientBinningFilter: the image must be of type CV_8UC2");
    Mat src = contiguous(image);
    Mat dst(src.rows, src.cols, interpolate ? CV_8UC4 : CV_8UC2);
    check(fd_gradient_binning_image(context(), src.data, src.cols, src.rows, (int)bins, signedGradients, interpolate, dst.data));
    filtered = dst;
    return filtered;
}
Mat LbpFilter::applyTo(const Mat& image, Mat& filtered) const {
    if (image.type() != CV_8UC1) throw std::invalid_argument("LbpFilter: the image must be of type CV_8UC1");
    Mat src = contiguous(image);
    Mat dst(src.rows, src.cols, CV_8UC1);
    check(fd_lbp_image(context(), src.data, src.cols, src.rows, (int)type, dst.data));
    filtered = dst;
    return filtered;
}
unsigned int LbpFilter::getBinCount() const {
    switch (type) {
        case Type::LBP8: return 256;
        case Type::LBP8_UNIFORM: return 59;
        default: return 16;
    }
}
Mat WhiteningFilter::applyTo(const Mat& image, Mat& filtered) const {
    if (image.channels() > 1) throw std::invalid_argument("WhiteningFilter: the image must have exactly one channel");
    if (image.type() != CV_8UC1) throw std::invalid_argument("WhiteningFilter: CV_8UC1 images are supported on this backend");
    Mat src = contiguous(image);
    Mat dst(src.rows, src.cols, CV_8UC1);
    check(fd_whitening_batch(context(), src.data, 1, src.cols, src.rows, alpha, cutoffFrequency, dst.data));
    filtered = dst;
    return filtered;
}
Mat ConversionFilter::applyTo(const Mat& image, Mat& filtered) const {
    const int sd = image.depth(), dd = type & 7;
    if ((sd != CV_8U && sd != CV_32F) || (dd != CV_8U && dd != CV_32F))
        throw std::invalid_argument("ConversionFilter: CV_8U and CV_32F are supported on this backend");
    Mat src = contiguous(image);
    Mat dst(src.rows, src.cols, CV_MAKETYPE(dd, src.channels()));
    check(fd_convert_batch(context(), src.data, sd == CV_32F ? FD_DTYPE_F32 : FD_DTYPE_U8, (int64_t)src.rows * src.cols * src.channels(), alpha, beta,
                           dst.data, dd == CV_32F ? FD_DTYPE_F32 : FD_DTYPE_U8));
    filtered = dst;
    return filtered;
}
Mat UnitNormFilter::applyTo(const Mat& image, Mat& filtered) const {
    if (image.channels() > 1) throw std::invalid_argument("UnitNormFilter: The image must have exactly one channel.");
    Mat f32 = image.depth() == CV_32F ? contiguous(image) : ConversionFilter(CV_32F).applyTo(image);   // image.convertTo(filtered, CV_32F)
    Mat dst(f32.rows, f32.cols, CV_32FC1);
    check(fd_unit_norm_batch(context(), f32.ptr<float>(0), 1, f32.rows * f32.cols, normType, dst.ptr<float>(0)));
    filtered = dst;
    return filtered;
}
Mat ZeroMeanUnitVarianceFilter::applyTo(const Mat& image, Mat& filtered) const {   // ZeroMeanUnitVarianceFilter.cpp:21-34
    if (image.channels() > 1) throw std::invalid_argument("ZeroMeanUnitVarianceFilter: The image must have exactly one channel.");
    Mat f32 = image.depth() == CV_32F ? contiguous(image).clone() : ConversionFilter(CV_32F).applyTo(image);   // image.convertTo(filtered, CV_32F)
    const size_t n = (size_t)f32.rows * f32.cols;
    const float* x = f32.ptr<float>(0);
    double s = 0, sq = 0;   // cv::meanStdDev: sums in double, population deviation
    for (size_t i = 0; i < n; ++i) { s += x[i]; sq += (double)x[i] * x[i]; }
    const double mean = n ? s / (double)n : 0.0;
    const double dev = n ? std::sqrt(std::max(sq / (double)n - mean * mean, 0.0)) : 0.0;
    Mat dst(f32.rows, f32.cols, CV_32FC1);
    float* o = dst.ptr<float>(0);
    for (size_t i = 0; i < n; ++i) o[i] = dev == 0 ? 0.f : (float)(((double)x[i] - mean) / dev);
    filtered = dst;
    return filtered;
}
Mat ReshapingFilter::applyTo(const Mat& image, Mat& filtered) const {   // Mat::reshape(cn, rows) of a continuous matrix
    Mat src = contiguous(image).clone();
    const int cn = channels == 0 ? src.channels() : channels;
    const size_t total = (size_t)src.rows * src.cols * src.channels();
    const int r = rows == 0 ? src.rows : rows;
    if (r <= 0 || total % ((size_t)r * cn) != 0) throw std::invalid_argument("ReshapingFilter: the matrix cannot be reshaped to the requested number of rows");
    Mat dst(r, (int)(total / ((size_t)r * cn)), CV_MAKETYPE(src.depth(), cn));
    std::memcpy(dst.data, src.data, total * Mat::elemSizeOf(src.depth()));
    filtered = dst;
    return filtered;
}
Mat HistogramEqualizationFilter::applyTo(const Mat& image, Mat& filtered) const {
    if (image.type() != CV_8UC1) throw std::invalid_argument("HistogramEqualizationFilter: the image must be of type CV_8UC1");
    Mat src = image.isContinuous() ? image : image.clone();
    filtered.create(src.rows, src.cols, CV_8UC1);
    check(fd_equalize_hist_batch(context(), src.ptr<uchar>(0), 1, src.cols, src.rows, filtered.ptr<uchar>(0)));
    return filtered;
}
HogFilter::HogFilter(int binCount, int cellSize, int blockSize, bool interpolate, bool signedAndUnsigned)
    : HogFilter(binCount, cellSize, cellSize, blockSize, blockSize, interpolate, signedAndUnsigned) {}
HogFilter::HogFilter(int binCount, int cellWidth, int cellHeight, int blockWidth, int blockHeight, bool interpolate, bool signedAndUnsigned)
    : HistogramFilter(Normalization::L2NORM), binCount(binCount), cellWidth(cellWidth), cellHeight(cellHeight), blockWidth(blockWidth),
      blockHeight(blockHeight), interpolate(interpolate), signedAndUnsigned(signedAndUnsigned) {
    if (binCount <= 0) throw std::invalid_argument("HogFilter: binCount must be greater than zero");
    if (cellWidth <= 0) throw std::invalid_argument("HogFilter: cellWidth must be greater than zero");
    if (cellHeight <= 0) throw std::invalid_argument("HogFilter: cellHeight must be greater than zero");
    if (blockWidth <= 0) throw std::invalid_argument("HogFilter: blockWidth must be greater than zero");
    if (blockHeight <= 0) throw std::invalid_argument("HogFilter: blockHeight must be greater than zero");
    if (signedAndUnsigned && binCount % 2 != 0)
        throw std::invalid_argument("HogFilter: the bin size must be even for signed and unsigned gradients to be combined");
}
SpatialHistogramFilter::SpatialHistogramFilter(int binCount, int cellSize, int blockSize, bool interpolate, bool concatenate, Normalization normalization)
    : SpatialHistogramFilter(binCount, cellSize, cellSize, blockSize, blockSize, interpolate, concatenate, normalization) {}
SpatialHistogramFilter::SpatialHistogramFilter(int binCount, int cellWidth, int cellHeight, int blockWidth, int blockHeight, bool interpolate,
                                               bool concatenate, Normalization normalization)
    : HistogramFilter(normalization), binCount(binCount), cellWidth(cellWidth), cellHeight(cellHeight), blockWidth(blockWidth),
      blockHeight(blockHeight), interpolate(interpolate), concatenate(concatenate) {
    if (binCount <= 0) throw std::invalid_argument("SpatialHistogramFilter: binCount must be greater than zero");
    if (cellWidth <= 0) throw std::invalid_argument("SpatialHistogramFilter: cellWidth must be greater than zero");
    if (cellHeight <= 0) throw std::invalid_argument("SpatialHistogramFilter: cellHeight must be greater than zero");
    if (blockWidth <= 0) throw std::invalid_argument("SpatialHistogramFilter: blockWidth must be greater than zero");
    if (blockHeight <= 0) throw std::invalid_argument("SpatialHistogramFilter: blockHeight must be greater than zero");
}
static Mat hist_apply(const HistogramFilter& f, const Mat& image, Mat& filtered);
Mat SpatialHistogramFilter::applyTo(const Mat& image, Mat& filtered) const { return hist_apply(*this, image, filtered); }
PyramidHogFilter::PyramidHogFilter(int binCount, int levelCount, bool interpolate, bool signedAndUnsigned)
    : HistogramFilter(Normalization::L2NORM), binCount(binCount), levelCount(levelCount), interpolate(interpolate),
      signedAndUnsigned(signedAndUnsigned) {
    if (binCount <= 0) throw std::invalid_argument("PyramidHogFilter: binCount must be greater than zero");
    if (levelCount <= 0) throw std::invalid_argument("PyramidHogFilter: levelCount must be greater than zero");
    if (signedAndUnsigned && binCount % 2 != 0)
        throw std::invalid_argument("PyramidHogFilter: the bin size must be even for signed and unsigned gradients to be combined");
}
Mat PyramidHogFilter::applyTo(const Mat& image, Mat& filtered) const { return hist_apply(*this, image, filtered); }
SpatialPyramidHistogramFilter::SpatialPyramidHistogramFilter(int binCount, int levelCount, bool interpolate, Normalization normalization)
    : HistogramFilter(normalization), binCount(binCount), levelCount(levelCount), interpolate(interpolate) {
    if (binCount <= 0) throw std::invalid_argument("SpatialPyramidHistogramFilter: binCount must be greater than zero");
    if (levelCount <= 0) throw std::invalid_argument("SpatialPyramidHistogramFilter: levelCount must be greater than zero");
}
Mat SpatialPyramidHistogramFilter::applyTo(const Mat& image, Mat& filtered) const { return hist_apply(*this, image, filtered); }

// parameters of the fused histogram kernels for a patch filter
static fd_hist_params hist_params_of(const HistogramFilter& f, int pw, int ph, int stepX, int stepY) {
    fd_hist_params hp;
    std::memset(&hp, 0, sizeof(hp));
    hp.patch_w = pw; hp.patch_h = ph; hp.step_x = stepX; hp.step_y = stepY;
    hp.normalization = (int)f.normalization;
    if (auto h = dynamic_cast<const HogFilter*>(&f)) {
        hp.kind = FD_HIST_HOG; hp.bins = h->binCount; hp.cell_size = h->cellWidth; hp.cell_h = h->cellHeight;
        hp.block_size = h->blockWidth; hp.block_h = h->blockHeight; hp.interpolate = h->interpolate; hp.signed_and_unsigned = h->signedAndUnsigned;
    } else if (auto s = dynamic_cast<const SpatialHistogramFilter*>(&f)) {
        hp.kind = FD_HIST_SPATIAL; hp.bins = s->binCount; hp.cell_size = s->cellWidth; hp.cell_h = s->cellHeight;
        hp.block_size = s->blockWidth; hp.block_h = s->blockHeight; hp.interpolate = s->interpolate; hp.concatenate = s->concatenate;
    } else if (auto q = dynamic_cast<const PyramidHogFilter*>(&f)) {
        hp.kind = FD_HIST_PYRAMID_HOG; hp.bins = q->binCount; hp.levels = q->levelCount; hp.interpolate = q->interpolate;
        hp.signed_and_unsigned = q->signedAndUnsigned;
    } else if (auto y = dynamic_cast<const SpatialPyramidHistogramFilter*>(&f)) {
        hp.kind = FD_HIST_SPATIAL_PYRAMID; hp.bins = y->binCount; hp.levels = y->levelCount; hp.interpolate = y->interpolate;
    } else {
        throw std::logic_error("unsupported HistogramFilter subclass");
    }
    return hp;
}
// HistogramFilter::applyTo(const Mat&): the bin image of one patch (CV_8UC1 bins, CV_8UC2 bin + weight, CV_8UC4 two bins + weights)
// -> 1 x F CV_32F feature vector
static Mat hist_apply(const HistogramFilter& f, const Mat& image, Mat& filtered) {
    if (image.depth() != CV_8U || (image.channels() != 1 && image.channels() != 2 && image.channels() != 4))
        throw std::invalid_argument("HistogramFilter: The image must have one, two or four channels and be of depth CV_8U");
    Mat src = contiguous(image);
    fd_hist_params hp = hist_params_of(f, src.cols, src.rows, 1, 1);
    const int F = fd_hist_feature_length(&hp, src.channels());
    if (F < 0) throw std::invalid_argument("HistogramFilter: invalid parameters for this patch size");
    Mat dst(1, F, CV_32FC1);
    check(fd_hist_patch_batch(context(), src.data, 1, src.channels(), &hp, dst.ptr<float>(0)));
    filtered = dst;
    return filtered;
}
Mat HogFilter::applyTo(const Mat& image, Mat& filtered) const { return hist_apply(*this, image, filtered); }

// ---- ImagePyramid -------------------------------------------------------------------------------
ImagePyramid::ImagePyramid(size_t octaveLayerCount, double minS, double maxS)
    : handle(nullptr), minScaleFactor(minS), maxScaleFactor(maxS), ctorOctaveLayers(octaveLayerCount), layersValid(false) {
    check(fd_pyramid_create(context(), (int)octaveLayerCount, minS, maxS, &handle));
}
ImagePyramid::ImagePyramid(double inc, double minS, double maxS)
    : handle(nullptr), minScaleFactor(minS), maxScaleFactor(maxS), ctorIncremental(inc), layersValid(false) {
    check(fd_pyramid_create_inc(context(), inc, minS, maxS, &handle));
}
fd_pyramid* ImagePyramid::createFramesPyramid(int frames) const {
    if (sourcePyramid || !handle || gradient || binning || lbp) return nullptr;
    fd_pyramid* p = nullptr;
    if (ctorOctaveLayers) check(fd_pyramid_create(context(), (int)ctorOctaveLayers, minScaleFactor, maxScaleFactor, &p));
    else check(fd_pyramid_create_inc(context(), ctorIncremental, minScaleFactor, maxScaleFactor, &p));
    const int rc = fd_pyramid_set_frames(p, frames);
    if (rc != FD_OK) { fd_pyramid_destroy(p); check(rc); }
    return p;
}
ImagePyramid::ImagePyramid(shared_ptr<ImagePyramid> pyramid, double minS, double maxS)
    : handle(nullptr), sourcePyramid(pyramid), minScaleFactor(minS), maxScaleFactor(maxS), layersValid(false) {
    if (!pyramid) throw std::invalid_argument("ImagePyramid: the source pyramid must not be null");
}
ImagePyramid::~ImagePyramid() { if (handle) fd_pyramid_destroy(handle); }
double ImagePyramid::getIncrementalScaleFactor() const { return fd_pyramid_incremental_scale(native()); }
static long g_pyramidBuilds = 0;
long ImagePyramid::buildCount() { return g_pyramidBuilds; }

void ImagePyramid::addImageFilter(const shared_ptr<ImageFilter>& filter) {
    if (!std::dynamic_pointer_cast<GrayscaleFilter>(filter))
        throw std::logic_error("ImagePyramid: only GrayscaleFilter is supported as image filter on this backend");
    if (sourcePyramid) sourcePyramid->addImageFilter(filter);   // ImagePyramid.cpp:108-110
}
void ImagePyramid::setSource(const shared_ptr<VersionedImage>& image) {
    if (sourcePyramid && !handle)
        throw std::logic_error("ImagePyramid: a pyramid that was constructed on another pyramid has no layer parameters of its own to build from an image");
    sourcePyramid.reset();
    sourceImage = image;
}
void ImagePyramid::setSource(const shared_ptr<ImagePyramid>& pyramid) {
    if (!pyramid) throw std::invalid_argument("ImagePyramid: the source pyramid must not be null");
    if (gradient || binning || lbp) throw std::logic_error("ImagePyramid: layer filters on top of a source pyramid are not available on this backend");
    sourceImage.reset();
    sourcePyramid = pyramid;
    version = Version();
    layersValid = false;
}
void ImagePyramid::update() {   // ImagePyramid.cpp:146-168
    if (sourcePyramid) {
        if (version != sourcePyramid->version) { version = sourcePyramid->version; imageSize = sourcePyramid->imageSize; layersValid = false; }
    } else if (sourceImage) {
        update(sourceImage);
    }
}
// layer indices of the source pyramid whose scale factors lie inside [minScaleFactor, maxScaleFactor] (ImagePyramid.cpp:217-222)
void ImagePyramid::viewRange(int& first, int& last) const {
    first = -1; last = -1;
    if (!sourcePyramid) return;
    first = 1 << 30; last = -2;   // empty unless a layer qualifies
    for (const auto& sc : sourcePyramid->getLayerScales())
        if (sc.second >= minScaleFactor && sc.second <= maxScaleFactor) { first = std::min(first, sc.first); last = std::max(last, sc.first); }
}
ImagePyramid::Selection::Selection(fd_pyramid* h, int first, int last, int step, const cv::Rect* roi, int viewFirst, int viewLast) : handle(h) {
    int r[4] = {0, 0, 0, 0};
    if (roi) { r[0] = roi->x; r[1] = roi->y; r[2] = roi->width; r[3] = roi->height; }
    // an empty range (first > last) selects nothing: express it as an index range no layer has
    if (last != -1 && first > last) { first = 1 << 30; last = 1 << 30; }
    check(fd_pyramid_select(handle, first, last, step, roi ? r : nullptr));
    // the layer step of extract() walks getLayers() of THIS pyramid: a view starts counting at its own first layer
    check(fd_pyramid_select_view(handle, viewFirst, viewLast));
}
ImagePyramid::Selection::~Selection() {
    if (handle) { fd_pyramid_select(handle, -1, -1, 1, nullptr); fd_pyramid_select_view(handle, -1, -1); }
}
ImagePyramid::Selection ImagePyramid::select(int firstLayer, int lastLayer, int stepLayer, const cv::Rect* roi) const {
    if (stepLayer < 1) throw std::invalid_argument("DirectPyramidFeatureExtractor: stepLayer has to be greater than zero");
    int vf, vl;
    viewRange(vf, vl);
    if (sourcePyramid) {   // intersect with the view's scale range
        firstLayer = firstLayer < 0 ? vf : std::max(firstLayer, vf);
        lastLayer = lastLayer < 0 ? vl : std::min(lastLayer, vl);
        if (vl == -2) { firstLayer = 1; lastLayer = 0; }
    }
    const cv::Rect* r = (roi && (roi->x != 0 || roi->y != 0 || roi->width != 0 || roi->height != 0)) ? roi : nullptr;
    const bool view = sourcePyramid && vl != -2;
    return Selection(native(), firstLayer, lastLayer, stepLayer, r, view ? vf : -1, view ? vl : -1);
}
void ImagePyramid::addLayerFilter(const shared_ptr<ImageFilter>& filter) {
    if (sourcePyramid) throw std::logic_error("ImagePyramid: layer filters on top of a source pyramid are not available on this backend");
    if (auto g = std::dynamic_pointer_cast<GradientFilter>(filter)) gradient = g;
    else if (auto b = std::dynamic_pointer_cast<GradientBinningFilter>(filter)) binning = b;
    else if (auto l = std::dynamic_pointer_cast<LbpFilter>(filter)) lbp = l;
    else throw std::logic_error("ImagePyramid: unsupported layer filter (GradientFilter, GradientBinningFilter, LbpFilter are available)");
    applyLayerFilterConfig();
}
void ImagePyramid::applyLayerFilterConfig() {
    if (gradient && binning)
    {
        check(fd_pyramid_set_layer_filter(handle, FD_LAYER_GRADBIN, (int)binning->bins, binning->signedGradients, binning->interpolate,
                                          gradient->kernelSize, 0));
        check(fd_pyramid_set_gradient_blur(handle, gradient->blurKernelSize > 0 ? gradient->blurKernelSize : 0));
    }
    else if (lbp)
        check(fd_pyramid_set_layer_filter(handle, FD_LAYER_LBP, 0, 0, 0, 1, (int)lbp->type));
    version = Version();
}
void ImagePyramid::update(const Mat& image) { update(make_shared<VersionedImage>(image)); }
void ImagePyramid::update(const shared_ptr<VersionedImage>& image) {
    if (sourcePyramid) {   // ImagePyramid.cpp:121-124: the source pyramid is updated (once per image version), this one follows
        sourcePyramid->update(image);
        update();
        return;
    }
    if ((gradient != nullptr) != (binning != nullptr))
        throw std::logic_error("ImagePyramid: GradientFilter and GradientBinningFilter have to be added together");
    sourceImage = image;
    if (version == image->getVersion()) return;   // ImagePyramid.cpp:150
    Mat src = contiguous(image->getData());
    check(fd_pyramid_update(handle, src.data, src.cols, src.rows, src.channels(), 0));
    ++g_pyramidBuilds;
    imageSize = cv::Size(src.cols, src.rows);
    version = image->getVersion();
    layersValid = false;
}
const vector<shared_ptr<ImagePyramidLayer>>& ImagePyramid::getLayers() const {
    if (sourcePyramid) {
        if (!layersValid) {
            layers.clear();
            for (const auto& l : sourcePyramid->getLayers())
                if (l->getScaleFactor() >= minScaleFactor && l->getScaleFactor() <= maxScaleFactor) layers.push_back(l);
            layersValid = true;
        }
        return layers;
    }
    if (!layersValid) {
        layers.clear();
        const int n = fd_pyramid_layer_count(handle);
        for (int i = 0; i < n; ++i) {
            int index, w, h, ch;
            double scale;
            check(fd_pyramid_layer_info(handle, i, &index, &scale, &w, &h, &ch));
            Mat img(h, w, CV_MAKETYPE(CV_8U, ch));
            check(fd_pyramid_layer_download(handle, i, img.data));
            layers.push_back(make_shared<ImagePyramidLayer>(index, scale, (double)w / imageSize.width, (double)h / imageSize.height, img));
        }
        layersValid = true;
    }
    return layers;
}
const shared_ptr<ImagePyramidLayer> ImagePyramid::getLayer(int index) const {
    const auto& ls = getLayers();
    if (ls.empty()) return shared_ptr<ImagePyramidLayer>();
    int real = index - ls.front()->getIndex();
    if (real < 0 || real >= (int)ls.size()) return shared_ptr<ImagePyramidLayer>();
    return ls[real];
}
vector<std::pair<int, double>> ImagePyramid::getLayerScales() const {
    vector<std::pair<int, double>> out;
    if (sourcePyramid) {
        for (const auto& sc : sourcePyramid->getLayerScales())
            if (sc.second >= minScaleFactor && sc.second <= maxScaleFactor) out.push_back(sc);
        return out;
    }
    for (int i = 0; i < fd_pyramid_layer_count(handle); ++i) {
        int index, w, h, ch; double scale;
        fd_pyramid_layer_info(handle, i, &index, &scale, &w, &h, &ch);
        out.emplace_back(index, scale);
    }
    return out;
}
vector<cv::Size> ImagePyramid::getLayerSizes() const {
    vector<cv::Size> out;
    if (sourcePyramid) {
        auto scales = sourcePyramid->getLayerScales();
        auto sizes = sourcePyramid->getLayerSizes();
        for (size_t i = 0; i < scales.size(); ++i)
            if (scales[i].second >= minScaleFactor && scales[i].second <= maxScaleFactor) out.push_back(sizes[i]);
        return out;
    }
    for (int i = 0; i < fd_pyramid_layer_count(handle); ++i) {
        int index, w, h, ch; double scale;
        fd_pyramid_layer_info(handle, i, &index, &scale, &w, &h, &ch);
        out.push_back(cv::Size(w, h));
    }
    return out;
}

namespace filtering {
FhogFilter::FhogFilter(int cellSize, int unsignedBinCount, bool interpolateBins, bool interpolateCells, float alpha)
    : cellSize(cellSize), unsignedBinCount(unsignedBinCount), interpolateBins(interpolateBins), interpolateCells(interpolateCells), alpha(alpha) {
    if (unsignedBinCount < 1) throw std::invalid_argument("FhogFilter: unsignedBinCount must be bigger than zero, but was: " + std::to_string(unsignedBinCount));
    if (alpha <= 0) throw std::invalid_argument("FhogAggregationFilter: alpha must be bigger than zero, but was: " + std::to_string(alpha));
}
Mat FhogFilter::applyTo(const Mat& image, Mat& descriptors) const {
    if (image.type() != CV_8UC1 && image.type() != CV_8UC3)
        throw std::invalid_argument("FhogFilter: the image type must be CV_8UC1 or CV_8UC3, but was " + std::to_string(image.type()));
    Mat src = image.isContinuous() ? image : image.clone();
    fd_fhog_params fp = {cellSize, unsignedBinCount, interpolateBins, interpolateCells, alpha};
    const int rows = src.rows / cellSize, cols = src.cols / cellSize, D = 3 * unsignedBinCount + 4;
    descriptors.create(rows, cols * D, CV_32FC1);   // the compat Mat has no CV_32FC(n) with n > 4: channels are interleaved in the row
    if (rows > 0 && cols > 0)
        check(fd_fhog_image_channels(context(), src.ptr<uchar>(0), src.cols, src.rows, src.channels(), &fp, descriptors.ptr<float>(0)));
    return descriptors;
}
}  // namespace filtering

// ---- DirectPyramidFeatureExtractor ----------------------------------------------------------------
DirectPyramidFeatureExtractor::DirectPyramidFeatureExtractor(shared_ptr<ImagePyramid> pyramid, int width, int height)
    : pyramid(pyramid), patchWidth(width), patchHeight(height) {}
void DirectPyramidFeatureExtractor::addPatchFilter(shared_ptr<ImageFilter> filter) {
    if (auto rf = std::dynamic_pointer_cast<ReshapingFilter>(filter)) {   // the fused kernels work on flat vectors: nothing to do
        if (rf->rows != 1 || rf->channels > 1) throw std::logic_error("DirectPyramidFeatureExtractor: ReshapingFilter(1) (row vectors) is available in a fused chain");
        reshaping = rf;
        chain->add(filter);
        return;
    }
    if (auto h = std::dynamic_pointer_cast<HistEq64Filter>(filter)) histeq = h;
    else if (auto wf = std::dynamic_pointer_cast<WhiteningFilter>(filter)) {
        if (whiStage != 0) throw std::logic_error("DirectPyramidFeatureExtractor: WhiteningFilter must be the first filter of the whi chain");
        whitening = wf; whiStage = 1;
    } else if (auto ef = std::dynamic_pointer_cast<HistogramEqualizationFilter>(filter)) {
        if (whiStage != 0 && whiStage != 1) throw std::logic_error("DirectPyramidFeatureExtractor: unsupported patch filter order");
        equalization = ef;
        if (whiStage == 1) whiStage = 2;
    } else if (auto cf = std::dynamic_pointer_cast<ConversionFilter>(filter)) {
        if (whiStage == 0 && !hist && cf->type == CV_32F) {   // u8 feature space -> f32 (input of an RVM / f32 SVM)
            conversion = cf;
            chain->add(filter);
            return;
        }
        if (whiStage != 2 || cf->type != CV_32F || cf->alpha != 1.0 / 127.5 || cf->beta != -1.0)
            throw std::logic_error("DirectPyramidFeatureExtractor: ConversionFilter is available as ConversionFilter(CV_32F, alpha, beta) after a u8 "
                                   "feature space, or as ConversionFilter(CV_32F, 1.0/127.5, -1.0) inside the whi chain");
        whiStage = 3;
    } else if (auto uf = std::dynamic_pointer_cast<UnitNormFilter>(filter)) {
        if (whiStage != 3 || uf->normType != cv::NORM_L2)
            throw std::logic_error("DirectPyramidFeatureExtractor: UnitNormFilter is available as UnitNormFilter(cv::NORM_L2) at the end of the whi chain only");
        whiStage = 4;
    } else if (auto hf = std::dynamic_pointer_cast<HistogramFilter>(filter)) {
        hist = hf;
        auto g = std::dynamic_pointer_cast<HogFilter>(filter);
        // the tuned k_hog_tile path covers the square, non-interpolating HogFilter; everything else runs k_hist_features
        hog = (g && !g->interpolate && g->cellWidth == g->cellHeight && g->blockWidth == g->blockHeight) ? g : nullptr;
    } else throw std::logic_error("DirectPyramidFeatureExtractor: unsupported patch filter (HistEq64Filter and the HistogramFilter family are available)");
    chain->add(filter);
}
vector<cv::Size> DirectPyramidFeatureExtractor::getPatchSizes() const {
    vector<cv::Size> sizes;
    for (const auto& sc : pyramid->getLayerScales())
        sizes.push_back(cv::Size(cv::cvRound(patchWidth / sc.second), cv::cvRound(patchHeight / sc.second)));
    return sizes;
}
int DirectPyramidFeatureExtractor::getLayerIndex(int width, int) const {
    double scaleFactor = (double)patchWidth / (double)width;
    int idx = (int)std::round(std::log(scaleFactor) / std::log(pyramid->getIncrementalScaleFactor()));
    return pyramid->getLayer(idx) ? idx : -1;
}
shared_ptr<Patch> DirectPyramidFeatureExtractor::extractFromLayer(const ImagePyramidLayer& layer, cv::Rect b) const {
    const Mat& image = layer.getScaledImage();
    if (b.x < 0 || b.y < 0 || b.x + b.width > image.cols || b.y + b.height > image.rows) return shared_ptr<Patch>();
    int ow = layer.getOriginal(b.width), oh = layer.getOriginal(b.height);
    int ox = layer.getOriginal(b.x) + ow / 2, oy = layer.getOriginal(b.y) + oh / 2;
    Mat data = Mat(image, b).clone();
    if (hasPatchFilters()) data = chain->applyTo(data);   // per Mat, in the order the filters were added (DirectPyramidFeatureExtractor.cpp:75-123)
    return make_shared<Patch>(ox, oy, ow, oh, data);
}
shared_ptr<Patch> DirectPyramidFeatureExtractor::extract(int x, int y, int width, int height) const {
    int idx = getLayerIndex(width, height);
    auto layer = idx < 0 ? shared_ptr<ImagePyramidLayer>() : pyramid->getLayer(idx);
    if (!layer) return shared_ptr<Patch>();
    return extractFromLayer(*layer, cv::Rect(layer->getScaled(x - width / 2), layer->getScaled(y - height / 2), patchWidth, patchHeight));
}
shared_ptr<Patch> DirectPyramidFeatureExtractor::extract(int layerIndex, int x, int y) const {
    auto layer = pyramid->getLayer(layerIndex);
    if (!layer) return shared_ptr<Patch>();
    return extractFromLayer(*layer, cv::Rect(x - patchWidth / 2, y - patchHeight / 2, patchWidth, patchHeight));
}
vector<shared_ptr<Patch>> DirectPyramidFeatureExtractor::extract(int stepX, int stepY, cv::Rect roi, int firstLayer, int lastLayer,
                                                                 int stepLayer) const {
    // DirectPyramidFeatureExtractor.cpp:75-123: the layer sub-range, the layer step and the region of interest apply to the window
    // enumeration of every chain (fd_pyramid_select); a pyramid built on another pyramid adds its scale range
    auto selection = pyramid->select(firstLayer, lastLayer, stepLayer, &roi);
    int64_t n = 0;
    check(fd_pyramid_window_count(pyramid->native(), patchWidth, patchHeight, stepX, stepY, nullptr, &n));
    vector<int32_t> wins((size_t)n * 7);
    if (n) check(fd_pyramid_windows(pyramid->native(), patchWidth, patchHeight, stepX, stepY, nullptr, wins.data(), n, &n));
    vector<shared_ptr<Patch>> patches;
    patches.reserve((size_t)n);
    if (hog) {
        fd_hog_params hp = {patchWidth, patchHeight, stepX, stepY, hog->binCount, hog->cellWidth, hog->blockWidth, hog->signedAndUnsigned};
        const int F = fd_hog_feature_length(&hp);
        Mat all((int)n, F, CV_32FC1);
        int64_t cnt = 0;
        if (n) check(fd_extract_hog(context(), pyramid->native(), &hp, all.ptr<float>(0), n, &cnt));
        for (int64_t i = 0; i < n; ++i) {
            const int32_t* w = &wins[7 * i];
            patches.push_back(make_shared<Patch>(w[3], w[4], w[5], w[6], Mat(all, cv::Rect(0, (int)i, F, 1))));
        }
        return patches;
    }
    if (whiStage != 0 && whiStage != 4) throw std::logic_error("DirectPyramidFeatureExtractor: incomplete whi filter chain");
    if (whiStage == 4) {
        fd_whi_params wp = {patchWidth, patchHeight, stepX, stepY, whitening->alpha, whitening->cutoffFrequency};
        const int F = patchWidth * patchHeight;
        Mat all((int)std::max<int64_t>(n, 1), F, CV_32FC1);
        int64_t cnt = 0;
        if (n) check(fd_extract_whi(context(), pyramid->native(), &wp, all.ptr<float>(0), n, &cnt));
        for (int64_t i = 0; i < n; ++i) {
            const int32_t* w = &wins[7 * i];
            Mat data(patchHeight, patchWidth, CV_32FC1);
            std::memcpy(data.data, all.ptr<float>((int)i), sizeof(float) * (size_t)F);
            patches.push_back(make_shared<Patch>(w[3], w[4], w[5], w[6], data));
        }
        return patches;
    }
    if (hist) {
        fd_hist_params hp = hist_params_of(*hist, patchWidth, patchHeight, stepX, stepY);
        int index, lw, lh, ch = 1; double scale;
        if (fd_pyramid_layer_count(pyramid->native()) > 0) fd_pyramid_layer_info(pyramid->native(), 0, &index, &scale, &lw, &lh, &ch);
        const int F = fd_hist_feature_length(&hp, ch);
        if (F < 0) throw std::invalid_argument("DirectPyramidFeatureExtractor: invalid histogram filter parameters for this patch size");
        Mat all((int)std::max<int64_t>(n, 1), F, CV_32FC1);
        int64_t cnt = 0;
        if (n) check(fd_extract_hist(context(), pyramid->native(), &hp, all.ptr<float>(0), n, &cnt));
        for (int64_t i = 0; i < n; ++i) {
            const int32_t* w = &wins[7 * i];
            patches.push_back(make_shared<Patch>(w[3], w[4], w[5], w[6], Mat(all, cv::Rect(0, (int)i, F, 1))));
        }
        return patches;
    }
    const auto& layers = (pyramid->getSourcePyramid() ? pyramid->getSourcePyramid() : pyramid)->getLayers();   // w[0] = position in the underlying pyramid
    const int d = patchWidth * patchHeight;
    Mat raw((int)std::max<int64_t>(n, 1), d, CV_8UC1), eq;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t* w = &wins[7 * i];
        const Mat& img = layers[w[0]]->getScaledImage();
        for (int y = 0; y < patchHeight; ++y) std::memcpy(raw.ptr<uchar>((int)i) + y * patchWidth, img.ptr<uchar>(w[2] + y) + w[1], patchWidth);
    }
    if (histeq && n) {
        eq.create((int)n, d, CV_8UC1);
        check(fd_histeq64_batch(context(), raw.data, n, patchWidth, patchHeight, eq.data));
    } else if (equalization && n) {   // feature space "histeq" (ffpDetectApp.cpp:446-448)
        eq.create((int)n, d, CV_8UC1);
        check(fd_equalize_hist_batch(context(), raw.data, n, patchWidth, patchHeight, eq.data));
    } else {
        eq = raw;
    }
    for (int64_t i = 0; i < n; ++i) {
        const int32_t* w = &wins[7 * i];
        if (conversion) {   // cv::Mat::convertTo(CV_32F, alpha, beta): float(u8) * float(alpha) + float(beta)
            Mat data(patchHeight, patchWidth, CV_32FC1);
            const uchar* srcp = eq.ptr<uchar>((int)i);
            float* dstp = data.ptr<float>(0);
            const float a = (float)conversion->alpha, b = (float)conversion->beta;
            for (int k = 0; k < d; ++k) dstp[k] = (float)srcp[k] * a + b;
            patches.push_back(make_shared<Patch>(w[3], w[4], w[5], w[6], data));
            continue;
        }
        Mat data(patchHeight, patchWidth, CV_8UC1);
        std::memcpy(data.data, eq.ptr<uchar>((int)i), d);
        patches.push_back(make_shared<Patch>(w[3], w[4], w[5], w[6], data));
    }
    return patches;
}

// ---- FilteringPyramidFeatureExtractor (FilteringPyramidFeatureExtractor.hpp:20-90) ------------------------------------------
FilteringPyramidFeatureExtractor::FilteringPyramidFeatureExtractor(shared_ptr<PyramidFeatureExtractor> extractor)
    : extractor(extractor), patchFilter(make_shared<ChainedFilter>()) {
    if (!extractor) throw std::invalid_argument("FilteringPyramidFeatureExtractor: the underlying extractor must not be null");
    auto direct = std::dynamic_pointer_cast<DirectPyramidFeatureExtractor>(extractor);
    if (direct && !direct->hasPatchFilters())
        fused = make_shared<DirectPyramidFeatureExtractor>(direct->getPyramid(), direct->getPatchWidth(), direct->getPatchHeight());
}
void FilteringPyramidFeatureExtractor::addPatchFilter(shared_ptr<ImageFilter> filter) {
    patchFilter->add(filter);
    if (fused) {
        try { fused->addPatchFilter(filter); }
        catch (const std::logic_error&) { fused.reset(); }   // not a chain the kernels fuse: per-Mat composition from now on
    }
}
shared_ptr<Patch> FilteringPyramidFeatureExtractor::extract(int x, int y, int width, int height) const {
    shared_ptr<Patch> patch = extractor->extract(x, y, width, height);
    if (patch) patchFilter->applyInPlace(patch->getData());
    return patch;
}
vector<shared_ptr<Patch>> FilteringPyramidFeatureExtractor::extract(int stepX, int stepY, cv::Rect roi, int firstLayer, int lastLayer, int stepLayer) const {
    if (fused) return fused->extract(stepX, stepY, roi, firstLayer, lastLayer, stepLayer);   // batched kernels, same values
    vector<shared_ptr<Patch>> patches = extractor->extract(stepX, stepY, roi, firstLayer, lastLayer, stepLayer);
    for (shared_ptr<Patch>& patch : patches) patchFilter->applyInPlace(patch->getData());
    return patches;
}
shared_ptr<Patch> FilteringPyramidFeatureExtractor::extract(int layer, int x, int y) const {
    shared_ptr<Patch> patch = extractor->extract(layer, x, y);
    if (patch) patchFilter->applyInPlace(patch->getData());
    return patch;
}

}  // namespace imageprocessing

// =================================================================================================
namespace classification {

double Kernel::compute(const Mat& lhs, const Mat& rhs) const {
    if (!lhs.isContinuous() || !rhs.isContinuous()) throw std::invalid_argument("Kernel: arguments have to be continuous");
    if (lhs.flags != rhs.flags) throw std::invalid_argument("Kernel: arguments have to have the same type");
    if (lhs.total() * lhs.channels() != rhs.total() * rhs.channels()) throw std::invalid_argument("Kernel: arguments have to have the same length");
    if (lhs.depth() != CV_8U && lhs.depth() != CV_32F) throw std::invalid_argument("Kernel: arguments have to be of depth CV_8U or CV_32F on this backend");
    fd_svm_model m;
    std::memset(&m, 0, sizeof(m));
    m.kernel = abiKernel();
    abiParams(m.p0, m.p1, m.p2);
    m.num_sv = 1;
    m.dim = (int)(rhs.total() * rhs.channels());
    m.dtype = rhs.depth() == CV_8U ? FD_DTYPE_U8 : FD_DTYPE_F32;
    m.support_vectors = rhs.data;
    const float one = 1.f;
    m.coefficients = &one;
    fd_svm* h = nullptr;
    check(fd_svm_create(context(), &m, &h));
    double out = 0;
    int rc = fd_svm_distance_batch(context(), h, lhs.data, 1, &out);
    fd_svm_destroy(h);
    check(rc);
    return out;
}

SvmClassifier::SvmClassifier(shared_ptr<Kernel> kernel) : VectorMachineClassifier(kernel), handle(nullptr), dirty(true) {}
SvmClassifier::~SvmClassifier() { fd_svm_destroy(handle); }
void SvmClassifier::setSvmParameters(vector<Mat> sv, vector<float> coeff, double b) {
    supportVectors = sv;
    coefficients = coeff;
    bias = (float)b;
    dirty = true;
}
const fd_svm* SvmClassifier::native(double la, double lb) const {
    if (dirty || !handle) {
        fd_svm_destroy(handle);
        handle = nullptr;
        if (supportVectors.empty() || supportVectors.size() != coefficients.size())
            throw std::runtime_error("SvmClassifier: no support vectors / coefficient count mismatch");
        const Mat& first = supportVectors.front();
        const int dim = (int)(first.total() * first.channels());
        const int depth = first.depth();
        if (depth != CV_8U && depth != CV_32F) throw std::runtime_error("SvmClassifier: support vectors must be CV_8U or CV_32F on this backend");
        const size_t es = depth == CV_8U ? 1 : 4;
        vector<unsigned char> flat(supportVectors.size() * dim * es);
        for (size_t i = 0; i < supportVectors.size(); ++i) {
            Mat s = contiguous(supportVectors[i]);
            if ((int)(s.total() * s.channels()) != dim || s.depth() != depth) throw std::runtime_error("SvmClassifier: inconsistent support vectors");
            std::memcpy(flat.data() + i * dim * es, s.data, dim * es);
        }
        fd_svm_model m;
        std::memset(&m, 0, sizeof(m));
        m.kernel = kernel->abiKernel();
        kernel->abiParams(m.p0, m.p1, m.p2);
        m.num_sv = (int)supportVectors.size();
        m.dim = dim;
        m.dtype = depth == CV_8U ? FD_DTYPE_U8 : FD_DTYPE_F32;
        m.support_vectors = flat.data();
        m.coefficients = coefficients.data();
        m.bias = bias;
        m.threshold = threshold;
        m.logistic_a = la;
        m.logistic_b = lb;
        check(fd_svm_create(context(), &m, &handle));
        dirty = false;
    }
    return handle;
}
double SvmClassifier::computeHyperplaneDistance(const Mat& featureVector) const {
    Mat x = contiguous(featureVector);
    double out = 0;
    check(fd_svm_distance_batch(context(), native(), x.data, 1, &out));
    return out;
}
bool SvmClassifier::classify(const Mat& featureVector) const { return classify(computeHyperplaneDistance(featureVector)); }
std::pair<bool, double> SvmClassifier::getConfidence(const Mat& featureVector) const {
    double d = computeHyperplaneDistance(featureVector);
    return classify(d) ? std::make_pair(true, d) : std::make_pair(false, -d);
}
void SvmClassifier::store(std::ofstream& file) {   // SvmClassifier.cpp:68-107
    if (!file) throw std::runtime_error("SvmClassifier: Cannot write into stream");
    file << "Kernel ";
    if (dynamic_cast<LinearKernel*>(kernel.get())) file << "Linear\n";
    else if (auto* k = dynamic_cast<PolynomialKernel*>(kernel.get())) file << "Polynomial " << k->getDegree() << ' ' << k->getConstant() << ' ' << k->getAlpha() << '\n';
    else if (auto* k = dynamic_cast<RbfKernel*>(kernel.get())) file << "RBF " << std::setprecision(17) << k->getGamma() << '\n';
    else if (dynamic_cast<HistogramIntersectionKernel*>(kernel.get())) file << "HIK\n";
    else throw std::runtime_error("SvmClassifier: cannot write kernel parameters (unknown kernel type)");
    file << std::setprecision(9) << "Bias " << getBias() << '\n';
    file << "Coefficients " << coefficients.size() << '\n';
    for (float c : coefficients) file << c << '\n';
    const Mat& v = supportVectors.front();
    file << "SupportVectors " << supportVectors.size() << ' ' << v.rows << ' ' << v.cols << ' ' << v.channels() << ' ' << v.depth() << '\n';
    for (const Mat& s : supportVectors) {
        const size_t n = s.total() * s.channels();
        Mat c = contiguous(s);
        for (size_t i = 0; i < n; ++i) {
            if (v.depth() == CV_8U) file << (int)c.ptr<uchar>(0)[i] << ' ';
            else file << c.ptr<float>(0)[i] << ' ';
        }
        file << '\n';
    }
}
shared_ptr<SvmClassifier> SvmClassifier::load(std::ifstream& file) {   // SvmClassifier.cpp:109-159
    if (!file) throw std::runtime_error("SvmClassifier: Cannot read from stream");
    string tmp, kernelType;
    file >> tmp >> kernelType;
    shared_ptr<Kernel> kernel;
    if (kernelType == "Linear") kernel.reset(new LinearKernel());
    else if (kernelType == "Polynomial") { int degree; double constant, scale; file >> degree >> constant >> scale; kernel.reset(new PolynomialKernel(scale, constant, degree)); }
    else if (kernelType == "RBF") { double gamma; file >> gamma; kernel.reset(new RbfKernel(gamma)); }
    else if (kernelType == "HIK") kernel.reset(new HistogramIntersectionKernel());
    else throw std::runtime_error("SvmClassifier: Invalid kernel type: " + kernelType);
    auto svm = make_shared<SvmClassifier>(kernel);
    file >> tmp >> svm->bias;
    size_t count;
    file >> tmp >> count;
    svm->coefficients.resize(count);
    for (size_t i = 0; i < count; ++i) file >> svm->coefficients[i];
    int rows, cols, channels, depth;
    file >> tmp >> count >> rows >> cols >> channels >> depth;
    if (depth != CV_8U && depth != CV_32F)
        throw std::runtime_error("SvmClassifier: cannot load support vectors of depth other than CV_8U or CV_32F on this backend");
    for (size_t i = 0; i < count; ++i) {
        Mat v(rows, cols, CV_MAKETYPE(depth, channels));
        const size_t n = (size_t)rows * cols * channels;
        for (size_t k = 0; k < n; ++k) {
            if (depth == CV_8U) { int t; file >> t; v.ptr<uchar>(0)[k] = (uchar)t; }
            else file >> v.ptr<float>(0)[k];
        }
        svm->supportVectors.push_back(v);
    }
    if (!file) throw std::runtime_error("SvmClassifier: Invalid classifier file");
    svm->dirty = true;
    return svm;
}

std::pair<bool, double> ProbabilisticSvmClassifier::getProbability(const Mat& featureVector) const {
    return getProbability(svm->computeHyperplaneDistance(featureVector));
}
std::pair<bool, double> ProbabilisticSvmClassifier::getProbability(double d) const {   // ProbabilisticSvmClassifier.cpp:54-58
    double fABp = logisticA + logisticB * d;
    double p = fABp >= 0 ? std::exp(-fABp) / (1.0 + std::exp(-fABp)) : 1.0 / (1.0 + std::exp(fABp));
    return std::make_pair(svm->classify(d), p);
}
void ProbabilisticSvmClassifier::store(std::ofstream& file) {
    svm->store(file);
    file << std::setprecision(17) << "Logistic " << logisticA << ' ' << logisticB << '\n';
}
shared_ptr<ProbabilisticSvmClassifier> ProbabilisticSvmClassifier::load(std::ifstream& file) {
    auto svm = SvmClassifier::load(file);
    string tmp;
    double a, b;
    file >> tmp >> a >> b;
    return make_shared<ProbabilisticSvmClassifier>(svm, a, b);
}
// SvmClassifier::loadFromText (SvmClassifier.cpp:161-239): the line-based text format of the reference's polynomial SVMs --
//   FullPolynomial <degree> <constant> <scale> / Number of SV : n / Dim of SV : d / B0 : b / alphas[i]=a (n lines) /
//   "Support vectors: " / n lines of d floats -- CV_32F support vectors, PolynomialKernel(scale, constant, degree)
shared_ptr<SvmClassifier> SvmClassifier::loadFromText(const string& classifierFilename) {
    std::ifstream file(classifierFilename.c_str());
    if (!file.is_open()) throw std::runtime_error("SvmClassifier: Invalid classifier file");
    string line;
    if (!std::getline(file, line)) throw std::runtime_error("SvmClassifier: Invalid classifier file");
    shared_ptr<Kernel> kernel;
    {
        std::istringstream lineStream(line);
        string kernelType;
        lineStream >> kernelType;
        if (kernelType != "FullPolynomial") throw std::runtime_error("SvmClassifier: Invalid kernel type: " + kernelType);
        int degree = 0;
        double constant = 0, scale = 0;
        lineStream >> degree >> constant >> scale;
        kernel.reset(new PolynomialKernel(scale, constant, degree));
    }
    auto svm = make_shared<SvmClassifier>(kernel);
    auto next = [&]() { if (!std::getline(file, line)) throw std::runtime_error("SvmClassifier: Invalid classifier file"); };
    int svCount = 0, dimensionCount = 0;
    float bias = 0;
    next();
    if (std::sscanf(line.c_str(), "Number of SV : %d", &svCount) != 1 || svCount < 1) throw std::runtime_error("SvmClassifier: Invalid classifier file");
    next();
    if (std::sscanf(line.c_str(), "Dim of SV : %d", &dimensionCount) != 1 || dimensionCount < 1) throw std::runtime_error("SvmClassifier: Invalid classifier file");
    next();
    if (std::sscanf(line.c_str(), "B0 : %f", &bias) != 1) throw std::runtime_error("SvmClassifier: Invalid classifier file");
    vector<float> coefficients((size_t)svCount, 0.f);
    for (int i = 0; i < svCount; ++i) {
        float alpha;
        int index;
        next();
        if (std::sscanf(line.c_str(), "alphas[%d]=%f", &index, &alpha) != 2 || index < 0 || index >= svCount)
            throw std::runtime_error("SvmClassifier: Invalid classifier file");
        coefficients[(size_t)index] = alpha;
    }
    next();   // "Support vectors: "
    vector<Mat> supportVectors;
    supportVectors.reserve((size_t)svCount);
    for (int i = 0; i < svCount; ++i) {
        Mat vec(1, dimensionCount, CV_32FC1);
        next();
        std::istringstream lineStream(line);
        float* values = vec.ptr<float>(0);
        for (int j = 0; j < dimensionCount; ++j)
            if (!(lineStream >> values[j])) throw std::runtime_error("SvmClassifier: Invalid classifier file");
        supportVectors.push_back(vec);
    }
    svm->setSvmParameters(supportVectors, coefficients, (double)bias);
    return svm;
}

// ProbabilisticSvmClassifier.cpp:80-104.  Non-.mat files are text models: the reference's own text format (loadFromText:
// polynomial SVMs, first token "FullPolynomial") -- or, as the stand-in for the Matlab models the reference loads through libmat
// (RBF / HIK SVMs on u8 patches exist upstream only as .mat files), the stream format of SvmClassifier::store (first token
// "Kernel"; see DESIGN.md section "model formats").
shared_ptr<ProbabilisticSvmClassifier> ProbabilisticSvmClassifier::load(const boost::property_tree::ptree& subtree) {
    string classifierFile = subtree.get<string>("classifierFile");
    if (classifierFile.size() > 4 && classifierFile.substr(classifierFile.size() - 4) == ".mat")
        throw std::runtime_error("ProbabilisticSvmClassifier: Cannot load a Matlab classifier (the reference needs libmat; this backend reads text models)");
    shared_ptr<ProbabilisticSvmClassifier> psvm;
    {
        std::ifstream probe(classifierFile.c_str());
        if (!probe.is_open()) throw std::runtime_error("SvmClassifier: Invalid classifier file");
        string first;
        probe >> first;
        if (first == "Kernel") {
            std::ifstream f(classifierFile.c_str());
            psvm = load(f);
        } else {
            psvm = make_shared<ProbabilisticSvmClassifier>(SvmClassifier::loadFromText(classifierFile));
        }
    }
    double la = subtree.get("logisticA", 0.0), lb = subtree.get("logisticB", 0.0);
    if (la != 0.0 && lb != 0.0) psvm->setLogisticParameters(la, lb);
    psvm->getSvm()->setThreshold(subtree.get("threshold", 0.0f));
    return psvm;
}

// ---- WVM ------------------------------------------------------------------------------------------
WvmClassifier::WvmClassifier() : VectorMachineClassifier(nullptr), limitReliabilityFilter(0.f), handle(nullptr), dirty(true) {}
WvmClassifier::~WvmClassifier() { fd_wvm_destroy(handle); }
void WvmClassifier::setModel(const Model& m) {
    model = m;
    bias = m.bias;
    setNumUsedFilters(m.num_used);
    setLimitReliabilityFilter(limitReliabilityFilter);
}
void WvmClassifier::setNumUsedFilters(int var) {   // WvmClassifier.cpp:151-158
    model.num_used = (var > model.num_filters || var == 0) ? model.num_filters : var;
    dirty = true;
}
void WvmClassifier::setLimitReliabilityFilter(float var) {   // WvmClassifier.cpp:165-181
    limitReliabilityFilter = var;
    hierarchicalThresholds = model.thresholdsFromFile;
    if (var != 0.0f)
        for (float& t : hierarchicalThresholds) t = t + limitReliabilityFilter;
    dirty = true;
}
const fd_wvm* WvmClassifier::native(double la, double lb) const {
    if (dirty || !handle) {
        fd_wvm_destroy(handle);
        handle = nullptr;
        fd_wvm_model m;
        std::memset(&m, 0, sizeof(m));
        m.filter_w = model.filter_w; m.filter_h = model.filter_h; m.num_filters = model.num_filters; m.num_used = model.num_used;
        m.num_per_level = model.num_per_level; m.basis_param = model.basis_param; m.bias = model.bias;
        m.thresholds = hierarchicalThresholds.data(); m.hk_weights = model.hk_weights.data(); m.pp = model.pp.data();
        m.val_off = model.val_off.data(); m.val = model.val.data(); m.rec_off = model.rec_off.data(); m.rects = model.rects.data();
        m.logistic_a = la; m.logistic_b = lb;
        m.num_vals = (int32_t)model.val.size(); m.num_rects = (int32_t)(model.rects.size() / 4);   // a truncated model file is rejected
        // a file with too short offset tables would be read out of bounds before the library sees it
        if ((int)model.val_off.size() != model.num_filters + 1 || model.val_off.back() < 0 || (int)model.rec_off.size() != model.val_off.back() + 1)
            throw std::invalid_argument("WvmClassifier: inconsistent offset tables in the model");
        check(fd_wvm_create(context(), &m, &handle));
        dirty = false;
    }
    return handle;
}
std::pair<int, double> WvmClassifier::computeHyperplaneDistance(const Mat& featureVector) const {
    if (featureVector.depth() != CV_8U || (int)(featureVector.total() * featureVector.channels()) != model.filter_w * model.filter_h)
        throw std::invalid_argument("WvmClassifier: feature vector must be a CV_8U patch of the filter size");
    Mat x = contiguous(featureVector);
    int32_t level = 0;
    float fout = 0;
    check(fd_wvm_eval_batch(context(), native(), x.data, 1, &level, &fout));
    return std::make_pair((int)level, (double)fout);
}
bool WvmClassifier::classify(std::pair<int, double> lad) const {   // WvmClassifier.cpp:91-98
    return lad.first + 1 == model.num_filters && lad.second >= hierarchicalThresholds[lad.first];
}
bool WvmClassifier::classify(const Mat& featureVector) const { return classify(computeHyperplaneDistance(featureVector)); }
std::pair<bool, double> WvmClassifier::getConfidence(const Mat& featureVector) const {
    auto lad = computeHyperplaneDistance(featureVector);
    return classify(lad) ? std::make_pair(true, lad.second) : std::make_pair(false, -lad.second);
}
// Binary model file "FDWVM1": int32 header {fw, fh, F, used, nper, nval, nrect}, float basis, float bias,
// then thresholds[F] f32, hk[F*F] f32, pp[F] f64, val_off[F+1] i32, val[nval] f64, rec_off[nval+1] i32, rects[4*nrect] u8
shared_ptr<WvmClassifier> WvmClassifier::loadFromFile(const string& filename) {
    std::ifstream f(filename.c_str(), std::ios::binary);
    if (!f.is_open()) throw std::invalid_argument("WvmClassifier: Could not open the provided classifier filename: " + filename);
    char magic[8];
    f.read(magic, 8);
    if (std::memcmp(magic, "FDWVM1\0\0", 8) != 0) throw std::runtime_error("WvmClassifier: not a FDWVM1 model file: " + filename);
    int32_t hdr[7];
    f.read((char*)hdr, sizeof(hdr));
    Model m;
    m.filter_w = hdr[0]; m.filter_h = hdr[1]; m.num_filters = hdr[2]; m.num_used = hdr[3]; m.num_per_level = hdr[4];
    const int nval = hdr[5], nrect = hdr[6], F = m.num_filters;
    if (F < 1 || nval < F || nrect < 0) throw std::runtime_error("WvmClassifier: corrupt model header");
    f.read((char*)&m.basis_param, 4);
    f.read((char*)&m.bias, 4);
    m.thresholdsFromFile.resize(F); m.hk_weights.resize((size_t)F * F); m.pp.resize(F); m.val_off.resize(F + 1); m.val.resize(nval);
    m.rec_off.resize(nval + 1); m.rects.resize((size_t)4 * nrect);
    f.read((char*)m.thresholdsFromFile.data(), 4 * F);
    f.read((char*)m.hk_weights.data(), 4 * (size_t)F * F);
    f.read((char*)m.pp.data(), 8 * F);
    f.read((char*)m.val_off.data(), 4 * (F + 1));
    f.read((char*)m.val.data(), 8 * nval);
    f.read((char*)m.rec_off.data(), 4 * (nval + 1));
    f.read((char*)m.rects.data(), 4 * (size_t)nrect);
    if (!f) throw std::runtime_error("WvmClassifier: truncated model file: " + filename);
    auto wvm = make_shared<WvmClassifier>();
    wvm->setModel(m);
    return wvm;
}

// ---- RVM ------------------------------------------------------------------------------------------
RvmClassifier::RvmClassifier(shared_ptr<Kernel> kernel, bool) : VectorMachineClassifier(kernel) {}
RvmClassifier::~RvmClassifier() { fd_rvm_destroy(handle); }
void RvmClassifier::setNumFiltersToUse(unsigned int numFilters) {   // RvmClassifier.cpp:119-126
    numFiltersToUse = (numFilters == 0 || numFilters > (unsigned int)model.num_filters) ? (unsigned int)model.num_filters : numFilters;
    dirty = true;
}
const fd_rvm* RvmClassifier::native(double la, double lb) const {
    if (dirty || !handle || la != builtA || lb != builtB) {
        fd_rvm_destroy(handle);
        handle = nullptr;
        fd_rvm_model m;
        std::memset(&m, 0, sizeof(m));
        m.kernel = model.kernel; m.p0 = model.p0; m.p1 = model.p1; m.p2 = model.p2;
        m.num_filters = model.num_filters; m.num_used = (int)numFiltersToUse; m.filter_w = model.filter_w; m.filter_h = model.filter_h;
        m.support_vectors = model.support_vectors.data(); m.coefficients = model.coefficients.data(); m.thresholds = model.thresholds.data();
        m.bias = model.bias; m.logistic_a = la; m.logistic_b = lb;
        check(fd_rvm_create(context(), &m, &handle));
        dirty = false; builtA = la; builtB = lb;
    }
    return handle;
}
std::pair<int, double> RvmClassifier::computeHyperplaneDistance(const Mat& featureVector) const {
    if (featureVector.depth() != CV_32F || (int)(featureVector.total() * featureVector.channels()) != model.filter_w * model.filter_h)
        throw std::invalid_argument("RbfKernel: arguments have to have the same type");   // Kernel::compute contract (RbfKernel.hpp:35-38)
    Mat x = contiguous(featureVector);
    int32_t level = 0;
    double dist = 0;
    check(fd_rvm_eval_batch(context(), native(), x.ptr<float>(0), 1, &level, &dist));
    return std::make_pair((int)level, dist);
}
bool RvmClassifier::classify(std::pair<int, double> lad) const {   // RvmClassifier.cpp:66-73
    return lad.first + 1 == (int)numFiltersToUse && lad.second >= model.thresholds[lad.first];
}
bool RvmClassifier::classify(const Mat& featureVector) const { return classify(computeHyperplaneDistance(featureVector)); }
std::pair<bool, double> RvmClassifier::getConfidence(std::pair<int, double> lad) const {
    return classify(lad) ? std::make_pair(true, lad.second) : std::make_pair(false, -lad.second);
}
std::pair<bool, double> RvmClassifier::getConfidence(const Mat& featureVector) const { return getConfidence(computeHyperplaneDistance(featureVector)); }
shared_ptr<RvmClassifier> RvmClassifier::loadFromFile(const string& filename) {
    std::ifstream f(filename.c_str(), std::ios::binary);
    if (!f.is_open()) throw std::invalid_argument("RvmClassifier: Could not open the provided classifier filename: " + filename);
    char magic[8];
    f.read(magic, 8);
    if (std::memcmp(magic, "FDRVM1\0\0", 8) != 0) throw std::runtime_error("RvmClassifier: not a FDRVM1 model file: " + filename);
    int32_t hdr[5];
    double prm[3];
    Model m;
    f.read((char*)hdr, sizeof(hdr));
    f.read((char*)prm, sizeof(prm));
    f.read((char*)&m.bias, 4);
    m.kernel = hdr[0]; m.filter_w = hdr[1]; m.filter_h = hdr[2]; m.num_filters = hdr[3];
    m.p0 = prm[0]; m.p1 = prm[1]; m.p2 = prm[2];
    const int F = m.num_filters, dim = m.filter_w * m.filter_h;
    if (F < 1 || dim < 1 || m.kernel < 0 || m.kernel > 3) throw std::runtime_error("RvmClassifier: corrupt model header");
    m.support_vectors.resize((size_t)F * dim); m.coefficients.resize((size_t)F * (F + 1) / 2); m.thresholds.resize(F);
    f.read((char*)m.support_vectors.data(), 4 * m.support_vectors.size());
    f.read((char*)m.coefficients.data(), 4 * m.coefficients.size());
    f.read((char*)m.thresholds.data(), 4 * (size_t)F);
    if (!f) throw std::runtime_error("RvmClassifier: truncated model file: " + filename);
    shared_ptr<Kernel> kernel;
    if (m.kernel == FD_KERNEL_RBF) kernel = make_shared<RbfKernel>(m.p0);
    else if (m.kernel == FD_KERNEL_POLY) kernel = make_shared<PolynomialKernel>(m.p0, m.p1, (int)m.p2);
    else if (m.kernel == FD_KERNEL_HIK) kernel = make_shared<HistogramIntersectionKernel>();
    else kernel = make_shared<LinearKernel>();
    auto rvm = make_shared<RvmClassifier>(kernel);
    rvm->model = m;
    rvm->bias = m.bias;
    rvm->setNumFiltersToUse((unsigned int)hdr[4]);
    return rvm;
}
shared_ptr<RvmClassifier> RvmClassifier::load(const boost::property_tree::ptree& subtree) {
    string classifierFile = subtree.get<string>("classifierFile");
    if (classifierFile.size() > 4 && classifierFile.substr(classifierFile.size() - 4) == ".mat")
        throw std::runtime_error("RvmClassifier: Cannot load a Matlab classifier (the reference needs libmat; this backend reads the FDRVM1 format)");
    return loadFromFile(classifierFile);
}
std::pair<bool, double> ProbabilisticRvmClassifier::getProbability(const Mat& featureVector) const {
    return getProbability(rvm->computeHyperplaneDistance(featureVector));
}
std::pair<bool, double> ProbabilisticRvmClassifier::getProbability(std::pair<int, double> lad) const {   // ProbabilisticRvmClassifier.cpp:62
    double probability = 1.0f / (1.0f + std::exp(logisticA + logisticB * lad.second));
    return std::make_pair(rvm->classify(lad), probability);
}
shared_ptr<ProbabilisticRvmClassifier> ProbabilisticRvmClassifier::load(const boost::property_tree::ptree& subtree) {
    auto rvm = RvmClassifier::load(subtree);
    auto prvm = make_shared<ProbabilisticRvmClassifier>(rvm, subtree.get("logisticA", 0.00556), subtree.get("logisticB", -2.95));
    if (int nf = subtree.get("numFiltersToUse", 0)) rvm->setNumFiltersToUse((unsigned int)nf);
    return prvm;
}

std::pair<bool, double> ProbabilisticWvmClassifier::getProbability(const Mat& featureVector) const {
    return getProbability(wvm->computeHyperplaneDistance(featureVector));
}
std::pair<bool, double> ProbabilisticWvmClassifier::getProbability(std::pair<int, double> lad) const {   // ProbabilisticWvmClassifier.cpp:52
    double probability = 1.0f / (1.0f + std::exp(logisticA + logisticB * lad.second));
    return std::make_pair(wvm->classify(lad), probability);
}
shared_ptr<ProbabilisticWvmClassifier> ProbabilisticWvmClassifier::load(const boost::property_tree::ptree& subtree) {
    auto wvm = WvmClassifier::loadFromFile(subtree.get<string>("classifierFile"));
    auto pwvm = make_shared<ProbabilisticWvmClassifier>(wvm, subtree.get("logisticA", 0.00556), subtree.get("logisticB", -2.95));
    pwvm->getWvm()->setLimitReliabilityFilter(subtree.get("threshold", 0.0f));
    return pwvm;
}

}  // namespace classification

// =================================================================================================
namespace detection {
using classification::ProbabilisticSvmClassifier;
using classification::ProbabilisticWvmClassifier;
using imageprocessing::DirectPyramidFeatureExtractor;
using imageprocessing::Patch;

static shared_ptr<ClassifiedPatch> to_patch(const fd_detection& d) {
    return make_shared<ClassifiedPatch>(make_shared<Patch>(d.cx, d.cy, d.w, d.h, Mat()), d.positive != 0, d.probability);
}
static fd_detection from_patch(const ClassifiedPatch& p) {
    fd_detection d;
    std::memset(&d, 0, sizeof(d));
    d.cx = p.getPatch()->getX(); d.cy = p.getPatch()->getY(); d.w = p.getPatch()->getWidth(); d.h = p.getPatch()->getHeight();
    d.positive = p.isPositive(); d.probability = p.getProbability();
    return d;
}

vector<shared_ptr<ClassifiedPatch>> OverlapElimination::eliminate(vector<shared_ptr<ClassifiedPatch>>& classifiedPatches) {
    vector<shared_ptr<ClassifiedPatch>> out;
    if (classifiedPatches.empty()) return out;
    vector<fd_detection> dets;
    for (const auto& p : classifiedPatches) dets.push_back(from_patch(*p));
    vector<int32_t> keep(dets.size());
    int n = 0;
    if (fd_overlap_elimination(dets.data(), (int)dets.size(), dist, ratio, keep.data(), &n) != FD_OK)
        throw std::runtime_error("OverlapElimination: invalid arguments");
    for (int i = 0; i < n; ++i) out.push_back(classifiedPatches[keep[i]]);
    return out;
}

vector<Detection> NonMaximumSuppression::eliminateRedundantDetections(vector<Detection> candidates) const {
    vector<fd_box> in(candidates.size()), out(candidates.size());
    for (size_t i = 0; i < candidates.size(); ++i)
        in[i] = fd_box{candidates[i].score, candidates[i].bounds.x, candidates[i].bounds.y, candidates[i].bounds.width, candidates[i].bounds.height};
    int n = 0;
    const int rc = fd_nms_iou(in.data(), (int)in.size(), overlapThreshold, (int)maximumType, out.data(), &n);
    if (rc == FD_ERR_RUNTIME) throw std::runtime_error("NonMaximumSuppression: the overlap threshold must not exceed one");
    if (rc != FD_OK) throw std::invalid_argument("NonMaximumSuppression: invalid arguments");
    vector<Detection> res;
    for (int i = 0; i < n; ++i) res.push_back(Detection{out[i].score, cv::Rect(out[i].x, out[i].y, out[i].w, out[i].h)});
    return res;
}

AggregatedFeaturesDetector::AggregatedFeaturesDetector(shared_ptr<imageprocessing::ImageFilter> imageFilter, shared_ptr<imageprocessing::ImageFilter> layerFilter,
                                                       int cellSize, cv::Size windowSize, int octaveLayerCount,
                                                       shared_ptr<classification::SvmClassifier> svm, shared_ptr<NonMaximumSuppression> nms,
                                                       float widthScale, float heightScale, int minWindowWidth)
    : scoreThreshold(svm->getThreshold()) {
    if (!dynamic_cast<classification::LinearKernel*>(svm->getKernel().get()))
        throw std::invalid_argument("AggregatedFeaturesDetector: the SVM must use a LinearKernel");
    auto fhog = std::dynamic_pointer_cast<imageprocessing::filtering::FhogFilter>(layerFilter);
    if (!std::dynamic_pointer_cast<imageprocessing::GrayscaleFilter>(imageFilter) || !fhog)
        throw std::logic_error("AggregatedFeaturesDetector: this backend needs a GrayscaleFilter image filter and a filtering::FhogFilter layer filter");
    if (fhog->cellSize != cellSize) throw std::invalid_argument("AggregatedFeaturesDetector: cellSize differs from the FhogFilter's");
    const int D = 3 * fhog->unsignedBinCount + 4;
    if (svm->getSupportVectors().size() != 1) throw std::invalid_argument("AggregatedFeaturesDetector: a linear SVM with one support vector is needed");
    Mat sv = contiguous(svm->getSupportVectors()[0]);
    if (sv.depth() != CV_32F || (int)(sv.total() * sv.channels()) != windowSize.width * windowSize.height * D)
        throw std::invalid_argument("AggregatedFeaturesDetector: the support vector must hold windowSize x (3 * unsignedBinCount + 4) floats");
    fd_aggregated_params prm;
    std::memset(&prm, 0, sizeof(prm));
    prm.fhog = fd_fhog_params{fhog->cellSize, fhog->unsignedBinCount, fhog->interpolateBins, fhog->interpolateCells, fhog->alpha};
    prm.window_w = windowSize.width; prm.window_h = windowSize.height; prm.octave_layer_count = octaveLayerCount;
    prm.min_window_width = minWindowWidth; prm.width_scale = widthScale; prm.height_scale = heightScale;
    // SvmClassifier: distance = -bias + coefficient * <sv, x>; the reference convolves the raw support vector (coefficients are 1)
    prm.svm_weights = sv.ptr<float>(0);
    prm.svm_bias = svm->getBias(); prm.score_threshold = svm->getThreshold();
    prm.nms_overlap_threshold = nms->getOverlapThreshold(); prm.nms_maximum_type = (int)nms->getMaximumType();
    check(fd_aggregated_create(context(), &prm, &handle));
}
AggregatedFeaturesDetector::~AggregatedFeaturesDetector() { fd_aggregated_destroy(handle); }
vector<std::pair<cv::Rect, float>> AggregatedFeaturesDetector::detectWithScores(shared_ptr<imageprocessing::VersionedImage> image) {
    Mat img = contiguous(image->getData());
    vector<fd_box> out(1 << 14);
    int n = 0;
    int rc = fd_aggregated_detect(context(), handle, img.data, img.cols, img.rows, img.channels(), 0, out.data(), (int)out.size(), &n, nullptr, 0, nullptr);
    if (rc == FD_ERR_CAPACITY) {
        out.resize((size_t)n);
        rc = fd_aggregated_detect(context(), handle, img.data, img.cols, img.rows, img.channels(), 0, out.data(), n, &n, nullptr, 0, nullptr);
    }
    check(rc);
    vector<std::pair<cv::Rect, float>> res;
    for (int i = 0; i < n; ++i) res.emplace_back(cv::Rect(out[i].x, out[i].y, out[i].w, out[i].h), out[i].score);
    return res;
}
vector<cv::Rect> AggregatedFeaturesDetector::detect(shared_ptr<imageprocessing::VersionedImage> image) {
    vector<cv::Rect> res;
    for (const auto& d : detectWithScores(image)) res.push_back(d.first);
    return res;
}

SlidingWindowDetector::SlidingWindowDetector(shared_ptr<classification::ProbabilisticClassifier> classifier,
                                             shared_ptr<imageprocessing::PyramidFeatureExtractor> featureExtractor, int sx, int sy)
    : classifier(classifier), featureExtractor(featureExtractor), stepSizeX(sx), stepSizeY(sy) {}

void Detector::fillPatchData(const imageprocessing::PyramidFeatureExtractor& extractor, vector<shared_ptr<ClassifiedPatch>>& patches) const {
    for (auto& cp : patches) {
        auto p = cp->getPatch();
        if (!p || !p->getData().empty()) continue;
        auto q = extractor.extract(p->getX(), p->getY(), p->getWidth(), p->getHeight());
        if (q) p->getData() = q->getData();
    }
}
vector<shared_ptr<ClassifiedPatch>> SlidingWindowDetector::detect(const cv::Rect* roi) const {
    auto out = detectWindows(roi);
    if (patchData) fillPatchData(*featureExtractor, out);
    return out;
}
vector<shared_ptr<ClassifiedPatch>> SlidingWindowDetector::detectWindows(const cv::Rect* roi) const {
    vector<shared_ptr<ClassifiedPatch>> out;
    auto direct = std::dynamic_pointer_cast<DirectPyramidFeatureExtractor>(featureExtractor);
    if (auto filtering = std::dynamic_pointer_cast<imageprocessing::FilteringPyramidFeatureExtractor>(featureExtractor))
        direct = filtering->getFusedExtractor();   // ffpDetectApp.cpp:445: same pyramid, the chain as patch filters; null = generic composition
    // the region of interest (SlidingWindowDetector.cpp:53-78) and the scale range of a pyramid built on another pyramid
    // apply to the window enumeration of every fused chain
    std::unique_ptr<imageprocessing::ImagePyramid::Selection> selection;
    if (direct) selection.reset(new imageprocessing::ImagePyramid::Selection(direct->getPyramid()->select(-1, -1, 1, roi)));
    auto pwvm = std::dynamic_pointer_cast<ProbabilisticWvmClassifier>(classifier);
    auto psvm = std::dynamic_pointer_cast<ProbabilisticSvmClassifier>(classifier);
    int r[4] = {0, 0, 0, 0};
    if (roi) { r[0] = roi->x; r[1] = roi->y; r[2] = roi->width; r[3] = roi->height; }
    if (direct && pwvm && direct->hasHistEq64()) {   // fused extract + HistEq64 + WVM cascade
        const fd_wvm* w = pwvm->getWvm()->native(pwvm->getLogisticA(), pwvm->getLogisticB());
        int64_t cnt = 0, cap = 1 << 16;
        vector<fd_detection> dets((size_t)cap);
        int rc = fd_detect_wvm(context(), direct->getPyramid()->native(), w, stepSizeX, stepSizeY, roi ? r : nullptr, dets.data(), cap, &cnt, nullptr, nullptr);
        if (rc == FD_ERR_CAPACITY) {
            dets.resize((size_t)cnt);
            rc = fd_detect_wvm(context(), direct->getPyramid()->native(), w, stepSizeX, stepSizeY, roi ? r : nullptr, dets.data(), cnt, &cnt, nullptr, nullptr);
        }
        check(rc);
        for (int64_t i = 0; i < cnt; ++i) out.push_back(to_patch(dets[i]));
        return out;
    }
    const bool f32sv = psvm && !psvm->getSvm()->getSupportVectors().empty() && psvm->getSvm()->getSupportVectors()[0].depth() == CV_32F;
    if (direct && psvm && direct->getHogFilter() && f32sv &&
        std::dynamic_pointer_cast<classification::RbfKernel>(psvm->getSvm()->getKernel())) {   // fused HOG + MFMA RBF-SVM
        auto hog = direct->getHogFilter();
        fd_hog_params hp = {direct->getPatchWidth(), direct->getPatchHeight(), stepSizeX, stepSizeY, hog->binCount, hog->cellWidth, hog->blockWidth,
                            hog->signedAndUnsigned};
        const fd_svm* s = psvm->getSvm()->native(psvm->getLogisticA(), psvm->getLogisticB());
        int64_t cnt = 0, cap = 1 << 16;
        vector<fd_detection> dets((size_t)cap);
        int rc = fd_detect_hog_svm(context(), direct->getPyramid()->native(), s, &hp, dets.data(), cap, &cnt, nullptr);
        if (rc == FD_ERR_CAPACITY) {
            dets.resize((size_t)cnt);
            rc = fd_detect_hog_svm(context(), direct->getPyramid()->native(), s, &hp, dets.data(), cnt, &cnt, nullptr);
        }
        check(rc);
        for (int64_t i = 0; i < cnt; ++i) out.push_back(to_patch(dets[i]));
        return out;
    }
    auto prvm = std::dynamic_pointer_cast<classification::ProbabilisticRvmClassifier>(classifier);
    if (direct && prvm && direct->getConversion() && !direct->getHistogramFilter() && !direct->getWhiChain()) {   // fused RVM cascade ("prvm")
        auto cf = direct->getConversion();
        fd_rvm_detect_params dp = {direct->getU8FeatureSpace(), (float)cf->alpha, (float)cf->beta, stepSizeX, stepSizeY};
        const fd_rvm* rv = prvm->getRvm()->native(prvm->getLogisticA(), prvm->getLogisticB());
        int64_t cnt = 0, cap = 1 << 16;
        vector<fd_detection> dets((size_t)cap);
        int rc = fd_detect_rvm(context(), direct->getPyramid()->native(), rv, &dp, roi ? r : nullptr, dets.data(), cap, &cnt, nullptr, nullptr);
        if (rc == FD_ERR_CAPACITY) {
            dets.resize((size_t)cnt);
            rc = fd_detect_rvm(context(), direct->getPyramid()->native(), rv, &dp, roi ? r : nullptr, dets.data(), cnt, &cnt, nullptr, nullptr);
        }
        check(rc);
        for (int64_t i = 0; i < cnt; ++i) out.push_back(to_patch(dets[i]));
        return out;
    }
    if (direct && psvm && direct->getWhiChain() && f32sv) {   // fused whi chain + SVM (ffpDetectApp.cpp:449-454, "psvm")
        auto wf = direct->getWhiChain();
        fd_whi_params wp = {direct->getPatchWidth(), direct->getPatchHeight(), stepSizeX, stepSizeY, wf->alpha, wf->cutoffFrequency};
        const fd_svm* s = psvm->getSvm()->native(psvm->getLogisticA(), psvm->getLogisticB());
        int64_t cnt = 0, cap = 1 << 16;
        vector<fd_detection> dets((size_t)cap);
        int rc = fd_detect_whi_svm(context(), direct->getPyramid()->native(), s, &wp, dets.data(), cap, &cnt, nullptr);
        if (rc == FD_ERR_CAPACITY) {
            dets.resize((size_t)cnt);
            rc = fd_detect_whi_svm(context(), direct->getPyramid()->native(), s, &wp, dets.data(), cnt, &cnt, nullptr);
        }
        check(rc);
        for (int64_t i = 0; i < cnt; ++i) out.push_back(to_patch(dets[i]));
        return out;
    }
    if (direct && psvm && direct->getHistogramFilter() && f32sv) {   // fused histogram features + SVM (any kernel)
        fd_hist_params hp = hist_params_of(*direct->getHistogramFilter(), direct->getPatchWidth(), direct->getPatchHeight(), stepSizeX, stepSizeY);
        const fd_svm* s = psvm->getSvm()->native(psvm->getLogisticA(), psvm->getLogisticB());
        int64_t cnt = 0, cap = 1 << 16;
        vector<fd_detection> dets((size_t)cap);
        int rc = fd_detect_hist_svm(context(), direct->getPyramid()->native(), s, &hp, dets.data(), cap, &cnt, nullptr);
        if (rc == FD_ERR_CAPACITY) {
            dets.resize((size_t)cnt);
            rc = fd_detect_hist_svm(context(), direct->getPyramid()->native(), s, &hp, dets.data(), cnt, &cnt, nullptr);
        }
        check(rc);
        for (int64_t i = 0; i < cnt; ++i) out.push_back(to_patch(dets[i]));
        return out;
    }
    // generic composition (SlidingWindowDetector.cpp:87-98): extract all, classify each through the per-Mat interface
    selection.reset();
    auto patches = roi ? featureExtractor->extract(stepSizeX, stepSizeY, *roi) : featureExtractor->extract(stepSizeX, stepSizeY);
    for (auto& p : patches) {
        auto res = classifier->getProbability(p->getData());
        if (res.first) out.push_back(make_shared<ClassifiedPatch>(p, res));
    }
    return out;
}
vector<shared_ptr<ClassifiedPatch>> SlidingWindowDetector::detect(const Mat& image) {
    featureExtractor->update(image);
    return detect((const cv::Rect*)nullptr);
}
vector<shared_ptr<ClassifiedPatch>> SlidingWindowDetector::detect(const Mat& image, const cv::Rect& roi) {
    featureExtractor->update(image);
    return detect(&roi);
}
vector<shared_ptr<ClassifiedPatch>> SlidingWindowDetector::detect(shared_ptr<imageprocessing::VersionedImage> image) {
    featureExtractor->update(image);
    return detect((const cv::Rect*)nullptr);
}

FiveStageSlidingWindowDetector::FiveStageSlidingWindowDetector(shared_ptr<SlidingWindowDetector> swd, shared_ptr<OverlapElimination> oe,
                                                               shared_ptr<classification::ProbabilisticClassifier> strong)
    : slidingWindowDetector(swd), overlapElimination(oe), strongClassifier(strong) {}

vector<shared_ptr<ClassifiedPatch>> FiveStageSlidingWindowDetector::run(const Mat& image, const cv::Rect* roi) {
    auto direct = std::dynamic_pointer_cast<DirectPyramidFeatureExtractor>(slidingWindowDetector->getPyramidFeatureExtractor());
    if (auto filtering = std::dynamic_pointer_cast<imageprocessing::FilteringPyramidFeatureExtractor>(slidingWindowDetector->getPyramidFeatureExtractor()))
        direct = filtering->getFusedExtractor();
    auto pwvm = std::dynamic_pointer_cast<ProbabilisticWvmClassifier>(slidingWindowDetector->getClassifier());
    auto psvm = std::dynamic_pointer_cast<ProbabilisticSvmClassifier>(strongClassifier);
    if (!direct || !pwvm || !psvm || !direct->hasHistEq64())
        throw std::logic_error("FiveStageSlidingWindowDetector: this backend needs DirectPyramidFeatureExtractor + HistEq64Filter, a "
                               "ProbabilisticWvmClassifier first stage and a ProbabilisticSvmClassifier second stage (ffpDetectApp.cpp:398-419)");
    direct->update(image);
    auto selection = direct->getPyramid()->select();   // the scale range of a pyramid built on another pyramid
    int r[4] = {0, 0, 0, 0};
    if (roi) { r[0] = roi->x; r[1] = roi->y; r[2] = roi->width; r[3] = roi->height; }
    int cnt = 0, cap = 4096;
    vector<fd_detection> dets((size_t)cap);
    int rc = fd_detect_five_stage(context(), direct->getPyramid()->native(), pwvm->getWvm()->native(pwvm->getLogisticA(), pwvm->getLogisticB()),
                                  psvm->getSvm()->native(psvm->getLogisticA(), psvm->getLogisticB()), overlapElimination->getDist(),
                                  overlapElimination->getRatio(), slidingWindowDetector->getStepSizeX(), slidingWindowDetector->getStepSizeY(),
                                  roi ? r : nullptr, dets.data(), cap, &cnt, nullptr);
    if (rc == FD_ERR_CAPACITY) {
        dets.resize((size_t)cnt);
        rc = fd_detect_five_stage(context(), direct->getPyramid()->native(), pwvm->getWvm()->native(), psvm->getSvm()->native(),
                                  overlapElimination->getDist(), overlapElimination->getRatio(), slidingWindowDetector->getStepSizeX(),
                                  slidingWindowDetector->getStepSizeY(), roi ? r : nullptr, dets.data(), cnt, &cnt, nullptr);
    }
    check(rc);
    vector<shared_ptr<ClassifiedPatch>> out;
    for (int i = 0; i < cnt; ++i) out.push_back(to_patch(dets[i]));
    if (patchData) fillPatchData(*slidingWindowDetector->getPyramidFeatureExtractor(), out);   // the patches of the shared extractor (FiveStageSlidingWindowDetector.cpp:187-380)
    return out;
}
FiveStageSlidingWindowDetector::~FiveStageSlidingWindowDetector() { if (framesPyramid) fd_pyramid_destroy(framesPyramid); }

vector<vector<shared_ptr<ClassifiedPatch>>> FiveStageSlidingWindowDetector::detectFrames(const vector<Mat>& images) {
    vector<vector<shared_ptr<ClassifiedPatch>>> out(images.size());
    auto direct = std::dynamic_pointer_cast<DirectPyramidFeatureExtractor>(slidingWindowDetector->getPyramidFeatureExtractor());
    if (auto filtering = std::dynamic_pointer_cast<imageprocessing::FilteringPyramidFeatureExtractor>(slidingWindowDetector->getPyramidFeatureExtractor()))
        direct = filtering->getFusedExtractor();
    auto pwvm = std::dynamic_pointer_cast<ProbabilisticWvmClassifier>(slidingWindowDetector->getClassifier());
    auto psvm = std::dynamic_pointer_cast<ProbabilisticSvmClassifier>(strongClassifier);
    size_t i = 0;
    while (i < images.size()) {
        // the longest run of images with the size and type of images[i], 64 at most
        size_t j = i + 1;
        while (j < images.size() && j - i < 64 && images[j].rows == images[i].rows && images[j].cols == images[i].cols && images[j].type() == images[i].type()) ++j;
        const int n = (int)(j - i);
        const int ch = images[i].channels();
        // (keepPatchData: the patches are cut from the extractor's own pyramid, which the multi-frame path never fills: one image at a time)
        bool fused = !patchData && direct && pwvm && psvm && direct->hasHistEq64() && n > 1 && images[i].depth() == CV_8U && (ch == 1 || ch == 3);
        if (fused && (!framesPyramid || framesCount != n)) {
            if (framesPyramid) { fd_pyramid_destroy(framesPyramid); framesPyramid = nullptr; }
            framesPyramid = direct->getPyramid()->createFramesPyramid(n);
            framesCount = n;
            fused = framesPyramid != nullptr;
        }
        if (!fused) {
            for (size_t k = i; k < j; ++k) out[k] = detect(images[k]);
            i = j;
            continue;
        }
        vector<Mat> cont((size_t)n);
        vector<const uint8_t*> ptrs((size_t)n);
        for (int k = 0; k < n; ++k) {
            cont[(size_t)k] = images[i + (size_t)k].isContinuous() ? images[i + (size_t)k] : images[i + (size_t)k].clone();
            ptrs[(size_t)k] = cont[(size_t)k].data;
        }
        check(fd_pyramid_update_frames(framesPyramid, ptrs.data(), n, images[i].cols, images[i].rows, ch, 0));
        int cap = 1024;
        vector<fd_detection> dets;
        vector<int32_t> counts((size_t)n);
        for (;;) {
            dets.resize((size_t)cap * (size_t)n);
            const int rc = fd_detect_five_stage_frames(context(), framesPyramid, pwvm->getWvm()->native(pwvm->getLogisticA(), pwvm->getLogisticB()),
                                                       psvm->getSvm()->native(psvm->getLogisticA(), psvm->getLogisticB()), overlapElimination->getDist(),
                                                       overlapElimination->getRatio(), slidingWindowDetector->getStepSizeX(),
                                                       slidingWindowDetector->getStepSizeY(), nullptr, dets.data(), cap, counts.data(), nullptr);
            // FD_ERR_CAPACITY: the output buffer was too small; counts[] holds every frame's count, so ONE retry fits.  (An overflow
            // of a device-side buffer is FD_ERR_DEVICE_CAPACITY: a larger output buffer would not help, no retry.)
            if (rc == FD_ERR_CAPACITY) {
                const int need = *std::max_element(counts.begin(), counts.end());
                if (need > cap) { cap = need; continue; }
            }
            check(rc);
            break;
        }
        for (int k = 0; k < n; ++k)
            for (int q = 0; q < counts[(size_t)k]; ++q) out[i + (size_t)k].push_back(to_patch(dets[(size_t)k * (size_t)cap + (size_t)q]));
        i = j;
    }
    return out;
}
vector<shared_ptr<ClassifiedPatch>> FiveStageSlidingWindowDetector::detect(const Mat& image) { return run(image, nullptr); }
vector<shared_ptr<ClassifiedPatch>> FiveStageSlidingWindowDetector::detect(const Mat& image, const cv::Rect& roi) { return run(image, &roi); }
vector<shared_ptr<ClassifiedPatch>> FiveStageSlidingWindowDetector::detect(shared_ptr<imageprocessing::VersionedImage>) {
    // FiveStageSlidingWindowDetector.cpp:324-329: "not yet implemented for a VersionedImage", returns empty
    return vector<shared_ptr<ClassifiedPatch>>();
}

}  // namespace detection

// =================================================================================================
namespace superviseddescent {

Mat VlHogDescriptorExtractor::getDescriptors(const Mat image, vector<cv::Point2f> locations, int windowSizeHalf) {
    if (image.channels() != 1 || image.depth() != CV_8U)
        throw std::invalid_argument("VlHogDescriptorExtractor: this backend expects the CV_8UC1 (gray) image the callers pass (detect-landmarks.cpp:245)");
    Mat img = contiguous(image);
    const int n = (int)locations.size();
    vector<float> px(n), py(n);
    for (int i = 0; i < n; ++i) { px[i] = locations[i].x; py[i] = locations[i].y; }
    const int variant = hogType == VlHogType::Uoctti ? 1 : 0;
    int len = 0;
    check(fd_sdm_descriptors(context(), img.data, img.cols, img.rows, px.data(), py.data(), n, windowSizeHalf, variant, numCells, cellSize, numBins, nullptr, &len));
    Mat out(n, len, CV_32FC1);
    if (n) check(fd_sdm_descriptors(context(), img.data, img.cols, img.rows, px.data(), py.data(), n, windowSizeHalf, variant, numCells, cellSize, numBins,
                                    out.ptr<float>(0), &len));
    return out;
}
string VlHogDescriptorExtractor::getParameterString() const {
    std::ostringstream s;
    s << "numCells " << numCells << " cellSize " << cellSize << " numBins " << numBins;
    return s.str();
}

SdmLandmarkModel::SdmLandmarkModel(Mat mean, vector<string> ids, vector<Mat> regs, vector<shared_ptr<DescriptorExtractor>> ex, vector<string> types)
    : meanLandmarks(mean), landmarkIdentifier(ids), regressorData(regs), descriptorExtractors(ex), descriptorTypes(types) {}
Mat SdmLandmarkModel::getMeanShape() const {   // SdmLandmarkModel.cpp:53-56: clone().t()
    const int n = meanLandmarks.cols;
    Mat col(n, 1, CV_32FC1);
    for (int i = 0; i < n; ++i) col.at<float>(i, 0) = meanLandmarks.at<float>(0, i);
    return col;
}
vector<cv::Point2f> SdmLandmarkModel::getMeanAsPoints() const {
    vector<cv::Point2f> pts;
    const int L = getNumLandmarks();
    for (int i = 0; i < L; ++i) pts.push_back(cv::Point2f(meanLandmarks.at<float>(0, i), meanLandmarks.at<float>(0, i + L)));
    return pts;
}
cv::Point2f SdmLandmarkModel::getLandmarkAsPoint(string id, Mat inst) const {
    auto it = std::find(landmarkIdentifier.begin(), landmarkIdentifier.end(), id);
    if (it == landmarkIdentifier.end()) throw std::invalid_argument("SdmLandmarkModel: unknown landmark " + id);
    const int index = (int)(it - landmarkIdentifier.begin()), L = getNumLandmarks();
    if (inst.empty()) return cv::Point2f(meanLandmarks.at<float>(0, index), meanLandmarks.at<float>(0, index + L));
    return cv::Point2f(inst.at<float>(index), inst.at<float>(index + L));
}
void SdmLandmarkModel::save(string filename, string comment) {   // SdmLandmarkModel.cpp:98-128
    std::ofstream file(filename.c_str());
    file << "# " << comment << std::endl;
    file << "numLandmarks " << getNumLandmarks() << std::endl;
    for (const auto& id : landmarkIdentifier) file << id << std::endl;
    file << std::setprecision(9);
    for (int i = 0; i < 2 * getNumLandmarks(); ++i) file << meanLandmarks.at<float>(0, i) << std::endl;
    file << "numCascadeSteps " << getNumCascadeSteps() << std::endl;
    for (int s = 0; s < getNumCascadeSteps(); ++s) {
        const Mat& R = regressorData[s];
        file << "cascadeStep " << s << " rows " << R.rows << " cols " << R.cols << std::endl;
        file << "descriptorType " << descriptorTypes[s] << std::endl;
        file << "descriptorPostprocessing none" << std::endl;
        file << "descriptorParameters " << descriptorExtractors[s]->getParameterString() << std::endl;
        for (int r = 0; r < R.rows; ++r) {
            for (int c = 0; c < R.cols; ++c) file << R.at<float>(r, c) << " ";
            file << std::endl;
        }
    }
}
static vector<string> split_ws(const string& line) {
    vector<string> out;
    std::istringstream ss(line);
    string t;
    while (std::getline(ss, t, ' ')) out.push_back(t);
    return out;
}
static void chomp(string& s) { while (!s.empty() && s.back() == '\r') s.pop_back(); }
SdmLandmarkModel SdmLandmarkModel::load(string filename) {   // SdmLandmarkModel.cpp:130-233
    SdmLandmarkModel model;
    std::ifstream file(filename.c_str());
    if (!file.is_open()) throw std::runtime_error("Given SDM model file could not be opened: " + filename);
    string line;
    std::getline(file, line);   // description
    std::getline(file, line); chomp(line);
    const int L = std::stoi(split_ws(line).at(1));
    for (int i = 0; i < L; ++i) { std::getline(file, line); chomp(line); model.landmarkIdentifier.push_back(line); }
    model.meanLandmarks = Mat(1, 2 * L, CV_32FC1);
    for (int i = 0; i < 2 * L; ++i) { std::getline(file, line); chomp(line); model.meanLandmarks.at<float>(0, i) = std::stof(line); }
    std::getline(file, line); chomp(line);
    const int S = std::stoi(split_ws(line).at(1));
    for (int s = 0; s < S; ++s) {
        std::getline(file, line); chomp(line);
        auto hdr = split_ws(line);
        const int rows = std::stoi(hdr.at(3)), cols = std::stoi(hdr.at(5));
        std::getline(file, line); chomp(line);
        const string type = split_ws(line).at(1);
        std::getline(file, line);   // descriptorPostprocessing
        std::getline(file, line); chomp(line);
        auto par = split_ws(line);
        if (type == "vlhog-dt") {
            if (par.size() != 7) throw std::logic_error("descriptorParameters must contain numCells, cellSize and numBins.");
            model.descriptorExtractors.push_back(make_shared<VlHogDescriptorExtractor>(VlHogDescriptorExtractor::VlHogType::DalalTriggs, std::stoi(par[2]),
                                                                                       std::stoi(par[4]), std::stoi(par[6])));
        } else if (type == "vlhog-uoctti") {
            if (par.size() <= 2) model.descriptorExtractors.push_back(make_shared<VlHogDescriptorExtractor>(VlHogDescriptorExtractor::VlHogType::Uoctti));
            else if (par.size() == 7)
                model.descriptorExtractors.push_back(make_shared<VlHogDescriptorExtractor>(VlHogDescriptorExtractor::VlHogType::Uoctti, std::stoi(par[2]),
                                                                                           std::stoi(par[4]), std::stoi(par[6])));
            else throw std::logic_error("descriptorParameters must either be empty (=face-size adaptive parameters) or contain numCells, cellSize and numBins.");
        } else {
            throw std::logic_error("descriptorType does not match 'vlhog-dt' or 'vlhog-uoctti' (OpenCVSift is not available on this backend).");
        }
        model.descriptorTypes.push_back(type);
        Mat R(rows, cols, CV_32FC1);
        for (int r = 0; r < rows; ++r) {
            std::getline(file, line);
            std::istringstream ss(line);
            for (int c = 0; c < cols; ++c) ss >> R.at<float>(r, c);
        }
        model.regressorData.push_back(R);
    }
    return model;
}

SdmLandmarkModelFitting::SdmLandmarkModelFitting(SdmLandmarkModel m, bool adaptive) : model(m), handle(nullptr) {
    const int L = model.getNumLandmarks(), S = model.getNumCascadeSteps();
    if (S == 0) return;
    Mat mean = model.getMeanShape();
    vector<float> meanv(2 * L);
    for (int i = 0; i < 2 * L; ++i) meanv[i] = mean.at<float>(i, 0);
    vector<Mat> regs;
    vector<const float*> R;
    vector<int32_t> rows;
    for (int s = 0; s < S; ++s) { regs.push_back(contiguous(model.getRegressorData(s))); R.push_back(regs.back().ptr<float>(0)); rows.push_back(regs.back().rows); }
    auto vl = std::dynamic_pointer_cast<VlHogDescriptorExtractor>(model.getDescriptorExtractor(0));
    if (!vl) throw std::logic_error("SdmLandmarkModelFitting: only VlHog descriptors are available on this backend");
    fd_sdm_model md = {};
    md.num_landmarks = L; md.num_steps = S; md.mean = meanv.data(); md.R = R.data(); md.R_rows = rows.data();
    md.hog_variant = vl->getType() == VlHogDescriptorExtractor::VlHogType::Uoctti ? 1 : 0;
    vector<int32_t> dp;
    if (!adaptive) {   // SdmLandmarkModel.hpp:236-238: "non-adaptive, the descriptorExtractor has all necessary params"
        for (int s = 0; s < S; ++s) {
            auto e = std::dynamic_pointer_cast<VlHogDescriptorExtractor>(model.getDescriptorExtractor(s));
            if (!e || e->getType() != vl->getType()) throw std::logic_error("SdmLandmarkModelFitting: the cascade steps must share one VlHog type");
            dp.push_back(e->getNumCells()); dp.push_back(e->getCellSize()); dp.push_back(e->getNumBins());
        }
        md.desc_params = dp.data();
    }
    check(fd_sdm_create(context(), &md, &handle));
}
SdmLandmarkModelFitting::~SdmLandmarkModelFitting() { fd_sdm_destroy(handle); }
Mat SdmLandmarkModelFitting::alignRigid(Mat modelShape, cv::Rect faceBox) const {   // SdmLandmarkModel.hpp:156-192
    if (modelShape.cols != 1) throw std::runtime_error("The supplied model shape does not have one column (i.e. it doesn't seem to be a column-vector).");
    const int L = modelShape.rows / 2;
    // cv::MatExpr "(x + 0.5f) * w + bx" evaluates as x * float(w) + float(0.5 * w + bx) (convertTo with alpha/beta)
    const float ax = (float)(double)faceBox.width, bx = (float)(0.5 * faceBox.width + faceBox.x);
    const float ay = (float)(double)faceBox.height, by = (float)(0.5 * faceBox.height + faceBox.y);
    for (int i = 0; i < L; ++i) {
        modelShape.at<float>(i, 0) = modelShape.at<float>(i, 0) * ax + bx;
        modelShape.at<float>(i + L, 0) = modelShape.at<float>(i + L, 0) * ay + by;
    }
    return modelShape;
}
vector<Mat> SdmLandmarkModelFitting::optimize(const vector<Mat>& shapes, const vector<Mat>& images) {
    if (!handle) throw std::logic_error("SdmLandmarkModelFitting: model has no cascade steps");
    const int B = (int)images.size(), L = model.getNumLandmarks();
    if (B == 0 || (int)shapes.size() != B) throw std::invalid_argument("SdmLandmarkModelFitting: need one shape per image");
    const int W = images[0].cols, H = images[0].rows;
    vector<unsigned char> stack((size_t)B * W * H);
    vector<float> sh((size_t)B * 2 * L);
    for (int b = 0; b < B; ++b) {
        if (images[b].cols != W || images[b].rows != H || images[b].type() != CV_8UC1)
            throw std::invalid_argument("SdmLandmarkModelFitting: images of a batch must be CV_8UC1 and of equal size");
        Mat im = contiguous(images[b]);
        std::memcpy(stack.data() + (size_t)b * W * H, im.data, (size_t)W * H);
        for (int i = 0; i < 2 * L; ++i) sh[(size_t)b * 2 * L + i] = shapes[b].at<float>(i, 0);
    }
    vector<int32_t> status(B);
    check(fd_sdm_optimize_batch(context(), handle, stack.data(), W, H, B, 0, sh.data(), status.data()));
    vector<Mat> out;
    for (int b = 0; b < B; ++b) {
        if (status[b]) throw std::runtime_error("VlHogDescriptorExtractor: patch window leaves the zero-extended image (cv::Mat roi assertion in the reference)");
        Mat s(2 * L, 1, CV_32FC1);
        for (int i = 0; i < 2 * L; ++i) s.at<float>(i, 0) = sh[(size_t)b * 2 * L + i];
        out.push_back(s);
    }
    return out;
}
Mat SdmLandmarkModelFitting::optimize(Mat modelShape, Mat image) {
    return optimize(vector<Mat>{modelShape}, vector<Mat>{image})[0];
}

}  // namespace superviseddescent

// =================================================================================================
#include "condensation/condensation_all.hpp"
namespace condensation {

double Sample::aspectRatio = 1;

WvmSvmModel::WvmSvmModel(shared_ptr<imageprocessing::FeatureExtractor> featureExtractor, shared_ptr<classification::ProbabilisticWvmClassifier> wvm,
                         shared_ptr<classification::ProbabilisticSvmClassifier> svm)
    : featureExtractor(featureExtractor), wvm(wvm), svm(svm) {}
void WvmSvmModel::update(shared_ptr<imageprocessing::VersionedImage> image) { featureExtractor->update(image); }
static imageprocessing::DirectPyramidFeatureExtractor* fused_extractor(const shared_ptr<imageprocessing::FeatureExtractor>& fe) {
    auto* d = dynamic_cast<imageprocessing::DirectPyramidFeatureExtractor*>(fe.get());
    if (!d || !d->hasHistEq64())
        throw std::logic_error("WvmSvmModel: this backend needs a DirectPyramidFeatureExtractor with a HistEq64Filter patch filter");
    return d;
}
void WvmSvmModel::evaluate(Sample& sample) const {   // WvmSvmModel.cpp:44-67 (single sample: no top-8 selection)
    auto patch = featureExtractor->extract(sample.getX(), sample.getY(), sample.getWidth(), sample.getHeight());
    if (!patch) {
        sample.setTarget(false);
        sample.setWeight(0);
        return;
    }
    auto wvmResult = wvm->getProbability(patch->getData());
    if (wvmResult.first) {
        auto svmResult = svm->getProbability(patch->getData());
        sample.setTarget(svmResult.first);
        sample.setWeight(wvmResult.second * svmResult.second);
    } else {
        sample.setTarget(false);
        sample.setWeight(0.5 * wvmResult.second);
    }
}
void WvmSvmModel::evaluate(shared_ptr<imageprocessing::VersionedImage> image, vector<shared_ptr<Sample>>& samples) {   // :69-118
    update(image);
    auto* direct = fused_extractor(featureExtractor);
    const int n = (int)samples.size();
    if (n == 0) return;
    vector<int32_t> xywh((size_t)4 * n);
    for (int i = 0; i < n; ++i) {
        xywh[4 * i] = samples[i]->getX(); xywh[4 * i + 1] = samples[i]->getY();
        xywh[4 * i + 2] = samples[i]->getWidth(); xywh[4 * i + 3] = samples[i]->getHeight();
    }
    vector<uint8_t> target((size_t)n);
    vector<double> weight((size_t)n);
    check(fd_wvm_svm_evaluate_samples(context(), direct->getPyramid()->native(), wvm->getWvm()->native(wvm->getLogisticA(), wvm->getLogisticB()),
                                      svm->getSvm()->native(svm->getLogisticA(), svm->getLogisticB()), n, xywh.data(), target.data(), weight.data()));
    for (int i = 0; i < n; ++i) {
        samples[i]->setTarget(target[i] != 0);
        samples[i]->setWeight(weight[i]);
    }
}

}  // namespace condensation
