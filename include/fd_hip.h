/* include/fd_hip.h -- C ABI of the MI355X-native FeatureDetection hot path (libfd_hip.so).
 *
 * The reference (elador/FeatureDetection) has no FFI/plugin layer: its extension points are the
 * C++ abstract classes wired by shared_ptr in each app's main().  The drop-in boundary is therefore
 *   C++ subclass of the reference interface (host/ in this repo)  ->  this C ABI  ->  HIP kernels.
 * Every entry point names the reference interface (file:line) whose work it performs.
 *
 * Conventions: opaque handles, int status (0 = FD_OK), caller-owned buffers, no exceptions across
 * the ABI, one HIP stream per context (thread-compatible, not thread-safe -- like the reference,
 * whose const methods mutate scratch buffers, WvmClassifier.hpp:129-130).  Pointers named dev_* are
 * device (HBM) pointers, everything else is host memory.  The library has NO CPU fallback: every
 * compute entry point fails with FD_ERR_HIP when no gfx950 device is usable.
 */
#ifndef FD_HIP_H_
#define FD_HIP_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum {
    FD_OK = 0,
    FD_ERR_INVALID_ARGUMENT = 1, /* std::invalid_argument in the reference */
    FD_ERR_RUNTIME = 2,          /* std::runtime_error */
    FD_ERR_LOGIC = 3,            /* std::logic_error */
    FD_ERR_HIP = 4,              /* HIP runtime failure / no device */
    FD_ERR_CAPACITY = 5,         /* caller buffer too small; required count is still reported */
    FD_ERR_DEVICE_CAPACITY = 6   /* a device-side buffer of the library overflowed (FD_WVM_POS_CAP / FD_WVM_DEEP_CAP in the
                                    environment raise them); a larger caller buffer does not help */
};

typedef struct fd_ctx fd_ctx;
typedef struct fd_pyramid fd_pyramid;
typedef struct fd_wvm fd_wvm;
typedef struct fd_svm fd_svm;
typedef struct fd_sdm fd_sdm;

/* ---- context ------------------------------------------------------------------------------ */
/* hip_stream: an existing hipStream_t to launch on (e.g. torch.cuda.current_stream().cuda_stream),
 * or NULL to create a private stream. */
int fd_ctx_create(int device_id, void* hip_stream, fd_ctx** out);
int fd_device_count(int* n);   /* visible HIP devices (a multi-rank launcher checks its --gpus against it before any rank starts) */
void fd_ctx_destroy(fd_ctx* ctx);
/* Optional: creates the streams the batch entry points otherwise create on first use (eight batch streams, the high-priority stream of
 * the SVM stage, an auxiliary one) now.  The HIP runtime deals streams to its hardware queues (four by default) in creation order, so
 * an application that creates other streams in between gets a different -- and, measured, up to 12 % slower -- mapping for them; calling
 * this straight after fd_ctx_create makes the mapping the same in every run. */
int fd_ctx_warm_streams(fd_ctx* ctx);
const char* fd_last_error(const fd_ctx* ctx); /* valid until the next failing call on ctx */
int fd_ctx_synchronize(fd_ctx* ctx);
const char* fd_version(void);

/* ---- image pyramid: imageprocessing::ImagePyramid (ImagePyramid.cpp:67-92 ctors, :116-128 update,
 *      :170-198 createLayers) with the GrayscaleFilter image filter (GrayscaleFilter.cpp:18-24) ---- */
int fd_pyramid_create(fd_ctx* ctx, int octave_layer_count, double min_scale, double max_scale, fd_pyramid** out);
int fd_pyramid_create_inc(fd_ctx* ctx, double incremental_scale, double min_scale, double max_scale, fd_pyramid** out);
void fd_pyramid_destroy(fd_pyramid* p);
/* Layer filters (ImagePyramid::addLayerFilter, ImagePyramid.cpp:112-114):
 *  FD_LAYER_NONE       gray layers (u8, 1 channel)
 *  FD_LAYER_GRADBIN    GradientFilter(grad_kernel, blur) -> GradientBinningFilter(bins, signed, interpolate)
 *                      (GradientFilter.cpp:16-59, GradientBinningFilter.cpp:18-93); 2 or 4 channels.
 *                      grad_kernel: 1, 3, 5, 7 or FD_GRAD_SCHARR (the reference's CV_SCHARR = -1);
 *                      blur: fd_pyramid_set_gradient_blur (blurKernelSize of the reference's constructor, default 0 = none)
 *  FD_LAYER_LBP        LbpFilter(lbp_type) (LbpFilter.cpp:56-85); 1 channel */
enum { FD_LAYER_NONE = 0, FD_LAYER_GRADBIN = 1, FD_LAYER_LBP = 2 };
enum { FD_LBP8 = 0, FD_LBP8_UNIFORM = 1, FD_LBP4 = 2, FD_LBP4_ROTATED = 3 };
enum { FD_GRAD_SCHARR = -1 };
int fd_pyramid_set_layer_filter(fd_pyramid* p, int kind, int bins, int signed_gradients, int interpolate,
                                int grad_kernel, int lbp_type);
int fd_pyramid_set_gradient_blur(fd_pyramid* p, int blur_kernel);
/* channels 1 (gray) or 3 (BGR, interleaved).  is_device != 0: image already resident in HBM. */
int fd_pyramid_update(fd_pyramid* p, const uint8_t* image, int width, int height, int channels, int is_device);
int fd_pyramid_octave_layer_count(const fd_pyramid* p);
double fd_pyramid_incremental_scale(const fd_pyramid* p);
int fd_pyramid_layer_count(const fd_pyramid* p);
/* ImagePyramidLayer.hpp:34-35: index, theoretical scale, size; channels of the filtered layer */
int fd_pyramid_layer_info(const fd_pyramid* p, int i, int* index, double* scale, int* width, int* height, int* channels);
/* copy the (filtered) layer i to host: width*height*channels bytes */
int fd_pyramid_layer_download(fd_pyramid* p, int i, uint8_t* host_dst);
/* Layer sub-range and default region of interest of every window enumeration that follows on this pyramid -- the firstLayer /
 * lastLayer / stepLayer / roi arguments of DirectPyramidFeatureExtractor::extract(stepX, stepY, roi, firstLayer, lastLayer,
 * stepLayer) (:75-123), for every extraction / detection entry point (also those without a roi argument), and the layer range of
 * an ImagePyramid built on another pyramid (ImagePyramid.cpp:100-104,200-235: same layers, scale factors within [min, max]).
 * first_layer / last_layer: pyramid layer indices, -1 = open; step_layer >= 1 counts over the kept layers from the first one;
 * roi: {x, y, w, h} or NULL (none; an explicit roi argument of a call takes precedence).  Reset with (-1, -1, 1, NULL). */
int fd_pyramid_select(fd_pyramid* p, int first_layer, int last_layer, int step_layer, const int* roi);
/* ImagePyramid(shared_ptr<ImagePyramid> pyramid, minScale, maxScale) (ImagePyramid.cpp:100-104,200-235): a pyramid built on another
 * one shares its layers but exposes only those inside its scale range; its getLayers() -- and therefore the layer step of
 * DirectPyramidFeatureExtractor::extract (:99) -- starts at the first layer of that range.  first_layer / last_layer: pyramid layer
 * indices of the range, -1 = open.  Applies to the window enumerations that follow; reset with (-1, -1). */
int fd_pyramid_select_view(fd_pyramid* p, int first_layer, int last_layer);
/* Window enumeration of DirectPyramidFeatureExtractor::extract(stepX, stepY, roi) (:75-123).
 * roi = {x,y,w,h} or NULL (whole image).  rows of out: {layerPos, lx, ly, cx, cy, ow, oh}. */
int fd_pyramid_window_count(const fd_pyramid* p, int patch_w, int patch_h, int step_x, int step_y,
                            const int* roi, int64_t* count);
int fd_pyramid_windows(const fd_pyramid* p, int patch_w, int patch_h, int step_x, int step_y, const int* roi,
                       int32_t* out, int64_t cap, int64_t* count);

/* Stand-alone image filters (ImageFilter::applyTo): GreyWorldNormalizationFilter.cpp:20-71 */
int fd_greyworld(fd_ctx* ctx, const uint8_t* bgr, int width, int height, uint8_t* dst, int is_device);
/* HistEq64Filter::applyTo (HistEq64Filter.cpp:32-125) on n contiguous patches of w*h bytes (host buffers) */
int fd_histeq64_batch(fd_ctx* ctx, const uint8_t* patches, int64_t n, int w, int h, uint8_t* dst);

/* Stand-alone ImageFilter::applyTo(const Mat&) forms (ImageFilter.hpp:18-57) of the filters the detection kernels fuse into the
 * pyramid / feature kernels; host buffers in, host buffers out.
 *  fd_gradient_image          GradientFilter(grad_kernel 1|3|5|7|FD_GRAD_SCHARR, no blur) (GradientFilter.cpp:16-59): CV_8UC1 -> CV_8UC2 (x, y)
 *  fd_gradient_binning_image  GradientBinningFilter(bins, signed, interpolate) (GradientBinningFilter.cpp:62-93): CV_8UC2 -> CV_8UC2
 *                             (bin, weight) or CV_8UC4 (two bins + weights)
 *  fd_lbp_image               LbpFilter(type) (LbpFilter.cpp:56-85): CV_8UC1 -> CV_8UC1 codes */
int fd_gradient_image(fd_ctx* ctx, const uint8_t* gray, int width, int height, int grad_kernel, uint8_t* dst2ch);
/* ... with GradientFilter's blurKernelSize (cv::blur before the derivatives, GradientFilter.cpp:43-46; 0 = none).
 * grad_kernel: 1, 3, 5, 7 or FD_GRAD_SCHARR (CV_SCHARR), as in the reference */
int fd_gradient_filter_image(fd_ctx* ctx, const uint8_t* gray, int width, int height, int grad_kernel, int blur_kernel, uint8_t* dst2ch);
int fd_gradient_binning_image(fd_ctx* ctx, const uint8_t* grad2ch, int width, int height, int bins, int signed_gradients,
                              int interpolate, uint8_t* dst);
int fd_lbp_image(fd_ctx* ctx, const uint8_t* gray, int width, int height, int lbp_type, uint8_t* dst);

/* ---- classification::WvmClassifier / ProbabilisticWvmClassifier --------------------------------
 * Field meaning as in WvmClassifier.hpp / the Matlab loader WvmClassifier.cpp:352-769 (values already
 * converted to the 0..255 domain as the loader does). */
typedef struct {
    int32_t filter_w, filter_h;  /* filter_size_x/y */
    int32_t num_filters;         /* numLinFilters */
    int32_t num_used;            /* numUsedFilters (0 or > num_filters => num_filters, :151-158) */
    int32_t num_per_level;       /* numFiltersPerLevel */
    float basis_param;           /* basisParam */
    float bias;                  /* lin_thresholds[i] == bias for all i (:567-570) */
    const float* thresholds;     /* hierarchicalThresholds[num_filters] (limitReliabilityFilter already added) */
    const float* hk_weights;     /* [num_filters][num_filters] row-major; hk_weights[k][p], p <= k */
    const double* pp;            /* app_rsv_convol[num_filters] */
    const int32_t* val_off;      /* [num_filters+1]; grey values of filter k are val[val_off[k]..val_off[k+1]) */
    const double* val;           /* area[k]->val[v] */
    const int32_t* rec_off;      /* [val_off[num_filters]+1]; rects of (k,v) */
    const uint8_t* rects;        /* {x1,y1,x2,y2} inclusive patch coordinates, 4 bytes per rect */
    double logistic_a, logistic_b; /* ProbabilisticWvmClassifier.hpp:36 */
    int32_t num_vals, num_rects; /* lengths of val / rects (entries), so that a truncated model is rejected instead of read out
                                  * of bounds; 0 = not stated (offsets are still checked for order) */
} fd_wvm_model;
int fd_wvm_create(fd_ctx* ctx, const fd_wvm_model* model, fd_wvm** out);
void fd_wvm_destroy(fd_wvm* m);
/* WvmClassifier::computeHyperplaneDistance (WvmClassifier.cpp:100-149) for n feature vectors (already
 * HistEq64-filtered filter_w x filter_h u8 patches, contiguous, host buffers): (lastLevel, fout) each.
 * Backs the per-Mat BinaryClassifier::classify / ProbabilisticClassifier::getProbability. */
int fd_wvm_eval_batch(fd_ctx* ctx, const fd_wvm* m, const uint8_t* patches, int64_t n, int32_t* out_level, float* out_score);

/* ---- classification::SvmClassifier / ProbabilisticSvmClassifier with Kernel{Linear,Polynomial,Rbf,HIK} */
enum { FD_KERNEL_LINEAR = 0, FD_KERNEL_POLY = 1, FD_KERNEL_RBF = 2, FD_KERNEL_HIK = 3 };
enum { FD_DTYPE_U8 = 0, FD_DTYPE_F32 = 1 };
typedef struct {
    int32_t kernel;       /* FD_KERNEL_* */
    double p0, p1, p2;    /* rbf: gamma; poly: alpha, constant, degree */
    int32_t num_sv, dim;
    int32_t dtype;        /* FD_DTYPE_* of the support vectors (and of the feature vectors) */
    const void* support_vectors; /* [num_sv][dim] */
    const float* coefficients;   /* [num_sv] */
    float bias, threshold;       /* VectorMachineClassifier.hpp */
    double logistic_a, logistic_b;
} fd_svm_model;
int fd_svm_create(fd_ctx* ctx, const fd_svm_model* model, fd_svm** out);
void fd_svm_destroy(fd_svm* m);
/* SvmClassifier::computeHyperplaneDistance (SvmClassifier.cpp:55-60) for n feature vectors (host buffers).
 * Backs the per-Mat BinaryClassifier::classify / ProbabilisticClassifier::getProbability. */
int fd_svm_distance_batch(fd_ctx* ctx, const fd_svm* m, const void* features, int64_t n, double* out_distance);

/* ---- detection ------------------------------------------------------------------------------ */
typedef struct {
    int32_t cx, cy, w, h;   /* imageprocessing::Patch centre and size in the original image (Patch.hpp:64-65) */
    int32_t layer, lx, ly;  /* layer position (pyramid order) and top-left corner inside the layer */
    int32_t level;          /* WVM: last evaluated filter (WvmClassifier.cpp:149); else -1 */
    int32_t positive;       /* ClassifiedPatch::isPositive */
    float score;            /* WVM fout / (float) SVM hyperplane distance */
    double probability;     /* ClassifiedPatch::getProbability */
} fd_detection;

/* detection::SlidingWindowDetector::detect (SlidingWindowDetector.cpp:40-98) with
 * DirectPyramidFeatureExtractor + HistEq64Filter patch filter + ProbabilisticWvmClassifier
 * (ffpDetectApp.cpp:407-417).  Positives are returned in extraction order.
 * all_level/all_score (host, may be NULL) receive every window's (lastLevel, fout). */
int fd_detect_wvm(fd_ctx* ctx, fd_pyramid* p, const fd_wvm* wvm, int step_x, int step_y, const int* roi,
                  fd_detection* out, int64_t cap, int64_t* count, int32_t* all_level, float* all_score);

/* detection::FiveStageSlidingWindowDetector::detect(image) (:187-320, roi == NULL) and
 * detect(image, roi) (:331-380): WVM -> OverlapElimination(dist, ratio) -> SVM classify ->
 * positives -> [block NMS, no-roi variant only] -> sort.  stage_counts[4] (may be NULL) =
 * {WVM positives, after OE, SVM positives, final}. */
int fd_detect_five_stage(fd_ctx* ctx, fd_pyramid* p, const fd_wvm* wvm, const fd_svm* svm, float oe_dist,
                         float oe_ratio, int step_x, int step_y, const int* roi, fd_detection* out, int cap,
                         int* count, int32_t* stage_counts);
/* detection::Detector::detect(const Mat& image) (Detector.hpp:59; FiveStageSlidingWindowDetector.cpp:187-190 updates its extractor
 * with the image and detects): fd_pyramid_update + fd_detect_five_stage in one call. */
int fd_detect_five_stage_image(fd_ctx* ctx, fd_pyramid* p, const fd_wvm* wvm, const fd_svm* svm, const uint8_t* image, int width,
                               int height, int channels, int image_is_device, float oe_dist, float oe_ratio, int step_x, int step_y,
                               const int* roi, fd_detection* out, int cap, int* count, int32_t* stage_counts);
/* Several frames of identical size in ONE pyramid, for the small-frame regime where the per-frame chain of dependent launches
 * (pyramid, cascade, SVM) bounds the throughput: fd_pyramid_set_frames(p, n) (1..64; gray pyramids only) makes p hold n frames,
 * fd_pyramid_update_frames builds all of them with one launch per pyramid stage, and fd_detect_five_stage_frames runs
 * FiveStageSlidingWindowDetector::detect (:187-320 / :331-380) on every frame with one cascade run and one SVM launch for the
 * whole call.  out: frames x cap_per_frame records (frame f's detections start at out + f * cap_per_frame); counts[frames];
 * stage_counts (may be NULL): frames x 4.  The results equal fd_detect_five_stage on each frame.  The other entry points
 * reject multi-frame pyramids.  fd_pyramid_frame_layer_download copies layer i of frame f to the host. */
int fd_pyramid_set_frames(fd_pyramid* p, int frames);
int fd_pyramid_update_frames(fd_pyramid* p, const uint8_t* const* images, int n, int width, int height, int channels, int is_device);
int fd_pyramid_frame_layer_download(fd_pyramid* p, int frame, int i, uint8_t* host_dst);
int fd_detect_five_stage_frames(fd_ctx* ctx, fd_pyramid* p, const fd_wvm* wvm, const fd_svm* svm, float oe_dist, float oe_ratio,
                                int step_x, int step_y, const int* roi, fd_detection* out, int cap_per_frame, int32_t* counts,
                                int32_t* stage_counts);
/* The same in two halves (frames in flight: further pyramid + WVM + SVM handles on further contexts): begin queues the cascade run
 * of all frames and returns; the host stages (ordering the positives, overlap elimination, the SVM launch, NMS) follow on the
 * library's queue threads as soon as the kernels retire (FD_FRAMES_ASYNC=0: inside end); end waits for them and fills out /
 * counts / stage_counts, reporting whatever they failed with.  Between begin and end the ticket's pyramid, classifier handles
 * and the context's stream must not be used by the caller.  Every ticket must be ended (end releases it, whatever it returns). */
typedef struct fd_five_stage_frames fd_five_stage_frames;
int fd_detect_five_stage_frames_begin(fd_ctx* ctx, fd_pyramid* p, const fd_wvm* wvm, const fd_svm* svm, float oe_dist, float oe_ratio,
                                      int step_x, int step_y, const int* roi, fd_five_stage_frames** ticket);
int fd_detect_five_stage_frames_end(fd_ctx* ctx, fd_five_stage_frames* ticket, fd_detection* out, int cap_per_frame, int32_t* counts,
                                    int32_t* stage_counts);
/* condensation::WvmSvmModel::evaluate(image, samples) (WvmSvmModel.cpp:69-118): the particle-filter
 * measurement model of the tracking apps on an updated gray pyramid.  xywh: n samples {x, y, width, height} (Sample::getX/
 * getY/getWidth/getHeight).  Every sample maps to one window (DirectPyramidFeatureExtractor::extract(x, y, width, height)),
 * weight = 0.5 p_wvm; the (at most 8) most probable WVM positives are re-scored: target = SVM decision, weight = p_wvm p_svm;
 * samples without a patch get weight 0. */
int fd_wvm_svm_evaluate_samples(fd_ctx* ctx, fd_pyramid* p, const fd_wvm* wvm, const fd_svm* svm, int n, const int32_t* xywh,
                                uint8_t* target, double* weight);

/* Several five-stage detectors in one call (the detector loop of ffpDetectApp.cpp:557-600; BASELINE config 3).  All WVM
 * stages are queued first (spread over a few internal streams, so that the small kernels of different jobs overlap);
 * the host-side stages of detector i overlap the GPU work of detectors i+1...  Two jobs may share a pyramid
 * (identical layers) but not a WVM handle.  Per job: count / stage_counts / status are outputs. */
typedef struct {
    fd_pyramid* pyramid;
    const fd_wvm* wvm;
    const fd_svm* svm;
    float oe_dist, oe_ratio;
    int32_t step_x, step_y;
    const int* roi;             /* {x,y,w,h} or NULL */
    fd_detection* out;
    int32_t cap;
    int32_t count;              /* out */
    int32_t stage_counts[4];    /* out */
    int32_t status;             /* out: FD_OK or the job's error code */
    /* optional: update the job's pyramid with this frame first (fd_pyramid_update arguments), on the job's stream;
     * NULL: the pyramid has been updated by the caller.  A pyramid may be updated by one job of a batch only. */
    const uint8_t* image;
    int32_t image_w, image_h, image_channels, image_is_device;
} fd_five_stage_job;
int fd_detect_five_stage_batch(fd_ctx* ctx, fd_five_stage_job* jobs, int n);
/* The same in two halves, for callers that keep more than one frame in flight: begin validates the jobs, queues the optional
 * pyramid updates and all cascades and returns at once; the host stages (read-back, overlap elimination, the SVM stage, NMS) of
 * every job follow on the library's own threads as its cascade completes (FD_BATCH_THREADS of them, shared by all batches in flight;
 * FD_BATCH_ASYNC=0: inside end) and fill out / count / stage_counts of the job; end waits for them and sets every job's status.
 * `jobs` (and the buffers it points to) must stay alive and untouched from begin until end returns -- the library writes to them in
 * between; every batch that has begun must be ended (end also releases the ticket, whatever it returns).
 * Two batches in flight must not share pyramid or WVM handles: their scratch buffers belong to one run at a time. */
typedef struct fd_five_stage_batch fd_five_stage_batch;
int fd_five_stage_batch_begin(fd_ctx* ctx, fd_five_stage_job* jobs, int n, fd_five_stage_batch** ticket);
int fd_five_stage_batch_end(fd_ctx* ctx, fd_five_stage_batch* ticket);

/* detection::OverlapElimination::eliminate (OverlapElimination.cpp:44-105); host-side, deterministic */
int fd_overlap_elimination(const fd_detection* in, int n, float dist, float ratio, int32_t* keep_idx, int* count);
/* nonMaximaSuppression (FiveStageSlidingWindowDetector.cpp:143-184) applied to the probability map that
 * :276-285 builds from the detections (max probability per centre pixel; masked != 0: only entries > 0.3).
 * maxima_xy receives (x, y) pairs in row-major order (cv::findNonZero order); host-side. */
int fd_block_nms(const fd_detection* in, int n, int image_w, int image_h, int sz, int masked, int32_t* maxima_xy,
                 int cap_pairs, int* count);

/* Single-stage SlidingWindowDetector with the HOG feature chain of benchmarkApp
 * (BenchmarkRunner.cpp:118-126,235-242): pyramid layer filter FD_LAYER_GRADBIN, patch filter
 * HogFilter(bins, cell, block, interpolate=false, signedAndUnsigned) (HogFilter.cpp:58-122) and a
 * ProbabilisticSvmClassifier on the f32 feature vectors.
 * all_distance (host, may be NULL): every window's hyperplane distance, extraction order. */
typedef struct {
    int32_t patch_w, patch_h, step_x, step_y;
    int32_t bins, cell_size, block_size, signed_and_unsigned;
} fd_hog_params;
int fd_hog_feature_length(const fd_hog_params* hp);
int fd_detect_hog_svm(fd_ctx* ctx, fd_pyramid* p, const fd_svm* svm, const fd_hog_params* hp, fd_detection* out,
                      int64_t cap, int64_t* count, double* all_distance);
/* The same in two halves, for callers that keep more than one frame in flight (one context per frame in flight: a context's
 * scratch buffers belong to one run at a time).  begin queues pyramid-layer reads, HOG tiles, the SVM, the positive selection
 * and the read-back of the positives on the context's stream and returns at once; end waits, orders the positives by
 * extraction order and fills out / count.  Every ticket must be ended (end releases it, whatever it returns). */
typedef struct fd_hog_svm_ticket fd_hog_svm_ticket;
int fd_detect_hog_svm_begin(fd_ctx* ctx, fd_pyramid* p, const fd_svm* svm, const fd_hog_params* hp, fd_hog_svm_ticket** ticket);
int fd_detect_hog_svm_end(fd_ctx* ctx, fd_hog_svm_ticket* ticket, fd_detection* out, int64_t cap, int64_t* count);
/* FeatureExtractor::extract for every window: n_windows x feature_length floats to host (tests) */
int fd_extract_hog(fd_ctx* ctx, fd_pyramid* p, const fd_hog_params* hp, float* features, int64_t cap_windows,
                   int64_t* count);

/* detection::NonMaximumSuppression::eliminateRedundantDetections (NonMaximumSuppression.cpp:27-118):
 * IoU clustering around the best remaining detection; maximum_type 0 MAX_SCORE, 1 AVERAGE, 2 WEIGHTED_AVERAGE.  Host only.
 * out: room for n boxes.  FD_ERR_RUNTIME for an overlap threshold > 1 (the reference does not terminate). */
typedef struct {
    float score;
    int32_t x, y, w, h;   /* cv::Rect bounds */
} fd_box;
int fd_nms_iou(const fd_box* in, int n, double overlap_threshold, int maximum_type, fd_box* out, int* count);

/* imageprocessing::filtering::FhogFilter(cellSize, unsignedBinCount, interpolateBins, interpolateCells, alpha)
 * (FhogFilter.cpp:20-132, FhogFilter.hpp:120-207; descriptors by FhogAggregationFilter.cpp:38-168) on CV_8UC1 images:
 * rows = height / cell_size, cols = width / cell_size cells of 3 * unsigned_bins + 4 floats (2B signed, B unsigned
 * orientation features, 4 energy features).  Reference defaults: 8, 9, false, true, 0.2 (FhogFilter.hpp:55-56). */
typedef struct {
    int32_t cell_size, unsigned_bins, interpolate_bins, interpolate_cells;
    float alpha;
} fd_fhog_params;
int fd_fhog_size(const fd_fhog_params* fp, int width, int height, int* rows, int* cols, int* channels);
/* gray: host image (width * height bytes); out: rows * cols * channels floats (host) */
int fd_fhog_image(fd_ctx* ctx, const uint8_t* gray, int width, int height, const fd_fhog_params* fp, float* out);
/* the same for CV_8UC1 (channels 1) or CV_8UC3 (channels 3) host images: with three channels every pixel votes with the channel
 * of the largest gradient magnitude (getBinCoefficients<false>, FhogFilter.hpp:144-172); image: width * height * channels bytes */
int fd_fhog_image_channels(fd_ctx* ctx, const uint8_t* image, int width, int height, int channels, const fd_fhog_params* fp, float* out);
/* the same on kept layer `layer` of an updated gray pyramid (the layer filter of AggregatedFeaturesExtractor's feature pyramid) */
int fd_pyramid_fhog_layer(fd_ctx* ctx, fd_pyramid* p, int layer, const fd_fhog_params* fp, float* out);

/* detection::AggregatedFeaturesDetector (AggregatedFeaturesDetector.cpp:37-128) with imageFilter = GrayscaleFilter,
 * layerFilter = FhogFilter on an extraction::AggregatedFeaturesExtractor (AggregatedFeaturesExtractor.cpp:34-130): a linear
 * SVM convolved over the FHOG cell pyramid (ConvolutionFilter.cpp:27-43), windows with score > threshold, bounds through
 * the layers' actual x / y scales, rescaleWindow, NonMaximumSuppression.  SURVEY.md 8(f) row 2. */
typedef struct fd_aggregated fd_aggregated;
typedef struct {
    fd_fhog_params fhog;              /* layer filter; fhog.cell_size is the detector's cellSize */
    int32_t window_w, window_h;       /* windowSize in cells */
    int32_t octave_layer_count;
    int32_t min_window_width;         /* minWindowWidth in pixels (0: none) */
    float width_scale, height_scale;  /* rescaleWindow */
    const float* svm_weights;         /* the linear SVM's support vector, [window_h][window_w][3 * unsigned_bins + 4] */
    float svm_bias, score_threshold;  /* delta = -bias; getThreshold() */
    double nms_overlap_threshold;
    int32_t nms_maximum_type;         /* 0 MAX_SCORE, 1 AVERAGE, 2 WEIGHTED_AVERAGE */
} fd_aggregated_params;
int fd_aggregated_create(fd_ctx* ctx, const fd_aggregated_params* prm, fd_aggregated** out);
void fd_aggregated_destroy(fd_aggregated* a);
/* detectWithScores(image): final detections in out; candidates (before NMS, layer / row / column order) optionally */
int fd_aggregated_detect(fd_ctx* ctx, fd_aggregated* a, const uint8_t* image, int width, int height, int channels, int is_device,
                         fd_box* out, int cap, int* count, fd_box* candidates, int cand_cap, int* cand_count);

/* Generic histogram patch filters on the pyramid's bin-image layers (FD_LAYER_GRADBIN: 2 or 4 channels,
 * FD_LAYER_LBP: 1 channel), all built on HistogramFilter::createCellHistograms (HistogramFilter.cpp:23-197,
 * interpolating and non-interpolating):
 *  FD_HIST_HOG             HogFilter(bins, cell_size, block_size, interpolate, signed_and_unsigned)  HogFilter.cpp:58-122
 *  FD_HIST_SPATIAL         SpatialHistogramFilter(bins, cell_size, block_size, interpolate, concatenate, normalization)
 *                          SpatialHistogramFilter.cpp:56-94
 *  FD_HIST_PYRAMID_HOG     PyramidHogFilter(bins, levels, interpolate, signed_and_unsigned)  PyramidHogFilter.cpp:33-113
 *  FD_HIST_SPATIAL_PYRAMID SpatialPyramidHistogramFilter(bins, levels, interpolate, normalization)
 *                          SpatialPyramidHistogramFilter.cpp:37-81
 * normalization (HistogramFilter::Normalization): 0 none, 1 L2NORM, 2 L2HYS, 3 L1NORM, 4 L1SQRT. */
enum { FD_HIST_HOG = 0, FD_HIST_SPATIAL = 1, FD_HIST_PYRAMID_HOG = 2, FD_HIST_SPATIAL_PYRAMID = 3 };
typedef struct {
    int32_t patch_w, patch_h, step_x, step_y;
    int32_t kind, bins;
    int32_t cell_size, block_size;   /* FD_HIST_HOG, FD_HIST_SPATIAL: cell width / block width (cells) */
    int32_t levels;                  /* pyramid kinds: levelCount */
    int32_t interpolate, signed_and_unsigned, concatenate, normalization;
    int32_t cell_h, block_h;         /* cell height / block height; 0 = same as cell_size / block_size */
} fd_hist_params;
/* feature length for a bin image with `channels` channels; -1 on invalid parameters */
int fd_hist_feature_length(const fd_hist_params* hp, int channels);
/* FeatureExtractor::extract for every window: n_windows x feature_length floats to host.
 * features == NULL: only *count is set. */
int fd_extract_hist(fd_ctx* ctx, fd_pyramid* p, const fd_hist_params* hp, float* features, int64_t cap_windows,
                    int64_t* count);
/* HistogramFilter::applyTo(const Mat&) of the same four patch filters on n contiguous bin-image patches (hp->patch_w x hp->patch_h
 * pixels, `channels` bytes per pixel: 1 bin, 2 bin + weight, 4 two bins + weights; hp->step_x / step_y are ignored).
 * out: n x fd_hist_feature_length(hp, channels) floats. */
int fd_hist_patch_batch(fd_ctx* ctx, const uint8_t* bin_patches, int64_t n, int channels, const fd_hist_params* hp, float* out);
/* SlidingWindowDetector::detect with the histogram patch filter + ProbabilisticSvmClassifier on the f32
 * vectors (any kernel; wiring of BenchmarkRunner.cpp:185-263) */
int fd_detect_hist_svm(fd_ctx* ctx, fd_pyramid* p, const fd_svm* svm, const fd_hist_params* hp, fd_detection* out,
                       int64_t cap, int64_t* count, double* all_distance);

/* The "whi" feature space of ffpDetectApp.cpp:449-454 on gray pyramid windows: WhiteningFilter(alpha, cutoff)
 * (WhiteningFilter.cpp:20-81) -> HistogramEqualizationFilter (cv::equalizeHist) -> ConversionFilter(CV_32F,
 * 1/127.5, -1) -> UnitNormFilter(NORM_L2).  Reference defaults: alpha 1, cutoff 0.390625 (WhiteningFilter.hpp:31). */
typedef struct {
    int32_t patch_w, patch_h, step_x, step_y;
    float alpha, cutoff;
} fd_whi_params;
/* the chain on n contiguous w x h u8 patches (host) -> n*w*h floats (host) */
int fd_whi_batch(fd_ctx* ctx, const uint8_t* patches, int64_t n, int w, int h, float alpha, float cutoff, float* dst);
/* HistogramEqualizationFilter::applyTo ("histeq" feature space, ffpDetectApp.cpp:446-448) on n patches */
int fd_equalize_hist_batch(fd_ctx* ctx, const uint8_t* patches, int64_t n, int w, int h, uint8_t* dst);
/* the single filters of that chain on n contiguous host patches / vectors (their per-Mat ImageFilter::applyTo):
 *  fd_whitening_batch  WhiteningFilter(alpha, cutoff) alone (WhiteningFilter.cpp:20-58): u8 -> u8 (convertTo(CV_8U, 1, 127))
 *  fd_convert_batch    ConversionFilter(type, alpha, beta) (cv::Mat::convertTo): src/dst dtype FD_DTYPE_U8 | FD_DTYPE_F32, `count` values
 *  fd_unit_norm_batch  UnitNormFilter(normType) (UnitNormFilter.cpp:24-43): n vectors of `len` floats, norm_type = cv::NORM_INF 1,
 *                      NORM_L1 2, NORM_L2 4 */
int fd_whitening_batch(fd_ctx* ctx, const uint8_t* patches, int64_t n, int w, int h, float alpha, float cutoff, uint8_t* dst);
int fd_convert_batch(fd_ctx* ctx, const void* src, int src_dtype, int64_t count, double alpha, double beta, void* dst, int dst_dtype);
int fd_unit_norm_batch(fd_ctx* ctx, const float* src, int64_t n, int len, int norm_type, float* dst);
/* every window of the pyramid: n_windows x (patch_w*patch_h) floats to host; features == NULL: count only */
int fd_extract_whi(fd_ctx* ctx, fd_pyramid* p, const fd_whi_params* wp, float* features, int64_t cap_windows, int64_t* count);
/* SlidingWindowDetector::detect with the whi feature space + ProbabilisticSvmClassifier (any kernel, f32 vectors) */
int fd_detect_whi_svm(fd_ctx* ctx, fd_pyramid* p, const fd_svm* svm, const fd_whi_params* wp, fd_detection* out, int64_t cap,
                      int64_t* count, double* all_distance);

/* ---- classification::RvmClassifier / ProbabilisticRvmClassifier -----------------
 * Cascaded reduced-vector machine (RvmClassifier.cpp:75-126): level k evaluates kernel(x, rsv_k) on the whole
 * vector; through the reference's cached path the running distance is d_0 = -bias + c[0][0] K_0,
 * d_k = d_{k-1} + c[k][k] K_k; a vector leaves at the first level with d_k < thresholds[k]; positive iff the last
 * used level is passed.  Probability = 1/(1+exp(a + b d)) for every vector (ProbabilisticRvmClassifier.cpp:62). */
typedef struct fd_rvm fd_rvm;
typedef struct {
    int32_t kernel;                 /* FD_KERNEL_* (the reference's loader builds RBF or polynomial, RvmClassifier.cpp:196-204) */
    double p0, p1, p2;              /* like fd_svm_model */
    int32_t num_filters, num_used;  /* num_used: setNumFiltersToUse (0 or > num_filters: all) */
    int32_t filter_w, filter_h;     /* reduced set vectors are filter_h x filter_w f32 images */
    const float* support_vectors;   /* [num_filters][filter_w*filter_h] */
    const float* coefficients;      /* lower triangle, coefficients[k][i] (i <= k) at k(k+1)/2 + i */
    const float* thresholds;        /* [num_filters] hierarchicalThresholds */
    float bias;
    double logistic_a, logistic_b;
} fd_rvm_model;
int fd_rvm_create(fd_ctx* ctx, const fd_rvm_model* model, fd_rvm** out);
void fd_rvm_destroy(fd_rvm* m);
/* computeHyperplaneDistance (RvmClassifier.cpp:75-85) of n f32 vectors (host): last level and distance */
int fd_rvm_eval_batch(fd_ctx* ctx, const fd_rvm* rvm, const float* features, int64_t n, int32_t* out_level, double* out_distance);
/* u8 patch feature spaces of ffpDetectApp.cpp:446-461 */
enum { FD_FEATURE_GRAY = 0, FD_FEATURE_HQ64 = 1, FD_FEATURE_HISTEQ = 2 };
typedef struct {
    int32_t feature_space;          /* FD_FEATURE_* */
    float conv_scale, conv_shift;   /* ConversionFilter(CV_32F, scale, shift): x = float(u8) * scale + shift */
    int32_t step_x, step_y;
} fd_rvm_detect_params;
/* SlidingWindowDetector::detect (SlidingWindowDetector.cpp:87-98) with a ProbabilisticRvmClassifier ("prvm",
 * ffpDetectApp.cpp:484).  all_level / all_distance (may be NULL): per window in extraction order. */
int fd_detect_rvm(fd_ctx* ctx, fd_pyramid* p, const fd_rvm* rvm, const fd_rvm_detect_params* dp, const int* roi, fd_detection* out,
                  int64_t cap, int64_t* count, int32_t* all_level, double* all_distance);

/* ---- supervised descent: superviseddescent::SdmLandmarkModel / SdmLandmarkModelFitting ---------- */
typedef struct {
    int32_t num_landmarks;      /* L */
    int32_t num_steps;          /* S (numCascadeSteps) */
    const float* mean;          /* 2L: x0..xL-1, y0..yL-1 (SdmLandmarkModel.cpp:53-56) */
    const float* const* R;      /* per step: (feature_dim+1) x 2L row-major regressor (last row = bias) */
    const int32_t* R_rows;      /* per step: feature_dim + 1 */
    int32_t hog_variant;        /* 0 DalalTriggs, 1 UoCTTI (VlHogDescriptorExtractor::VlHogType) */
    const int32_t* desc_params; /* NULL: optimize() as the reference compiles it -- `if (true) { // adaptive` (SdmLandmarkModel.hpp:209,
                                 * 243): every step uses the 30x30 / 3x3 cells / 9 bins descriptor whatever the model file says
                                 * (DescriptorExtractor.hpp:132-139), R_rows[s] = L*279 + 1 (UoCTTI) or L*324 + 1.  Non-NULL: the
                                 * non-adaptive `else` branches of :236-238 and :246-248 (present upstream, compiled out by the `if
                                 * (true)`): per step {numCells, cellSize, numBins} of VlHogDescriptorExtractor(type, numCells, cellSize,
                                 * numBins) (SdmLandmarkModel.cpp:188-204), getDescriptors(image, points) with windowSizeHalf = 0 and
                                 * modelShape + deltaShape.t() without the face-size factor.  This is the only way the regressors of the
                                 * reference's shipped model (detect-landmarks/share/models/SDM_Model_HOG_Zhenhua_11012014.txt: 144 / 144 /
                                 * 64 / 64 / 16 dimensions per landmark) can be applied at all.
                                 * The struct MUST be zero-initialised before its fields are set (`fd_sdm_model md = {0};`): this trailing
                                 * field was added in round 5, and a caller that fills the older fields one by one would otherwise hand
                                 * in a wild pointer.  Values are validated (1 <= numCells <= 8, 2 <= cellSize <= 64, 1 <= numBins <= 32). */
} fd_sdm_model;
int fd_sdm_create(fd_ctx* ctx, const fd_sdm_model* model, fd_sdm** out);
void fd_sdm_destroy(fd_sdm* m);
/* VlHogDescriptorExtractor::getDescriptors (DescriptorExtractor.hpp:106-219), adaptive parameters
 * (window_size_half > 0: 30x30 patch, 3x3 cells of 10, 9 bins) or fixed (num_cells, cell_size, num_bins).
 * gray: host image; out: n x len floats; *len receives the descriptor length. */
int fd_sdm_descriptors(fd_ctx* ctx, const uint8_t* gray, int width, int height, const float* px, const float* py,
                       int n, int window_size_half, int variant, int num_cells, int cell_size, int num_bins,
                       float* out, int* len);
/* SdmLandmarkModelFitting::alignRigid + optimize (SdmLandmarkModel.hpp:156-192,199-256) for a batch of
 * B gray images of identical size (contiguous, host or device) and one face box {x,y,w,h} each.
 * shapes_out: B x 2L floats.  status_out (may be NULL): per face 0 ok / 1 window left the image
 * (the reference would throw). */
int fd_sdm_fit_batch(fd_ctx* ctx, const fd_sdm* m, const uint8_t* gray_images, int width, int height, int batch,
                     const int32_t* face_boxes, int images_on_device, float* shapes_out, int32_t* status_out);
/* SdmLandmarkModelFitting::optimize only (SdmLandmarkModel.hpp:199-256): shapes_inout holds the B initial
 * shapes (2L floats each: x0..xL-1, y0..yL-1) and receives the optimised ones. */
int fd_sdm_optimize_batch(fd_ctx* ctx, const fd_sdm* m, const uint8_t* gray_images, int width, int height, int batch,
                          int images_on_device, float* shapes_inout, int32_t* status_out);
/* Asynchronous form of fd_sdm_fit_batch (the five-stage path has _begin/_end, so has this): _begin queues the whole fit of a batch
 * (its S cascade steps and the read-back) and returns; _end waits and delivers shapes / status.  Several batches of one model can be
 * in flight from ONE host thread; each has its own scratch set.  Host images / boxes of a batch must stay valid until its _end, and the
 * model must outlive its tickets (a ticket holds a plain pointer to it).  Device-resident images are read behind everything queued on
 * the context's stream at the time of _begin. */
typedef struct fd_sdm_ticket fd_sdm_ticket;
int fd_sdm_fit_batch_begin(fd_ctx* ctx, const fd_sdm* model, const uint8_t* gray_images, int width, int height, int batch,
                           const int32_t* face_boxes, int images_on_device, fd_sdm_ticket** ticket);
int fd_sdm_fit_batch_end(fd_ctx* ctx, fd_sdm_ticket* ticket, float* shapes_out, int32_t* status_out);

/* ---- image-shard data parallelism (north_star: "images shard embarrassingly across the 8 GPUs of one node with a single RCCL gather
 * of detections over xGMI").  One process per GPU; image i belongs to rank i mod world; models are replicated; no data-path
 * collective.  The reference has no counterpart (it is single-threaded, ffpDetectApp.cpp:548-659 loops over the images of a
 * source); a maintainer binds these around that loop (ffp_detect_app --gpus N).
 *   fd_dist_unique_id   rank 0 creates the communicator id (ncclGetUniqueId, 128 bytes) and hands it to the other ranks by any means
 *   fd_dist_init        joins the communicator on the context's device (ncclCommInitRank); world 1 needs no id and no librccl: the
 *                       rank gathers from itself, whatever id holds.  (FD_DIST_FORCE_COMM=1 in the environment + an id: a real one-rank
 *                       communicator, the gather goes through ncclAllGather -- what the one-GPU tests use to run librccl itself.)
 *   fd_pack_records     fd_detection -> fixed-stride records {image, detector, cx, cy, w, h, score, probability} (all exact in fp64)
 *   fd_dist_gather_records  a 64-byte header exchange (every rank's count) and ONE ncclAllGather of max-count + 1 rows per rank, on the
 *                       handle's own stream (nothing queued on the context's stream is waited for or held up); every rank receives the
 *                       records of all ranks ordered by (image, detector, original order).  cap_per_rank: the most records a rank
 *                       may contribute; *truncated != 0: a rank had more (the surplus was dropped).
 *                       COLLECTIVE: every rank of the communicator makes the call, with the same cap_per_rank (a rank that packed with
 *                       another stride is reported as FD_ERR_INVALID_ARGUMENT on all ranks).  The collective runs once per set of
 *                       records: with all == NULL the call returns the count, with all_cap too small FD_ERR_CAPACITY -- either way the
 *                       gathered records stay in the handle, and the next call with a large enough buffer delivers them WITHOUT another
 *                       collective (so a retry on some ranks only cannot deadlock).  local / n_local of such a follow-up call are ignored.
 *                       FD_RCCL_LIB names another library with the five nccl entry points (tests: tests/stub_rccl).
 *   fd_dist_gather_begin / _end  the same gather in two halves: _begin exchanges the headers (a wait of tens of microseconds on the
 *                       gather stream) and queues the payload collective, _end waits for it and delivers -- the records of one
 *                       interval travel while the caller processes the next interval's images.  One gather in flight per handle; local
 *                       may be reused as soon as _begin returns.  _end has fd_dist_gather_records' count-only / capacity-retry rules.
 *   fd_dist_gather_discard  drops a gathered set the caller does not want to fetch (after a count-only call or FD_ERR_CAPACITY), so that
 *                       the next fd_dist_gather_records is a new collective.  Every rank must drop or take a set: a rank that still
 *                       holds one would answer the next call from its handle while the others enter ncclAllGather.
 *   fd_dist_gather_pending  1 while the handle holds a gathered set that has not been delivered or dropped. */
#define FD_DIST_ID_BYTES 128
typedef struct fd_dist fd_dist;
typedef struct fd_record {
    double image, detector, cx, cy, w, h, score, probability;
} fd_record;
int fd_dist_owner(int64_t image_index, int world);
int fd_dist_unique_id(uint8_t* id /* FD_DIST_ID_BYTES */);
int fd_dist_init(fd_ctx* ctx, int rank, int world, const uint8_t* id, fd_dist** out);
void fd_dist_destroy(fd_dist* d);
int fd_dist_rank(const fd_dist* d);
int fd_dist_world(const fd_dist* d);
int fd_pack_records(int64_t image_id, int32_t detector_id, const fd_detection* dets, int n, fd_record* out);
int fd_dist_gather_records(fd_dist* d, const fd_record* local, int n_local, int cap_per_rank, fd_record* all, int64_t all_cap,
                           int64_t* n_all, int* truncated);
int fd_dist_gather_begin(fd_dist* d, const fd_record* local, int n_local, int cap_per_rank);
int fd_dist_gather_end(fd_dist* d, fd_record* all, int64_t all_cap, int64_t* n_all, int* truncated);
void fd_dist_gather_discard(fd_dist* d);
int fd_dist_gather_pending(const fd_dist* d);

#ifdef __cplusplus
}
#endif
#endif /* FD_HIP_H_ */
