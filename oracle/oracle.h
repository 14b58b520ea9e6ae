/* oracle/oracle.h -- C API of the CPU oracle.  TEST INFRASTRUCTURE ONLY (see orc_common.h).
 * Every function cites the reference file:line it restates (paths relative to the
 * elador/FeatureDetection tree).  Loaded from Python tests through ctypes. */
#ifndef ORACLE_H_
#define ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- image primitives (OpenCV 2.4 semantics) ------------- */
void orc_bgr2gray(const uint8_t* bgr, int w, int h, uint8_t* gray);                 /* GrayscaleFilter.cpp:18-24 */
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh); /* ImagePyramid.cpp:177 */
void orc_pyrdown_u8(const uint8_t* src, int sw, int sh, uint8_t* dst);              /* ImagePyramid.cpp:186 */
void orc_resize_linear_f32(const float* src, int sw, int sh, float* dst, int dw, int dh); /* DescriptorExtractor.hpp:183 */
void orc_gradient_filter(const uint8_t* src, int w, int h, int ksize, int blur, uint8_t* dst2ch); /* GradientFilter.cpp:38-59 */
void orc_gradient_binning_lut(int bins, int signedGradients, int interpolate, uint8_t* lut); /* GradientBinningFilter.cpp:18-60; lut 65536*(2|4) bytes */
void orc_gradient_binning(const uint8_t* grad2ch, int n, int bins, int signedGradients, int interpolate, uint8_t* dst);
void orc_lbp(const uint8_t* src, int w, int h, int type, uint8_t* dst);              /* LbpFilter.cpp:56-85; type 0 LBP8, 1 LBP8_UNIFORM, 2 LBP4, 3 LBP4_ROTATED */
void orc_greyworld(const uint8_t* bgr, int w, int h, uint8_t* dst);                  /* GreyWorldNormalizationFilter.cpp:20-71 */
void orc_equalize_hist(const uint8_t* src, int w, int h, int stride, uint8_t* dst);  /* HistogramEqualizationFilter.cpp (cv::equalizeHist) */
void orc_histeq64(const uint8_t* src, int w, int h, int stride, uint8_t* dst);       /* HistEq64Filter.cpp:32-125 */
void orc_whi(const uint8_t* src, int w, int h, int stride, float alpha, float cutoff, float* dst); /* ffpDetectApp.cpp:449-454 chain */
void orc_whitening(const uint8_t* src, int w, int h, int stride, float alpha, float cutoff, uint8_t* dst); /* WhiteningFilter.cpp:20-58 alone */

/* ---------------- histogram features ---------------- */
/* HogFilter.cpp:58-122 on a bin image patch (ch = 1, 2 or 4).  Returns feature length. */
int orc_hog_filter(const uint8_t* img, int w, int h, int ch, int stride_bytes, int bins,
                   int cellW, int cellH, int blockW, int blockH, int interpolate,
                   int signedAndUnsigned, float* out);
/* SpatialHistogramFilter.cpp:56-94; normalization 0 none,1 L2,2 L2HYS,3 L1,4 L1SQRT */
int orc_spatial_histogram(const uint8_t* img, int w, int h, int ch, int stride_bytes, int bins,
                          int cellW, int cellH, int blockW, int blockH, int interpolate,
                          int concatenate, int normalization, float* out);

/* PyramidHogFilter.cpp:33-113 */
int orc_pyramid_hog(const uint8_t* img, int w, int h, int ch, int stride_bytes, int bins, int levels,
                    int interpolate, int signedAndUnsigned, float* out);
/* SpatialPyramidHistogramFilter.cpp:37-81 */
int orc_spatial_pyramid_histogram(const uint8_t* img, int w, int h, int ch, int stride_bytes, int bins, int levels,
                                  int interpolate, int normalization, float* out);

/* filtering::FhogFilter::applyTo on a CV_8UC1 image (FhogFilter.cpp:59-72 + FhogAggregationFilter.cpp:38-168):
 * rows = h / cellSize, cols = w / cellSize cells of 3 * unsignedBinCount + 4 floats; returns the number of floats */
int orc_fhog(const uint8_t* img, int w, int h, int stride, int cellSize, int unsignedBinCount, int interpolateBins,
             int interpolateCells, float alpha, float* out, int* rows, int* cols);
/* the same on a CV_8UC1 (channels 1) or CV_8UC3 (channels 3: per pixel the channel with the largest gradient magnitude,
   FhogFilter.hpp:144-172) image; stride in bytes */
int orc_fhog_channels(const uint8_t* img, int w, int h, int channels, int stride, int cellSize, int unsignedBinCount, int interpolateBins,
                      int interpolateCells, float alpha, float* out, int* rows, int* cols);

/* ---------------- pyramid + window enumeration ---------------- */
typedef struct orc_pyramid orc_pyramid;
orc_pyramid* orc_pyramid_create(int octaveLayerCount, double minScale, double maxScale); /* ImagePyramid.cpp:67-77 */
orc_pyramid* orc_pyramid_create_inc(double incScale, double minScale, double maxScale);  /* ImagePyramid.cpp:79-92 */
void orc_pyramid_destroy(orc_pyramid* p);
/* layer filter chain: kind 0 none; 1 GradientFilter(gradKernel, blurKernel)+GradientBinningFilter(bins,signed,interp);
 * 2 LbpFilter(lbpType) */
void orc_pyramid_set_layer_filter(orc_pyramid* p, int kind, int bins, int signedGradients,
                                  int interpolate, int gradKernel, int blurKernel, int lbpType);
void orc_pyramid_update(orc_pyramid* p, const uint8_t* img, int w, int h, int ch);   /* ImagePyramid.cpp:170-198 */
int orc_pyramid_octave_layers(const orc_pyramid* p);
double orc_pyramid_inc_scale(const orc_pyramid* p);
int orc_pyramid_num_layers(const orc_pyramid* p);
void orc_pyramid_layer_info(const orc_pyramid* p, int i, int* index, double* scale, int* w, int* h, int* ch);
const uint8_t* orc_pyramid_layer_data(const orc_pyramid* p, int i);
/* DirectPyramidFeatureExtractor.cpp:75-123.  out rows: {layerPos, lx, ly, cx, cy, ow, oh}; returns count
 * (writes at most cap rows). roi = {x,y,w,h}, all zero = whole image. */
int64_t orc_extract_windows(const orc_pyramid* p, int pw, int ph, int stepX, int stepY,
                            const int* roi, int32_t* out, int64_t cap);

/* ---------------- classifiers ---------------- */
typedef struct {
    int32_t filter_w, filter_h;
    int32_t num_filters;      /* numLinFilters */
    int32_t num_used;         /* numUsedFilters (after setNumUsedFilters clamp) */
    int32_t num_per_level;    /* numFiltersPerLevel */
    float basis_param;        /* basisParam */
    float bias;               /* lin_thresholds[i] (all equal to the bias, WvmClassifier.cpp:567-570) */
    const float* thresholds;  /* hierarchicalThresholds[num_filters] */
    const float* hk_weights;  /* [num_filters*num_filters] row-major, entries p<=k used */
    const double* pp;         /* app_rsv_convol[num_filters] */
    const int32_t* val_off;   /* [num_filters+1]: range of grey values of filter k */
    const double* val;        /* area[k]->val[v] concatenated */
    const int32_t* rec_off;   /* [val_off[num_filters]+1]: range of rects of (k,v) */
    const uint8_t* rects;     /* x1,y1,x2,y2 inclusive, 4 bytes per rect */
    double logistic_a, logistic_b;
} orc_wvm_desc;
typedef struct orc_wvm orc_wvm;
orc_wvm* orc_wvm_create(const orc_wvm_desc* d);
void orc_wvm_destroy(orc_wvm* m);
/* WvmClassifier.cpp:100-149 + 191-346.  patch = filter_w*filter_h contiguous u8. */
void orc_wvm_eval(const orc_wvm* m, const uint8_t* patch, int32_t* lastLevel, float* fout);
int orc_wvm_classify(const orc_wvm* m, int lastLevel, double fout);                 /* WvmClassifier.cpp:91-98 */
double orc_wvm_probability(const orc_wvm* m, double fout);                         /* ProbabilisticWvmClassifier.cpp:52 */
void orc_iimg(const uint8_t* patch, int w, int h, int sqr, float* out);            /* IImg.cpp:26-65 */

/* kernel: 0 linear, 1 polynomial(alpha=p0, constant=p1, degree=p2), 2 rbf(gamma=p0), 3 hik.
 * dtype: 0 u8, 1 f32 */
typedef struct orc_svm orc_svm;
orc_svm* orc_svm_create(int kernel, double p0, double p1, double p2, int nsv, int dim, int dtype,
                        const void* sv, const float* coeff, float bias, float threshold,
                        double logistic_a, double logistic_b);
void orc_svm_destroy(orc_svm* m);
double orc_svm_distance(const orc_svm* m, const void* x);                          /* SvmClassifier.cpp:55-60 */
int orc_svm_classify(const orc_svm* m, double dist);                               /* SvmClassifier.cpp:44-46 */
double orc_svm_probability(const orc_svm* m, double dist);                         /* ProbabilisticSvmClassifier.cpp:54-58 */
void orc_svm_distance_batch(const orc_svm* m, const void* x, int64_t n, double* out);

/* RvmClassifier (RvmClassifier.cpp:75-126) on f32 feature vectors; coeffPacked: coefficients[k][i] at k(k+1)/2 + i */
typedef struct orc_rvm orc_rvm;
orc_rvm* orc_rvm_create(int kernel, double p0, double p1, double p2, int numFilters, int numUse, int dim, const float* sv,
                        const float* coeffPacked, const float* thresholds, float bias, double logisticA, double logisticB);
void orc_rvm_destroy(orc_rvm* m);
void orc_rvm_eval(const orc_rvm* m, const float* x, int32_t* lastLevel, double* distance);   /* computeHyperplaneDistance :75-85 */
void orc_rvm_eval_batch(const orc_rvm* m, const float* x, int64_t n, int32_t* lastLevel, double* distance);
int orc_rvm_classify(const orc_rvm* m, int lastLevel, double distance);                      /* :68-73 */
double orc_rvm_probability(const orc_rvm* m, double distance);                               /* ProbabilisticRvmClassifier.cpp:62 */

/* ---------------- detection ---------------- */
typedef struct {
    int32_t cx, cy, w, h;     /* Patch centre/size in the original image (Patch.hpp) */
    int32_t layer, lx, ly;    /* layer position in pyramid order, top-left in the layer */
    int32_t level;            /* WVM last filter level (or -1) */
    int32_t positive;
    float fout;               /* classifier output (WVM fout or (float)SVM distance) */
    double prob;
} orc_det;
/* OverlapElimination.cpp:44-105; idx_out receives indices into the input, returns count */
int orc_overlap_elimination(int n, const orc_det* in, float dist, float ratio, int32_t* idx_out);
/* FiveStageSlidingWindowDetector.cpp:143-184; mask may be NULL */
void orc_block_nms(const float* map, int H, int W, int sz, const uint8_t* mask, uint8_t* dst);
/* SlidingWindowDetector.cpp:87-98 with a WVM classifier and the HistEq64 patch filter.
 * If all_level/all_fout non-NULL they receive per-window results for ALL windows in extraction order.
 * Returns number of positives written to out (at most cap). */
int64_t orc_sliding_wvm(const orc_pyramid* p, const orc_wvm* m, int stepX, int stepY, const int* roi,
                        orc_det* out, int64_t cap, int32_t* all_level, float* all_fout);
/* FiveStageSlidingWindowDetector.cpp:187-320 (roi==NULL) / :331-380 (roi!=NULL).
 * stage_counts[4] = {wvm positives, after OE, svm positives, final}. */
/* AggregatedFeaturesDetector::getPositiveWindows (AggregatedFeaturesDetector.cpp:87-106) for GrayscaleFilter + FhogFilter on
 * an AggregatedFeaturesExtractor: candidate windows (score, rescaled bounds) before NMS, in layer / row / column order.
 * Returns their number (may exceed cap), -1 when the feature pyramid has fewer than two layers (the reference throws). */
int orc_aggregated_candidates(const uint8_t* img, int w, int h, int ch, int cellSize, int unsignedBinCount, int interpolateBins,
                              int interpolateCells, float alpha, int windowW, int windowH, int octaveLayerCount, int minWindowWidth,
                              float widthScale, float heightScale, const float* svmWeights, float svmBias, float scoreThreshold,
                              float* outScore, int32_t* outXywh, int cap);
/* NonMaximumSuppression::eliminateRedundantDetections (NonMaximumSuppression.cpp:27-118); returns the number of detections (-1: overlap threshold > 1) */
int orc_nms_iou(int n, const float* score, const int32_t* xywh, double overlapThreshold, int maximumType, float* outScore, int32_t* outXywh);
/* DirectPyramidFeatureExtractor::extract(x, y, width, height) (:67-73,134-153): 1 + {layerPos, lx, ly, cx, cy, ow, oh} or 0 */
int orc_extract_single(const orc_pyramid* p, int pw, int ph, int x, int y, int width, int height, int32_t* out7);
/* condensation::WvmSvmModel::evaluate(image, samples) (WvmSvmModel.cpp:69-118): samples {x, y, width, height} */
void orc_wvm_svm_evaluate(const orc_pyramid* p, const orc_wvm* wvm, const orc_svm* svm, int n, const int32_t* xywh,
                          uint8_t* target, double* weight);
int orc_five_stage(const orc_pyramid* p, int imgW, int imgH, const orc_wvm* wvm, const orc_svm* svm,
                   float oeDist, float oeRatio, int stepX, int stepY, const int* roi,
                   orc_det* out, int cap, int32_t* stage_counts);
/* bench.py cpu_baseline only: per-thread phase timers (extract = patch filters, classify = classifiers + OE + NMS) */
void orc_phase_timing(int enable);
void orc_phase_get(double* extract_s, double* classify_s);
/* Single-stage SlidingWindowDetector with HOG (pyramid layer filter kind 1) + HogFilter patch filter + SVM
 * (BenchmarkRunner.cpp:235-242 recipe).  all_dist receives every window's hyperplane distance. */
int64_t orc_sliding_hog_svm(const orc_pyramid* p, const orc_svm* svm, int pw, int ph, int stepX, int stepY,
                            int bins, int cell, int block, int interpolate, int signedAndUnsigned,
                            orc_det* out, int64_t cap, double* all_dist, float* feat_out, int64_t feat_cap_windows);
/* bench.py cpu_baseline only: the same loop over windows first, first + step, ... (a bounded sample of one full-size frame) */
int64_t orc_sliding_hog_svm_sample(const orc_pyramid* p, const orc_svm* svm, int pw, int ph, int stepX, int stepY,
                                   int bins, int cell, int block, int interpolate, int signedAndUnsigned,
                                   orc_det* out, int64_t cap, double* all_dist, float* feat_out, int64_t feat_cap_windows,
                                   int64_t first, int64_t step, int64_t* visited);

/* ---------------- SDM ---------------- */
/* hog.c:174-…,596-721,858-1063 (VLFeat HOG), variant 0 DalalTriggs, 1 UoCTTI.  Returns dims; out = hw*hh*dim planar */
int orc_vlhog(const float* img, int w, int h, int cellSize, int numOrient, int variant, float* out,
              int* hogW, int* hogH);
/* DescriptorExtractor.hpp:106-219.  adaptive iff windowSizeHalf>0.  Returns descriptor length per point,
 * or -1 if the reference would throw (roi outside extended image).  out = n * len floats. */
int orc_sdm_descriptors(const uint8_t* gray, int w, int h, const float* px, const float* py, int n,
                        int windowSizeHalf, int variant, int numCells, int cellSize, int numBins, float* out);
/* SdmLandmarkModel.hpp:156-192 */
void orc_sdm_align_rigid(float* shape, int L, const int* faceBox);
/* SdmLandmarkModel.hpp:199-256 (adaptive branch).  R[s] = (featDim+1) x 2L row-major.  Returns 0 or -1 */
int orc_sdm_optimize(const uint8_t* gray, int w, int h, float* shape, int L, int S,
                     const float* const* R, const int* Rrows, int variant);
/* The non-adaptive branches of the same function (:236-238, :246-248; compiled out upstream by `if (true)`): getDescriptors(image,
 * points) with the extractor's own parameters descParams[3*step + {0,1,2}] = {numCells, cellSize, numBins}, shape += delta.
 * Used for the reference's one shipped model (detect-landmarks/share/models/SDM_Model_HOG_Zhenhua_11012014.txt), whose regressors
 * are 144/144/64/64/16 dimensions per landmark and cannot be multiplied with the adaptive 279-dimensional descriptors. */
int orc_sdm_optimize_fixed(const uint8_t* gray, int w, int h, float* shape, int L, int S,
                           const float* const* R, const int* Rrows, int variant, const int* descParams);

#ifdef __cplusplus
}
#endif
#endif
