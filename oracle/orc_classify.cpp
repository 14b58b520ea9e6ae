// oracle/orc_classify.cpp -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
// libClassification: integral image, WVM cascade, kernel SVM, logistic wrappers.
#include "orc_common.h"
#include "orc_internal.h"
#include "oracle.h"
#include <cstring>

namespace orc {

// IImg.cpp:26-65 (float accumulation, per-row running sum)
static void iimg(const uchar* in_img, int w, int h, bool sqr, float* data) {
    int c, r;
    long z, zb;
    float rowsum;
    if (sqr) {
        rowsum = 0;
        for (c = 0; c < w; c++) { rowsum += in_img[c] * in_img[c]; data[c] = rowsum; }
        z = w; zb = 0;
        for (r = 1; r < h; r++) {
            rowsum = 0;
            for (c = 0; c < w; c++) { rowsum += in_img[z + c] * in_img[z + c]; data[z + c] = data[zb + c] + rowsum; }
            z += w; zb += w;
        }
    } else {
        rowsum = 0;
        for (c = 0; c < w; c++) { rowsum += in_img[c]; data[c] = rowsum; }
        z = w; zb = 0;
        for (r = 1; r < h; r++) {
            rowsum = 0;
            for (c = 0; c < w; c++) { rowsum += in_img[z + c]; data[z + c] = data[zb + c] + rowsum; }
            z += w; zb += w;
        }
    }
}

// WvmClassifier.cpp:100-149 (computeHyperplaneDistance) and :191-346 (linEvalWvmHisteq64)
void Wvm::eval(const uchar* patch, int& lastLevel, float& foutOut) const {
    const int d = fw * fh;
    std::vector<float> iix(d), iixx(d);
    iimg(patch, fw, fh, false, iix.data());
    iimg(patch, fw, fh, true, iixx.data());
    std::vector<float> filter_output(numFilters, 0.f), u_kernel_eval(numFilters, 0.f);
    for (int n = 0; n < numPerLevel; n++) u_kernel_eval[n] = 0.0f;
    int filter_level = -1;
    float fout = 0.0;
    const int lx = fw - 1, ly = fh - 1;
    do {
        filter_level++;
        const int level = filter_level, n = filter_level % numPerLevel;
        // ---- linEvalWvmHisteq64 ----
        const float* this_weight = hkWeights.data() + (size_t)level * numFilters;
        float res = -bias;  // -lin_thresholds[level]
        double norm_new = 0.0F, sum_xp = 0.0F;
        float sumv = 0.0f, sumv0 = 0.0f;
        const int dr = ly * fw + lx;
        norm_new = iixx[dr];
        sumv0 = iix[dr];
        const int v0 = valOff[level], cntval = valOff[level + 1] - v0;
        for (int v = 1; v < cntval; v++) {
            sumv = 0;
            for (int r = recOff[v0 + v]; r < recOff[v0 + v + 1]; r++) {
                const uchar* rec = rects.data() + 4 * (size_t)r;  // x1,y1,x2,y2
                int ax1 = rec[0] - 1, ax2 = rec[2], ay1 = rec[1];
                int ay1w = (ay1 - 1) * fw, ay2w = rec[3] * fw;
                if (ax1 + 1 > 0 && ay1 > 0)
                    sumv += iix[ay2w + ax2] - iix[ay1w + ax2] - iix[ay2w + ax1] + iix[ay1w + ax1];
                else if (ax1 + 1 > 0)
                    sumv += iix[ay2w + ax2] - iix[ay2w + ax1];
                else if (ay1 > 0)
                    sumv += iix[ay2w + ax2] - iix[ay1w + ax2];
                else
                    sumv += iix[ay2w + ax2];
            }
            sumv0 -= sumv;
            sum_xp += sumv * val[v0 + v];
        }
        sum_xp += sumv0 * val[v0];
        sum_xp += u_kernel_eval[n];
        u_kernel_eval[n] = (float)sum_xp;
        norm_new -= 2 * sum_xp;
        norm_new += pp[level];
        filter_output[level] = (float)(std::exp(-basisParam * norm_new));
        for (int p = 0; p <= level; ++p) res += this_weight[p] * filter_output[p];
        fout = res;
    } while (fout >= thresholds[filter_level] && filter_level + 1 < numUsed);
    lastLevel = filter_level;
    foutOut = fout;
}

// RbfKernel.hpp:78-108, HistogramIntersectionKernel.hpp:67-93, LinearKernel.hpp:27-29 (cv::Mat::dot
// accumulates in double; 8U uses an exact int dot), PolynomialKernel.hpp:35-37,73-81
double Svm::kernelValue(const void* xv, int i) const {
    if (dtype == 0) {
        const uchar* x = (const uchar*)xv;
        const uchar* s = svU8.data() + (size_t)i * dim;
        if (kernel == 2) {
            int sum = 0;
            for (int k = 0; k < dim; ++k) { int diff = x[k] - s[k]; sum += diff * diff; }
            return std::exp(-p0 * sum);
        }
        if (kernel == 3) {
            int sum = 0;
            for (int k = 0; k < dim; ++k) sum += std::min(x[k], s[k]);
            return sum;
        }
        double dot = 0;
        for (int k = 0; k < dim; ++k) dot += (double)(x[k] * s[k]);
        if (kernel == 0) return dot;
        double base = p0 * dot + p1, tmp = base, ret = 1.0;
        for (int t = (int)p2; t > 0; t /= 2) { if (t % 2 == 1) ret *= tmp; tmp = tmp * tmp; }
        return ret;
    } else {
        const float* x = (const float*)xv;
        const float* s = svF32.data() + (size_t)i * dim;
        if (kernel == 2) {
            float sum = 0;
            for (int k = 0; k < dim; ++k) { float diff = x[k] - s[k]; sum += diff * diff; }
            return std::exp(-p0 * sum);
        }
        if (kernel == 3) {
            float sum = 0;
            for (int k = 0; k < dim; ++k) sum += std::min(x[k], s[k]);
            return sum;
        }
        double dot = 0;
        for (int k = 0; k < dim; ++k) dot += (double)x[k] * s[k];
        if (kernel == 0) return dot;
        double base = p0 * dot + p1, tmp = base, ret = 1.0;
        for (int t = (int)p2; t > 0; t /= 2) { if (t % 2 == 1) ret *= tmp; tmp = tmp * tmp; }
        return ret;
    }
}

// SvmClassifier.cpp:55-60
double Svm::distance(const void* x) const {
    double distance = -bias;
    for (int i = 0; i < nsv; ++i) distance += coeff[i] * kernelValue(x, i);
    return distance;
}

// RvmClassifier::computeHyperplaneDistance (RvmClassifier.cpp:75-85) through computeHyperplaneDistanceCached
// (:94-112).  Note the quirk of the cached path, kept here: filterEvalCache is constructed with numFiltersToUse
// elements (:78), level 0 clears it and computes -bias + c[0][0] K_0, and from then on `size == filterLevel`
// holds, so level k only adds c[k][k] K_k to the level k-1 value (the off-diagonal coefficients c[k][i], i < k,
// are never used unless numFiltersToUse == 1).
void Rvm::eval(const float* x, int& lastLevel, double& distanceOut) const {
    int filterLevel = -1;
    double hyperplaneDistance = 0;
    std::vector<double> cache((size_t)numUse);
    do {
        ++filterLevel;
        if (cache.size() == (size_t)filterLevel && filterLevel != 0) {
            double distance = cache[filterLevel - 1];
            distance += coeff[(size_t)filterLevel * (filterLevel + 1) / 2 + filterLevel] * store.kernelValue(x, filterLevel);
            cache.push_back(distance);
            hyperplaneDistance = distance;
        } else {
            cache.clear();
            double distance = -bias;
            for (int i = 0; i <= filterLevel; ++i)
                distance += coeff[(size_t)filterLevel * (filterLevel + 1) / 2 + i] * store.kernelValue(x, i);
            cache.push_back(distance);
            hyperplaneDistance = distance;
        }
    } while (hyperplaneDistance >= thresholds[filterLevel] && filterLevel + 1 < numUse);
    lastLevel = filterLevel;
    distanceOut = hyperplaneDistance;
}

}  // namespace orc

using namespace orc;
extern "C" {
void orc_iimg(const uint8_t* patch, int w, int h, int sqr, float* out) { iimg(patch, w, h, sqr != 0, out); }

orc_wvm* orc_wvm_create(const orc_wvm_desc* d) {
    Wvm* m = new Wvm();
    m->fw = d->filter_w; m->fh = d->filter_h; m->numFilters = d->num_filters;
    m->numUsed = (d->num_used > d->num_filters || d->num_used == 0) ? d->num_filters : d->num_used;  // WvmClassifier.cpp:151-158
    m->numPerLevel = d->num_per_level; m->basisParam = d->basis_param; m->bias = d->bias;
    m->thresholds.assign(d->thresholds, d->thresholds + d->num_filters);
    m->hkWeights.assign(d->hk_weights, d->hk_weights + (size_t)d->num_filters * d->num_filters);
    m->pp.assign(d->pp, d->pp + d->num_filters);
    m->valOff.assign(d->val_off, d->val_off + d->num_filters + 1);
    int nval = d->val_off[d->num_filters];
    m->val.assign(d->val, d->val + nval);
    m->recOff.assign(d->rec_off, d->rec_off + nval + 1);
    m->rects.assign(d->rects, d->rects + 4 * (size_t)d->rec_off[nval]);
    m->logisticA = d->logistic_a; m->logisticB = d->logistic_b;
    return (orc_wvm*)m;
}
void orc_wvm_destroy(orc_wvm* m) { delete (Wvm*)m; }
void orc_wvm_eval(const orc_wvm* m, const uint8_t* patch, int32_t* lastLevel, float* fout) {
    int l; float f;
    ((const Wvm*)m)->eval(patch, l, f);
    *lastLevel = l; *fout = f;
}
int orc_wvm_classify(const orc_wvm* m, int lastLevel, double fout) { return ((const Wvm*)m)->classify(lastLevel, fout); }
double orc_wvm_probability(const orc_wvm* m, double fout) { return ((const Wvm*)m)->probability(fout); }

orc_svm* orc_svm_create(int kernel, double p0, double p1, double p2, int nsv, int dim, int dtype, const void* sv,
                        const float* coeff, float bias, float threshold, double la, double lb) {
    Svm* m = new Svm();
    m->kernel = kernel; m->p0 = p0; m->p1 = p1; m->p2 = p2; m->nsv = nsv; m->dim = dim; m->dtype = dtype;
    if (dtype == 0) m->svU8.assign((const uchar*)sv, (const uchar*)sv + (size_t)nsv * dim);
    else m->svF32.assign((const float*)sv, (const float*)sv + (size_t)nsv * dim);
    m->coeff.assign(coeff, coeff + nsv);
    m->bias = bias; m->threshold = threshold; m->logisticA = la; m->logisticB = lb;
    return (orc_svm*)m;
}
void orc_svm_destroy(orc_svm* m) { delete (Svm*)m; }
double orc_svm_distance(const orc_svm* m, const void* x) { return ((const Svm*)m)->distance(x); }
int orc_svm_classify(const orc_svm* m, double d) { return ((const Svm*)m)->classify(d); }
double orc_svm_probability(const orc_svm* m, double d) { return ((const Svm*)m)->probability(d); }
void orc_svm_distance_batch(const orc_svm* m_, const void* x, int64_t n, double* out) {
    const Svm* m = (const Svm*)m_;
    size_t es = m->dtype == 0 ? 1 : 4;
    for (int64_t i = 0; i < n; ++i) out[i] = m->distance((const char*)x + (size_t)i * m->dim * es);
}
orc_rvm* orc_rvm_create(int kernel, double p0, double p1, double p2, int numFilters, int numUse, int dim, const float* sv,
                        const float* coeffPacked, const float* thresholds, float bias, double la, double lb) {
    Rvm* m = new Rvm();
    m->store.kernel = kernel; m->store.p0 = p0; m->store.p1 = p1; m->store.p2 = p2;
    m->store.nsv = numFilters; m->store.dim = dim; m->store.dtype = 1;
    m->store.svF32.assign(sv, sv + (size_t)numFilters * dim);
    m->coeff.assign(coeffPacked, coeffPacked + (size_t)numFilters * (numFilters + 1) / 2);
    m->thresholds.assign(thresholds, thresholds + numFilters);
    m->numFilters = numFilters;
    m->numUse = (numUse == 0 || numUse > numFilters) ? numFilters : numUse;   // setNumFiltersToUse, RvmClassifier.cpp:119-126
    m->bias = bias; m->logisticA = la; m->logisticB = lb;
    return (orc_rvm*)m;
}
void orc_rvm_destroy(orc_rvm* m) { delete (Rvm*)m; }
void orc_rvm_eval(const orc_rvm* m, const float* x, int32_t* lastLevel, double* distance) {
    int l; double d;
    ((const Rvm*)m)->eval(x, l, d);
    *lastLevel = l; *distance = d;
}
void orc_rvm_eval_batch(const orc_rvm* m_, const float* x, int64_t n, int32_t* lastLevel, double* distance) {
    const Rvm* m = (const Rvm*)m_;
    for (int64_t i = 0; i < n; ++i) {
        int l; double d;
        m->eval(x + (size_t)i * m->store.dim, l, d);
        lastLevel[i] = l; distance[i] = d;
    }
}
int orc_rvm_classify(const orc_rvm* m, int lastLevel, double d) { return ((const Rvm*)m)->classify(lastLevel, d); }
double orc_rvm_probability(const orc_rvm* m, double d) { return ((const Rvm*)m)->probability(d); }
}
