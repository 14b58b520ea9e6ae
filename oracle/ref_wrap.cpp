// oracle/ref_wrap.cpp -- TEST INFRASTRUCTURE ONLY.
// Thin C wrappers (our code) around the reference translation units that compile standalone.
// The reference SOURCES are compiled where they lie under /root/reference by oracle/Makefile;
// outputs go to oracle/_ref/ (git-ignored).  Used only to pin the oracle restatements.
#include <cstdint>
#include <vector>
#include <cstring>
#include "classification/IImg.hpp"   // libClassification/include (reference header, via -I)
#include "svm.h"                     // libSvm/include (reference header, via -I)

extern "C" {

// IImg::calIImgPatch (libClassification/src/classification/IImg.cpp:26-65)
void ref_iimg(const uint8_t* patch, int w, int h, int sqr, float* out) {
    classification::IImg ii(w, h, 8);
    ii.calIImgPatch(patch, sqr != 0);
    std::memcpy(out, ii.data, sizeof(float) * (size_t)w * h);
}

// libsvm 3.17 (+HIK) decision value for a 2-class C_SVC model assembled from flat arrays
// (libSvm/src/svm.cpp svm_predict_values).  kernel: 0 linear, 1 poly, 2 rbf, 5 hik.
// Returns sum_i coef_i K(x, sv_i) - rho.
double ref_svm_decision(int kernel, int degree, double gamma, double coef0, int nsv, int dim,
                        const double* sv, const double* coef, double rho, const double* x) {
    svm_model m;
    std::memset(&m, 0, sizeof(m));
    m.param.svm_type = C_SVC;
    m.param.kernel_type = kernel;
    m.param.degree = degree;
    m.param.gamma = gamma;
    m.param.coef0 = coef0;
    m.nr_class = 2;
    m.l = nsv;
    std::vector<std::vector<svm_node>> nodes(nsv, std::vector<svm_node>(dim + 1));
    std::vector<svm_node*> svp(nsv);
    for (int i = 0; i < nsv; ++i) {
        for (int k = 0; k < dim; ++k) { nodes[i][k].index = k + 1; nodes[i][k].value = sv[(size_t)i * dim + k]; }
        nodes[i][dim].index = -1;
        svp[i] = nodes[i].data();
    }
    m.SV = svp.data();
    std::vector<double> c(coef, coef + nsv);
    double* cp = c.data();
    m.sv_coef = &cp;
    m.rho = &rho;
    int label[2] = {1, -1};
    int nSV[2] = {nsv, 0};
    m.label = label;
    m.nSV = nSV;
    std::vector<svm_node> xn(dim + 1);
    for (int k = 0; k < dim; ++k) { xn[k].index = k + 1; xn[k].value = x[k]; }
    xn[dim].index = -1;
    double dec = 0;
    svm_predict_values(&m, xn.data(), &dec);
    return dec;
}
}
