// fdcompat/cv.hpp -- the tiny subset of OpenCV's core types that the preserved reference interfaces
// are written in (cv::Mat, cv::Rect, cv::Size, cv::Point, cv::Point2f).  OpenCV is not
// available in this environment; define FD_USE_OPENCV to build the host layer against the real
// headers instead (the class definitions below are then skipped).
#pragma once
#ifdef FD_USE_OPENCV
#include "opencv2/core/core.hpp"
#else
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#define CV_8U 0
#define CV_32S 4
#define CV_32F 5
#define CV_SCHARR -1   /* cv::Sobel kernel size constant used by GradientFilter (GradientFilter.cpp:17) */
#define CV_64F 6
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) (((depth) & 7) + (((cn)-1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC2 CV_MAKETYPE(CV_8U, 2)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_8UC4 CV_MAKETYPE(CV_8U, 4)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)

namespace cv {
enum { NORM_INF = 1, NORM_L1 = 2, NORM_L2 = 4 };

typedef unsigned char uchar;

template <class T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<int> Point;
typedef Point_<int> Point2i;
typedef Point_<float> Point2f;

template <class T> struct Size_ {
    T width, height;
    Size_() : width(0), height(0) {}
    Size_(T w, T h) : width(w), height(h) {}
    T area() const { return width * height; }
};
typedef Size_<int> Size;

template <class T> struct Rect_ {
    T x, y, width, height;
    Rect_() : x(0), y(0), width(0), height(0) {}
    Rect_(T x_, T y_, T w, T h) : x(x_), y(y_), width(w), height(h) {}
    T area() const { return width * height; }
};
typedef Rect_<int> Rect;

static inline int cvRound(double v);

// Reference-counted dense 2-D matrix header with ROI support (the container part of cv::Mat).
class Mat {
public:
    int flags = 0, rows = 0, cols = 0;
    size_t step = 0;       // bytes per row
    uchar* data = nullptr;

    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void* external, size_t step_ = 0)
        : flags(type), rows(r), cols(c), step(step_ ? step_ : (size_t)c * elemSizeOf(type)), data((uchar*)external) {}
    Mat(const Mat& m, const Rect& roi) : flags(m.flags), rows(roi.height), cols(roi.width), step(m.step), buf(m.buf) {
        if (roi.x < 0 || roi.y < 0 || roi.width < 0 || roi.height < 0 || roi.x + roi.width > m.cols || roi.y + roi.height > m.rows)
            throw std::runtime_error("cv::Mat: ROI outside of the matrix");
        data = m.data + (size_t)roi.y * m.step + (size_t)roi.x * m.elemSize();
    }
    void create(int r, int c, int type) {
        if (data && rows == r && cols == c && flags == type && isContinuous() && buf) return;
        flags = type; rows = r; cols = c;
        step = (size_t)c * elemSize();
        buf = std::shared_ptr<std::vector<uchar>>(new std::vector<uchar>((size_t)r * step + 16));
        data = buf->data();
    }
    static Mat zeros(int r, int c, int type) { Mat m(r, c, type); std::memset(m.data, 0, (size_t)r * m.step); return m; }
    int type() const { return flags & 0xfff; }
    int depth() const { return flags & 7; }
    int channels() const { return ((flags >> CV_CN_SHIFT) & 511) + 1; }
    static size_t elemSizeOf(int type) {
        static const int sz[8] = {1, 1, 2, 2, 4, 4, 8, 0};
        return (size_t)sz[type & 7] * (((type >> CV_CN_SHIFT) & 511) + 1);
    }
    size_t elemSize() const { return elemSizeOf(flags); }
    size_t total() const { return (size_t)rows * cols; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return rows <= 1 || step == (size_t)cols * elemSize(); }
    Size size() const { return Size(cols, rows); }
    template <class T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
    template <class T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
    template <class T> T& at(int r, int c) { return ptr<T>(r)[c]; }
    template <class T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }
    template <class T> T& at(int i) { return rows == 1 ? ptr<T>(0)[i] : ptr<T>(i)[0]; }
    template <class T> const T& at(int i) const { return rows == 1 ? ptr<T>(0)[i] : ptr<T>(i)[0]; }
    Mat operator()(const Rect& roi) const { return Mat(*this, roi); }
    Mat clone() const {
        Mat m;
        if (empty()) return m;
        m.create(rows, cols, flags);
        for (int r = 0; r < rows; ++r) std::memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, (size_t)cols * elemSize());
        return m;
    }
    void copyTo(Mat& dst) const { dst = clone(); }

private:
    std::shared_ptr<std::vector<uchar>> buf;  // owner (null for external data)
};

static inline int cvRound(double v) { return (int)__builtin_lrint(v); }

}  // namespace cv
#endif
