// compat/cv_min.hpp — the handful of OpenCV types the four SIVO class headers mention,
// for builds where OpenCV is not installed (this container).  With a real OpenCV, define
// SIVO_HAVE_OPENCV and the headers include <opencv2/core/core.hpp> instead: the class
// sources only use members that exist in both.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)

namespace cv {

typedef unsigned char uchar;

struct Size {
    int width = 0, height = 0;
    Size() {}
    Size(int w, int h) : width(w), height(h) {}
    bool operator==(const Size &o) const { return width == o.width && height == o.height; }
};

template <class T>
struct Point_ {
    T x = 0, y = 0;
    Point_() {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<float> Point2f;
typedef Point_<int> Point2i;
typedef Point2i Point;

struct KeyPoint {   // 28 bytes, the layout SivoKeyPoint mirrors
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};

struct Vec3b {
    uchar v[3];
    Vec3b(uchar a = 0, uchar b = 0, uchar c = 0) : v{a, b, c} {}
    uchar &operator[](int i) { return v[i]; }
    uchar operator[](int i) const { return v[i]; }
};

class Mat {
 public:
    int rows = 0, cols = 0;
    size_t step = 0;
    uchar *data = nullptr;

    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(Size s, int type) { create(s.height, s.width, type); }
    // header over user memory (no ownership), like cv::Mat(rows, cols, type, void*)
    Mat(int r, int c, int type, void *ptr, size_t step_ = 0) : rows(r), cols(c), type_(type) {
        step = step_ ? step_ : (size_t)c * elemSize();
        data = static_cast<uchar *>(ptr);
    }
    static Mat zeros(int r, int c, int type) {
        Mat m(r, c, type);
        if (m.data) std::memset(m.data, 0, m.step * (size_t)r);
        return m;
    }
    void create(int r, int c, int type) {
        if (r == rows && c == cols && type == type_ && data) return;
        rows = r; cols = c; type_ = type;
        step = (size_t)c * elemSize();
        buf_.reset(new uchar[step * (size_t)(r > 0 ? r : 1)], std::default_delete<uchar[]>());
        data = buf_.get();
    }
    void release() { rows = cols = 0; step = 0; data = nullptr; buf_.reset(); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    int channels() const { return (type_ >> 3) + 1; }
    int depth() const { return type_ & 7; }
    size_t elemSize() const {
        static const int sz[8] = {1, 1, 2, 2, 4, 4, 8, 0};
        return (size_t)sz[depth()] * channels();
    }
    Size size() const { return Size(cols, rows); }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }
    template <class T> T *ptr(int r = 0) { return reinterpret_cast<T *>(data + step * (size_t)r); }
    template <class T> const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(data + step * (size_t)r); }
    uchar *ptr(int r = 0) { return data + step * (size_t)r; }
    const uchar *ptr(int r = 0) const { return data + step * (size_t)r; }
    template <class T> T &at(int r, int c) { return ptr<T>(r)[c]; }
    template <class T> const T &at(int r, int c) const { return ptr<T>(r)[c]; }
    Mat row(int r) const {
        Mat m; m.rows = 1; m.cols = cols; m.type_ = type_; m.step = step; m.data = data + step * (size_t)r; m.buf_ = buf_;
        return m;
    }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; ++r) std::memcpy(m.ptr(r), ptr(r), (size_t)cols * elemSize());
        return m;
    }

 private:
    int type_ = 0;
    std::shared_ptr<uchar> buf_;
};

// The reference passes cv::InputArray / cv::OutputArray; every call site hands a cv::Mat.
typedef const Mat &InputArray;
typedef Mat &OutputArray;

}  // namespace cv
