// compat/cv_min.hpp — the handful of OpenCV types the four SIVO class headers mention,
// for builds where OpenCV is not installed (this container).  With a real OpenCV, define
// SIVO_HAVE_OPENCV and the headers include <opencv2/core/core.hpp> instead: the class
// sources only use members that exist in both.
#pragma once
#include <cmath>
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
#define CV_32FC3 CV_MAKETYPE(CV_32F, 3)
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
    template <class S> Point_ &operator*=(S s) { x = (T)(x * s); y = (T)(y * s); return *this; }
};
struct Rect {
    int x = 0, y = 0, width = 0, height = 0;
    Rect() {}
    Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
};
typedef Point_<float> Point2f;
typedef Point_<int> Point2i;
typedef Point2i Point;
template <class T>
struct Point3_ {
    T x = 0, y = 0, z = 0;
    Point3_() {}
    Point3_(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
};
typedef Point3_<float> Point3f;

struct KeyPoint {   // 28 bytes, the layout SivoKeyPoint mirrors
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
    KeyPoint() {}
    KeyPoint(float x, float y, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
        : pt(x, y), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
};

struct Vec3b {
    uchar v[3];
    Vec3b(uchar a = 0, uchar b = 0, uchar c = 0) : v{a, b, c} {}
    uchar &operator[](int i) { return v[i]; }
    uchar operator[](int i) const { return v[i]; }
};

class Mat;
// `A.t()`, `-A`, `s * A`, `A / s`: a matrix with a pending transpose / scale factor, as cv::MatExpr keeps them, because
// OpenCV's arithmetic depends on it (see gemm below).  Converts to Mat where a Mat is expected.
struct MatScaled;
// `A * B (+ C)`: one cv::gemm call, evaluated when it is converted to Mat.
struct MatProduct;

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
    // Mat::zeros is an initializer expression in OpenCV: ASSIGNED to a matrix that already has that size and type it
    // fills the existing storage (a view stays a view — ORBextractor.cc:1012 relies on it), otherwise it allocates.
    struct Zeros { int rows, cols, type; };
    static Zeros zeros(int r, int c, int type) { return Zeros{r, c, type}; }
    Mat(const Zeros &z) { *this = z; }
    Mat &operator=(const Zeros &z) {
        create(z.rows, z.cols, z.type);
        for (int r = 0; r < rows; ++r) std::memset(ptr(r), 0, (size_t)cols * elemSize());
        return *this;
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
    // element i of a row or column vector
    template <class T> T &at(int i) { return rows == 1 ? ptr<T>(0)[i] : ptr<T>(i)[0]; }
    template <class T> const T &at(int i) const { return rows == 1 ? ptr<T>(0)[i] : ptr<T>(i)[0]; }
    // views (share the buffer)
    Mat rowRange(int r0, int r1) const { Mat m = row(r0); m.rows = r1 - r0; return m; }
    Mat colRange(int c0, int c1) const {
        Mat m; m.rows = rows; m.cols = c1 - c0; m.type_ = type_; m.step = step; m.data = data + (size_t)c0 * elemSize(); m.buf_ = buf_;
        return m;
    }
    Mat col(int c) const { return colRange(c, c + 1); }
    Mat operator()(const Rect &r) const { return rowRange(r.y, r.y + r.height).colRange(r.x, r.x + r.width); }
    Mat(const Mat &m, const Rect &r) { *this = m(r); }
    size_t step1() const { static const int sz[8] = {1, 1, 2, 2, 4, 4, 8, 0}; return step / (size_t)sz[depth()]; }
    // cv::InputArray / cv::OutputArray are plain Mat references here
    Mat getMat() const { return *this; }
    // append rows (a view becomes an owning matrix, as in OpenCV)
    void push_back(const Mat &m) {
        if (empty()) { *this = m.clone(); return; }
        Mat grown(rows + m.rows, cols, type_);
        for (int r = 0; r < rows; ++r) std::memcpy(grown.ptr(r), ptr(r), (size_t)cols * elemSize());
        for (int r = 0; r < m.rows; ++r) std::memcpy(grown.ptr(rows + r), m.ptr(r), (size_t)cols * elemSize());
        *this = grown;
    }
    // convertTo without scaling: CV_8U / CV_32F -> CV_32F, any channel count (dst may be *this)
    void convertTo(Mat &dst, int type) const {
        Mat out(rows, cols, type);
        const int n = cols * channels();
        for (int r = 0; r < rows; ++r) {
            float *o = out.ptr<float>(r);
            for (int c = 0; c < n; ++c) o[c] = depth() == CV_8U ? (float)ptr<uchar>(r)[c] : ptr<float>(r)[c];
        }
        dst = out;
    }
    static Mat eye(int r, int c, int type) {          // CV_32F
        Mat m(r, c, type);
        for (int i = 0; i < r; ++i)
            for (int j = 0; j < c; ++j) m.at<float>(i, j) = i == j ? 1.0f : 0.0f;
        return m;
    }
    static Mat ones(int r, int c, int type) {
        Mat m(r, c, type);
        for (int i = 0; i < r; ++i)
            for (int j = 0; j < c; ++j) m.at<float>(i, j) = 1.0f;
        return m;
    }
    Mat reshape(int) const { return *this; }      // only reached behind cv::undistortPoints, which is not provided
    void copyTo(Mat &dst) const { dst = clone(); }
    // CV_32F algebra (defined below the class)
    inline MatScaled t() const;
    inline double dot(const Mat &o) const;
    inline Mat(const MatScaled &e);
    inline Mat(const MatProduct &e);

 private:
    int type_ = 0;
    std::shared_ptr<uchar> buf_;
};

// ---------------------------------------------------------------------------------------------------------------------
// CV_32F matrix algebra with the rounding of OpenCV 3.x (modules/core/src/matmul.cpp, convert.cpp, stat.cpp), which is
// what decides last-bit results of expressions like `Rcw * x3Dw + tcw` or `-Rcw.t() * tcw`:
//   * A * B + C with no transposed operand and an inner dimension of 2..4 takes gemm's small-matrix path: every dot
//     product is accumulated left to right in float, then (float)(t * alpha + c * beta) with alpha / beta double;
//   * any other product (a transposed operand) goes through GEMMSingleMul<float, double>: products and sum in double,
//     (float)(alpha * sum + beta * c);
//   * s * A, A / s, -A are convertTo with the factor narrowed to float: a * (float)alpha;
//   * Mat::dot and cv::norm accumulate in double.
struct MatScaled {
    Mat m;
    bool transposed;
    double alpha;
};
struct MatProduct {
    MatScaled a, b;
    Mat c;          // empty: none
    double beta;
};

inline MatScaled Mat::t() const { return MatScaled{*this, true, 1.0}; }
inline double Mat::dot(const Mat &o) const {
    double s = 0.0;
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) s += (double)at<float>(r, c) * (double)o.at<float>(r, c);
    return s;
}
inline Mat::Mat(const MatScaled &e) {
    const int R = e.transposed ? e.m.cols : e.m.rows, C = e.transposed ? e.m.rows : e.m.cols;
    create(R, C, CV_32F);
    const float a = (float)e.alpha;
    for (int r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c) {
            const float v = e.transposed ? e.m.at<float>(c, r) : e.m.at<float>(r, c);
            at<float>(r, c) = e.alpha == 1.0 ? v : v * a;
        }
}
inline Mat::Mat(const MatProduct &e) {
    const bool tA = e.a.transposed, tB = e.b.transposed;
    const Mat &A = e.a.m, &B = e.b.m;
    const int M = tA ? A.cols : A.rows, K = tA ? A.rows : A.cols, N = tB ? B.rows : B.cols;
    const double alpha = e.a.alpha * e.b.alpha;
    create(M, N, CV_32F);
    const bool small = !tA && !tB && K >= 2 && K <= 4 && (K == N || K == M);
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            const double c = e.c.empty() ? 0.0 : (double)e.c.at<float>(i, j) * e.beta;
            if (small) {
                float t = A.at<float>(i, 0) * B.at<float>(0, j);
                for (int k = 1; k < K; ++k) t = t + A.at<float>(i, k) * B.at<float>(k, j);
                at<float>(i, j) = (float)((double)t * alpha + c);
            } else {
                double s = 0.0;
                for (int k = 0; k < K; ++k)
                    s += (double)(tA ? A.at<float>(k, i) : A.at<float>(i, k)) * (double)(tB ? B.at<float>(j, k) : B.at<float>(k, j));
                at<float>(i, j) = (float)(alpha * s + c);
            }
        }
}

inline MatScaled operator-(const Mat &a) { return MatScaled{a, false, -1.0}; }
inline MatScaled operator-(const MatScaled &a) { return MatScaled{a.m, a.transposed, -a.alpha}; }
inline MatScaled operator*(double s, const Mat &a) { return MatScaled{a, false, s}; }
inline MatScaled operator*(const Mat &a, double s) { return MatScaled{a, false, s}; }
inline MatScaled operator/(const Mat &a, double s) { return MatScaled{a, false, 1.0 / s}; }
inline MatScaled operator*(double s, const MatScaled &a) { return MatScaled{a.m, a.transposed, a.alpha * s}; }
inline MatProduct operator*(const Mat &a, const Mat &b) { return MatProduct{MatScaled{a, false, 1.0}, MatScaled{b, false, 1.0}, Mat(), 0.0}; }
inline MatProduct operator*(const MatScaled &a, const Mat &b) { return MatProduct{a, MatScaled{b, false, 1.0}, Mat(), 0.0}; }
inline MatProduct operator*(const Mat &a, const MatScaled &b) { return MatProduct{MatScaled{a, false, 1.0}, b, Mat(), 0.0}; }
inline MatProduct operator+(const MatProduct &p, const Mat &c) { MatProduct q = p; q.c = c; q.beta = 1.0; return q; }
inline MatProduct operator-(const MatProduct &p, const Mat &c) { MatProduct q = p; q.c = c; q.beta = -1.0; return q; }
inline Mat operator+(const Mat &a, const Mat &b) {
    Mat m(a.rows, a.cols, CV_32F);
    for (int r = 0; r < a.rows; ++r)
        for (int c = 0; c < a.cols; ++c) m.at<float>(r, c) = a.at<float>(r, c) + b.at<float>(r, c);
    return m;
}
inline Mat operator-(const Mat &a, const Mat &b) {
    Mat m(a.rows, a.cols, CV_32F);
    for (int r = 0; r < a.rows; ++r)
        for (int c = 0; c < a.cols; ++c) m.at<float>(r, c) = a.at<float>(r, c) - b.at<float>(r, c);
    return m;
}
// A - s * B: addWeighted(A, 1, B, -s) with the weights narrowed to float
inline Mat operator-(const Mat &a, const MatScaled &b) {
    Mat m(a.rows, a.cols, CV_32F);
    const float s = (float)b.alpha;
    for (int r = 0; r < a.rows; ++r)
        for (int c = 0; c < a.cols; ++c) m.at<float>(r, c) = a.at<float>(r, c) - s * (b.transposed ? b.m.at<float>(c, r) : b.m.at<float>(r, c));
    return m;
}
inline double norm(const Mat &a) { return std::sqrt(a.dot(a)); }
enum { NORM_L1 = 2, NORM_L2 = 4 };
inline double norm(const Mat &a, const Mat &b, int type) {     // CV_32F, sums in double
    double s = 0.0;
    for (int r = 0; r < a.rows; ++r)
        for (int c = 0; c < a.cols; ++c) {
            const double d = (double)a.at<float>(r, c) - (double)b.at<float>(r, c);
            s += type == NORM_L1 ? std::fabs(d) : d * d;
        }
    return type == NORM_L1 ? s : std::sqrt(s);
}
// cv::Mat_<float>(r, c) << a, b, c
template <class T>
class Mat_ : public Mat {
 public:
    Mat_(int r, int c) : Mat(r, c, sizeof(T) == 4 ? CV_32F : sizeof(T) == 8 ? CV_64F : CV_8U) {}
    struct Filler {
        Mat_ *m;
        int i;
        Filler &operator,(T v) { m->template ptr<T>(i / m->cols)[i % m->cols] = v; ++i; return *this; }
        operator Mat() const { return *m; }
    };
    Filler operator<<(T v) { Filler f{this, 0}; f, v; return f; }
};

// The reference passes cv::InputArray / cv::OutputArray; every call site hands a cv::Mat.
typedef const Mat &InputArray;
typedef Mat &OutputArray;

}  // namespace cv
