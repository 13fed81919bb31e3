// Stand-in for <Eigen/Core> + <Eigen/Dense>, exactly as far as the reference's src/sivo_helpers/sivo_helpers.cpp needs them,
// so that file can be compiled UNTOUCHED into oracle/_ref/libref_helpers.so (Eigen is not installed here).  Dense double
// matrices, evaluated eagerly in the order Eigen 3.3 evaluates the same expressions without vectorisation:
//   product          coefficient (i, j) = sum over k = 0, 1, ... in that order;  A * B * C = (A * B) * C
//   determinant      2 x 2, 3 x 3, 4 x 4: the closed forms of Eigen/src/LU/Determinant.h;
//                    larger: PartialPivLU (first largest pivot, l = a / pivot, rank-1 update), sign * prod(diagonal) with the
//                    product taken in the halving order of Eigen's unrolled reduction
//   inverse          dynamic size: PartialPivLU, then column-oriented forward / backward substitution of the identity
//   Affine3d         rotation() returns the linear part (Eigen extracts it by SVD; equal for a rotation up to rounding)
// This restates Eigen; what the resulting library pins are the reference's OWN formulas (Jacobian entries, covariance
// assembly, mutual information).  Test infrastructure only.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <type_traits>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

namespace Eigen {

const int Dynamic = -1;

template <class T, int R, int C>
class Matrix {
 public:
    Matrix() : rows_(R > 0 ? R : 0), cols_(C > 0 ? C : 0), v_((size_t)(rows_ * cols_)) {}
    Matrix(int r, int c) : rows_(r), cols_(c), v_((size_t)(r * c)) {}
    Matrix(T a, T b, T c) : rows_(R > 0 ? R : 3), cols_(C > 0 ? C : 1), v_((size_t)(rows_ * cols_)) { v_[0] = a; v_[1] = b; v_[2] = c; }   // Vector3d(x, y, z)
    template <int R2, int C2>
    Matrix(const Matrix<T, R2, C2> &o) : rows_(o.rows()), cols_(o.cols()), v_((size_t)(o.rows() * o.cols())) {
        if ((R > 0 && R != rows_) || (C > 0 && C != cols_)) std::abort();
        for (int i = 0; i < rows_; ++i)
            for (int j = 0; j < cols_; ++j) (*this)(i, j) = o(i, j);
    }
    int rows() const { return rows_; }
    int cols() const { return cols_; }
    T &operator()(int i, int j) { return v_[(size_t)(i * cols_ + j)]; }
    const T &operator()(int i, int j) const { return v_[(size_t)(i * cols_ + j)]; }
    T &operator()(int i) { return v_[(size_t)i]; }
    const T &operator()(int i) const { return v_[(size_t)i]; }

    Matrix &operator*=(T s) { for (T &x : v_) x *= s; return *this; }
    static Matrix Zero() { return Matrix(); }
    static Matrix Identity() {
        Matrix m;
        for (int i = 0; i < m.rows_ && i < m.cols_; ++i) m(i, i) = T(1);
        return m;
    }

    // m << a, b, c ...: coefficients in row-major order
    struct Comma {
        Matrix *m;
        int n;
        Comma &operator,(T x) { m->v_[(size_t)n++] = x; return *this; }
    };
    Comma operator<<(T x) { Comma c{this, 0}; c, x; return c; }

    Matrix<T, C, R> transpose() const {
        Matrix<T, C, R> t(cols_, rows_);
        for (int i = 0; i < rows_; ++i)
            for (int j = 0; j < cols_; ++j) t(j, i) = (*this)(i, j);
        return t;
    }

    template <int BR, int BC>
    struct BlockRef {
        Matrix *m;
        int i0, j0;
        template <int R2, int C2>
        BlockRef &operator=(const Matrix<T, R2, C2> &o) {
            for (int i = 0; i < BR; ++i)
                for (int j = 0; j < BC; ++j) (*m)(i0 + i, j0 + j) = o(i, j);
            return *this;
        }
        operator Matrix<T, BR, BC>() const {
            Matrix<T, BR, BC> out;
            for (int i = 0; i < BR; ++i)
                for (int j = 0; j < BC; ++j) out(i, j) = (*m)(i0 + i, j0 + j);
            return out;
        }
    };
    template <int BR, int BC> BlockRef<BR, BC> block(int i, int j) { return BlockRef<BR, BC>{this, i, j}; }
    template <int BR, int BC> Matrix<T, BR, BC> block(int i, int j) const {
        Matrix<T, BR, BC> out;
        for (int a = 0; a < BR; ++a)
            for (int b = 0; b < BC; ++b) out(a, b) = (*this)(i + a, j + b);
        return out;
    }

    T determinant() const {
        const Matrix &m = *this;
        const int n = rows_;
        if (n == 1) return m(0, 0);
        if (n == 2) return m(0, 0) * m(1, 1) - m(1, 0) * m(0, 1);
        if (n == 3) {
            auto h = [&m](int a, int b, int c) { return m(0, a) * (m(1, b) * m(2, c) - m(1, c) * m(2, b)); };
            return h(0, 1, 2) - h(1, 0, 2) + h(2, 0, 1);
        }
        if (n == 4) {
            auto d2 = [&m](int i0, int i1, int j0, int j1) { return m(i0, j0) * m(i1, j1) - m(i1, j0) * m(i0, j1); };
            auto d3 = [&m, &d2](int j, int k, int p, int q) { return d2(j, k, p, q) * (m(p, 2) * m(q, 3) - m(q, 2) * m(p, 3)); };   // not used by sivo_helpers
            (void)d3;
            auto helper = [&m](int j, int k, int p, int q) { return (m(j, 0) * m(k, 1) - m(k, 0) * m(j, 1)) * (m(p, 2) * m(q, 3) - m(q, 2) * m(p, 3)); };
            return helper(0, 1, 2, 3) - helper(0, 2, 1, 3) + helper(0, 3, 1, 2) + helper(1, 2, 0, 3) - helper(1, 3, 0, 2) + helper(2, 3, 0, 1);
        }
        std::vector<T> a(v_);
        std::vector<int> perm;
        const int sign = lu_inplace(a, n, perm);
        std::vector<T> d((size_t)n);
        for (int i = 0; i < n; ++i) d[(size_t)i] = a[(size_t)(i * n + i)];
        return T(sign) * halving_product(d, 0, n);
    }

    Matrix inverse() const {
        const int n = rows_;
        std::vector<T> a(v_);
        std::vector<int> perm;
        lu_inplace(a, n, perm);
        Matrix inv(n, n);
        for (int c = 0; c < n; ++c) {
            std::vector<T> x((size_t)n, T(0));
            for (int i = 0; i < n; ++i) x[(size_t)i] = perm[(size_t)i] == c ? T(1) : T(0);       // P * e_c
            for (int k = 0; k < n; ++k)                                                           // L y = P e_c, unit lower, by columns
                for (int i = k + 1; i < n; ++i) x[(size_t)i] -= x[(size_t)k] * a[(size_t)(i * n + k)];
            for (int k = n - 1; k >= 0; --k) {                                                    // U x = y, by columns
                x[(size_t)k] /= a[(size_t)(k * n + k)];
                for (int i = 0; i < k; ++i) x[(size_t)i] -= x[(size_t)k] * a[(size_t)(i * n + k)];
            }
            for (int i = 0; i < n; ++i) inv(i, c) = x[(size_t)i];
        }
        return inv;
    }

 private:
    // PartialPivLU, unblocked (sizes <= 16): returns the permutation sign; perm[i] = source row of row i
    static int lu_inplace(std::vector<T> &a, int n, std::vector<int> &perm) {
        perm.resize((size_t)n);
        for (int i = 0; i < n; ++i) perm[(size_t)i] = i;
        int sign = 1;
        for (int k = 0; k < n; ++k) {
            int piv = k;
            T best = std::fabs(a[(size_t)(k * n + k)]);
            for (int i = k + 1; i < n; ++i)
                if (std::fabs(a[(size_t)(i * n + k)]) > best) { best = std::fabs(a[(size_t)(i * n + k)]); piv = i; }
            if (piv != k) {
                for (int j = 0; j < n; ++j) std::swap(a[(size_t)(k * n + j)], a[(size_t)(piv * n + j)]);
                std::swap(perm[(size_t)k], perm[(size_t)piv]);
                sign = -sign;
            }
            if (best != T(0)) {
                for (int i = k + 1; i < n; ++i) a[(size_t)(i * n + k)] /= a[(size_t)(k * n + k)];
                for (int i = k + 1; i < n; ++i)
                    for (int j = k + 1; j < n; ++j) a[(size_t)(i * n + j)] -= a[(size_t)(i * n + k)] * a[(size_t)(k * n + j)];
            }
        }
        return sign;
    }
    static T halving_product(const std::vector<T> &d, int start, int len) {
        if (len == 1) return d[(size_t)start];
        const int half = len / 2;
        return halving_product(d, start, half) * halving_product(d, start + half, len - half);
    }

    int rows_, cols_;
    std::vector<T> v_;
};

template <class T, int R1, int C1, int R2, int C2>
Matrix<T, R1, C2> operator*(const Matrix<T, R1, C1> &a, const Matrix<T, R2, C2> &b) {
    if (a.cols() != b.rows()) std::abort();
    Matrix<T, R1, C2> out(a.rows(), b.cols());
    for (int i = 0; i < a.rows(); ++i)
        for (int j = 0; j < b.cols(); ++j) {
            T s = a(i, 0) * b(0, j);
            for (int k = 1; k < a.cols(); ++k) s += a(i, k) * b(k, j);
            out(i, j) = s;
        }
    return out;
}
template <class T, int R, int C> Matrix<T, R, C> operator*(const Matrix<T, R, C> &a, T s) {
    Matrix<T, R, C> out(a.rows(), a.cols());
    for (int i = 0; i < a.rows(); ++i)
        for (int j = 0; j < a.cols(); ++j) out(i, j) = a(i, j) * s;
    return out;
}
template <class T, int R, int C> Matrix<T, R, C> operator*(T s, const Matrix<T, R, C> &a) {
    Matrix<T, R, C> out(a.rows(), a.cols());
    for (int i = 0; i < a.rows(); ++i)
        for (int j = 0; j < a.cols(); ++j) out(i, j) = s * a(i, j);
    return out;
}
// a scalar of another arithmetic type (Eigen converts it to the matrix's scalar type first): `Matrix2d::Identity() * invSigma2` with a float
template <class T, int R, int C, class S, class = typename std::enable_if<std::is_arithmetic<S>::value && !std::is_same<S, T>::value>::type>
Matrix<T, R, C> operator*(const Matrix<T, R, C> &a, S s) { return a * (T)s; }
template <class T, int R, int C, class S, class = typename std::enable_if<std::is_arithmetic<S>::value && !std::is_same<S, T>::value>::type>
Matrix<T, R, C> operator*(S s, const Matrix<T, R, C> &a) { return (T)s * a; }
template <class T, int R1, int C1, int R2, int C2>
Matrix<T, R1, C1> operator+(const Matrix<T, R1, C1> &a, const Matrix<T, R2, C2> &b) {
    Matrix<T, R1, C1> out(a.rows(), a.cols());
    for (int i = 0; i < a.rows(); ++i)
        for (int j = 0; j < a.cols(); ++j) out(i, j) = a(i, j) + b(i, j);
    return out;
}
template <class T, int R1, int C1, int R2, int C2>
Matrix<T, R1, C1> operator-(const Matrix<T, R1, C1> &a, const Matrix<T, R2, C2> &b) {
    Matrix<T, R1, C1> out(a.rows(), a.cols());
    for (int i = 0; i < a.rows(); ++i)
        for (int j = 0; j < a.cols(); ++j) out(i, j) = a(i, j) - b(i, j);
    return out;
}

typedef Matrix<double, 2, 2> Matrix2d;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, Dynamic, Dynamic> MatrixXd;

class Affine3d {
 public:
    Matrix3d linear_ = Matrix3d::Identity();
    Vector3d translation_;
    Vector3d translation() const { return translation_; }
    Matrix3d rotation() const { return linear_; }
};

}  // namespace Eigen
