// compat/eigen_min.hpp — row-major dynamic matrix with the members of
// Eigen::Matrix<T, Dynamic, Dynamic, RowMajor> the SIVO headers use (MatXu / MatXd,
// reference include/bayesian_segnet/bayesian_segnet.hpp:46-50).  With a real Eigen,
// define SIVO_HAVE_EIGEN and the aliases below become the Eigen types.
#pragma once
#include <cstddef>
#include <vector>

namespace sivo_compat {

template <class T>
class RowMatrix {
 public:
    RowMatrix() {}
    RowMatrix(std::ptrdiff_t r, std::ptrdiff_t c) { resize(r, c); }
    void resize(std::ptrdiff_t r, std::ptrdiff_t c) { rows_ = r; cols_ = c; v_.resize((size_t)(r * c)); }
    std::ptrdiff_t rows() const { return rows_; }
    std::ptrdiff_t cols() const { return cols_; }
    std::ptrdiff_t size() const { return rows_ * cols_; }
    T *data() { return v_.data(); }
    const T *data() const { return v_.data(); }
    T &operator()(std::ptrdiff_t r, std::ptrdiff_t c) { return v_[(size_t)(r * cols_ + c)]; }
    const T &operator()(std::ptrdiff_t r, std::ptrdiff_t c) const { return v_[(size_t)(r * cols_ + c)]; }
    T minCoeff() const { T m = v_[0]; for (const T &x : v_) m = x < m ? x : m; return m; }
    T maxCoeff() const { T m = v_[0]; for (const T &x : v_) m = x > m ? x : m; return m; }

 private:
    std::ptrdiff_t rows_ = 0, cols_ = 0;
    std::vector<T> v_;
};

}  // namespace sivo_compat
