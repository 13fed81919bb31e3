// Eigen stand-in for compiling the reference's src/orbslam/Optimizer.cc and Converter.cc UNTOUCHED (oracle/Makefile `ref`): the dense
// double matrices of ref_shims_eigen/Eigen/eigen_mini.hpp plus the few further pieces those two files name — Quaterniond (matrix
// -> quaternion -> matrix with Eigen 3.3's formulas), aligned_allocator, Matrix4d.  Test infrastructure only.
#pragma once
#include <memory>

#include "../ref_shims_eigen/Eigen/eigen_mini.hpp"

namespace Eigen {

template <class T> using aligned_allocator = std::allocator<T>;
typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<double, 7, 7> Matrix7d;

// Eigen::Quaterniond as far as Converter.cc / g2o::SE3Quat / g2o::Sim3 use it (Eigen/src/Geometry/Quaternion.h)
class Quaterniond {
 public:
    Quaterniond() {}
    Quaterniond(double w, double x, double y, double z) { q_[0] = x; q_[1] = y; q_[2] = z; q_[3] = w; }
    explicit Quaterniond(const Matrix3d &m) {           // quaternionbase_assign_impl<Other, 3, 3>
        double t = m(0, 0) + m(1, 1) + m(2, 2);
        if (t > 0.0) {
            t = std::sqrt(t + 1.0);
            q_[3] = 0.5 * t;
            t = 0.5 / t;
            q_[0] = (m(2, 1) - m(1, 2)) * t; q_[1] = (m(0, 2) - m(2, 0)) * t; q_[2] = (m(1, 0) - m(0, 1)) * t;
        } else {
            int i = 0;
            if (m(1, 1) > m(0, 0)) i = 1;
            if (m(2, 2) > m(i, i)) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
            q_[i] = 0.5 * t;
            t = 0.5 / t;
            q_[3] = (m(k, j) - m(j, k)) * t; q_[j] = (m(j, i) + m(i, j)) * t; q_[k] = (m(k, i) + m(i, k)) * t;
        }
    }
    double x() const { return q_[0]; }
    double y() const { return q_[1]; }
    double z() const { return q_[2]; }
    double w() const { return q_[3]; }
    void normalize() {
        const double n = std::sqrt(q_[0] * q_[0] + q_[1] * q_[1] + q_[2] * q_[2] + q_[3] * q_[3]);
        for (double &v : q_) v /= n;
    }
    void negate() { for (double &v : q_) v = -v; }
    Matrix3d toRotationMatrix() const {                   // QuaternionBase::toRotationMatrix
        const double tx = 2 * q_[0], ty = 2 * q_[1], tz = 2 * q_[2], twx = tx * q_[3], twy = ty * q_[3], twz = tz * q_[3], txx = tx * q_[0],
                     txy = ty * q_[0], txz = tz * q_[0], tyy = ty * q_[1], tyz = tz * q_[1], tzz = tz * q_[2];
        Matrix3d r;
        r(0, 0) = 1 - (tyy + tzz); r(0, 1) = txy - twz; r(0, 2) = txz + twy;
        r(1, 0) = txy + twz; r(1, 1) = 1 - (txx + tzz); r(1, 2) = tyz - twx;
        r(2, 0) = txz - twy; r(2, 1) = tyz + twx; r(2, 2) = 1 - (txx + tyy);
        return r;
    }

 private:
    double q_[4] = {0, 0, 0, 1};
};

}  // namespace Eigen
