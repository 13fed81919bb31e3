// C entry points over the reference's OWN src/sivo_helpers/sivo_helpers.cpp, compiled untouched into
// oracle/_ref/libref_helpers.so against the Eigen stand-in of ref_shims_eigen/ (oracle/Makefile `ref`).  The formulas —
// projection Jacobians, joint covariance assembly, mutual information, the two covariance updates — are the reference's;
// the dense linear algebra under them is the stand-in's.  Test infrastructure: tests/test_pin_helpers.py.
#include <cstring>

#include "include/sivo_helpers/sivo_helpers.hpp"

#define REF_API extern "C" __attribute__((visibility("default")))

template <int R, int C> static Eigen::Matrix<double, R, C> load(const double *p) {
    Eigen::Matrix<double, R, C> m;
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < C; ++j) m(i, j) = p[i * C + j];
    return m;
}
template <int R, int C> static void store(const Eigen::Matrix<double, R, C> &m, double *p) {
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < C; ++j) p[i * C + j] = m(i, j);
}

REF_API void ref_mono_jacobian_pose(double fx, double fy, double X, double Y, double Z, double *J /*2x6*/) {
    store<2, 6>(SIVO::SIVO::computeMonocularJacobianPose(fx, fy, X, Y, Z), J);
}
REF_API void ref_stereo_jacobian_pose(double fx, double fy, double bl, double X, double Y, double Z, double *J /*3x6*/) {
    store<3, 6>(SIVO::SIVO::computeStereoJacobianPose(fx, fy, bl, X, Y, Z), J);
}
REF_API void ref_mono_jacobian_point(double fx, double fy, double X, double Y, double Z, const double *Ccw, double *J /*2x3*/) {
    store<2, 3>(SIVO::SIVO::computeMonocularJacobianPoint(fx, fy, X, Y, Z, load<3, 3>(Ccw)), J);
}
REF_API void ref_stereo_jacobian_point(double fx, double fy, double bl, double X, double Y, double Z, const double *Ccw, double *J /*3x3*/) {
    store<3, 3>(SIVO::SIVO::computeStereoJacobianPoint(fx, fy, bl, X, Y, Z, load<3, 3>(Ccw)), J);
}
REF_API void ref_mono_covariance(const double *S, const double *J, const double *N, double *out /*8x8*/) {
    store<8, 8>(SIVO::SIVO::computeMonocularCovariance(load<6, 6>(S), load<2, 6>(J), load<2, 2>(N)), out);
}
REF_API void ref_stereo_covariance(const double *S, const double *J, const double *N, double *out /*9x9*/) {
    store<9, 9>(SIVO::SIVO::computeStereoCovariance(load<6, 6>(S), load<3, 6>(J), load<3, 3>(N)), out);
}
REF_API double ref_mono_mutual_information(const double *cov /*8x8*/) { return SIVO::SIVO::computeMonocularMutualInformation(load<8, 8>(cov)); }
REF_API double ref_stereo_mutual_information(const double *cov /*9x9*/) { return SIVO::SIVO::computeStereoMutualInformation(load<9, 9>(cov)); }
REF_API void ref_update_covariance_stereo(const double *S, const double *J, const double *N, double *out /*6x6*/) {
    store<6, 6>(SIVO::SIVO::updateStateCovarianceStereo(load<6, 6>(S), load<3, 6>(J), load<3, 3>(N)), out);
}
REF_API void ref_update_covariance_motion(const double *S, const double *R /*3x3*/, const double *t /*3*/, double *out /*6x6*/) {
    Eigen::Affine3d T;
    T.linear_ = load<3, 3>(R);
    for (int i = 0; i < 3; ++i) T.translation_(i) = t[i];
    store<6, 6>(SIVO::SIVO::updateStateCovarianceMotion(load<6, 6>(S), T), out);
}
