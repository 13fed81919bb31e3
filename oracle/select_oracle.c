/*
 * oracle/select_oracle.c — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of SIVO's information-theoretic feature-selection gate:
 *   computeStereoJacobianPose          src/sivo_helpers/sivo_helpers.cpp:64-88
 *   computeStereoCovariance            src/sivo_helpers/sivo_helpers.cpp:160-180
 *   computeStereoMutualInformation     src/sivo_helpers/sivo_helpers.cpp:201-219
 * applied per semantic keypoint as Tracking::CreateNewKeyFrame does
 * (src/orbslam/Tracking.cc:934-1023): entropy lookup at the truncated keypoint position, depth > 0,
 * MI - entropy > ThEntropyReduction.
 *
 * The reference takes determinants with Eigen (absent here): fixed 3x3 by cofactors, 6x6 and 9x9 by
 * partial-pivot LU with the diagonal multiplied in the order of Eigen's unrolled reduction — restated as such.
 * PARITY: the FORMULAS are pinned against the reference's own sivo_helpers.cpp, compiled untouched into
 * oracle/_ref/libref_helpers.so (Eigen replaced by the stand-in of oracle/ref_shims_eigen): orc_stereo_mutual_information
 * equals the reference's Jacobian -> covariance -> mutual-information chain bit for bit on 512 cases
 * (tests/test_pin_helpers.py; fixture tests/golden/helpers_reference.json).  Eigen's arithmetic itself and the call sites in
 * Tracking.cc / LocalMapping.cc (which keypoints, which thresholds) stay restated.  Further anchors in
 * tests/test_oracle_select.py: numpy slogdet and the Schur identity det S9 = det Sx * det R.
 */
#include <math.h>
#include <stdint.h>

typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } OrcKeyPoint;

/* prod() of a fixed-size vector in the order of Eigen's unrolled reduction: halves, recursively */
static double halving_product(const double *d, int start, int len) {
    if (len == 1) return d[start];
    const int half = len / 2;
    return halving_product(d, start, half) * halving_product(d, start + half, len - half);
}

/* determinant by LU with partial pivoting (Eigen::PartialPivLU::determinant: first largest pivot, l = a / pivot,
 * rank-1 update, sign * diagonal().prod()) */
static double det_lu(double *a, int n) {
    double diag[16];
    int sign = 1;
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double best = fabs(a[k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (fabs(a[i * n + k]) > best) { best = fabs(a[i * n + k]); piv = i; }
        if (best == 0.0) return 0.0;
        if (piv != k) {
            for (int j = 0; j < n; ++j) { const double t = a[k * n + j]; a[k * n + j] = a[piv * n + j]; a[piv * n + j] = t; }
            sign = -sign;
        }
        diag[k] = a[k * n + k];
        for (int i = k + 1; i < n; ++i) {
            const double f = a[i * n + k] / a[k * n + k];
            for (int j = k + 1; j < n; ++j) a[i * n + j] -= f * a[k * n + j];
        }
    }
    return (double)sign * halving_product(diag, 0, n);
}

static double det3(const double *m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

/* sivo_helpers.cpp:64-88; row-major 3x6, translation columns first */
static void stereo_jacobian_pose(double fx, double fy, double bl, double X, double Y, double Z, double *J) {
    for (int i = 0; i < 18; ++i) J[i] = 0.0;
    if (Z != 0) {
        J[0] = fx / Z; J[1] = 0.0; J[2] = -fx * X / (Z * Z);
        J[3] = -fx * X * Y / (Z * Z); J[4] = fx * (1.0 + (X * X) / (Z * Z)); J[5] = -fx * Y / Z;
        J[6] = 0.0; J[7] = fy / Z; J[8] = -fy * Y / (Z * Z);
        J[9] = -fy * (1 + (Y * Y) / (Z * Z)); J[10] = fy * X * Y / (Z * Z); J[11] = fy * X / Z;
        J[12] = fx / Z; J[13] = 0.0; J[14] = -fx * (X - bl) / (Z * Z);
        J[15] = -fx * (X - bl) * Y / (Z * Z); J[16] = fx * (1.0 + (X * (X - bl)) / (Z * Z)); J[17] = -fx * Y / Z;
    }
}

double orc_stereo_mutual_information(const double *Sx /*6x6*/, double fx, double fy, double bl, double X, double Y,
                                     double Z, double sigma2) {
    double J[18], S9[81], JS[18], Sz[9], Sx_copy[36];
    stereo_jacobian_pose(fx, fy, bl, X, Y, Z, J);
    /* computeStereoCovariance: blocks (0,0)=Sx, (6,6)=J Sx J' + R, (0,6)=Sx J', (6,0)=J Sx */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
            for (int k = 0; k < 6; ++k) s += J[i * 6 + k] * Sx[k * 6 + j];
            JS[i * 6 + j] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 6; ++k) s += JS[i * 6 + k] * J[j * 6 + k];
            Sz[i * 3 + j] = s + (i == j ? sigma2 : 0.0);
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) { S9[i * 9 + j] = Sx[i * 6 + j]; Sx_copy[i * 6 + j] = Sx[i * 6 + j]; }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 6; ++k) s += Sx[i * 6 + k] * J[j * 6 + k];
            S9[i * 9 + 6 + j] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 6; ++j) S9[(6 + i) * 9 + j] = JS[i * 6 + j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) S9[(6 + i) * 9 + 6 + j] = Sz[i * 3 + j];
    /* computeStereoMutualInformation */
    const double state_det = det_lu(Sx_copy, 6);
    const double meas_det = det3(Sz);
    const double cov_det = det_lu(S9, 9);
    return 0.5 * log2(state_det * meas_det / cov_det);
}

/* The gate of Tracking.cc:934-1023 over n keypoints.  accept[i] = 1 iff depth > 0 and
 * MI - entropy(row, col) > th.  mi / reduction are NaN-free only where depth > 0 (0 elsewhere). */
void orc_entropy_gate(int n, const OrcKeyPoint *kps, const float *depth, const double *xyz, const double *entropy,
                      int rows, int cols, const double *Sx, double fx, double fy, double bl,
                      const float *level_sigma2, double th, double *mi, double *reduction, uint8_t *accept) {
    for (int i = 0; i < n; ++i) {
        mi[i] = 0.0; reduction[i] = 0.0; accept[i] = 0;
        const int col = (int)kps[i].x, row = (int)kps[i].y;
        if (!(depth[i] > 0) || row < 0 || row >= rows || col < 0 || col >= cols) continue;
        const double e = entropy[(int64_t)row * cols + col];
        const double sigma2 = level_sigma2[kps[i].octave];
        const double m = orc_stereo_mutual_information(Sx, fx, fy, bl, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], sigma2);
        mi[i] = m;
        reduction[i] = m - e;
        accept[i] = (m - e) > th;
    }
}

/* LocalMapping::CheckSemantics(pKF, idx, wP, compute_information = true) (reference src/orbslam/LocalMapping.cc:474-538) over n
 * keypoints: detected_class[i] = the class at the truncated position, or VOID (255) unless depth > 0, class <= TERRAIN (8),
 * confidence >= th_conf, and the entropy reduction is not below th (`if (entropy_reduction < mThEntropyReduction)` rejects). */
void orc_check_semantics(int n, const OrcKeyPoint *kps, const float *depth, const double *xyz, const double *entropy,
                         const double *confidence, const uint8_t *classes, int rows, int cols, const double *Sx, double fx, double fy,
                         double bl, const float *level_sigma2, double th, double th_conf, double *mi, double *reduction,
                         uint8_t *detected_class) {
    for (int i = 0; i < n; ++i) {
        mi[i] = 0.0; reduction[i] = 0.0; detected_class[i] = 255;
        const int col = (int)kps[i].x, row = (int)kps[i].y;
        if (row < 0 || row >= rows || col < 0 || col >= cols) continue;
        const uint8_t cls = classes[(int64_t)row * cols + col];
        const int depth_criteria = depth[i] > 0, class_criteria = cls <= 8;
        const int confidence_criteria = confidence[(int64_t)row * cols + col] >= th_conf;
        if (!(depth_criteria && class_criteria && confidence_criteria)) continue;
        const double sigma2 = level_sigma2[kps[i].octave];
        const double m = orc_stereo_mutual_information(Sx, fx, fy, bl, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], sigma2);
        mi[i] = m;
        reduction[i] = m - entropy[(int64_t)row * cols + col];
        detected_class[i] = reduction[i] < th ? 255 : cls;
    }
}
