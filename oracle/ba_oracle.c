/*
 * oracle/ba_oracle.c — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the per-edge reprojection residual / Jacobian that g2o
 * evaluates for every edge Optimizer::LocalBundleAdjustment and
 * Optimizer::PoseOptimization build (src/orbslam/Optimizer.cc:318-409,
 * 651-755): EdgeSE3ProjectXYZ, EdgeStereoSE3ProjectXYZ and their OnlyPose
 * variants over VertexSE3Expmap / VertexSBAPointXYZ with RobustKernelHuber.
 *
 * The arithmetic lives in navganti/g2o (fork of RainerKuemmerle/g2o carrying
 * ORB-SLAM2's types_six_dof_expmap; un-vendored submodule, .gitmodules:4-6,
 * commit not recorded in the tree).  Restated here from the published
 * ORB-SLAM2 edge definitions:
 *   p = R*Xw + t = (x,y,z)
 *   proj = (fx*x/z + cx, fy*y/z + cy [, fx*x/z + cx - bf/z])
 *   err  = obs - proj
 *   J_point = -dproj/dp * R               (vertex 0, VertexSBAPointXYZ)
 *   J_pose  = -dproj/dp * [ -p^ | I ]     (vertex 1, VertexSE3Expmap, rotation first)
 *   chi2 = err' * (invSigma2*I) * err
 *   Huber(delta): rho = chi2 if chi2 <= delta^2 else 2*sqrt(chi2)*delta - delta^2,
 *                 w = 1 if chi2 <= delta^2 else delta/sqrt(chi2)
 *
 * PARITY UNPINNED: g2o is absent and the reference has no optimizer tests.
 * Anchors: the in-tree stereo Jacobians of src/sivo_helpers/sivo_helpers.cpp:
 * 64-88,113-136 (algebraic cross-check, tests/test_oracle_ba.py) and central
 * finite differences of the residual.
 */
#include <math.h>
#include <stdint.h>

typedef struct {
    int32_t pose;     /* index into poses (12 doubles each: R row-major, t) */
    int32_t point;    /* index into points (3 doubles each) */
    int32_t stereo;   /* 0: mono (u,v), 1: stereo (u,v,uR) */
    int32_t pad_;
    double obs[3];    /* u, v, uR (uR ignored for mono) */
    double inv_sigma2;
} OrcEdge;

/* Outputs per edge e:
 *   err[3e..]   residual (err[2]=0 for mono)
 *   Jx[9e..]    3x3 row-major, d err / d point   (row 2 zero for mono)
 *   Jp[18e..]   3x6 row-major, d err / d pose    (row 2 zero for mono)
 *   chi2[e], rho[e] (robustified cost), w[e] (Huber weight rho'), depth_ok[e] (z>0)
 */
void orc_ba_linearize(const double *poses, const double *points, const OrcEdge *edges,
                      int64_t nE, const double *intr /* fx,fy,cx,cy,bf */,
                      double delta_mono, double delta_stereo,
                      double *err, double *Jx, double *Jp, double *chi2, double *rho,
                      double *w, uint8_t *depth_ok) {
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3], bf = intr[4];
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < nE; ++e) {
        const OrcEdge *ed = &edges[e];
        const double *R = poses + 12 * (int64_t)ed->pose, *t = R + 9;
        const double *X = points + 3 * (int64_t)ed->point;
        const double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
        const double y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
        const double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
        const double invz = 1.0 / z, z_2 = z * z;
        double *er = err + 3 * e, *jx = Jx + 9 * e, *jp = Jp + 18 * e;

        er[0] = ed->obs[0] - (x * invz * fx + cx);
        er[1] = ed->obs[1] - (y * invz * fy + cy);
        er[2] = ed->stereo ? ed->obs[2] - (x * invz * fx + cx - bf * invz) : 0.0;

        for (int j = 0; j < 3; ++j) {
            jx[j] = -fx * R[j] / z + fx * x * R[6 + j] / z_2;
            jx[3 + j] = -fy * R[3 + j] / z + fy * y * R[6 + j] / z_2;
            jx[6 + j] = ed->stereo ? jx[j] - bf * R[6 + j] / z_2 : 0.0;
        }

        jp[0] = x * y / z_2 * fx;
        jp[1] = -(1 + (x * x / z_2)) * fx;
        jp[2] = y / z * fx;
        jp[3] = -1. / z * fx;
        jp[4] = 0;
        jp[5] = x / z_2 * fx;

        jp[6] = (1 + y * y / z_2) * fy;
        jp[7] = -x * y / z_2 * fy;
        jp[8] = -x / z * fy;
        jp[9] = 0;
        jp[10] = -1. / z * fy;
        jp[11] = y / z_2 * fy;

        if (ed->stereo) {
            jp[12] = jp[0] - bf * y / z_2;
            jp[13] = jp[1] + bf * x / z_2;
            jp[14] = jp[2];
            jp[15] = jp[3];
            jp[16] = 0;
            jp[17] = jp[5] - bf / z_2;
        } else {
            for (int j = 12; j < 18; ++j) jp[j] = 0.0;
        }

        const double c2 = (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]) * ed->inv_sigma2;
        const double delta = ed->stereo ? delta_stereo : delta_mono;
        const double dsqr = delta * delta;
        chi2[e] = c2;
        if (c2 <= dsqr) { rho[e] = c2; w[e] = 1.0; }
        else { const double s = sqrt(c2); rho[e] = 2 * s * delta - dsqr; w[e] = delta / s; }
        depth_ok[e] = z > 0.0;
    }
}
