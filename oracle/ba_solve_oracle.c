/*
 * oracle/ba_solve_oracle.c — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of what g2o does for Optimizer::PoseOptimization
 * (src/orbslam/Optimizer.cc:273-491) and Optimizer::LocalBundleAdjustment
 * (src/orbslam/Optimizer.cc:493-926) once the graph is built: Levenberg-Marquardt
 * (g2o::OptimizationAlgorithmLevenberg) over BlockSolver_6_3 with the landmark
 * block marginalised (Schur complement), the chi2 re-classification schedule of the
 * two reference functions, and BlockSolver::computeMarginals for the pose covariance
 * (Optimizer.cc:482-487, 900-907).
 *
 * g2o is an un-vendored submodule (navganti/g2o, .gitmodules:4-6, commit not recorded);
 * its published algorithm is restated:
 *   - robustified normal equations: H += w J' Omega J, b -= w J' Omega e  with
 *     w = rho'(chi2) (the second-order term of robustInformation is disabled in g2o);
 *   - lambda_0 = 1e-5 * max |diag H| at iteration 0 of every optimize() call;
 *   - trial: (H + lambda I) dx = b ; x <- x (+) dx ; rho = (chi - chi_new) /
 *     (sum dx_j (lambda dx_j + b_j) + 1e-3); accepted when rho > 0:
 *     lambda *= max(1/3, min(2/3, 1 - (2 rho - 1)^3)), ni = 2; otherwise the state is
 *     restored, lambda *= ni, ni *= 2; at most 10 trials; optimize() stops when all 10
 *     trials failed or rho == 0;
 *   - VertexSE3Expmap: T <- exp([omega, upsilon]) * T (SE3Quat::exp, small-angle branch
 *     at theta < 1e-5); VertexSBAPointXYZ: X <- X + dx;
 *   - the per-edge error vector is whatever the LAST computeActiveErrors left, i.e. the
 *     error at a rejected trial state if the final trial was rejected (g2o does not
 *     recompute after pop()); chi2() read by the caller afterwards sees that;
 *   - computeMarginals factorises Hpp of the LAST buildSystem (lambda removed), not
 *     the Schur complement; with no pose-pose edges the block is inv(Hpp_ii).
 *
 * PARITY UNPINNED (no g2o, no optimizer tests in the reference).  Anchors used by the
 * tests: convergence to the generating poses/points on noise-free scenes, agreement of
 * the Schur solve with a dense solve of the full normal equations (numpy), finite
 * differences for the edge Jacobians (ba_oracle.c).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t pose, point, stereo, pad_;
    double obs[3];
    double inv_sigma2;
} OrcEdge;

static void edge_eval(const double *P, const double *X, const OrcEdge *ed, const double *intr, double *er, double *jp,
                      double *jx, int *depth_ok) {
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3], bf = intr[4];
    const double *R = P, *t = P + 9;
    const double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
    const double y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
    const double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
    const double invz = 1.0 / z, z_2 = z * z;
    er[0] = ed->obs[0] - (x * invz * fx + cx);
    er[1] = ed->obs[1] - (y * invz * fy + cy);
    er[2] = ed->stereo ? ed->obs[2] - (x * invz * fx + cx - bf * invz) : 0.0;
    if (depth_ok) *depth_ok = z > 0.0;
    if (jx)
        for (int j = 0; j < 3; ++j) {
            jx[j] = -fx * R[j] / z + fx * x * R[6 + j] / z_2;
            jx[3 + j] = -fy * R[3 + j] / z + fy * y * R[6 + j] / z_2;
            jx[6 + j] = ed->stereo ? jx[j] - bf * R[6 + j] / z_2 : 0.0;
        }
    if (jp) {
        jp[0] = x * y / z_2 * fx;  jp[1] = -(1 + (x * x / z_2)) * fx;  jp[2] = y / z * fx;
        jp[3] = -1. / z * fx;      jp[4] = 0;                          jp[5] = x / z_2 * fx;
        jp[6] = (1 + y * y / z_2) * fy;  jp[7] = -x * y / z_2 * fy;    jp[8] = -x / z * fy;
        jp[9] = 0;                 jp[10] = -1. / z * fy;              jp[11] = y / z_2 * fy;
        if (ed->stereo) {
            jp[12] = jp[0] - bf * y / z_2;  jp[13] = jp[1] + bf * x / z_2;  jp[14] = jp[2];
            jp[15] = jp[3];                 jp[16] = 0;                     jp[17] = jp[5] - bf / z_2;
        } else
            for (int j = 12; j < 18; ++j) jp[j] = 0.0;
    }
}

static void huber(double c2, double delta, double *rho, double *w) {
    const double dsqr = delta * delta;
    if (c2 <= dsqr) { *rho = c2; *w = 1.0; }
    else { const double s = sqrt(c2); *rho = 2 * s * delta - dsqr; *w = delta / s; }
}

/* T <- exp(update) * T, update = [omega(3), upsilon(3)]; T = R row-major (9) then t (3). */
static void se3_oplus(double *T, const double *u) {
    const double wx = u[0], wy = u[1], wz = u[2];
    const double theta = sqrt(wx * wx + wy * wy + wz * wz);
    const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double O2[9], R[9], V[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; ++i) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i]; V[i] = R[i]; }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / (theta * theta * theta);
        for (int i = 0; i < 9; ++i) {
            const double I = (i % 4 == 0 ? 1.0 : 0.0);
            R[i] = I + a * O[i] + b * O2[i];
            V[i] = I + b * O[i] + c * O2[i];
        }
    }
    double tn[3], Rn[9];
    for (int i = 0; i < 3; ++i) tn[i] = V[3 * i] * u[3] + V[3 * i + 1] * u[4] + V[3 * i + 2] * u[5];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) Rn[3 * i + j] = R[3 * i] * T[j] + R[3 * i + 1] * T[3 + j] + R[3 * i + 2] * T[6 + j];
        tn[i] += R[3 * i] * T[9] + R[3 * i + 1] * T[10] + R[3 * i + 2] * T[11];
    }
    memcpy(T, Rn, sizeof Rn);
    memcpy(T + 9, tn, sizeof tn);
}

/* in-place lower Cholesky of the n x n row-major A; returns 0 when A is not positive definite */
static int chol(double *A, int n) {
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0)) return 0;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    return 1;
}
static void chol_solve(const double *L, int n, double *x) {
    for (int i = 0; i < n; ++i) { double s = x[i]; for (int k = 0; k < i; ++k) s -= L[i * n + k] * x[k]; x[i] = s / L[i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double s = x[i]; for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k]; x[i] = s / L[i * n + i]; }
}
static int inv3(const double *A, double *Ai) {
    const double c0 = A[4] * A[8] - A[5] * A[7], c1 = A[5] * A[6] - A[3] * A[8], c2 = A[3] * A[7] - A[4] * A[6];
    const double det = A[0] * c0 + A[1] * c1 + A[2] * c2, id = 1.0 / det;
    Ai[0] = c0 * id; Ai[1] = (A[2] * A[7] - A[1] * A[8]) * id; Ai[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    Ai[3] = c1 * id; Ai[4] = (A[0] * A[8] - A[2] * A[6]) * id; Ai[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    Ai[6] = c2 * id; Ai[7] = (A[1] * A[6] - A[0] * A[7]) * id; Ai[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    return det != 0.0 && isfinite(id);
}

/* ---------------------------------------------------------------------------------------------
 * The optimisation problem: nP poses (fixed[i] != 0: not optimised), nX points (all free and
 * marginalised unless points_fixed), nE edges with a level (0 = active) and a robust flag.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    double *poses; const uint8_t *fixed; int nP;
    double *points; int nX, points_fixed;
    const OrcEdge *edges; int64_t nE;
    const double *intr; double delta_mono, delta_stereo;
    uint8_t *level, *robust;
    double *err;                 /* 3 per edge: what the last computeActiveErrors left */
    int *slot;                   /* pose -> free slot or -1 */
    int nF;
    double *Hpp_last;            /* 36 per free pose: Hpp blocks of the last buildSystem */
    const volatile int *stop;
    const volatile uint8_t *stop8;   /* the same as one byte (the reference's `bool *pbStopFlag`, read while the solve runs) */
} Problem;
#define STOPPED(p) (((p)->stop && *(p)->stop) || ((p)->stop8 && *(p)->stop8))

static double active_errors(Problem *p) {
    double chi = 0;
    for (int64_t e = 0; e < p->nE; ++e) {
        if (p->level[e]) continue;
        const OrcEdge *ed = &p->edges[e];
        double *er = p->err + 3 * e;
        edge_eval(p->poses + 12 * ed->pose, p->points + 3 * ed->point, ed, p->intr, er, 0, 0, 0);
        const double c2 = (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]) * ed->inv_sigma2;
        if (p->robust[e]) { double r, w; huber(c2, ed->stereo ? p->delta_stereo : p->delta_mono, &r, &w); chi += r; }
        else chi += c2;
    }
    return chi;
}

/* g2o::SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg; returns iterations run */
static int lm_optimize(Problem *p, int iterations, int *trials_out) {
    const int nF = p->nF, n6 = 6 * nF, nX = p->points_fixed ? 0 : p->nX;
    double *Hpp = calloc((size_t)(nF ? nF : 1) * 36, 8), *bp = calloc((size_t)(n6 ? n6 : 1), 8);
    double *Hll = calloc((size_t)(nX ? nX : 1) * 9, 8), *bl = calloc((size_t)(nX ? nX : 1) * 3, 8);
    double *W = calloc((size_t)(p->nE ? p->nE : 1) * 18, 8);   /* Hpl block of edge e: 6x3 */
    double *Hinv = calloc((size_t)(nX ? nX : 1) * 9, 8);
    double *S = calloc((size_t)(n6 ? n6 * n6 : 1), 8), *xs = calloc((size_t)(n6 ? n6 : 1), 8), *xl = calloc((size_t)(nX ? nX : 1) * 3, 8);
    double *bk_pose = malloc((size_t)p->nP * 12 * 8), *bk_pts = malloc((size_t)(p->nX ? p->nX : 1) * 3 * 8);
    double lambda = 0, ni = 2;
    int it = 0, trials = 0;
    for (; it < iterations; ++it) {
        if (STOPPED(p)) break;
        double current = active_errors(p);
        /* buildSystem */
        memset(Hpp, 0, (size_t)(nF ? nF : 1) * 36 * 8); memset(bp, 0, (size_t)(n6 ? n6 : 1) * 8);
        memset(Hll, 0, (size_t)(nX ? nX : 1) * 72); memset(bl, 0, (size_t)(nX ? nX : 1) * 24);
        for (int64_t e = 0; e < p->nE; ++e) {
            if (p->level[e]) continue;
            const OrcEdge *ed = &p->edges[e];
            double er[3], jp[18], jx[9];
            edge_eval(p->poses + 12 * ed->pose, p->points + 3 * ed->point, ed, p->intr, er, jp, jx, 0);
            const double c2 = (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]) * ed->inv_sigma2;
            double r = c2, w = 1.0;
            if (p->robust[e]) huber(c2, ed->stereo ? p->delta_stereo : p->delta_mono, &r, &w);
            const double wo = w * ed->inv_sigma2;
            const int s = p->slot[ed->pose];
            if (s >= 0) {
                for (int a = 0; a < 6; ++a) {
                    for (int b = 0; b < 6; ++b)
                        Hpp[36 * s + 6 * a + b] += wo * (jp[a] * jp[b] + jp[6 + a] * jp[6 + b] + jp[12 + a] * jp[12 + b]);
                    bp[6 * s + a] -= wo * (jp[a] * er[0] + jp[6 + a] * er[1] + jp[12 + a] * er[2]);
                }
            }
            if (nX) {
                const int q = ed->point;
                for (int a = 0; a < 3; ++a) {
                    for (int b = 0; b < 3; ++b)
                        Hll[9 * q + 3 * a + b] += wo * (jx[a] * jx[b] + jx[3 + a] * jx[3 + b] + jx[6 + a] * jx[6 + b]);
                    bl[3 * q + a] -= wo * (jx[a] * er[0] + jx[3 + a] * er[1] + jx[6 + a] * er[2]);
                }
                if (s >= 0)
                    for (int a = 0; a < 6; ++a)
                        for (int b = 0; b < 3; ++b)
                            W[18 * e + 3 * a + b] = wo * (jp[a] * jx[b] + jp[6 + a] * jx[3 + b] + jp[12 + a] * jx[6 + b]);
            }
        }
        if (p->Hpp_last) memcpy(p->Hpp_last, Hpp, (size_t)nF * 36 * 8);
        if (it == 0) {
            double md = 0;
            for (int s = 0; s < nF; ++s) for (int a = 0; a < 6; ++a) md = fmax(md, fabs(Hpp[36 * s + 7 * a]));
            for (int q = 0; q < nX; ++q) for (int a = 0; a < 3; ++a) md = fmax(md, fabs(Hll[9 * q + 4 * a]));
            lambda = 1e-5 * md; ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        do {
            memcpy(bk_pose, p->poses, (size_t)p->nP * 96);
            if (nX) memcpy(bk_pts, p->points, (size_t)p->nX * 24);
            /* Schur complement: S = Hpp + lambda I - sum_q W Hll^-1 W', bs = bp - sum W Hll^-1 bl */
            memset(S, 0, (size_t)(n6 ? n6 * n6 : 1) * 8);
            for (int s = 0; s < nF; ++s)
                for (int a = 0; a < 6; ++a) {
                    for (int b = 0; b < 6; ++b) S[(6 * s + a) * n6 + 6 * s + b] = Hpp[36 * s + 6 * a + b];
                    S[(6 * s + a) * n6 + 6 * s + a] += lambda;
                    xs[6 * s + a] = bp[6 * s + a];
                }
            int ok = 1;
            if (nX) {
                for (int q = 0; q < nX; ++q) {
                    double D[9];
                    memcpy(D, Hll + 9 * q, 72);
                    D[0] += lambda; D[4] += lambda; D[8] += lambda;
                    inv3(D, Hinv + 9 * q);
                }
                /* edges grouped by point: O(nE * deg) via per-point lists */
                int64_t *head = malloc((size_t)(nX + 1) * 8), *list = malloc((size_t)(p->nE ? p->nE : 1) * 8);
                memset(head, 0, (size_t)(nX + 1) * 8);
                for (int64_t e = 0; e < p->nE; ++e) if (!p->level[e] && p->slot[p->edges[e].pose] >= 0) head[p->edges[e].point + 1]++;
                for (int q = 0; q < nX; ++q) head[q + 1] += head[q];
                int64_t *fill = malloc((size_t)(nX + 1) * 8);
                memcpy(fill, head, (size_t)(nX + 1) * 8);
                for (int64_t e = 0; e < p->nE; ++e) if (!p->level[e] && p->slot[p->edges[e].pose] >= 0) list[fill[p->edges[e].point]++] = e;
                for (int q = 0; q < nX; ++q)
                    for (int64_t i1 = head[q]; i1 < head[q + 1]; ++i1) {
                        const int64_t e1 = list[i1];
                        const int s1 = p->slot[p->edges[e1].pose];
                        double Y[18];     /* W_e1 * Hll^-1 */
                        for (int a = 0; a < 6; ++a)
                            for (int b = 0; b < 3; ++b)
                                Y[3 * a + b] = W[18 * e1 + 3 * a] * Hinv[9 * q + b] + W[18 * e1 + 3 * a + 1] * Hinv[9 * q + 3 + b] + W[18 * e1 + 3 * a + 2] * Hinv[9 * q + 6 + b];
                        for (int a = 0; a < 6; ++a)
                            xs[6 * s1 + a] -= Y[3 * a] * bl[3 * q] + Y[3 * a + 1] * bl[3 * q + 1] + Y[3 * a + 2] * bl[3 * q + 2];
                        for (int64_t i2 = head[q]; i2 < head[q + 1]; ++i2) {
                            const int64_t e2 = list[i2];
                            const int s2 = p->slot[p->edges[e2].pose];
                            for (int a = 0; a < 6; ++a)
                                for (int b = 0; b < 6; ++b)
                                    S[(6 * s1 + a) * n6 + 6 * s2 + b] -= Y[3 * a] * W[18 * e2 + 3 * b] + Y[3 * a + 1] * W[18 * e2 + 3 * b + 1] + Y[3 * a + 2] * W[18 * e2 + 3 * b + 2];
                        }
                    }
                free(head); free(list); free(fill);
            }
            if (n6) { ok = chol(S, n6); if (ok) chol_solve(S, n6, xs); }
            if (nX && ok) {
                for (int q = 0; q < nX; ++q) { xl[3 * q] = bl[3 * q]; xl[3 * q + 1] = bl[3 * q + 1]; xl[3 * q + 2] = bl[3 * q + 2]; }
                for (int64_t e = 0; e < p->nE; ++e) {
                    if (p->level[e]) continue;
                    const int s = p->slot[p->edges[e].pose];
                    if (s < 0) continue;
                    const int q = p->edges[e].point;
                    for (int b = 0; b < 3; ++b)
                        for (int a = 0; a < 6; ++a) xl[3 * q + b] -= W[18 * e + 3 * a + b] * xs[6 * s + a];
                }
                for (int q = 0; q < nX; ++q) {
                    double v[3] = {xl[3 * q], xl[3 * q + 1], xl[3 * q + 2]};
                    for (int a = 0; a < 3; ++a) xl[3 * q + a] = Hinv[9 * q + 3 * a] * v[0] + Hinv[9 * q + 3 * a + 1] * v[1] + Hinv[9 * q + 3 * a + 2] * v[2];
                }
            }
            double scale = 0;
            if (ok) {
                for (int i = 0; i < p->nP; ++i) if (p->slot[i] >= 0) se3_oplus(p->poses + 12 * i, xs + 6 * p->slot[i]);
                for (int i = 0; i < 3 * nX; ++i) p->points[i] += xl[i];
                for (int i = 0; i < n6; ++i) scale += xs[i] * (lambda * xs[i] + bp[i]);
                for (int i = 0; i < 3 * nX; ++i) scale += xl[i] * (lambda * xl[i] + bl[i]);
            }
            double temp = active_errors(p);
            if (!ok) temp = DBL_MAX;
            rho = (current - temp) / (scale + 1e-3);
            if (rho > 0 && isfinite(temp)) {
                double alpha = 1. - pow(2 * rho - 1, 3);
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha);
                ni = 2; current = temp;
            } else {
                lambda *= ni; ni *= 2;
                memcpy(p->poses, bk_pose, (size_t)p->nP * 96);
                if (nX) memcpy(p->points, bk_pts, (size_t)p->nX * 24);
            }
            ++qmax; ++trials;
        } while (rho < 0 && qmax < 10 && !STOPPED(p));
        if (qmax == 10 || rho == 0) { ++it; break; }
    }
    free(Hpp); free(bp); free(Hll); free(bl); free(W); free(Hinv); free(S); free(xs); free(xl); free(bk_pose); free(bk_pts);
    if (trials_out) *trials_out += trials;
    return it;
}

static int inv6_spd(const double *H, double *out) {
    double L[36];
    memcpy(L, H, sizeof L);
    if (!chol(L, 6)) return 0;
    for (int c = 0; c < 6; ++c) {
        double e[6] = {0, 0, 0, 0, 0, 0};
        e[c] = 1;
        chol_solve(L, 6, e);
        for (int r = 0; r < 6; ++r) out[6 * r + c] = e[r];
    }
    return 1;
}

/* Optimizer::PoseOptimization (Optimizer.cc:273-491) on arrays.  edges[e].pose is ignored (one vertex);
 * edges[e].point indexes `points` (the map points' world positions, held fixed).  outlier[e] is written
 * (Frame::mvbOutlier); mono edges are never re-classified (the reference only loops over the stereo edges,
 * Optimizer.cc:432-467).  Returns nInitialCorrespondences - nBad, or 0 when there are fewer than 3 edges. */
int orc_pose_optimize(const double *pose0, const double *points, const OrcEdge *edges_in, int64_t nE, const double *intr,
                      uint8_t *outlier, double *pose_out, double *cov, int *cov_ok, double *chi2_out, int *iters, int *trials) {
    memcpy(pose_out, pose0, 96);
    if (cov_ok) *cov_ok = 0;
    if (iters) *iters = 0;
    if (trials) *trials = 0;
    if (nE < 3) return 0;
    OrcEdge *edges = malloc((size_t)nE * sizeof(OrcEdge));
    memcpy(edges, edges_in, (size_t)nE * sizeof(OrcEdge));
    int maxpt = 0;
    for (int64_t e = 0; e < nE; ++e) { edges[e].pose = 0; if (edges[e].point > maxpt) maxpt = edges[e].point; }
    uint8_t fixed = 0;
    int slot = 0;
    Problem p;
    memset(&p, 0, sizeof p);
    double pose[12], Hlast[36];
    p.poses = pose; p.fixed = &fixed; p.nP = 1; p.points = (double *)points; p.nX = maxpt + 1; p.points_fixed = 1;
    p.edges = edges; p.nE = nE; p.intr = intr;
    p.delta_mono = (double)sqrtf(5.991f); p.delta_stereo = (double)sqrtf(7.815f);      /* const float delta = std::sqrt(5.991f) (:307-308) */
    p.level = calloc((size_t)nE, 1); p.robust = malloc((size_t)nE); memset(p.robust, 1, (size_t)nE);
    p.err = calloc((size_t)nE * 3, 8); p.slot = &slot; p.nF = 1; p.Hpp_last = Hlast;
    memset(Hlast, 0, sizeof Hlast);
    memset(outlier, 0, (size_t)nE);
    int nBad = 0;
    for (int round = 0; round < 4; ++round) {
        memcpy(pose, pose0, 96);
        int tr = 0;
        const int n = lm_optimize(&p, 10, &tr);
        if (iters) *iters += n;
        if (trials) *trials += tr;
        nBad = 0;
        for (int64_t e = 0; e < nE; ++e) {
            if (!edges[e].stereo) continue;
            double *er = p.err + 3 * e;
            if (outlier[e]) edge_eval(pose, points + 3 * edges[e].point, &edges[e], intr, er, 0, 0, 0);
            const float chi2 = (float)((er[0] * er[0] + er[1] * er[1] + er[2] * er[2]) * edges[e].inv_sigma2);
            if (chi2 > 7.815f) { outlier[e] = 1; p.level[e] = 1; ++nBad; }
            else { outlier[e] = 0; p.level[e] = 0; }
            if (round == 2) p.robust[e] = 0;
        }
        if (nE < 10) break;
    }
    memcpy(pose_out, pose, 96);
    if (chi2_out)
        for (int64_t e = 0; e < nE; ++e) { const double *er = p.err + 3 * e; chi2_out[e] = (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]) * edges[e].inv_sigma2; }
    if (cov) { const int ok = inv6_spd(Hlast, cov); if (cov_ok) *cov_ok = ok; }
    free(p.level); free(p.robust); free(p.err); free(edges);
    return (int)nE - nBad;
}

/* Optimizer::LocalBundleAdjustment (Optimizer.cc:757-926) on arrays: optimize(5) with Huber kernels; unless
 * stopped, edges with chi2 > 5.991 (mono) / 7.815 (stereo) or non-positive depth go to level 1 and every kernel
 * is dropped; optimize(10); final classification -> outlier[] (the observations the caller erases).  poses and
 * points are updated in place.  cov (36) = marginal block of pose `cov_pose` (ignored when < 0 or fixed). */
int orc_local_ba(double *poses, const uint8_t *fixed, int nP, double *points, int nX, const OrcEdge *edges, int64_t nE,
                 const double *intr, const int *stop, uint8_t *outlier, int cov_pose, double *cov, int *cov_ok,
                 int *iters, int *trials) {
    Problem p;
    memset(&p, 0, sizeof p);
    p.poses = poses; p.fixed = fixed; p.nP = nP; p.points = points; p.nX = nX; p.edges = edges; p.nE = nE; p.intr = intr;
    p.delta_mono = (double)sqrtf(5.991f); p.delta_stereo = (double)sqrtf(7.815f);   /* const float thHuberMono = sqrt(5.991f) (:646-647) */
    p.level = calloc((size_t)(nE ? nE : 1), 1); p.robust = malloc((size_t)(nE ? nE : 1)); memset(p.robust, 1, (size_t)nE);
    p.err = calloc((size_t)(nE ? nE : 1) * 3, 8); p.slot = malloc((size_t)(nP ? nP : 1) * sizeof(int));
    p.stop = (const volatile int *)stop;
    int nF = 0;
    for (int i = 0; i < nP; ++i) p.slot[i] = fixed[i] ? -1 : nF++;
    p.nF = nF; p.Hpp_last = calloc((size_t)(nF ? nF : 1) * 36, 8);
    if (iters) *iters = 0;
    if (trials) *trials = 0;
    if (cov_ok) *cov_ok = 0;
    if (outlier) memset(outlier, 0, (size_t)nE);
    if (stop && *stop) goto done;
    {
        int tr = 0;
        int n = lm_optimize(&p, 5, &tr);
        if (!(stop && *stop)) {
            for (int64_t e = 0; e < nE; ++e) {
                const double *er = p.err + 3 * e;
                const double c2 = (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]) * edges[e].inv_sigma2;
                int dok;
                double tmp[3];
                edge_eval(poses + 12 * edges[e].pose, points + 3 * edges[e].point, &edges[e], intr, tmp, 0, 0, &dok);
                if (c2 > (edges[e].stereo ? 7.815 : 5.991) || !dok) p.level[e] = 1;
                p.robust[e] = 0;
            }
            n += lm_optimize(&p, 10, &tr);
        }
        if (iters) *iters = n;
        if (trials) *trials = tr;
        if (outlier)
            for (int64_t e = 0; e < nE; ++e) {
                const double *er = p.err + 3 * e;
                const double c2 = (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]) * edges[e].inv_sigma2;
                int dok;
                double tmp[3];
                edge_eval(poses + 12 * edges[e].pose, points + 3 * edges[e].point, &edges[e], intr, tmp, 0, 0, &dok);
                outlier[e] = (c2 > (edges[e].stereo ? 7.815 : 5.991) || !dok);
            }
        if (cov && cov_pose >= 0 && cov_pose < nP && p.slot[cov_pose] >= 0) {
            const int ok = inv6_spd(p.Hpp_last + 36 * p.slot[cov_pose], cov);
            if (cov_ok) *cov_ok = ok;
        }
    }
done:
    free(p.level); free(p.robust); free(p.err); free(p.slot); free(p.Hpp_last);
    return 0;
}

/* One g2o optimize(iterations) call on arrays (what Optimizer::BundleAdjustment, Optimizer.cc:49-271, runs with
 * bRobust): level[] (0 = active) and robust[] per edge are the caller's; err (3 per edge) receives the error
 * vectors g2o would hold afterwards; hpp_last (36 per free pose) the Hpp blocks of the last buildSystem. */
int orc_ba_optimize(double *poses, const uint8_t *fixed, int nP, double *points, int nX, const OrcEdge *edges, int64_t nE,
                    const double *intr, double delta_mono, double delta_stereo, const uint8_t *level,
                    const uint8_t *robust, int iterations, double *err, double *hpp_last, int *trials) {
    Problem p;
    memset(&p, 0, sizeof p);
    p.poses = poses; p.fixed = fixed; p.nP = nP; p.points = points; p.nX = nX; p.edges = edges; p.nE = nE; p.intr = intr;
    p.delta_mono = delta_mono; p.delta_stereo = delta_stereo;
    p.level = (uint8_t *)level; p.robust = (uint8_t *)robust;
    p.err = err ? err : calloc((size_t)(nE ? nE : 1) * 3, 8);
    p.slot = malloc((size_t)(nP ? nP : 1) * sizeof(int));
    int nF = 0;
    for (int i = 0; i < nP; ++i) p.slot[i] = fixed[i] ? -1 : nF++;
    p.nF = nF; p.Hpp_last = hpp_last;
    int tr = 0;
    const int n = lm_optimize(&p, iterations, &tr);
    if (trials) *trials = tr;
    if (!err) free(p.err);
    free(p.slot);
    return n;
}

/* One g2o::SparseOptimizer::optimize(iterations) call for the g2o stand-in of oracle/ref_shims_g2o (the reference's own
 * Optimizer.cc compiled against it drives the schedule: which edges are active, which carry a kernel, how many iterations,
 * when the error vectors are read).  As orc_ba_optimize, plus: points_fixed (the only-pose edges of PoseOptimization hold
 * their map point as a constant: nothing is marginalised), err is IN/OUT (inactive edges keep the vector the previous call
 * left, as g2o's computeActiveErrors does), the stop flag is the caller's `bool *` read as a byte while the solve runs. */
int orc_g2o_optimize(double *poses, const uint8_t *fixed, int nP, double *points, int nX, int points_fixed, const OrcEdge *edges, int64_t nE,
                     const double *intr, double delta_mono, double delta_stereo, const uint8_t *level, const uint8_t *robust, int iterations,
                     double *err, double *hpp_last, const volatile uint8_t *stop_byte, int *trials) {
    Problem p;
    memset(&p, 0, sizeof p);
    p.poses = poses; p.fixed = fixed; p.nP = nP; p.points = points; p.nX = nX; p.points_fixed = points_fixed; p.edges = edges; p.nE = nE; p.intr = intr;
    p.delta_mono = delta_mono; p.delta_stereo = delta_stereo;
    p.level = (uint8_t *)level; p.robust = (uint8_t *)robust;
    p.err = err;
    p.slot = malloc((size_t)(nP ? nP : 1) * sizeof(int));
    int nF = 0;
    for (int i = 0; i < nP; ++i) p.slot[i] = fixed[i] ? -1 : nF++;
    p.nF = nF; p.Hpp_last = hpp_last;
    p.stop8 = stop_byte;
    int tr = 0;
    const int n = lm_optimize(&p, iterations, &tr);
    if (trials) *trials = tr;
    free(p.slot);
    return n;
}

/* inverse of a symmetric positive definite 6 x 6 block (BlockSolver::computeMarginals on a block-diagonal Hpp); 0 when the
 * Cholesky factorisation fails */
int orc_inv6_spd(const double *H, double *out) { return inv6_spd(H, out); }

/* the error vector / depth test of one edge at the current estimates (EdgeSE3ProjectXYZ::computeError, isDepthPositive) */
void orc_edge_error(const double *pose, const double *point, const OrcEdge *edge, const double *intr, double *err3, int *depth_positive) {
    edge_eval(pose, point, edge, intr, err3, 0, 0, depth_positive);
}
