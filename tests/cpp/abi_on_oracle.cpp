// The matcher subset of include/sivo_hip.h implemented over the CPU oracle (oracle/search_oracle.c) — TEST
// infrastructure: linked only into tests/cpp/pin_matcher_cpu, where it lets the SIVO::ORBmatcher templates (gather ->
// C ABI -> scatter) run without a GPU so that they, and the oracle behind them, can be compared with the reference's own
// ORBmatcher.cc (oracle/_ref).  The product never links this file; libsivo_hip.so implements the same entry points on
// the device (sivo_amd/csrc/search.hip).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/sivo_hip.h"

extern "C" {
struct OrcFrame;
struct OrcKp;
OrcFrame *orc_frame_create(const OrcKp *keys, int32_t N, const float *mvRight, const uint8_t *desc, float minX, float maxX, float minY,
                           float maxY, const float *scale, const float *sigma2, const float *inv_sigma2, int32_t nlevels);
void orc_frame_destroy(OrcFrame *F);
int orc_frame_features_in_area(const OrcFrame *F, float x, float y, float r, int minLevel, int maxLevel, int32_t *out, int cap);
int orc_search_by_projection_mappoints(const OrcFrame *F, int nMP, const uint8_t *track_in_view, const float *proj_x, const float *proj_y,
                                       const float *proj_xr, const int32_t *level, const float *view_cos, const uint8_t *mp_desc,
                                       const int32_t *mp_obs, float th, float mfNNratio, int32_t *occ_obs, int32_t *match);
int orc_search_by_projection_frame(const OrcFrame *Cur, int nLast, const uint8_t *valid, const float *pu, const float *pv, const float *pinvz,
                                   const int32_t *last_octave, const float *last_angle, const uint8_t *mp_desc, const int32_t *mp_obs,
                                   float th, int bForward, int bBackward, float mbf, int mbCheckOrientation, int32_t *occ_obs,
                                   int32_t *match);
int orc_search_by_projection_reloc(const OrcFrame *Cur, int nKF, const uint8_t *valid, const float *pu, const float *pv,
                                   const int32_t *pred_level, const float *kf_angle, const uint8_t *mp_desc, float th, int ORBdist,
                                   int mbCheckOrientation, uint8_t *occupied, int32_t *match);
int orc_search_by_projection_kf(const OrcFrame *KF, int nMP, const uint8_t *valid, const float *pu, const float *pv, const int32_t *pred_level,
                                const uint8_t *mp_desc, int th, uint8_t *matched, int32_t *match);
int orc_fuse(const OrcFrame *KF, int nMP, const uint8_t *valid, const float *pu, const float *pv, const float *pur, const int32_t *pred_level,
             const uint8_t *mp_desc, float th, int scw_variant, int32_t *best_idx, int32_t *best_dist);
void orc_search_by_sim3_dir(const OrcFrame *KF, int n, const uint8_t *valid, const float *pu, const float *pv, const int32_t *pred_level,
                            const uint8_t *mp_desc, float th, int32_t *vnMatch);
int orc_search_by_bow_kf_frame(int n_nodes, const int32_t *off1, const int32_t *idx1, const int32_t *off2, const int32_t *idx2,
                               const uint8_t *kf_valid, const OrcKp *keysKF, const uint8_t *descKF, const OrcKp *keysF, const uint8_t *descF,
                               int nF, float mfNNratio, int mbCheckOrientation, int32_t *match_f);
int orc_search_by_bow_kf_kf(int n_nodes, const int32_t *off1, const int32_t *idx1, const int32_t *off2, const int32_t *idx2,
                            const uint8_t *valid1, const OrcKp *keys1, const uint8_t *desc1, int n1, const uint8_t *valid2, const OrcKp *keys2,
                            const uint8_t *desc2, int n2, float mfNNratio, int mbCheckOrientation, int32_t *matches12);
struct OrcEdge;
int orc_pose_optimize(const double *pose0, const double *points, const OrcEdge *edges_in, int64_t nE, const double *intr, uint8_t *outlier,
                      double *pose_out, double *cov, int *cov_ok, double *chi2_out, int *iters, int *trials);
int orc_local_ba(double *poses, const uint8_t *fixed, int nP, double *points, int nX, const OrcEdge *edges, int64_t nE, const double *intr,
                 const int *stop, uint8_t *outlier, int cov_pose, double *cov, int *cov_ok, int *iters, int *trials);
int orc_g2o_optimize(double *poses, const uint8_t *fixed, int nP, double *points, int nX, int points_fixed, const OrcEdge *edges, int64_t nE,
                     const double *intr, double delta_mono, double delta_stereo, const uint8_t *level, const uint8_t *robust, int iterations,
                     double *err, double *hpp_last, const volatile uint8_t *stop_byte, int *trials);
int orc_search_for_triangulation(int n_nodes, const int32_t *off1, const int32_t *idx1, const int32_t *off2, const int32_t *idx2,
                                 const OrcKp *keys1, const float *mvRight1, const uint8_t *has_mp1, const uint8_t *desc1, int n1,
                                 const OrcKp *keys2, const float *mvRight2, const uint8_t *has_mp2, const uint8_t *desc2, int n2,
                                 const float *F12, float ex, float ey, const float *scale2, const float *sigma2_2, int bOnlyStereo,
                                 int mbCheckOrientation, int32_t *matches12);
}

// A frame handle on the CPU: owned copies of the arrays + the oracle's grid.
struct sivo_mframe {
    std::vector<SivoKeyPoint> keys;
    std::vector<float> right, scale, sigma2, inv_sigma2;
    std::vector<uint8_t> desc;
    OrcFrame *F = nullptr;
    const OrcKp *kp() const { return reinterpret_cast<const OrcKp *>(keys.data()); }
};

extern "C" {

const char *sivo_last_error(void) { return "abi_on_oracle: error"; }

// (test hook: how many frame views were built / are alive — tests/cpp/test_frame_cache.cpp counts what ORBmatcher.h's cache saves)
static int g_mframes_created = 0, g_mframes_alive = 0;
int abi_on_oracle_mframes_created(void) { return g_mframes_created; }
int abi_on_oracle_mframes_alive(void) { return g_mframes_alive; }
int sivo_mframe_create(const SivoKeyPoint *keys, int n, const float *u_right, const uint8_t *descriptors, float min_x, float max_x,
                       float min_y, float max_y, const float *scale_factors, const float *level_sigma2, const float *inv_level_sigma2,
                       int nlevels, int, sivo_mframe_t *out) {
    ++g_mframes_created; ++g_mframes_alive;
    sivo_mframe *h = new sivo_mframe;
    h->keys.assign(keys, keys + n);
    if (u_right) h->right.assign(u_right, u_right + n);
    else h->right.assign((size_t)n, -1.0f);
    h->desc.assign(descriptors, descriptors + 32 * (size_t)n);
    h->scale.assign(scale_factors, scale_factors + nlevels);
    h->sigma2.assign(level_sigma2, level_sigma2 + nlevels);
    h->inv_sigma2.assign(inv_level_sigma2, inv_level_sigma2 + nlevels);
    h->F = orc_frame_create(h->kp(), n, h->right.data(), h->desc.data(), min_x, max_x, min_y, max_y, h->scale.data(), h->sigma2.data(),
                            h->inv_sigma2.data(), nlevels);
    *out = h;
    return SIVO_OK;
}
int sivo_mframe_destroy(sivo_mframe_t h) {
    if (h) { orc_frame_destroy(h->F); delete h; --g_mframes_alive; }
    return SIVO_OK;
}
int sivo_mframe_features_in_area(sivo_mframe_t h, float x, float y, float r, int min_level, int max_level, int32_t *out, int capacity,
                                 int *n_out) {
    *n_out = orc_frame_features_in_area(h->F, x, y, r, min_level, max_level, out, capacity);
    return SIVO_OK;
}
int sivo_search_by_projection_mappoints(sivo_mframe_t F, int n_mp, const uint8_t *track_in_view, const float *proj_x, const float *proj_y,
                                        const float *proj_xr, const int32_t *level, const float *view_cos, const uint8_t *mp_desc,
                                        const int32_t *mp_obs, float th, float nn_ratio, int32_t *occ_obs, int32_t *match, int *n_matches) {
    *n_matches = orc_search_by_projection_mappoints(F->F, n_mp, track_in_view, proj_x, proj_y, proj_xr, level, view_cos, mp_desc, mp_obs, th,
                                                    nn_ratio, occ_obs, match);
    return SIVO_OK;
}
int sivo_search_by_projection_frame(sivo_mframe_t current, int n_last, const uint8_t *valid, const float *u, const float *v, const float *inv_z,
                                    const int32_t *last_octave, const float *last_angle, const uint8_t *mp_desc, const int32_t *mp_obs,
                                    float th, int forward, int backward, float bf, int check_orientation, int32_t *occ_obs, int32_t *match,
                                    int *n_matches) {
    *n_matches = orc_search_by_projection_frame(current->F, n_last, valid, u, v, inv_z, last_octave, last_angle, mp_desc, mp_obs, th, forward,
                                                backward, bf, check_orientation, occ_obs, match);
    return SIVO_OK;
}
int sivo_search_by_projection_reloc(sivo_mframe_t current, int n_kf, const uint8_t *valid, const float *u, const float *v,
                                    const int32_t *pred_level, const float *kf_angle, const uint8_t *mp_desc, float th, int orb_dist,
                                    int check_orientation, uint8_t *occupied, int32_t *match, int *n_matches) {
    *n_matches = orc_search_by_projection_reloc(current->F, n_kf, valid, u, v, pred_level, kf_angle, mp_desc, th, orb_dist, check_orientation,
                                                occupied, match);
    return SIVO_OK;
}
int sivo_search_by_projection_kf(sivo_mframe_t kf, int n_mp, const uint8_t *valid, const float *u, const float *v, const int32_t *pred_level,
                                 const uint8_t *mp_desc, int th, uint8_t *matched, int32_t *match, int *n_matches) {
    *n_matches = orc_search_by_projection_kf(kf->F, n_mp, valid, u, v, pred_level, mp_desc, th, matched, match);
    return SIVO_OK;
}
int sivo_fuse(sivo_mframe_t kf, int n_mp, const uint8_t *valid, const float *u, const float *v, const float *ur, const int32_t *pred_level,
              const uint8_t *mp_desc, float th, int scw_variant, int32_t *best_idx, int32_t *best_dist, int *n_fused) {
    std::vector<int32_t> dist((size_t)(n_mp > 0 ? n_mp : 1));
    std::vector<float> no_ur;
    if (!ur) { no_ur.assign((size_t)(n_mp > 0 ? n_mp : 1), 0.f); ur = no_ur.data(); }
    const int n = orc_fuse(kf->F, n_mp, valid, u, v, ur, pred_level, mp_desc, th, scw_variant, best_idx, best_dist ? best_dist : dist.data());
    if (n_fused) *n_fused = n;
    return SIVO_OK;
}
int sivo_search_by_sim3_dir(sivo_mframe_t kf, int n, const uint8_t *valid, const float *u, const float *v, const int32_t *pred_level,
                            const uint8_t *mp_desc, float th, int32_t *match_out) {
    orc_search_by_sim3_dir(kf->F, n, valid, u, v, pred_level, mp_desc, th, match_out);
    return SIVO_OK;
}
int sivo_search_by_bow_kf_frame(int n_nodes, const int32_t *off1, const int32_t *idx1, const int32_t *off2, const int32_t *idx2,
                                const uint8_t *kf_valid, const SivoKeyPoint *keys_kf, const uint8_t *desc_kf, int, sivo_mframe_t frame,
                                float nn_ratio, int check_orientation, int32_t *match_f, int *n_matches) {
    *n_matches = orc_search_by_bow_kf_frame(n_nodes, off1, idx1, off2, idx2, kf_valid, reinterpret_cast<const OrcKp *>(keys_kf), desc_kf,
                                            frame->kp(), frame->desc.data(), (int)frame->keys.size(), nn_ratio, check_orientation, match_f);
    return SIVO_OK;
}
int sivo_search_by_bow_kf_kf(int n_nodes, const int32_t *off1, const int32_t *idx1, const int32_t *off2, const int32_t *idx2,
                             const uint8_t *valid1, const SivoKeyPoint *keys1, const uint8_t *desc1, int n1, const uint8_t *valid2,
                             sivo_mframe_t kf2, float nn_ratio, int check_orientation, int32_t *matches12, int *n_matches) {
    *n_matches = orc_search_by_bow_kf_kf(n_nodes, off1, idx1, off2, idx2, valid1, reinterpret_cast<const OrcKp *>(keys1), desc1, n1, valid2,
                                         kf2->kp(), kf2->desc.data(), (int)kf2->keys.size(), nn_ratio, check_orientation, matches12);
    return SIVO_OK;
}
int sivo_search_for_triangulation(int n_nodes, const int32_t *off1, const int32_t *idx1, const int32_t *off2, const int32_t *idx2,
                                  const SivoKeyPoint *keys1, const float *u_right1, const uint8_t *has_mp1, const uint8_t *desc1, int n1,
                                  sivo_mframe_t kf2, const uint8_t *has_mp2, const float F12[9], float ex, float ey, int only_stereo,
                                  int check_orientation, int32_t *matches12, int *n_matches) {
    std::vector<float> no_right;
    if (!u_right1) { no_right.assign((size_t)(n1 > 0 ? n1 : 1), -1.0f); u_right1 = no_right.data(); }
    *n_matches = orc_search_for_triangulation(n_nodes, off1, idx1, off2, idx2, reinterpret_cast<const OrcKp *>(keys1), u_right1, has_mp1, desc1,
                                              n1, kf2->kp(), kf2->right.data(), has_mp2, kf2->desc.data(), (int)kf2->keys.size(), F12, ex, ey,
                                              kf2->scale.data(), kf2->sigma2.data(), only_stereo, check_orientation, matches12);
    return SIVO_OK;
}
// ---- the optimisation loops (include/sivo_hip.h: sivo_pose_optimize / sivo_local_ba / sivo_ba_optimize) over
// oracle/ba_solve_oracle.c, for tests/cpp/pin_optimizer.cpp: the SIVO::Optimizer member templates (gather -> C ABI -> scatter) run
// without a GPU and are compared with the reference's own Optimizer.cc (oracle/_ref/ref_optimizer.o over the g2o stand-in).
int sivo_pose_optimize(const double pose0[12], const double *points, int, const SivoEdge *edges, int64_t n_edges, const double intr[5],
                       uint8_t *outlier, double pose_out[12], double cov[36], int *cov_ok, double *chi2, int *n_inliers, int *iterations,
                       int *trials) {
    const int n = orc_pose_optimize(pose0, points, reinterpret_cast<const OrcEdge *>(edges), n_edges, intr, outlier, pose_out, cov, cov_ok, chi2,
                                    iterations, trials);
    if (n_inliers) *n_inliers = n;
    return SIVO_OK;
}
int sivo_local_ba(double *poses, const uint8_t *pose_fixed, int n_poses, double *points, int n_points, const SivoEdge *edges, int64_t n_edges,
                  const double intr[5], const volatile uint8_t *stop_flag, uint8_t *outlier, int cov_pose, double *cov, int *cov_ok,
                  int *iterations, int *trials) {
    // (the oracle entry point polls an int; the reference's flag is a bool that the tests here never raise mid-solve)
    int stop = stop_flag && *stop_flag ? 1 : 0;
    orc_local_ba(poses, pose_fixed, n_poses, points, n_points, reinterpret_cast<const OrcEdge *>(edges), n_edges, intr, stop_flag ? &stop : nullptr,
                 outlier, cov_pose, cov, cov_ok, iterations, trials);
    return SIVO_OK;
}
int sivo_ba_optimize(double *poses, const uint8_t *pose_fixed, int n_poses, double *points, int n_points, const SivoEdge *edges, int64_t n_edges,
                     const double intr[5], double delta_mono, double delta_stereo, const uint8_t *level, const uint8_t *robust, int iterations,
                     const volatile uint8_t *stop_flag, double *err_out, double *hpp_last_out, int *iterations_run, int *trials) {
    std::vector<uint8_t> lv(level ? level : nullptr, level ? level + n_edges : nullptr), rb(robust ? robust : nullptr, robust ? robust + n_edges : nullptr);
    if (!level) lv.assign((size_t)n_edges, 0);
    if (!robust) rb.assign((size_t)n_edges, 1);
    std::vector<double> err((size_t)(n_edges ? n_edges : 1) * 3, 0.0);
    const int n = orc_g2o_optimize(poses, pose_fixed, n_poses, points, n_points, 0, reinterpret_cast<const OrcEdge *>(edges), n_edges, intr, delta_mono,
                                   delta_stereo, lv.data(), rb.data(), iterations, err.data(), hpp_last_out, stop_flag, trials);
    if (err_out) std::memcpy(err_out, err.data(), (size_t)n_edges * 24);
    if (iterations_run) *iterations_run = n;
    return SIVO_OK;
}
// SIVO::Optimizer::LinearizeEdges is not part of this comparison.
int sivo_ba_linearize(const double *, int, const double *, int, const SivoEdge *, int64_t, const double *, double, double, double *, double *, double *,
                      double *, double *, double *, uint8_t *) {
    std::abort();
}
// SIVO::ORBmatcher::BestTwo is not part of this comparison.
int sivo_hamming_argmin2(const uint8_t *, int, const uint8_t *, int, const int32_t *, const int32_t *, int32_t *, int32_t *, int32_t *,
                         int32_t *) {
    std::abort();
}
}
