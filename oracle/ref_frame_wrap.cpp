// C entry point over the reference's OWN SIVO::Frame stereo constructor (src/orbslam/Frame.cc:86-175, compiled
// untouched into oracle/_ref/libref_frame.so together with the reference's ORBextractor.cc): SegmentImage -> two
// ExtractORB threads -> SelectSemanticKeys -> ComputeStereoMatches -> ComputeImageBounds -> AssignFeaturesToGrid, then
// GetFeaturesInArea / isInFrustum / UnprojectStereo on the finished frame.  Stand-ins (ref_shims_frame/): the network
// (hands over the class map the caller prepared), the vocabulary, MapPoint.  OpenCV primitives: oracle/orb_oracle.c.
// Test infrastructure: tests/test_pin_frame.py.
#include <cstdint>
#include <cstring>
#include <vector>

#include "include/orbslam/Frame.h"
#include "ref_arena.inc"

struct RefFrameIn {
    const uint8_t *left, *right;       // rows x cols, 8UC1
    int32_t rows, cols;
    const uint8_t *classes;            // rows x cols
    int32_t nfeatures, nlevels, iniThFAST, minThFAST;
    float scaleFactor, fx, fy, cx, cy, bf;
    // window queries on the finished frame
    int32_t n_queries;
    const float *qx, *qy, *qr;
    const int32_t *qmin, *qmax;
    // isInFrustum on map points
    const float *Tcw;                  // 16 floats, row-major
    int32_t n_points;
    const float *pos, *normal, *min_dist, *max_dist;     // 3n, 3n, n, n
    float cos_limit;
};
struct RefFrameOut {
    int32_t capacity;                  // keys
    int32_t n_left, n_right, n_semantic;
    void *keys_semantic;               // cv::KeyPoint x capacity
    uint8_t *desc_semantic;            // capacity x 32
    float *right, *depth, *unprojected;   // capacity, capacity, 3 x capacity (NaN where depth <= 0)
    int32_t query_capacity;
    int32_t *query_off, *query_idx;    // n_queries + 1, query_capacity
    uint8_t *in_view;                  // n_points
    float *track;                      // 4 x n_points: mTrackProjX, mTrackProjY, mTrackProjXR, mTrackViewCos
    int32_t *track_level;              // n_points
    float bounds[4];                   // mnMinX, mnMaxX, mnMinY, mnMaxY
    float grid_inv[2];
};

extern "C" int ref_frame_build(const RefFrameIn *in, RefFrameOut *out) {
    using namespace SIVO;
    arena_begin(true);
    int rc = 0;
    {
        ORBextractor left(in->nfeatures, in->scaleFactor, in->nlevels, in->iniThFAST, in->minThFAST);
        ORBextractor right(in->nfeatures, in->scaleFactor, in->nlevels, in->iniThFAST, in->minThFAST);
        ORBVocabulary voc;
        BayesianSegNet net;
        net.classes.resize(in->rows, in->cols); net.confidence.resize(in->rows, in->cols); net.entropy.resize(in->rows, in->cols);
        std::memcpy(net.classes.data(), in->classes, (size_t)in->rows * in->cols);
        cv::Mat imL(in->rows, in->cols, CV_8UC1, const_cast<uint8_t *>(in->left)), imR(in->rows, in->cols, CV_8UC1, const_cast<uint8_t *>(in->right));
        cv::Mat K = cv::Mat::zeros(3, 3, CV_32F), dist = cv::Mat::zeros(4, 1, CV_32F);
        K.at<float>(0, 0) = in->fx; K.at<float>(1, 1) = in->fy; K.at<float>(0, 2) = in->cx; K.at<float>(1, 2) = in->cy; K.at<float>(2, 2) = 1.f;
        Frame::mbInitialComputations = true;
        Frame F(imL, imL, imR, 0.0, &left, &right, &voc, &net, K, dist, in->bf, 35.f, 0.f, 0.f);
        out->n_left = (int)F.mvKeysLeft.size(); out->n_right = (int)F.mvKeysRight.size(); out->n_semantic = (int)F.mvKeysSemantic.size();
        const int n = out->n_semantic;
        if (n > out->capacity) rc = -1;
        else if (n > 0 && (int)F.mvRight.size() == n) {
            for (int i = 0; i < n; ++i) {
                std::memcpy(static_cast<uint8_t *>(out->keys_semantic) + 28 * (size_t)i, &F.mvKeysSemantic[(size_t)i], 28);
                std::memcpy(out->desc_semantic + 32 * (size_t)i, F.mDescriptorsSemantic.ptr(i), 32);
                out->right[i] = F.mvRight[(size_t)i]; out->depth[i] = F.mvDepth[(size_t)i];
            }
            out->bounds[0] = Frame::mnMinX; out->bounds[1] = Frame::mnMaxX; out->bounds[2] = Frame::mnMinY; out->bounds[3] = Frame::mnMaxY;
            out->grid_inv[0] = Frame::mfGridElementWidthInv; out->grid_inv[1] = Frame::mfGridElementHeightInv;
            int total = 0;
            out->query_off[0] = 0;
            for (int q = 0; q < in->n_queries; ++q) {
                const std::vector<size_t> v = F.GetFeaturesInArea(in->qx[q], in->qy[q], in->qr[q], in->qmin[q], in->qmax[q]);
                for (size_t k : v) { if (total < out->query_capacity) out->query_idx[total] = (int32_t)k; ++total; }
                out->query_off[q + 1] = total;
            }
            if (total > out->query_capacity) rc = -2;
            cv::Mat Tcw(4, 4, CV_32F);
            std::memcpy(Tcw.data, in->Tcw, 64);
            F.SetPose(Tcw);
            for (int p = 0; p < in->n_points; ++p) {
                MapPoint mp;
                mp.mWorldPos = cv::Mat(3, 1, CV_32F); mp.mNormalVector = cv::Mat(3, 1, CV_32F);
                for (int k = 0; k < 3; ++k) { mp.mWorldPos.at<float>(k) = in->pos[3 * p + k]; mp.mNormalVector.at<float>(k) = in->normal[3 * p + k]; }
                mp.mfMinDistance = in->min_dist[p]; mp.mfMaxDistance = in->max_dist[p];
                out->in_view[p] = F.isInFrustum(&mp, in->cos_limit) ? 1 : 0;
                out->track[4 * p] = mp.mTrackProjX; out->track[4 * p + 1] = mp.mTrackProjY; out->track[4 * p + 2] = mp.mTrackProjXR;
                out->track[4 * p + 3] = mp.mTrackViewCos; out->track_level[p] = mp.mnTrackScaleLevel;
            }
            for (int i = 0; i < n; ++i) {
                const cv::Mat x = F.UnprojectStereo((unsigned long)i);
                for (int k = 0; k < 3; ++k) out->unprojected[3 * i + k] = x.empty() ? __builtin_nanf("") : x.at<float>(k);
            }
        }
    }
    arena_end();
    return rc;
}
