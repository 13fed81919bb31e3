// Stand-in for the reference's include/orbslam/Optimizer.h (TEST INFRASTRUCTURE ONLY): the class declaration with the reference's
// signatures (include/orbslam/Optimizer.h:43-79), over the stand-in SLAM types of ../../optimizer_standins.h instead of the real
// Map.h / MapPoint.h / KeyFrame.h / LoopClosing.h / Frame.h (which pull in OpenCV, DBoW2, Pangolin).  src/orbslam/Optimizer.cc is
// compiled untouched against this header; -DOptimizer=RefOptimizer renames the class so that it can share a test binary with this
// repository's SIVO::Optimizer.
#ifndef REF_SHIM_OPTIMIZER_H
#define REF_SHIM_OPTIMIZER_H

#include "../../optimizer_standins.h"

#include <Eigen/Core>
#include <Eigen/StdVector>

#include <g2o/types/sim3/types_seven_dof_expmap.h>

#include <iostream>
#include <map>
#include <set>

// Optimizer.cc names vector / map / min / max / make_pair / unique_lock / mutex without the qualifier in places: the real headers
// it includes carry `using namespace std;`
#include <algorithm>
#include <mutex>
#include <utility>
using namespace std;

namespace SIVO {

class LoopClosing;

class Optimizer {
 public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    void static BundleAdjustment(const std::vector<KeyFrame *> &vpKF, const std::vector<MapPoint *> &vpMP, int nIterations = 5,
                                 bool *pbStopFlag = nullptr, const unsigned long nLoopKF = 0ul, const bool bRobust = true);
    void static GlobalBundleAdjustment(Map *pMap, int nIterations = 5, bool *pbStopFlag = nullptr, const unsigned long nLoopKF = 0ul,
                                       const bool bRobust = true);
    void static LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap);
    int static PoseOptimization(Frame *pFrame);
    void static OptimizeEssentialGraph(Map *pMap, KeyFrame *pLoopKF, KeyFrame *pCurKF, const LoopClosing::KeyFrameAndPose &NonCorrectedSim3,
                                       const LoopClosing::KeyFrameAndPose &CorrectedSim3,
                                       const std::map<KeyFrame *, std::set<KeyFrame *>> &LoopConnections, const bool &bFixScale);
    static int OptimizeSim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches1, g2o::Sim3 &g2oS12, const float th2,
                            const bool bFixScale);
};

}  // namespace SIVO

#endif
